// k_frames3s: the wave-level frame kernel for n_fft = 2048 / hop = 512 (BASELINE.json configs[4], the "music"
// configuration of 04_feature_extraction_experiment/feature_extractor.py:19-23; librosa stft / mel / power_to_db
// semantics as in oracle/cpu_ref.py).  Same design as k_frames3 (afx_frames3.hip) -- autonomous waves, one 8.5 KB LDS
// image per wave, packed-f32 butterflies, samples read once before the trim decision -- with ONE frame per FFT:
//   * the 2048 real samples of a frame are packed as 1024 complex points c[n] = x[2n] + i x[2n+1]; a lane holds points
//     l + 64 u as the float2 rows it loads (8 bytes per lane, 512 contiguous bytes per wave-load), so window and
//     pre-emphasis are packed instructions on whole rows; hop = 512 samples = 4 rows: a frame keeps 12 rows of its
//     predecessor and takes in 4;
//   * the same 16 x 8 x 8 complex FFT as k_frames3; its last pass hands every lane Z[k] and Z[1024 - k], exactly the
//     operands of the real-FFT split X[k] = (E - i W^k O) / 2, X[1024 - k] = conj((E + i W^k O) / 2) with
//     E = Z[k] + conj Z[1024 - k], O = Z[k] - conj Z[1024 - k], W = exp(-2 pi i / 2048) -- one twiddle product per
//     bin pair, both powers formed as one packed pair;
//   * the power spectrum of the frame (1025 floats) takes the image's place; a mel lane reads four bins per 16-byte
//     read (schedule: afx_tables.cpp build_f3_mel with align = 4).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "afx_device.h"
#include "afx_devenv.h"
#include "afx_frames3.h"
#include "afx_frames3_dev.h"

namespace afx {

constexpr int kF3sTabFloats = 2048 + 1024 + 2048;      // FFT twiddles (as k_frames3) + split twiddles 8 x 64 float2 + window 16 x 64 float2

size_t frames3s_lds_bytes(int waves, const F3Tables& ft) {
  return (size_t)(waves * kF3ExFloats + kF3sTabFloats + ft.mel_wfloats + ft.mel_rounds * 64) * sizeof(float);
}

// DESC: instead of the mel / dB stage, the frame-level spectral descriptors the reference's experiment extractor takes
// from librosa at its defaults (n_fft 2048, hop 512: exactly this kernel's shape;
// 04_feature_extraction_experiment/feature_extractor.py:497-506) off the magnitude spectrum sqrt(P) in the image:
// spectral_centroid, spectral_bandwidth (p = 2, normalised), spectral_rolloff (85 %), and per octave band of
// spectral_contrast the mean of the `cnt` largest and smallest magnitudes (peak / valley; the dB difference and its
// clip-global top_db clamp are taken on the host).  17 floats per frame at desc_out[desc_offs[clip] + 17 t].
// NBS > 0: the mel schedule is known at compile time -- the batch counts (NBS) and lanes per filter (WDS) of up to four rounds
// as nibbles, first round lowest (the reference's 44.1 kHz / 128 mels plans batches 3, 1, 5, 5 at widths 1, 4, 2, 4:
// NBS 0x5513, WDS 0x4241) -- and is walked as straight-line code.  NBS = 0: any schedule, batches behind uniform branches.
template <int FMT, int WAVES, bool SPEC, bool DESC, int NBS = 0, int WDS = 0>
__global__ __launch_bounds__(WAVES * 64) void k_frames3s(const void* __restrict__ samples,
                                                         ClipInfo* __restrict__ info,
                                                         const BlockDesc* __restrict__ blocks, int nblocks,
                                                         const int* __restrict__ nblocks_dev,
                                                         F3Tables ft, KParams kp,
                                                         float* __restrict__ logmel,
                                                         float* __restrict__ blockmax,
                                                         float* __restrict__ bsum,
                                                         float* __restrict__ desc_out,
                                                         const int64_t* __restrict__ desc_offs,
                                                         SpecBands sb, int* __restrict__ work_ctr) {
  constexpr int N = 2048, HOP = 512;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (nblocks_dev) {                       // the list launch: an empty list (nothing was trimmed) costs no table set-up
    nblocks = *nblocks_dev;
    if (nblocks <= 0) return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* const tabs = smem + WAVES * kF3ExFloats;
  v2* const T2 = reinterpret_cast<v2*>(tabs);                 // [r][16]: W_128^(c r)
  v2* const T3a = reinterpret_cast<v2*>(tabs + 256);          // [r-1][lane]: W_1024^(ja r)
  v2* const T3b = reinterpret_cast<v2*>(tabs + 256 + 896);    // [r-1][lane]: W_1024^(jb r)
  v2* const TS = reinterpret_cast<v2*>(tabs + 2048);          // [s][lane]: W_2048^k of the lane's bin pair s
  v2* const WT = reinterpret_cast<v2*>(tabs + 2048 + 1024);   // [u][lane]: window pair of row u, x 0.5
  float* const MW = tabs + kF3sTabFloats;                     // mel weights [round][batch][lane][4]
  int* const MM = reinterpret_cast<int*>(MW + ft.mel_wfloats);   // mel meta [round][lane]
  v2* const E = reinterpret_cast<v2*>(smem + wave * kF3ExFloats);
  float* const XB = reinterpret_cast<float*>(E);

  {
    const v2* w1024 = reinterpret_cast<const v2*>(ft.w1024);       // exp(-2 pi i k / 1024), k < 512
    const v2* w2048 = reinterpret_cast<const v2*>(ft.w2048);       // exp(-2 pi i k / 2048), k < 1024
    auto W = [&](int m) { const v2 v = w1024[m & 511]; return (m & 512) ? -v : v; };
    if (tid < 128) T2[tid] = W(8 * (tid >> 4) * (tid & 15));
    if (tid < 64) {
      const int jbt = tid ? 128 - tid : 64;
#pragma unroll
      for (int r = 1; r < 8; ++r) { T3a[(r - 1) * 64 + tid] = W(tid * r); T3b[(r - 1) * 64 + tid] = W(jbt * r); }
      // bin pair s of a lane: k = lane + 128 s; lane 0 holds the self-mirrored butterflies 0 and 64, its pairs 4..7 are k = 64 + 128 s
#pragma unroll
      for (int s = 0; s < 8; ++s) TS[s * 64 + tid] = w2048[((tid == 0 && s >= 4) ? 64 : tid) + 128 * s];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int n = 2 * (tid + 64 * u);
        WT[u * 64 + tid] = v2{0.5f * ft.window[n], 0.5f * ft.window[n + 1]};
      }
    }
    for (int i = tid; i < ft.mel_wfloats; i += WAVES * 64) MW[i] = ft.mel_w[i];
    for (int i = tid; i < ft.mel_rounds * 64; i += WAVES * 64) MM[i] = ft.mel_meta[i];
    for (int i = lane; i < kF3ExFloats; i += 64) XB[i] = 0.f;
  }
  __syncthreads();

  const v2 H = {0.70710678118654752440f, 0.70710678118654752440f};
  const v2 W1 = {0.92387953251128675613f, -0.38268343236508977173f};      // W16^1
  const v2 W3 = {0.38268343236508977173f, -0.92387953251128675613f};      // W16^3
  const bool pre = (kp.flags & AFX_FLAG_PREEMPH) != 0;
  const float b1 = kp.preemph_b1;
  const int M = kp.n_mels;
  const int jb = lane ? 128 - lane : 64;
  const int lq = lane ? lane : 64;         // first bin of the lane's pairs 4..7 (minus 128 s)
  // exchange-1 image [k1][66] (transposed, row stride 66 = 2 mod 32): a lane stores its value k at row k, column = its own
  // index -- 16 consecutive lanes, consecutive slots: conflict-free -- and reads row k1 = lane & 15 at column (lane >> 4) + 4 u:
  // the two 16-lane quarters of a 32-lane read group differ by one column, i.e. fall on the even / the odd slots of
  // 2 k1 + column -- conflict-free as well (round 2's [l][17] image had one 2-way conflict per read: 16 LDS cycles a pass)
  v2* const e1w = E + lane;
  const v2* const e1r = E + 66 * (lane & 15) + (lane >> 4);
  v2* const e2w = E + 128 * (lane >> 4) + (lane & 15);
  const v2* const ea = E + lane;
  const v2* const eb = E + jb;

  auto raw_ld = [&](int64_t idx) -> float {
    if constexpr (FMT == AFX_FMT_S16) return (float)((const int16_t*)samples)[idx] * (1.0f / 32768.0f);
    else return ((const float*)samples)[idx];
  };
  typedef typename std::conditional<FMT == AFX_FMT_S16, int16_t, float>::type sample_t;
  auto row_ld = [&](const sample_t* base, unsigned idx) -> float {
    if constexpr (FMT == AFX_FMT_S16) return (float)base[idx] * (1.0f / 32768.0f);
    else return base[idx];
  };
  const int n_rounds = ft.mel_rounds;
  const float amin = kp.amin;

  // sum of squares of four float2 rows (one 512-sample sub-block of the pre-emphasised signal), wave-wide
  auto subblock = [&](v2 r0, v2 r1, v2 r2, v2 r3, const BlockDesc& bd, int j) {
    v2 a = r0 * r0; a = r1 * r1 + a; a = r2 * r2 + a; a = r3 * r3 + a;
    float q = a.x + a.y;
    q += F3_DPP(q, 0xB1); q += F3_DPP(q, 0x4E); q += F3_DPP(q, 0x141); q += F3_DPP(q, 0x140);
    const int qi = __float_as_int(q);
    const float t = (__int_as_float(__builtin_amdgcn_readlane(qi, 0)) + __int_as_float(__builtin_amdgcn_readlane(qi, 16))) +
                    (__int_as_float(__builtin_amdgcn_readlane(qi, 32)) + __int_as_float(__builtin_amdgcn_readlane(qi, 48)));
    if (j >= 0 && j < bd.pad_[1]) {
      if (lane == 0) bsum[bd.pad_[0] + j] = t;
      if (!(fabsf(t) < INFINITY)) {
        const bool bad = !(isfinite(r0.x) && isfinite(r0.y) && isfinite(r1.x) && isfinite(r1.y) &&
                           isfinite(r2.x) && isfinite(r2.y) && isfinite(r3.x) && isfinite(r3.y));
        if (__any(bad) && lane == 0) atomicOr(&info[bd.clip].nonfinite, 1u);
      }
    }
  };

  const int total_waves = gridDim.x * WAVES;
  F3Runs runs = f3_runs_init(work_ctr, nblocks, total_waves, blockIdx.x * WAVES + wave, false);
  do {
  for (int b = runs.b_lo; b < runs.b_hi; b += runs.stride) {
    BlockDesc bd = blocks[b];
    if (!bd.active) continue;
    // consecutive blocks of one clip inside a run continue the row pipeline across the block boundary (4 new rows, as
    // inside a block) instead of re-loading 16: no halo re-read, no exposed load latency at a block's start
    bool first = true;
    v2 R[16];
    for (;;) {                                            // the chained blocks
    const int Tleft = bd.T - bd.t0;
    const int nfr = Tleft >= 16 ? 16 : Tleft;
    bool chain = false;
    if (SPEC && nfr == 16 && runs.stride == 1 && b + 1 < runs.b_hi) {
      const BlockDesc* nx = blocks + b + 1;
      chain = nx->active && nx->clip == bd.clip && nx->t0 == bd.t0 + 16;
    }
    const int64_t sbase = bd.sample_base;
    const sample_t* const sp = (const sample_t*)samples + sbase;
    auto interior = [&](int j0, int j1) -> bool {
      return (j0 - 1 >= bd.have_lo) && (j1 <= bd.have_hi) && (j0 >= bd.keep_lo) && (j1 <= bd.keep_hi);
    };
    auto edge_sample = [&](int j) -> float {
      const int lo = bd.have_lo, hi = bd.have_hi - 1;
      const int jc = j < lo ? lo : (j > hi ? hi : j), jp = (j - 1) < lo ? lo : ((j - 1) > hi ? hi : (j - 1));
      const float y = (jc == j) ? raw_ld(sbase + jc) : 0.f;
      const float yp = (jp == j - 1) ? raw_ld(sbase + jp) : 0.f;
      float v = y;
      if (pre) {
        v = f3_pre1(y, yp, b1);
        if (j == lo) v = f3_pre0(raw_ld(bd.clip_off), raw_ld(bd.clip_off + 1));
      }
      return (j >= bd.keep_lo && j < bd.keep_hi) ? v : 0.f;
    };
    // one float2 row: samples j0 + 2 lane, + 1 (pre-emphasised)
    auto pre_row = [&](float x0, float x1, float xp) -> v2 {
      return pre ? v2{f3_pre1(x0, xp, b1), f3_pre1(x1, x0, b1)} : v2{x0, x1};
    };

    // ---- rows of the first frame: staged samples [0, 2048)
    if (first) {
    if (interior(0, N)) {
      float x0[16], x1[16], xp[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        x0[u] = row_ld(sp, 128 * u + 2 * lane); x1[u] = row_ld(sp, 128 * u + 2 * lane + 1); xp[u] = row_ld(sp - 1, 128 * u + 2 * lane);
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) R[u] = pre_row(x0[u], x1[u], xp[u]);
    } else {
#pragma unroll 1
      for (int u = 0; u < 16; ++u) {
        XB[128 * u + 2 * lane] = edge_sample(128 * u + 2 * lane);
        XB[128 * u + 2 * lane + 1] = edge_sample(128 * u + 2 * lane + 1);
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) R[u] = E[64 * u + lane];
    }
    if constexpr (SPEC) {
      // rows 8..11 / 12..15 are sub-blocks t0 / t0 + 1 of the clip; the former belongs to the previous block unless this is the clip's first
      if (bd.t0 == 0) subblock(R[8], R[9], R[10], R[11], bd, 0);
      subblock(R[12], R[13], R[14], R[15], bd, bd.t0 + 1);
    }
    }
    float lmax = -INFINITY;
    float* const tile = logmel + bd.frame_slot * (int64_t)M;     // [frame][mel]

#pragma unroll 1
    for (int f = 0; f < nfr; ++f) {
      // ---- c[n] = w (x[2n] + i x[2n+1])
      v2 z[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) z[u] = R[u] * ldv(WT + u * 64 + lane);
#pragma unroll
      for (int u = 0; u < 12; ++u) R[u] = R[u + 4];
      const bool more = f + 1 < nfr || chain;              // the next frame may be the next block's first
      const int jn = HOP * f + N;                          // the next frame's 4 new rows: staged samples [jn, jn + 512)
      const bool nint = more && interior(jn, jn + HOP);

      // ---- 1024-point complex FFT (k_frames3's schedule)
      f3_dft16(z, H, W1, W3);
#pragma unroll
      for (int k = 0; k < 16; ++k) stv(e1w + 66 * k, z[k]);       // unmerged 8-byte stores: 2 x 6 LDS cycles against 13 for ds_write2_b64
#pragma unroll
      for (int u = 0; u < 16; ++u) z[u] = ldv(e1r + 4 * u);
      {
        v2 tw[8];
#pragma unroll
        for (int r = 1; r < 8; ++r) tw[r] = ldv(T2 + r * 16 + (lane & 15));
        v2 xa[8], xb[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { xa[r] = z[2 * r]; xb[r] = z[2 * r + 1]; }
#pragma unroll
        for (int r = 1; r < 8; ++r) cmul2(xa[r], tw[r], xb[r], tw[r]);
        f3_dft8(xa, H); f3_dft8(xb, H);
#pragma unroll
        for (int r = 0; r < 8; ++r) { stv(e2w + 16 * r, xa[r]); stv(e2w + 16 * r + 512, xb[r]); }
      }
      v2 A[8], B[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) { A[r] = ldv(ea + 128 * r); B[r] = ldv(eb + 128 * r); }
#pragma unroll
      for (int r = 1; r < 8; ++r) cmul2(A[r], ldv(T3a + (r - 1) * 64 + lane), B[r], ldv(T3b + (r - 1) * 64 + lane));
      f3_dft8(A, H); f3_dft8(B, H);
      // A[s] = Z[lane + 128 s], B[s] = Z[jb + 128 s]; mirror of A[s] is B[7-s]; lane 0 re-seated as in k_frames3
      const v2 nyq = A[4];
      if (lane == 0) { A[4] = B[4]; B[4] = A[5]; A[5] = B[5]; B[5] = A[6]; A[6] = B[6]; B[6] = A[7]; A[7] = B[7]; B[7] = A[0]; }
      // ---- real-FFT split + power: pair s = (Z[k], Z[1024 - k]), k = lane + 128 s (lane 0, s >= 4: 64 + 128 s)
      //   E = Z[k] + conj Z[N-k], O = Z[k] - conj Z[N-k], T = W^k O;  |X[k]|^2 = |E - iT|^2, |X[N-k]|^2 = |E + iT|^2
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const v2 za = A[s], zb = B[7 - s];
        v2 Ev, Ov;
        asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(Ev) : "v"(za), "v"(zb));      // (za.x + zb.x, za.y - zb.y)
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(Ov) : "v"(za), "v"(zb));      // (za.x - zb.x, za.y + zb.y)
        const v2 T = cmul(Ov, ldv(TS + s * 64 + lane));
        v2 U, V;      // U = (Xp.x, Xm.x) = (E.x + T.y, E.x - T.y);  V = (Xp.y, Xm.y) = (E.y - T.x, E.y + T.x)
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(U) : "v"(Ev), "v"(T));
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(V) : "v"(Ev), "v"(T));
        const v2 P = U * U + V * V;                                  // (|X[k]|^2, |X[N-k]|^2)
        const int k0 = (s < 4 ? lane : lq) + 128 * s;
        XB[k0] = P.x; XB[1024 - k0] = P.y;
      }
      if (lane == 0) XB[512] = 4.f * (nyq.x * nyq.x + nyq.y * nyq.y);

      // ---- the next frame's new rows are fetched under the mel phase
      float nx0[4], nx1[4], nxp[4];
      if (nint) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          nx0[i] = row_ld(sp + jn, 128 * i + 2 * lane); nx1[i] = row_ld(sp + jn, 128 * i + 2 * lane + 1); nxp[i] = row_ld(sp + jn - 1, 128 * i + 2 * lane);
        }
      }
      if constexpr (DESC) {
        // ---- spectral descriptors: lane l holds the magnitudes of bins 16 l .. 16 l + 15 (lane 63 also bin 1024)
        float mg[17];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 p4 = *reinterpret_cast<const float4*>(XB + 16 * lane + 4 * q);
          mg[4 * q] = sqrtf(p4.x); mg[4 * q + 1] = sqrtf(p4.y); mg[4 * q + 2] = sqrtf(p4.z); mg[4 * q + 3] = sqrtf(p4.w);
        }
        mg[16] = lane == 63 ? sqrtf(XB[1024]) : 0.f;
        auto wsum = [&](float q) -> float {
          q += F3_DPP(q, 0xB1); q += F3_DPP(q, 0x4E); q += F3_DPP(q, 0x141); q += F3_DPP(q, 0x140);
          const int qi = __float_as_int(q);
          return (__int_as_float(__builtin_amdgcn_readlane(qi, 0)) + __int_as_float(__builtin_amdgcn_readlane(qi, 16))) +
                 (__int_as_float(__builtin_amdgcn_readlane(qi, 32)) + __int_as_float(__builtin_amdgcn_readlane(qi, 48)));
        };
        const float hz = sb.hz_per_bin, k0f = (float)(16 * lane);
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int j = 0; j < 17; ++j) { s0 += mg[j]; s1 = fmaf(mg[j], (k0f + (float)j) * hz, s1); }
        const float tot = wsum(s0);
        const float len = tot < 1.17549435e-38f ? 1.0f : tot;            // librosa.util.normalize(norm=1): a column below tiny keeps its scale
        const float cen = wsum(s1) / len;
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 17; ++j) { const float d = (k0f + (float)j) * hz - cen; s2 = fmaf(mg[j], d * d, s2); }
        const float bw = sqrtf(wsum(s2) / len);
        // roll-off: lowest bin whose running sum reaches 85 % of the total (lane totals scanned across the wave)
        float run[17], acc = 0.f;
#pragma unroll
        for (int j = 0; j < 17; ++j) { acc += mg[j]; run[j] = acc; }
        float pre = acc;                                   // inclusive scan of lane totals
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(pre, o); if (lane >= o) pre += t; }
        const float base = pre - acc, thr = sb.roll_percent * tot;
        int kroll = 1 << 20;
#pragma unroll
        for (int j = 16; j >= 0; --j) if ((j < 16 || lane == 63) && base + run[j] >= thr) kroll = 16 * lane + j;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { const int t = __shfl_xor(kroll, o); kroll = t < kroll ? t : kroll; }
        const float roll = (kroll < (1 << 20) ? (float)kroll : 0.f) * hz;
        float* const dst = desc_out + desc_offs[bd.clip] + (int64_t)(bd.t0 + f) * 17;
        if (lane == 0) { dst[0] = cen; dst[1] = bw; dst[2] = roll; }
        // contrast bands: repeated extraction of the band's smallest (largest) remaining magnitude
#pragma unroll 1
        for (int bnd = 0; bnd < 7; ++bnd) {
          const int lo = sb.lo[bnd], hi = sb.hi[bnd], cnt = sb.cnt[bnd];
          unsigned inb = 0;
#pragma unroll
          for (int j = 0; j < 17; ++j) { const int k = 16 * lane + j; if (k >= lo && k <= hi && (j < 16 || lane == 63)) inb |= 1u << j; }
#pragma unroll 1
          for (int side = 0; side < 2; ++side) {           // 0: valley (smallest), 1: peak (largest)
            unsigned avail = inb;
            float total = 0.f;
#pragma unroll 1
            for (int r = 0; r < cnt; ++r) {
              float m = side ? -1.f : INFINITY;
              int jm = 0;
#pragma unroll
              for (int j = 0; j < 17; ++j) {
                const bool ok = (avail >> j) & 1u;
                const bool better = ok && (side ? mg[j] > m : mg[j] < m);
                m = better ? mg[j] : m; jm = better ? j : jm;
              }
              float w = m;
              if (side) { w = fmaxf(w, F3_DPP(w, 0xB1)); w = fmaxf(w, F3_DPP(w, 0x4E)); w = fmaxf(w, F3_DPP(w, 0x141)); w = fmaxf(w, F3_DPP(w, 0x140)); }
              else { w = fminf(w, F3_DPP(w, 0xB1)); w = fminf(w, F3_DPP(w, 0x4E)); w = fminf(w, F3_DPP(w, 0x141)); w = fminf(w, F3_DPP(w, 0x140)); }
              const int wi = __float_as_int(w);
              const float r0 = __int_as_float(__builtin_amdgcn_readlane(wi, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(wi, 16));
              const float r2 = __int_as_float(__builtin_amdgcn_readlane(wi, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(wi, 48));
              const float best = side ? fmaxf(fmaxf(r0, r1), fmaxf(r2, r3)) : fminf(fminf(r0, r1), fminf(r2, r3));
              const bool have = side ? m >= 0.f : m < INFINITY;      // the lane still had a candidate
              const unsigned long long holders = __ballot(have && m == best);
              if (holders == 0ull) break;                   // band exhausted
              const int owner = __ffsll((long long)holders) - 1;
              if (lane == owner) avail &= ~(1u << jm);
              total += best;
            }
            if (lane == 0) dst[3 + 7 * side + bnd] = cnt > 0 ? total / (float)cnt : 0.f;     // [3..9] valley, [10..16] peak
          }
        }
      } else {
      // ---- mel + dB: a lane walks 4 * nb consecutive bins of its filter, four per 16-byte read
      const bool vF = f < Tleft;
      float* const rowF = tile + (unsigned)(f * M);
      __builtin_amdgcn_s_setprio(1);          // a wave in its mel phase (LDS reads) goes ahead of its SIMD mates' FFTs: k_frames3
      if constexpr (NBS > 0) {
        int woff = 0;
#pragma unroll
        for (int rd = 0; rd < 4; ++rd) {
          constexpr int kNbs = NBS, kWds = WDS;
          const int nb = (kNbs >> (4 * rd)) & 15, wd = (kWds >> (4 * rd)) & 15;
          if (nb == 0) break;
          const int meta = MM[rd * 64 + lane];
          const float4* pp = reinterpret_cast<const float4*>(XB + (meta & 2047));
          const float4* ww = reinterpret_cast<const float4*>(MW + woff) + lane;
          woff += nb * 256;
          v2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (i < nb) {
              const float4 c = ww[64 * i], q = pp[i];
              if (i == 0) { a0 = v2{q.x, q.y} * v2{c.x, c.y}; a1 = v2{q.z, q.w} * v2{c.z, c.w}; }
              else { a0 = v2{q.x, q.y} * v2{c.x, c.y} + a0; a1 = v2{q.z, q.w} * v2{c.z, c.w} + a1; }
            }
          }
          const v2 a = a0 + a1;
          float acc = a.x + a.y;
          if (wd >= 2) acc += F3_DPP(acc, 0xB1);
          if (wd >= 4) acc += F3_DPP(acc, 0x4E);
          if (wd >= 8) acc += F3_DPP(acc, 0x141);
          const float L = 3.01029995663981195f * __builtin_amdgcn_logf(f3_max(acc, amin));
          if (wd == 1 || (meta & (1 << 20))) {              // width 1: every lane owns its filter (launch_frames3s checks)
            const unsigned m = (meta >> 11) & 511;
            if (vF) { rowF[m] = L; lmax = f3_max(lmax, L); }
          }
        }
      } else {
#pragma unroll 1
      for (int rd = 0; rd < n_rounds; ++rd) {
        const uint32_t rp = ft.mel_rp[rd];
        const int meta = MM[rd * 64 + lane];
        const float4* pp = reinterpret_cast<const float4*>(XB + (meta & 2047));
        const float4* ww = reinterpret_cast<const float4*>(MW + (rp >> 8)) + lane;
        const int nb = rp & 15, wd = (rp >> 4) & 15;
        v2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#define F3_BATCH(i)                                                                  \
        if (nb > (i)) {                                                              \
          const float4 c = ww[64 * (i)];                                             \
          const float4 q = pp[(i)];                                                  \
          a0 = v2{q.x, q.y} * v2{c.x, c.y} + a0; a1 = v2{q.z, q.w} * v2{c.z, c.w} + a1; \
        }
        F3_BATCH(0) F3_BATCH(1) F3_BATCH(2) F3_BATCH(3) F3_BATCH(4) F3_BATCH(5) F3_BATCH(6) F3_BATCH(7)
#undef F3_BATCH
        const v2 a = a0 + a1;
        float acc = a.x + a.y;
        if (wd >= 2) acc += F3_DPP(acc, 0xB1);
        if (wd >= 4) acc += F3_DPP(acc, 0x4E);
        if (wd >= 8) acc += F3_DPP(acc, 0x141);
        const float L = 3.01029995663981195f * __builtin_amdgcn_logf(f3_max(acc, amin));
        if (meta & (1 << 20)) {
          const unsigned m = (meta >> 11) & 511;
          if (vF) { rowF[m] = L; lmax = f3_max(lmax, L); }
        }
      }
      }
      __builtin_amdgcn_s_setprio(0);
      }

      // ---- take in the next frame's 4 new rows (sub-block t0 + f + 2 of the clip)
      if (more) {
        if (nint) {
#pragma unroll
          for (int i = 0; i < 4; ++i) R[12 + i] = pre_row(nx0[i], nx1[i], nxp[i]);
        } else {
#pragma unroll 1
          for (int i = 0; i < 4; ++i) {
            XB[128 * i + 2 * lane] = edge_sample(jn + 128 * i + 2 * lane);
            XB[128 * i + 2 * lane + 1] = edge_sample(jn + 128 * i + 2 * lane + 1);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) R[12 + i] = E[64 * i + lane];
        }
        if constexpr (SPEC) subblock(R[12], R[13], R[14], R[15], bd, bd.t0 + f + 2);
      }
    }
    {
      float v = lmax;
      v = f3_max(v, F3_DPP(v, 0xB1)); v = f3_max(v, F3_DPP(v, 0x4E)); v = f3_max(v, F3_DPP(v, 0x141)); v = f3_max(v, F3_DPP(v, 0x140));
      const int vi = __float_as_int(v);
      const float r0 = __int_as_float(__builtin_amdgcn_readlane(vi, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(vi, 16));
      const float r2 = __int_as_float(__builtin_amdgcn_readlane(vi, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(vi, 48));
      const float mx = fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
      if constexpr (DESC) { (void)mx; }
      else if constexpr (SPEC) { if (lane == 0) blockmax[b] = mx; }
      else { if (lane == 0 && mx > -INFINITY) atomicMax(&info[bd.clip].lmax_ord, f3_ord(mx)); }
    }
    if (!chain) break;
    first = false;
    ++b;
    bd = blocks[b];
    }
  }
  } while (f3_runs_next(runs, work_ctr, nblocks, (int)(gridDim.x * WAVES), lane));
}

int frames3s_waves(const F3Tables& ft) {
  const int forced = dev_env().f3_waves;
  if (forced == 12 || forced == 16) return frames3s_lds_bytes(forced, ft) <= 160 * 1024 ? forced : 12;
  return frames3s_lds_bytes(16, ft) <= 160 * 1024 ? 16 : 12;
}

template <int FMT, int WAVES, bool SPEC, int NBS, int WDS>
static hipError_t launch_frames3s_t(hipStream_t s, const void* samples, ClipInfo* info, const BlockDesc* blocks,
                                    int nblocks, const int* nblocks_dev, const F3Tables& ft, const KParams& kp,
                                    float* logmel, float* blockmax, float* bsum, int* work_ctr, int n_cu) {
  static bool attr_set[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_frames3s<FMT, WAVES, SPEC, false, NBS, WDS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  const int grid = std::max(1, std::min(n_cu, (nblocks + WAVES - 1) / WAVES));
  hipLaunchKernelGGL((k_frames3s<FMT, WAVES, SPEC, false, NBS, WDS>), dim3(grid), dim3(WAVES * 64), frames3s_lds_bytes(WAVES, ft), s,
                     samples, info, blocks, nblocks, nblocks_dev, ft, kp, logmel, blockmax, bsum, nullptr, nullptr, SpecBands{}, work_ctr);
  return hipGetLastError();
}

template <int FMT>
static hipError_t launch_spectral_t(hipStream_t s, const void* samples, ClipInfo* info, const BlockDesc* blocks, int nblocks,
                                    const F3Tables& ft, const KParams& kp, float* desc_out, const int64_t* desc_offs,
                                    const SpecBands& sb, int n_cu) {
  static bool attr_set[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_frames3s<FMT, 12, false, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  const int grid = std::max(1, std::min(n_cu, (nblocks + 11) / 12));
  hipLaunchKernelGGL((k_frames3s<FMT, 12, false, true>), dim3(grid), dim3(12 * 64), frames3s_lds_bytes(12, ft), s,
                     samples, info, blocks, nblocks, nullptr, ft, kp, nullptr, nullptr, nullptr, desc_out, desc_offs, sb, nullptr);
  return hipGetLastError();
}

hipError_t launch_spectral(hipStream_t s, const void* samples, ClipInfo* info, const BlockDesc* blocks, int nblocks,
                           const F3Tables& ft, const KParams& kp, float* desc_out, const int64_t* desc_offs,
                           const SpecBands& sb, int n_cu) {
  if (kp.fmt == AFX_FMT_S16) return launch_spectral_t<AFX_FMT_S16>(s, samples, info, blocks, nblocks, ft, kp, desc_out, desc_offs, sb, n_cu);
  return launch_spectral_t<AFX_FMT_F32>(s, samples, info, blocks, nblocks, ft, kp, desc_out, desc_offs, sb, n_cu);
}

hipError_t launch_frames3s(hipStream_t s, const void* samples, ClipInfo* info, const BlockDesc* blocks, int nblocks,
                           const int* nblocks_dev, const F3Tables& ft, const KParams& kp, float* logmel,
                           float* blockmax, float* bsum, bool spec, int* work_ctr, int n_cu) {
  const int waves = frames3s_waves(ft);
  // the compiled-in schedule: batches 3, 1, 5, 5 at widths 1, 4, 2, 4, weights packed round after round; in a width-1 round
  // every lane must own a filter (the straight-line code stores without an owner test there)
  bool fixed = ft.mel_rounds == 4 && !dev_env().f3_generic_mel;
  const int want_nb[4] = {3, 1, 5, 5}, want_wd[4] = {1, 4, 2, 4};
  int woff = 0;
  for (int r = 0; r < 4 && fixed; ++r) {
    fixed = (int)(ft.mel_rp[r] & 15) == want_nb[r] && (int)((ft.mel_rp[r] >> 4) & 15) == want_wd[r] && (int)(ft.mel_rp[r] >> 8) == woff;
    woff += want_nb[r] * 256;
  }
  fixed = fixed && (ft.mel_own_w1 != 0);
#define AFX_F3S_GO2(FMT, W, NBS, WDS)                                                                                                    \
  (spec ? launch_frames3s_t<FMT, W, true, NBS, WDS>(s, samples, info, blocks, nblocks, nblocks_dev, ft, kp, logmel, blockmax, bsum, work_ctr, n_cu) \
        : launch_frames3s_t<FMT, W, false, NBS, WDS>(s, samples, info, blocks, nblocks, nblocks_dev, ft, kp, logmel, blockmax, bsum, work_ctr, n_cu))
#define AFX_F3S_GO(FMT, W) (fixed ? AFX_F3S_GO2(FMT, W, 0x5513, 0x4241) : AFX_F3S_GO2(FMT, W, 0, 0))
  if (kp.fmt == AFX_FMT_S16) return waves == 16 ? AFX_F3S_GO(AFX_FMT_S16, 16) : AFX_F3S_GO(AFX_FMT_S16, 12);
  return waves == 16 ? AFX_F3S_GO(AFX_FMT_F32, 16) : AFX_F3S_GO(AFX_FMT_F32, 12);
#undef AFX_F3S_GO
#undef AFX_F3S_GO2
}

}  // namespace afx
