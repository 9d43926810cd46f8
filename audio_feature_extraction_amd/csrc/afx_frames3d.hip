// k_frames3d: the wave-level frame kernel for n_fft = 512 / hop = 128 (BASELINE.json configs[2], the 16 kHz speech
// configuration of 04_feature_extraction_experiment/feature_extraction.py:35-41; librosa stft / mel / power_to_db
// semantics as in oracle/cpu_ref.py).  Same design as k_frames3 (afx_frames3.hip), with TWO frame pairs per wave:
//   * a 512-point complex FFT (two real frames, z = w yA + i w yB) keeps 16 points per lane on 32 lanes, so each
//     half-wave runs its own pair: half 0 frames (4i, 4i+1), half 1 frames (4i+2, 4i+3) of a 16-frame block -- four
//     frames per pass of the instruction stream, each half on its own half of the wave's LDS image;
//   * schedule 16 x 8 x 4: radix 16 over u (lane l' holds points l' + 32 u), two radix-8 butterflies (a = (l'>>4) + 2i,
//     twiddle W_128^(k1 r)), four radix-4 butterflies per lane in the last pass -- j = l', 128 - l', l' + 32, 96 - l' --
//     so that, as in k_frames3, the lane that owns Z[k] also owns Z[512 - k] and X_A, X_B follow with adds only;
//   * hop = 128 = 4 rows of 32 samples: frame B is frame A shifted by 4 rows (rows live as (A, B) pairs), the next
//     pass of a half starts 16 rows later -- all 20 rows of a pair are fetched per pass (the 4 shared ones hit L1);
//   * mel: the 32 lanes of a half walk the schedule of afx_tables.cpp build_f3_mel(lanes = 32) on their own spectrum.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "afx_device.h"
#include "afx_devenv.h"
#include "afx_frames3.h"
#include "afx_frames3_dev.h"

namespace afx {

constexpr int kF3dTabFloats = 256 + 768 + 512;      // pass-2 twiddles 128 float2, last-pass 3 x 4 x 32 float2, window 8 x 32 float2

size_t frames3d_lds_bytes(int waves, const F3Tables& ft) {
  return (size_t)(waves * kF3ExFloats + kF3dTabFloats + ft.mel_wfloats + ft.mel_rounds * 64) * sizeof(float);
}

// NBS > 0: the mel schedule is known at compile time -- up to four rounds of width 1 in which every lane owns a filter, their
// batch counts the nibbles of NBS from the first round up (the reference's 16 kHz / 128 mels plans 1, 1, 2, 4: NBS 0x4211) --
// and is walked as straight-line code, every read of a round in flight before its first FMA.  NBS = 0: any schedule.
template <int FMT, int WAVES, bool SPEC, int NBS>
__global__ __launch_bounds__(WAVES * 64) void k_frames3d(const void* __restrict__ samples,
                                                         ClipInfo* __restrict__ info,
                                                         const BlockDesc* __restrict__ blocks, int nblocks,
                                                         const int* __restrict__ nblocks_dev,
                                                         F3Tables ft, KParams kp,
                                                         float* __restrict__ logmel,
                                                         float* __restrict__ blockmax,
                                                         float* __restrict__ bsum, int* __restrict__ work_ctr) {
  constexpr int N = 512, HOP = 128;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (nblocks_dev) {                       // the list launch: an empty list (nothing was trimmed) costs no table set-up
    nblocks = *nblocks_dev;
    if (nblocks <= 0) return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lp = lane & 31, half = lane >> 5;
  float* const tabs = smem + WAVES * kF3ExFloats;
  v2* const T2 = reinterpret_cast<v2*>(tabs);                 // [r][16]: W_128^(c r)
  v2* const T3 = reinterpret_cast<v2*>(tabs + 256);           // [a-1][kind][l']: W_512^(a j_kind), kinds ja0, jb0, ja1, jb1
  v2* const WT = reinterpret_cast<v2*>(tabs + 256 + 768);     // [u/2][l']: (w[l' + 32 u], w[l' + 32 (u+1)]) x 0.5
  float* const MW = tabs + kF3dTabFloats;
  int* const MM = reinterpret_cast<int*>(MW + ft.mel_wfloats);
  v2* const Ew = reinterpret_cast<v2*>(smem + wave * kF3ExFloats);
  v2* const E = Ew + half * (kF3ExFloats / 4);                // this half's image: 544 float2
  float* const XBw = reinterpret_cast<float*>(Ew);

  {
    const v2* w512 = reinterpret_cast<const v2*>(ft.w1024);        // here: exp(-2 pi i k / 512), k < 256
    auto W = [&](int m) { const v2 v = w512[m & 255]; return (m & 256) ? -v : v; };
    if (tid < 128) T2[tid] = W(4 * (tid >> 4) * (tid & 15));       // W_128^(c r) = W_512^(4 c r)
    if (tid < 32) {
      const int j[4] = {tid, tid ? 128 - tid : 64, tid + 32, 96 - tid};
#pragma unroll
      for (int a = 1; a < 4; ++a)
#pragma unroll
        for (int k = 0; k < 4; ++k) T3[((a - 1) * 4 + k) * 32 + tid] = W(a * j[k]);
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        auto wn = [&](int u) { const int n = tid + 32 * u; return 0.5f * ft.window[n < N ? n : 0]; };
        WT[v * 32 + tid] = v2{wn(2 * v), wn(2 * v + 1)};
      }
    }
    for (int i = tid; i < ft.mel_wfloats; i += WAVES * 64) MW[i] = ft.mel_w[i];
    for (int i = tid; i < ft.mel_rounds * 64; i += WAVES * 64) MM[i] = ft.mel_meta[i];
    for (int i = lane; i < kF3ExFloats; i += 64) XBw[i] = 0.f;
  }
  __syncthreads();

  const v2 H = {0.70710678118654752440f, 0.70710678118654752440f};
  const v2 W1 = {0.92387953251128675613f, -0.38268343236508977173f};      // W16^1
  const v2 W3 = {0.38268343236508977173f, -0.92387953251128675613f};      // W16^3
  const bool pre = (kp.flags & AFX_FLAG_PREEMPH) != 0;
  const float b1 = kp.preemph_b1;
  const int M = kp.n_mels;
  const int jb0 = lp ? 128 - lp : 64, ja1 = lp + 32, jb1 = 96 - lp;
  // exchange-1 image [k1][34] (transposed, row stride 34 = 2 mod 32): a lane stores its value k at row k, column = its own
  // index -- 16 consecutive lanes, consecutive slots: conflict-free -- and reads row k1 = lp & 15 at column (lp >> 4) + 2 u:
  // the two 16-lane quarters of a 32-lane read group differ by one column, i.e. fall on the even / the odd slots of
  // 2 k1 + column -- conflict-free as well (round 2's [l][17] image had one 2-way conflict per read: 16 LDS cycles a pass)
  v2* const e1w = E + lp;
  const v2* const e1r = E + 34 * (lp & 15) + (lp >> 4);
  v2* const e2w = E + 128 * (lp >> 4) + (lp & 15);

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

  // sums of squares of four rows over the wave; `upper`: the upper half-wave's total (its rows are the pass's new sub-blocks)
  auto rowsum4 = [&](float r0, float r1, float r2, float r3, bool upper, bool& bad) -> float {
    float q = r0 * r0; q = fmaf(r1, r1, q); q = fmaf(r2, r2, q); q = fmaf(r3, r3, q);
    bad = !(isfinite(r0) && isfinite(r1) && isfinite(r2) && isfinite(r3));
    q += F3_DPP(q, 0xB1); q += F3_DPP(q, 0x4E); q += F3_DPP(q, 0x141); q += F3_DPP(q, 0x140);
    const int qi = __float_as_int(q);
    return upper ? __int_as_float(__builtin_amdgcn_readlane(qi, 32)) + __int_as_float(__builtin_amdgcn_readlane(qi, 48))
                 : __int_as_float(__builtin_amdgcn_readlane(qi, 0)) + __int_as_float(__builtin_amdgcn_readlane(qi, 16));
  };
  auto put_sum = [&](float t, bool bad, bool upper, const BlockDesc& bd, int j) {
    if (j >= 0 && j < bd.pad_[1]) {
      if (lane == 0) bsum[bd.pad_[0] + j] = t;
      if (!(fabsf(t) < INFINITY)) {
        const bool mine = bad && ((lane >> 5) == (upper ? 1 : 0));
        if (__any(mine) && lane == 0) atomicOr(&info[bd.clip].nonfinite, 1u);
      }
    }
  };

  const int total_waves = gridDim.x * WAVES;
  F3Runs runs = f3_runs_init(work_ctr, nblocks, total_waves, blockIdx.x * WAVES + wave, false);
  do {
  for (int b = runs.b_lo; b < runs.b_hi; b += runs.stride) {
    const BlockDesc bd = blocks[b];
    if (!bd.active) continue;
    const int Tleft = bd.T - bd.t0;
    const int nit = Tleft >= 16 ? 4 : (Tleft + 3) >> 2;
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
    float lmax = -INFINITY;
    float* const tile = logmel + bd.frame_slot * (int64_t)M;     // [frame][mel]

#pragma unroll 1
    for (int it = 0; it < nit; ++it) {
      // ---- rows of this half's pair: staged samples [512 it + 256 half, + 640), 20 rows of 32
      const int j0 = 512 * it;
      float rows[20];
      if (interior(j0, j0 + 256 + N + HOP)) {
        float y[20], yp[20];
#pragma unroll
        for (int u = 0; u < 20; ++u) {
          y[u] = row_ld(sp + j0, 256 * half + 32 * u + lp); yp[u] = row_ld(sp + j0 - 1, 256 * half + 32 * u + lp);
        }
#pragma unroll
        for (int u = 0; u < 20; ++u) rows[u] = pre ? f3_pre1(y[u], yp[u], b1) : y[u];
      } else {
#pragma unroll 1
        for (int u = 0; u < 20; ++u) XBw[640 * half + 32 * u + lp] = edge_sample(j0 + 256 * half + 32 * u + lp);
#pragma unroll
        for (int u = 0; u < 20; ++u) rows[u] = XBw[640 * half + 32 * u + lp];
      }
      if constexpr (SPEC) {
        // the upper half's rows 4..19 are sub-blocks g + 1 .. g + 4 of the clip (g = t0 + 4 it, 128 samples each); the
        // clip's first block also owns sub-block 0 = the lower half's rows 8..11
        const int g = bd.t0 + 4 * it;
        bool bad;
        if (g == 0) { const float t = rowsum4(rows[8], rows[9], rows[10], rows[11], false, bad); put_sum(t, bad, false, bd, 0); }
        // the four new sub-blocks at once: per-lane partials q[k], then a butterfly over the 32 lanes of the upper half that
        // halves the number of live values at every step (lane & 1 picks q0 | q1 and q2 | q3, lane & 2 picks between those),
        // so that lane 32 + k ends with the total of sub-block g + 1 + k -- 14 vector instructions and one store where four
        // separate wave reductions (4 DPP steps, two readlanes, a scalar store each) were 60
        float q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          q[k] = rows[4 + 4 * k] * rows[4 + 4 * k];
          q[k] = fmaf(rows[5 + 4 * k], rows[5 + 4 * k], q[k]); q[k] = fmaf(rows[6 + 4 * k], rows[6 + 4 * k], q[k]);
          q[k] = fmaf(rows[7 + 4 * k], rows[7 + 4 * k], q[k]);
        }
        const unsigned long long odd = 0xAAAAAAAAAAAAAAAAull, hi2 = 0xCCCCCCCCCCCCCCCCull;
        float s01 = f3_sel(q[0], q[1], odd) + F3_DPP(f3_sel(q[1], q[0], odd), 0xB1);       // lane ^ 1
        float s23 = f3_sel(q[2], q[3], odd) + F3_DPP(f3_sel(q[3], q[2], odd), 0xB1);
        float sq = f3_sel(s01, s23, hi2) + F3_DPP(f3_sel(s23, s01, hi2), 0x4E);            // lane ^ 2
        sq += F3_DPP(sq, 0x124); sq += F3_DPP(sq, 0x128);                                   // row_ror 4, 8: the four quads of a row
        sq = f3_add_xor16(sq);                                                              // the half's two rows
        const int jk = g + 1 + (lane & 3);
        if (lane >= 32 && lane < 36 && jk < bd.pad_[1]) bsum[bd.pad_[0] + jk] = sq;
        if (__any(lane >= 32 && !(fabsf(sq) < INFINITY))) {       // rare: a sum that is not finite -- is it the samples?
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const bool badk = !(isfinite(rows[4 + 4 * k]) && isfinite(rows[5 + 4 * k]) && isfinite(rows[6 + 4 * k]) && isfinite(rows[7 + 4 * k]));
            if (g + 1 + k < bd.pad_[1] && __any(badk && lane >= 32) && lane == 0) atomicOr(&info[bd.clip].nonfinite, 1u);
          }
        }
      }
      // ---- z = w yA + i w yB
      v2 z[16];
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        const v2 w = ldv(WT + v * 32 + lp);
        z[2 * v] = v2{rows[2 * v], rows[2 * v + 4]} * v2{w.x, w.x};
        z[2 * v + 1] = v2{rows[2 * v + 1], rows[2 * v + 5]} * v2{w.y, w.y};
      }
      // ---- pass 1 + exchange 1 (this half's image)
      f3_dft16(z, H, W1, W3);
#pragma unroll
      for (int k = 0; k < 16; ++k) stv(e1w + 34 * k, z[k]);       // unmerged 8-byte stores: 2 x 6 LDS cycles against 13 for ds_write2_b64
#pragma unroll
      for (int u = 0; u < 16; ++u) z[u] = ldv(e1r + 2 * u);
      // ---- pass 2 + exchange 2
      {
        v2 tw[8];
#pragma unroll
        for (int r = 1; r < 8; ++r) tw[r] = ldv(T2 + r * 16 + (lp & 15));
        v2 xa[8], xb[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { xa[r] = z[2 * r]; xb[r] = z[2 * r + 1]; }
#pragma unroll
        for (int r = 1; r < 8; ++r) cmul2(xa[r], tw[r], xb[r], tw[r]);
        f3_dft8(xa, H); f3_dft8(xb, H);
#pragma unroll
        for (int r = 0; r < 8; ++r) { stv(e2w + 16 * r, xa[r]); stv(e2w + 16 * r + 256, xb[r]); }
      }
      // ---- pass 3: four radix-4 butterflies, j = l', jb0, l' + 32, 96 - l'
      v2 A0[4], B0[4], A1[4], B1[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        A0[a] = ldv(E + 128 * a + lp); B0[a] = ldv(E + 128 * a + jb0);
        A1[a] = ldv(E + 128 * a + ja1); B1[a] = ldv(E + 128 * a + jb1);
      }
#pragma unroll
      for (int a = 1; a < 4; ++a) {
        cmul2(A0[a], ldv(T3 + ((a - 1) * 4 + 0) * 32 + lp), B0[a], ldv(T3 + ((a - 1) * 4 + 1) * 32 + lp));
        cmul2(A1[a], ldv(T3 + ((a - 1) * 4 + 2) * 32 + lp), B1[a], ldv(T3 + ((a - 1) * 4 + 3) * 32 + lp));
      }
      f3_dft4(A0[0], A0[1], A0[2], A0[3]); f3_dft4(B0[0], B0[1], B0[2], B0[3]);
      f3_dft4(A1[0], A1[1], A1[2], A1[3]); f3_dft4(B1[0], B1[1], B1[2], B1[3]);
      // A0[s] = Z[l' + 128 s], B0[s] = Z[jb0 + 128 s]: mirror of A0[s] is B0[3 - s]; same for A1 / B1 (j = l' + 32, 96 - l').
      // Lane l' = 0 owns the self-mirrored butterflies 0 and 64: pairs (A0[s], A0[4 - s]) and (B0[s], B0[3 - s]).
      const v2 nyq = A0[2];
      if (lp == 0) { const v2 t2 = B0[2], t3 = B0[3]; B0[2] = A0[3]; B0[3] = A0[0]; A0[2] = t2; A0[3] = t3; }
      // ---- |X_A|^2, |X_B|^2 -> this half's image as PB[bin] = (A, B), bins 0..256
      {
        v2 p0, p1;
        sqsum2(A0[0] + B0[3], A0[0] - B0[3], A0[1] + B0[2], A0[1] - B0[2], p0, p1);
        E[lp] = p0; E[lp + 128] = p1;
        sqsum2(A0[2] + B0[1], A0[2] - B0[1], A0[3] + B0[0], A0[3] - B0[0], p0, p1);
        E[jb0 + 128] = p0; E[jb0] = p1;
        sqsum2(A1[0] + B1[3], A1[0] - B1[3], A1[1] + B1[2], A1[1] - B1[2], p0, p1);
        E[ja1] = p0; E[ja1 + 128] = p1;
        sqsum2(A1[2] + B1[1], A1[2] - B1[1], A1[3] + B1[0], A1[3] - B1[0], p0, p1);
        E[jb1 + 128] = p0; E[jb1] = p1;
        if (lp == 0) E[256] = v2{4.f * nyq.x * nyq.x, 4.f * nyq.y * nyq.y};
      }
      // ---- mel + dB: frames 4 it + 2 half, + 1
      const int fA = 4 * it + 2 * half;
      const bool vA = fA < Tleft, vB = fA + 1 < Tleft;
      float* const rowA = tile + (unsigned)(fA * M);
      __builtin_amdgcn_s_setprio(1);          // a wave in its mel phase (LDS reads) goes ahead of its SIMD mates' FFTs: k_frames3
      if constexpr (NBS > 0) {
        int woff = 0;
#pragma unroll
        for (int rd = 0; rd < 4; ++rd) {
          constexpr int kNbs = NBS;
          const int nb = (kNbs >> (4 * rd)) & 15;
          if (nb == 0) break;
          const int meta = MM[rd * 64 + lane];
          const float4* pp = reinterpret_cast<const float4*>(E + (meta & 2047));
          const float4* ww = reinterpret_cast<const float4*>(MW + woff) + lane;
          woff += nb * 256;
          v2 sa[4];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (i < nb) {
              const float4 c = ww[64 * i], q0 = pp[2 * i], q1 = pp[2 * i + 1];
              const v2 c01 = {c.x, c.y}, c23 = {c.z, c.w};
              if (i == 0) {
                sa[0] = f3_mel_mul_lo(v2{q0.x, q0.y}, c01); sa[1] = f3_mel_mul_hi(v2{q0.z, q0.w}, c01);
                sa[2] = f3_mel_mul_lo(v2{q1.x, q1.y}, c23); sa[3] = f3_mel_mul_hi(v2{q1.z, q1.w}, c23);
              } else {
                sa[0] = f3_mel_fma_lo(v2{q0.x, q0.y}, c01, sa[0]); sa[1] = f3_mel_fma_hi(v2{q0.z, q0.w}, c01, sa[1]);
                sa[2] = f3_mel_fma_lo(v2{q1.x, q1.y}, c23, sa[2]); sa[3] = f3_mel_fma_hi(v2{q1.z, q1.w}, c23, sa[3]);
              }
            }
          }
          const v2 acc = (sa[0] + sa[1]) + (sa[2] + sa[3]);
          const float L0 = 3.01029995663981195f * __builtin_amdgcn_logf(f3_max(acc.x, amin));
          const float L1 = 3.01029995663981195f * __builtin_amdgcn_logf(f3_max(acc.y, amin));
          const unsigned m = (meta >> 11) & 511;          // every lane owns its filter (launch_frames3d checks)
          if (vA) { rowA[m] = L0; lmax = f3_max(lmax, L0); }
          if (vB) { rowA[M + m] = L1; lmax = f3_max(lmax, L1); }
        }
      } else {
#pragma unroll 1
      for (int rd = 0; rd < n_rounds; ++rd) {
        const uint32_t rp = ft.mel_rp[rd];
        const int meta = MM[rd * 64 + lane];
        const float4* pp = reinterpret_cast<const float4*>(E + (meta & 2047));
        const float4* ww = reinterpret_cast<const float4*>(MW + (rp >> 8)) + lane;
        const int nb = rp & 15, wd = (rp >> 4) & 15;
        v2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#define F3_BATCH(i)                                                                  \
        if (nb > (i)) {                                                              \
          const float4 c = ww[64 * (i)];                                             \
          const float4 q0 = pp[2 * (i)], q1 = pp[2 * (i) + 1];                       \
          a0 = v2{q0.x, q0.y} * v2{c.x, c.x} + a0; a1 = v2{q0.z, q0.w} * v2{c.y, c.y} + a1; \
          a0 = v2{q1.x, q1.y} * v2{c.z, c.z} + a0; a1 = v2{q1.z, q1.w} * v2{c.w, c.w} + a1; \
        }
        F3_BATCH(0) F3_BATCH(1) F3_BATCH(2) F3_BATCH(3) F3_BATCH(4) F3_BATCH(5) F3_BATCH(6) F3_BATCH(7)
#undef F3_BATCH
        v2 acc = a0 + a1;
        if (wd >= 2) { acc.x += F3_DPP(acc.x, 0xB1); acc.y += F3_DPP(acc.y, 0xB1); }
        if (wd >= 4) { acc.x += F3_DPP(acc.x, 0x4E); acc.y += F3_DPP(acc.y, 0x4E); }
        if (wd >= 8) { acc.x += F3_DPP(acc.x, 0x141); acc.y += F3_DPP(acc.y, 0x141); }
        const float L0 = 3.01029995663981195f * __builtin_amdgcn_logf(f3_max(acc.x, amin));
        const float L1 = 3.01029995663981195f * __builtin_amdgcn_logf(f3_max(acc.y, amin));
        if (meta & (1 << 20)) {
          const unsigned m = (meta >> 11) & 511;
          if (vA) { rowA[m] = L0; lmax = f3_max(lmax, L0); }
          if (vB) { rowA[M + m] = L1; lmax = f3_max(lmax, L1); }
        }
      }
      }
      __builtin_amdgcn_s_setprio(0);
    }
    {
      float v = lmax;
      v = f3_max(v, F3_DPP(v, 0xB1)); v = f3_max(v, F3_DPP(v, 0x4E)); v = f3_max(v, F3_DPP(v, 0x141)); v = f3_max(v, F3_DPP(v, 0x140));
      const int vi = __float_as_int(v);
      const float r0 = __int_as_float(__builtin_amdgcn_readlane(vi, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(vi, 16));
      const float r2 = __int_as_float(__builtin_amdgcn_readlane(vi, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(vi, 48));
      const float mx = fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
      if constexpr (SPEC) { if (lane == 0) blockmax[b] = mx; }
      else { if (lane == 0 && mx > -INFINITY) atomicMax(&info[bd.clip].lmax_ord, f3_ord(mx)); }
    }
  }
  } while (f3_runs_next(runs, work_ctr, nblocks, (int)(gridDim.x * WAVES), lane));
}

int frames3d_waves(const F3Tables& ft) {
  const int forced = dev_env().f3_waves;
  if (forced == 12 || forced == 16) return frames3d_lds_bytes(forced, ft) <= 160 * 1024 ? forced : 12;
  return frames3d_lds_bytes(16, ft) <= 160 * 1024 ? 16 : 12;
}

template <int FMT, int WAVES, bool SPEC, int NBS>
static hipError_t launch_frames3d_t(hipStream_t s, const void* samples, ClipInfo* info, const BlockDesc* blocks,
                                    int nblocks, const int* nblocks_dev, const F3Tables& ft, const KParams& kp,
                                    float* logmel, float* blockmax, float* bsum, int* work_ctr, int n_cu) {
  static bool attr_set[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_frames3d<FMT, WAVES, SPEC, NBS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  const int grid = std::max(1, std::min(n_cu, (nblocks + WAVES - 1) / WAVES));
  hipLaunchKernelGGL((k_frames3d<FMT, WAVES, SPEC, NBS>), dim3(grid), dim3(WAVES * 64), frames3d_lds_bytes(WAVES, ft), s,
                     samples, info, blocks, nblocks, nblocks_dev, ft, kp, logmel, blockmax, bsum, work_ctr);
  return hipGetLastError();
}

hipError_t launch_frames3d(hipStream_t s, const void* samples, ClipInfo* info, const BlockDesc* blocks, int nblocks,
                           const int* nblocks_dev, const F3Tables& ft, const KParams& kp, float* logmel,
                           float* blockmax, float* bsum, bool spec, int* work_ctr, int n_cu) {
  const int waves = frames3d_waves(ft);
  // the compiled-in schedule: rounds of width 1 with 1, 1, 2, 4 batches, every lane an owner, weights packed round after round
  bool fixed = ft.mel_rounds == 4 && ft.mel_all_own && !dev_env().f3_generic_mel;
  const int want_nb[4] = {1, 1, 2, 4};
  int woff = 0;
  for (int r = 0; r < 4 && fixed; ++r) {
    fixed = (int)(ft.mel_rp[r] & 15) == want_nb[r] && ((ft.mel_rp[r] >> 4) & 15) == 1 && (int)(ft.mel_rp[r] >> 8) == woff;
    woff += want_nb[r] * 256;
  }
#define AFX_F3D_GO2(FMT, W, NBS)                                                                                                    \
  (spec ? launch_frames3d_t<FMT, W, true, NBS>(s, samples, info, blocks, nblocks, nblocks_dev, ft, kp, logmel, blockmax, bsum, work_ctr, n_cu) \
        : launch_frames3d_t<FMT, W, false, NBS>(s, samples, info, blocks, nblocks, nblocks_dev, ft, kp, logmel, blockmax, bsum, work_ctr, n_cu))
#define AFX_F3D_GO(FMT, W) (fixed ? AFX_F3D_GO2(FMT, W, 0x4211) : AFX_F3D_GO2(FMT, W, 0))
  if (kp.fmt == AFX_FMT_S16) return waves == 16 ? AFX_F3D_GO(AFX_FMT_S16, 16) : AFX_F3D_GO(AFX_FMT_S16, 12);
  return waves == 16 ? AFX_F3D_GO(AFX_FMT_F32, 16) : AFX_F3D_GO(AFX_FMT_F32, 12);
#undef AFX_F3D_GO
#undef AFX_F3D_GO2
}

}  // namespace afx
