// k_frames3: the wave-autonomous frame kernel of libafx.so (gfx950) for n_fft = 1024 / hop = 256
// (reference call sites: audio_feature_extraction_toolkit/core/feature_extractor.py:127-134 -> librosa stft,
// |.|^2, filters.mel, power_to_db; the framing, window and mel semantics are those of oracle/cpu_ref.py).
//
// Why a third frame kernel.  tools/micro/valu_rate.hip measured on MI355X that ONE wave issues a vector
// instruction every ~8 cycles whatever the instruction, and that a SIMD only approaches its packed-f32 rate with
// four waves resident (4.7 / 3.7 / 3.3 cycles per v_pk_fma_f32 at 2 / 3 / 4 waves).  k_frames2 runs two waves per
// SIMD (204 VGPRs, 78 KB of LDS per 4-wave workgroup, two workgroup barriers per 16 frames) and was measured at
// ~86 % of that two-wave issue limit: it cannot get faster without more waves or fewer instructions.  This kernel
// does both:
//   * every wave is autonomous -- it takes runs of 16-frame blocks (a contiguous share up front, then tickets:
//     f3_runs_* in afx_frames3_dev.h), walks a block as 8 frame pairs, carries its rows from one block of a clip into
//     the next, and never meets a workgroup barrier after the table set-up.  Per wave the LDS holds one 8.5 KB image that serves, in turn, the
//     two FFT exchanges, the pair's power spectrum and the edge-row staging; 12 or 16 waves share one set of
//     twiddle / mel tables (one workgroup per CU);
//   * the butterflies are written as packed-f32 instructions with VOP3P op_sel / neg modifiers (inline asm): a
//     multiplication by +-i or the swap a complex product needs is a source modifier, not a v_mov pair -- the
//     compiler's own packing of k_frames2 spent 14 % of its instructions on such moves;
//   * sample rows (lane l holds samples l + 64 u) are loaded straight from global memory as dwords, one row per
//     instruction, together with their left neighbours (pre-emphasis), 8 new rows per pair; nothing is re-cut
//     through LDS;
//   * mel: per pair, lanes map to filters through a host-built schedule of "rounds" (width 1, 2, 4 or 8 lanes per
//     filter, afx_tables.cpp: build_f3_mel) -- packed FMAs on (frame A, frame B), quad / half-row DPP reductions;
//   * the log-mel spill is frame-major ([frame][mel], 512 B per frame at 128 mels) so that a pair's values leave
//     as whole 512-byte rows; k_dct16<.., true> reads that layout.
//
// FFT schedule (unchanged mathematics from k_frames2): z = w*yA + i*w*yB, 1024 points as 16 x 8 x 8,
//   pass 1  radix 16 over u (lane l holds points l + 64 u)                 -> Y[l][k1]
//   xchg 1  image [k1][66] (transposed; base + immediate on both sides, conflict-free both ways), lane reads l_src = (l>>4) + 4u, k1 = l & 15
//   pass 2  two radix-8 butterflies (a = (l>>4) + 4i, over r: l_src = a + 8r), twiddle W_128^(k1 r)
//   xchg 2  image [a][128] at j = k1 + 16 k2, plain layout (both sides conflict-free)
//   pass 3  radix 8 over a for butterflies ja = l and jb = 128 - l (lane 0: 0 and 64), twiddle W_1024^(a j):
//           the lane then owns Z[k] and Z[N-k], so X_A, X_B follow with adds only.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "afx_device.h"
#include "afx_devenv.h"
#include "afx_frames3.h"
#include "afx_frames3_dev.h"

namespace afx {

size_t frames3_lds_bytes(int waves, const F3Tables& ft) {
  return (size_t)(waves * kF3ExFloats + kF3TabFloats + ft.mel_wfloats + ft.mel_rounds * 64 + 1024) * sizeof(float);
}

// NB0, NB1 > 0: the mel schedule is known at compile time to be two rounds of width 1 with NB0 and NB1 batches (the
// reference's 22050 Hz / 128 mels is 2 and 7): the tap walk is then straight-line code, every LDS read of a round in
// flight before its first FMA.  NB0 = 0: any schedule, batches behind uniform branches.
// SPEC (the first launch of a batch, "speculative"): blocks are the host-built absolute 16-frame blocks of every
// clip, computed BEFORE the trim decision -- frames sit on the absolute hop grid because librosa.effects.trim cuts at
// multiples of its own hop (512), so an interior frame does not depend on where the cut falls.  The kernel then also
// emits what the trim decision needs, from the pre-emphasised rows it holds anyway: the sum of squares of every
// 256-sample sub-block (bsum) and the non-finite flag; and, instead of the clip-wide atomic maximum, one log-mel
// maximum per block (blockmax), from which k_trim_decide3 takes the clip maximum over the blocks that survive the
// cut.  !SPEC (second launch): the few blocks around a cut (device-built list, count in *nblocks_dev), frames
// recomputed with the trimmed span masked, maxima merged by atomicMax.  The samples are thus read once per batch.
#ifdef AFX_F3_DEBUG
// diagnostic build only: per wave of the speculative launch, wall clock (100 MHz) at entry, after the table set-up and at
// the end, and the hardware id (HW_ID | XCC_ID << 32)
__device__ unsigned long long g_f3_stamps[4 * 8192];
__device__ unsigned long long g_f3_blk[65536];       // wall clock at the end of block b
#endif

template <int FMT, int WAVES, int NB0, int NB1, bool SPEC>
__global__ __launch_bounds__(WAVES * 64) void k_frames3(const void* __restrict__ samples,
                                                        ClipInfo* __restrict__ info,
                                                        const BlockDesc* __restrict__ blocks, int nblocks,
                                                        const int* __restrict__ nblocks_dev,
                                                        F3Tables ft, KParams kp,
                                                        float* __restrict__ logmel,
                                                        float* __restrict__ blockmax,
                                                        float* __restrict__ bsum,
                                                        int* __restrict__ work_ctr) {
  constexpr int N = 1024, HOP = 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (nblocks_dev) {                       // the list launch: an empty list (nothing was trimmed) costs no table set-up
    nblocks = *nblocks_dev;
    if (nblocks <= 0) return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef AFX_F3_DEBUG
  const unsigned long long stamp0 = wall_clock64();
#endif
  float* const tabs = smem + WAVES * kF3ExFloats;
  // twiddles as 16-byte entries, one LDS read for two: an LDS instruction costs the SIMD ~8 cycles of issue whatever its width
  float4* const T2q = reinterpret_cast<float4*>(tabs);        // [rr][16]: (W_128^(c 2rr), W_128^(c (2rr + 1))), rr < 4
  float4* const T3q = reinterpret_cast<float4*>(tabs + 256);  // [r-1][lane]: (W_1024^(ja r), W_1024^(jb r))
  // with the schedule compiled in, every table sits at a compile-time LDS address: the lane-indexed ones (weights, window,
  // last-pass twiddles) can then share one address register (16 * lane) and differ by their immediate offsets
  const int mel_wfloats = NB0 > 0 ? (NB0 + NB1) * 256 : ft.mel_wfloats;
  const int mel_rounds = NB0 > 0 ? 2 : ft.mel_rounds;
  float* const MW = tabs + kF3TabFloats;                      // mel weights [round][batch][lane][4]
  int* const MM = reinterpret_cast<int*>(MW + mel_wfloats);   // mel meta [round][lane]
  v2* const E = reinterpret_cast<v2*>(smem + wave * kF3ExFloats);
  float* const XB = reinterpret_cast<float*>(E);

  // ---- once per workgroup: tables -> LDS, the wave's image zeroed (padded mel taps read whatever lies there)
  {
    const v2* w1024 = reinterpret_cast<const v2*>(ft.w1024);
    auto W = [&](int m) { const v2 v = w1024[m & 511]; return (m & 512) ? -v : v; };
    if (tid < 64) {
      const int rr = tid >> 4, c = tid & 15;
      const v2 a = W(8 * (2 * rr) * c), b = W(8 * (2 * rr + 1) * c);
      T2q[tid] = float4{a.x, a.y, b.x, b.y};
      const int jbt = tid ? 128 - tid : 64;
#pragma unroll
      for (int r = 1; r < 8; ++r) { const v2 ta = W(tid * r), tb = W(jbt * r); T3q[(r - 1) * 64 + tid] = float4{ta.x, ta.y, tb.x, tb.y}; }
    }
    for (int i = tid; i < mel_wfloats; i += WAVES * 64) MW[i] = ft.mel_w[i];
    for (int i = tid; i < mel_rounds * 64; i += WAVES * 64) MM[i] = ft.mel_meta[i];
    for (int i = lane; i < kF3ExFloats; i += 64) XB[i] = 0.f;
  }
  // the lane's 16 window values (w[n] = w[N - n]) x 0.5 as a 4 KB table [u / 4][lane] of quadruples, read back per frame
  // pair (four 16-byte reads): 16 registers less, so that the kernel fits 128 registers at 16 waves per CU and, at 12
  // waves, leaves the other streams' bandwidth-bound kernels (DCT 74 registers, statistics 56) room beside it
  float4* const WT = reinterpret_cast<float4*>(MM + mel_rounds * 64);
  if (tid < 64) {
    float wreg[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) wreg[u] = 0.5f * ft.window[u < 8 ? lane + 64 * u : (64 - lane) + 64 * (15 - u)];
#pragma unroll
    for (int v = 0; v < 4; ++v) WT[v * 64 + lane] = float4{wreg[4 * v], wreg[4 * v + 1], wreg[4 * v + 2], wreg[4 * v + 3]};
  }
  __syncthreads();
  const v2 H = {0.70710678118654752440f, 0.70710678118654752440f};
  const v2 W1 = {0.92387953251128675613f, -0.38268343236508977173f};      // W16^1
  const v2 W3 = {0.38268343236508977173f, -0.92387953251128675613f};      // W16^3
  const bool pre = (kp.flags & AFX_FLAG_PREEMPH) != 0;
  const float b1 = pre ? kp.preemph_b1 : 0.f;              // y + 0 * prev = y: interior rows need no select
  const int M = kp.n_mels;
  const int jb = lane ? 128 - lane : 64;
  // image slots (float2 units)
  // exchange-1 image [k1][66] (transposed, row stride 66 = 2 mod 32): a lane stores its value k at row k, column = its own
  // index -- 16 consecutive lanes, consecutive slots: conflict-free -- and reads row k1 = lane & 15 at column (lane >> 4) + 4 u:
  // the two 16-lane quarters of a 32-lane read group differ by one column, i.e. fall on the even / the odd slots of
  // 2 k1 + column -- conflict-free as well (round 2's [l][17] image had one 2-way conflict per read: 16 LDS cycles a pass)
  v2* const e1w = E + lane;
  const v2* const e1r = E + 66 * (lane & 15) + (lane >> 4);
  v2* const e2w = E + 128 * (lane >> 4) + (lane & 15);
  v2* const ea = E + lane;                 // pass-3 reads of butterfly ja; power-spectrum bins lane + 128 s
  v2* const eb = E + jb;                   // butterfly jb; bins jb + 384 - 128 (s - 4)

  auto raw_ld = [&](int64_t idx) -> float {
    if constexpr (FMT == AFX_FMT_S16) return (float)((const int16_t*)samples)[idx] * (1.0f / 32768.0f);
    else return ((const float*)samples)[idx];
  };
  // row loads of interior pairs: wave-uniform base pointer + 32-bit lane offset (global_load ... s[base] offset:imm)
  typedef typename std::conditional<FMT == AFX_FMT_S16, int16_t, float>::type sample_t;
  auto row_ld = [&](const sample_t* base, unsigned idx) -> float {
    if constexpr (FMT == AFX_FMT_S16) return (float)base[idx] * (1.0f / 32768.0f);
    else return base[idx];
  };
  const int n_rounds = mel_rounds;
  const int meta0 = MM[lane], meta1 = MM[64 + lane];       // straight-line schedule: the lane's two filters
  const unsigned mf0 = (meta0 >> 11) & 511, mf1 = (meta1 >> 11) & 511;
  const float amin = kp.amin;

  // Sums of squares of 256-sample sub-blocks of the pre-emphasised signal (four rows each), two at a time: `x` is the
  // lanes' share of an odd sub-block of the current block (index relative to its first frame), `y` of the even one after
  // it.  The totals land in lanes `idx`, `idx + 1` of `bs`, one lane per sub-block; the block stores them in one go.
  float bs = 0.f;
  unsigned long long bsmask = 0;
  // Returns true (wave-uniform) when a sum is not finite: the caller then looks at its rows (an overflowing sum of finite
  // samples is not a non-finite clip).
  auto sub2 = [&](float x, float y, int idx, unsigned long long two) -> bool {
    const float m = f3_sum2(x, y);
    const unsigned long long sel = two << idx;
    bs = f3_sel(bs, m, sel);
    bsmask |= sel;
    return __any(!(fabsf(m) < INFINITY));
  };
  auto flag_nonfinite = [&](const BlockDesc& bd, bool bad_rows) {
    if (__any(bad_rows) && lane == 0) atomicOr(&info[bd.clip].nonfinite, 1u);
  };
  auto sub_store = [&](const BlockDesc& bd) {
    const int j = bd.t0 + lane;
    if (((bsmask >> lane) & 1) && j < bd.pad_[1]) bsum[bd.pad_[0] + j] = bs;
    bsmask = 0;
  };

  // A wave takes a contiguous run of the block list.  Consecutive blocks of one clip then continue the row pipeline
  // across the block boundary (8 new rows, as inside a block) instead of re-loading 20 rows: no halo re-read between
  // the blocks of a run, no exposed load latency at a block's start.
  const int total_waves = gridDim.x * WAVES, wg = blockIdx.x * WAVES + wave;
#ifdef AFX_F3_DEBUG
  const unsigned long long stamp1 = wall_clock64();
#endif
  F3Runs runs = f3_runs_init(work_ctr, nblocks, total_waves, wg, true);     // afx_frames3_dev.h: shares, then tickets
  do {                                                    // the runs of this wave
  const int b_hi = runs.b_hi;
  for (int b = runs.b_lo; b < b_hi; ++b) {
    BlockDesc bd = blocks[b];
    if (!bd.active) continue;
    bool first = true;                                    // first block of a run: its 20 rows are loaded; later ones inherit them
    // Rows live as pairs R[u] = (row u, row u + 4): exactly the (frame A, frame B) operands of z[u], so the window multiply
    // is one packed instruction per point.  Ra = R[0..7], Rb = R[8..15]; a pair's successor shares rows 8..19, so its
    // R'[0..7] ARE Rb: the two arrays swap roles from pair to pair (the loop body below is written once and instantiated
    // for both roles) instead of being copied -- 16 register moves per pair less.
    v2 Ra[8], Rb[8];
    for (;;) {                                            // the blocks of one run
    const int Tleft = bd.T - bd.t0;                       // frames left from this block on (>= 1)
    const int npairs = Tleft >= 16 ? 8 : (Tleft + 1) >> 1;
    bool chain = false;                                   // block b + 1 is this clip's next block and this wave's
    if (SPEC && npairs == 8 && b + 1 < b_hi) {
      const BlockDesc* nx = blocks + b + 1;
      chain = nx->active && nx->clip == bd.clip && nx->t0 == bd.t0 + 16;
    }
    const int64_t sbase = bd.sample_base;
    const sample_t* const sp = (const sample_t*)samples + sbase;      // staged sample 0 (may lie before the clip: edge path)
    auto interior = [&](int j0, int j1) -> bool {         // every sample of [j0, j1) and its predecessor exists and is kept
      return (j0 - 1 >= bd.have_lo) && (j1 <= bd.have_hi) && (j0 >= bd.keep_lo) && (j1 <= bd.keep_hi);
    };
    // The paths a pair rarely takes (a clip edge, a trimmed span, a non-finite sum) read the block's record again through
    // an opaque index -- scalar loads where they are needed -- instead of pinning ten of its words in scalar registers
    // across the pair loop: the kernel sits at the scalar file's limit, and a spilled scalar register costs a vector one
    auto interior_r = [&](int j0, int j1) -> bool {
      const BlockDesc& r = blocks[f3_opaque(b)];
      return (j0 - 1 >= r.have_lo) && (j1 <= r.have_hi) && (j0 >= r.keep_lo) && (j1 <= r.keep_hi);
    };
    auto edge_sample = [&](int j) -> float {              // pre-emphasised, trim-masked sample j (clamped loads)
      const BlockDesc& bd = blocks[f3_opaque(b)];
      const int lo = bd.have_lo, hi = bd.have_hi - 1;
      const int jc = j < lo ? lo : (j > hi ? hi : j), jp = (j - 1) < lo ? lo : ((j - 1) > hi ? hi : (j - 1));
      const float y = (jc == j) ? raw_ld(sbase + jc) : 0.f;
      const float yp = (jp == j - 1) ? raw_ld(sbase + jp) : 0.f;
      float v = y;
      if (pre) {
        v = f3_pre1(y, yp, b1);
        if (j == lo) v = f3_pre0(raw_ld(bd.clip_off), raw_ld(bd.clip_off + 1));   // clip sample 0
      }
      return (j >= bd.keep_lo && j < bd.keep_hi) ? v : 0.f;
    };

    // ---- rows of the first pair: staged samples [0, 1280)
    if (first) {
      float rows[20];
      if (interior(0, N + HOP)) {
        float y[20], yp[20];
#pragma unroll
        for (int u = 0; u < 20; ++u) { y[u] = row_ld(sp, 64 * u + lane); yp[u] = row_ld(sp - 1, 64 * u + lane); }
#pragma unroll
        for (int u = 0; u < 20; ++u) rows[u] = f3_pre1(y[u], yp[u], b1);
      } else {
#pragma unroll 1
        for (int u = 0; u < 20; ++u) XB[64 * u + lane] = edge_sample(64 * u + lane);
#pragma unroll
        for (int u = 0; u < 20; ++u) rows[u] = XB[64 * u + lane];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { Ra[u] = v2{rows[u], rows[u + 4]}; Rb[u] = v2{rows[u + 8], rows[u + 12]}; }
      if constexpr (SPEC) {
        // rows 8..11 / 12..15 / 16..19 are sub-blocks t0, t0 + 1, t0 + 2 of the clip; the first belongs to the
        // previous block's last pair unless this is the clip's first block
        auto sq4 = [&](int r) { float q = rows[r] * rows[r]; q = fmaf(rows[r + 1], rows[r + 1], q); q = fmaf(rows[r + 2], rows[r + 2], q); return fmaf(rows[r + 3], rows[r + 3], q); };
        auto bad4 = [&](int r) { return !(isfinite(rows[r]) && isfinite(rows[r + 1]) && isfinite(rows[r + 2]) && isfinite(rows[r + 3])); };
        if (sub2(sq4(12), sq4(16), 1, 3ull)) flag_nonfinite(bd, bad4(12) || bad4(16));
        if (bd.t0 == 0 && sub2(0.f, sq4(8), 0, 1ull)) flag_nonfinite(bd, bad4(8));
      }
    }
    float lmax = -INFINITY;
    float* const tile = logmel + bd.frame_slot * (int64_t)M;     // [frame][mel]
    const bool blk_int = interior(0, 512 * 8 + 768 + 512);       // every row this block (and a chained successor's first pair) loads

    // One frame pair.  P = R[0..7], Q = R[8..15] of this pair; on return Q, P (in that order) are the next pair's rows.
    auto pair_body = [&](v2 (&P)[8], v2 (&Q)[8], const int p) __attribute__((always_inline)) {
      // ---- z = w yA + i w yB (frame A: rows 0..15, frame B: rows 4..19)
      v2 z[16];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float4 w = ldv4(WT + v * 64 + lane);
        const v2* const src = v < 2 ? P + 4 * v : Q + 4 * (v - 2);
        z[4 * v] = src[0] * v2{w.x, w.x}; z[4 * v + 1] = src[1] * v2{w.y, w.y};
        z[4 * v + 2] = src[2] * v2{w.z, w.z}; z[4 * v + 3] = src[3] * v2{w.w, w.w};
      }
      // the next pair shares rows 8..19 (Q) and brings 8 new ones, n[0..7] = rows 20..27 of this pair's window; they go
      // where P is (dead from here on, its 16 registers free across the FFT and the mel phase):
      //   P'[i] = (Q[4 + i].y, n[i]),  P'[4 + i] = (n[i], n[4 + i])
      const bool more = p + 1 < npairs || chain;           // the next pair may be the next block's first
      const int jn = 512 * (p + 1) + 768;                  // staged samples [jn, jn + 512)
      const bool nint = more && (blk_int || interior_r(jn, jn + 512));

      // ---- pass 1 + exchange 1
      f3_dft16(z, H, W1, W3);
      if (!F3_SKIP(0x200)) {
#pragma unroll
      for (int k = 0; k < 16; ++k) stv(e1w + 66 * k, z[k]);       // stv: two ds_write_b64 (2 x 6 cycles), not a merged ds_write2_b64 (13)
#pragma unroll
      for (int u = 0; u < 16; ++u) z[u] = ldv(e1r + 4 * u);
      }
      // ---- pass 2 + exchange 2
      {
        v2 tw[8];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const float4 t = ldv4(T2q + rr * 16 + (lane & 15));
          tw[2 * rr] = v2{t.x, t.y}; tw[2 * rr + 1] = v2{t.z, t.w};
        }
        v2 xa[8], xb[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { xa[r] = z[2 * r]; xb[r] = z[2 * r + 1]; }
#pragma unroll
        for (int r = 1; r < 8; ++r) cmul2(xa[r], tw[r], xb[r], tw[r]);
        f3_dft8(xa, H); f3_dft8(xb, H);
        if (!F3_SKIP(0x2000)) {
#pragma unroll
        for (int r = 0; r < 8; ++r) { stv(e2w + 16 * r, xa[r]); stv(e2w + 16 * r + 512, xb[r]); }
        } else {
#pragma unroll
          for (int r = 0; r < 8; ++r) { z[2 * r] = xa[r]; z[2 * r + 1] = xb[r]; }
        }
      }
      // ---- pass 3
      v2 A[8], B[8];
      if (!F3_SKIP(0x2000)) {
#pragma unroll
      for (int r = 0; r < 8; ++r) { A[r] = ldv(ea + 128 * r); B[r] = ldv(eb + 128 * r); }
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) { A[r] = z[r]; B[r] = z[r + 8]; }
      }
#pragma unroll
      for (int r = 1; r < 8; ++r) {
        const float4 t = ldv4(T3q + (r - 1) * 64 + lane);
        cmul2(A[r], v2{t.x, t.y}, B[r], v2{t.z, t.w});
      }
      f3_dft8(A, H); f3_dft8(B, H);
      // A[s] = Z[lane + 128 s], B[s] = Z[jb + 128 s]; the mirror of A[s] is B[7-s].  Lane 0 owns the self-mirrored
      // butterflies 0 and 64: its pairs are (A[s], A[8-s]) and (B[s], B[7-s]) -- re-seat its registers once so
      // that the generic pairing below yields them (bins 128 s from A, 64 + 384 - 128 (s-4) from B).
      v2 nyq = A[4];
      if (lane == 0) { A[4] = B[4]; B[4] = A[5]; A[5] = B[5]; B[5] = A[6]; A[6] = B[6]; B[6] = A[7]; A[7] = B[7]; B[7] = A[0]; }
      // ---- |X_A|^2, |X_B|^2 -> the image as PB[bin] = (A, B)
      if (!F3_SKIP(0x800)) {
#pragma unroll
      for (int s = 0; s < 4; s += 2) {
        v2 p0, p1;
        sqsum2(A[s] + B[7 - s], A[s] - B[7 - s], A[s + 1] + B[6 - s], A[s + 1] - B[6 - s], p0, p1);
        ea[128 * s] = p0; ea[128 * (s + 1)] = p1;
      }
#pragma unroll
      for (int s = 4; s < 8; s += 2) {
        v2 p0, p1;
        sqsum2(A[s] + B[7 - s], A[s] - B[7 - s], A[s + 1] + B[6 - s], A[s + 1] - B[6 - s], p0, p1);
        eb[384 - 128 * (s - 4)] = p0; eb[384 - 128 * (s - 3)] = p1;
      }
      if (lane == 0) E[512] = v2{4.f * nyq.x * nyq.x, 4.f * nyq.y * nyq.y};
      } else {
        asm volatile("" :: "v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]), "v"(A[4]), "v"(A[5]), "v"(A[6]), "v"(A[7]));
        asm volatile("" :: "v"(B[0]), "v"(B[1]), "v"(B[2]), "v"(B[3]), "v"(B[4]), "v"(B[5]), "v"(B[6]), "v"(B[7]), "v"(nyq));
      }

      // ---- the next pair's 8 new rows are fetched under the mel phase (issued here, not before the FFT: 16 registers
      // that would be live across its register peak)
      float ny[8], nyp[8];
      if (nint) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { ny[u] = row_ld(sp + jn, 64 * u + lane); nyp[u] = row_ld(sp + jn - 1, 64 * u + lane); }
      }
      // ---- mel + dB
      // The kernel waits on the LDS pipe (SQ_WAIT_INST_LDS is a quarter of a wave's life); a wave in the mel phase -- 27 reads
      // in front of 36 FMAs -- goes ahead of its three SIMD mates that are in their FFT (same-box A/B: frame kernel -2.5 %)
      __builtin_amdgcn_s_setprio(1);
      const bool vA = 2 * p < Tleft, vB = 2 * p + 1 < Tleft;
      float* const rowA = tile + (unsigned)(2 * p * M);
      if constexpr (NB0 > 0) {
        if (!F3_SKIP(0x400)) {
        const float4* pp0 = reinterpret_cast<const float4*>(E + (meta0 & 2047));
        const float4* pp1 = reinterpret_cast<const float4*>(E + (meta1 & 2047));
        const float4* ww0 = reinterpret_cast<const float4*>(MW) + lane;
        const float4* ww1 = reinterpret_cast<const float4*>(MW + NB0 * 256) + lane;
        // (A, B) bin pairs times a weight taken from either half of a register pair: the half is an op_sel modifier
        // (f3_mel_*), not a move (hipcc's own code for `q * v2{c.w, c.w}` copies c.w into an even register first)
        v2 s0[4], s1[4];
#pragma unroll
        for (int i = 0; i < NB0; ++i) {
          const float4 c = ww0[64 * i], q0 = pp0[2 * i], q1 = pp0[2 * i + 1];
          const v2 c01 = {c.x, c.y}, c23 = {c.z, c.w};
          if (i == 0) {
            s0[0] = f3_mel_mul_lo(v2{q0.x, q0.y}, c01); s0[1] = f3_mel_mul_hi(v2{q0.z, q0.w}, c01);
            s0[2] = f3_mel_mul_lo(v2{q1.x, q1.y}, c23); s0[3] = f3_mel_mul_hi(v2{q1.z, q1.w}, c23);
          } else {
            s0[0] = f3_mel_fma_lo(v2{q0.x, q0.y}, c01, s0[0]); s0[1] = f3_mel_fma_hi(v2{q0.z, q0.w}, c01, s0[1]);
            s0[2] = f3_mel_fma_lo(v2{q1.x, q1.y}, c23, s0[2]); s0[3] = f3_mel_fma_hi(v2{q1.z, q1.w}, c23, s0[3]);
          }
        }
#pragma unroll
        for (int i = 0; i < NB1; ++i) {
          const float4 c = ww1[64 * i], q0 = pp1[2 * i], q1 = pp1[2 * i + 1];
          const v2 c01 = {c.x, c.y}, c23 = {c.z, c.w};
          if (i == 0) {
            s1[0] = f3_mel_mul_lo(v2{q0.x, q0.y}, c01); s1[1] = f3_mel_mul_hi(v2{q0.z, q0.w}, c01);
            s1[2] = f3_mel_mul_lo(v2{q1.x, q1.y}, c23); s1[3] = f3_mel_mul_hi(v2{q1.z, q1.w}, c23);
          } else {
            s1[0] = f3_mel_fma_lo(v2{q0.x, q0.y}, c01, s1[0]); s1[1] = f3_mel_fma_hi(v2{q0.z, q0.w}, c01, s1[1]);
            s1[2] = f3_mel_fma_lo(v2{q1.x, q1.y}, c23, s1[2]); s1[3] = f3_mel_fma_hi(v2{q1.z, q1.w}, c23, s1[3]);
          }
        }
        const v2 m0 = (s0[0] + s0[1]) + (s0[2] + s0[3]), m1 = (s1[0] + s1[1]) + (s1[2] + s1[3]);
        const float L00 = 3.01029995663981195f * __builtin_amdgcn_logf(f3_max(m0.x, amin));
        const float L01 = 3.01029995663981195f * __builtin_amdgcn_logf(f3_max(m0.y, amin));
        const float L10 = 3.01029995663981195f * __builtin_amdgcn_logf(f3_max(m1.x, amin));
        const float L11 = 3.01029995663981195f * __builtin_amdgcn_logf(f3_max(m1.y, amin));
        // every lane owns its two filters here (launch_frames3_w checks), frame A of a pair always exists
        rowA[mf0] = L00; rowA[mf1] = L10;
        lmax = f3_max(lmax, f3_max(L00, L10));
        if (vB) {
          rowA[M + mf0] = L01; rowA[M + mf1] = L11;
          lmax = f3_max(lmax, f3_max(L01, L11));
        }
        }
      } else {
#pragma unroll 1
      for (int rd = 0; rd < (F3_SKIP(0x400) ? 0 : n_rounds); ++rd) {
        const uint32_t rp = ft.mel_rp[rd];                  // batches | width << 4 | weight offset << 8
        const int meta = MM[rd * 64 + lane];
        // taps come in fours: two 16-byte reads of (A, B) pairs (the lane's first bin is even), one 16-byte weight read
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
        if (wd >= 2) { acc.x += F3_DPP(acc.x, 0xB1); acc.y += F3_DPP(acc.y, 0xB1); }       // lane ^ 1
        if (wd >= 4) { acc.x += F3_DPP(acc.x, 0x4E); acc.y += F3_DPP(acc.y, 0x4E); }       // lane ^ 2
        if (wd >= 8) { acc.x += F3_DPP(acc.x, 0x141); acc.y += F3_DPP(acc.y, 0x141); }     // the other quad of 8
        const float L0 = 3.01029995663981195f * __builtin_amdgcn_logf(f3_max(acc.x, amin));
        const float L1 = 3.01029995663981195f * __builtin_amdgcn_logf(f3_max(acc.y, amin));
        if (meta & (1 << 20)) {                               // this lane owns filter m
          const unsigned m = (meta >> 11) & 511;
          if (vA) rowA[m] = L0;
          if (vB) rowA[M + m] = L1;
          if (vA) lmax = f3_max(lmax, L0);
          if (vB) lmax = f3_max(lmax, L1);
        }
      }
      }

      __builtin_amdgcn_s_setprio(0);
      // ---- take in the next pair's 8 new rows, straight into the registers P leaves free
      if (more) {
        float n[8];
        if (nint) {
#pragma unroll
          for (int u = 0; u < 8; ++u) n[u] = f3_pre1(ny[u], nyp[u], b1);
        } else {
          // (the rare path: its sample indices are rebuilt from an opaque copy of jn, or the compiler carries three
          // lane-indexed induction registers through every pair for it)
          const int jn_e = f3_opaque(jn);
#pragma unroll 1
          for (int u = 0; u < 8; ++u) XB[64 * u + lane] = edge_sample(jn_e + 64 * u + lane);
#pragma unroll
          for (int u = 0; u < 8; ++u) n[u] = XB[64 * u + lane];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { P[i] = v2{Q[4 + i].y, n[i]}; P[4 + i] = v2{n[i], n[4 + i]}; }
        if constexpr (SPEC) {          // rows 12..19 of pair p + 1: sub-blocks t0 + 2 (p + 1) + 1, + 2, as (x, y) = P'[4 + i]
          v2 q = P[4] * P[4]; q = P[5] * P[5] + q; q = P[6] * P[6] + q; q = P[7] * P[7] + q;
          if (sub2(q.x, q.y, 2 * p + 3, 3ull)) {
            bool bad = false;
#pragma unroll
            for (int u = 0; u < 8; ++u) bad |= !isfinite(n[u]);
            flag_nonfinite(blocks[f3_opaque(b)], bad);
          }
        }
      }
    };

#pragma unroll 1
    for (int p = 0; p < npairs; p += 2) {
      pair_body(Ra, Rb, p);
      if (p + 1 < npairs) pair_body(Rb, Ra, p + 1);       // roles swapped; eight pairs (a whole block) end where they began
    }
    // ---- clip maximum of the log-mel (power_to_db's top_db reference)
    {
      float v = lmax;
      v = f3_max(v, F3_DPP(v, 0xB1)); v = f3_max(v, F3_DPP(v, 0x4E)); v = f3_max(v, F3_DPP(v, 0x141)); v = f3_max(v, F3_DPP(v, 0x140));
      const int vi = __float_as_int(v);
      const float r0 = __int_as_float(__builtin_amdgcn_readlane(vi, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(vi, 16));
      const float r2 = __int_as_float(__builtin_amdgcn_readlane(vi, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(vi, 48));
      const float mx = fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
      const BlockDesc& be = blocks[f3_opaque(b)];          // read again: see interior_r
      if constexpr (SPEC) { if (lane == 0) blockmax[b] = mx; sub_store(be); }
      else { if (lane == 0 && mx > -INFINITY) atomicMax(&info[be.clip].lmax_ord, f3_ord(mx)); }
    }
#ifdef AFX_F3_DEBUG
    if (SPEC && lane == 0 && b < 65536) g_f3_blk[b] = wall_clock64();
#endif
    if (!chain) break;
    first = false;
    ++b;
    bd = blocks[b];
    }
  }
  } while (f3_runs_next(runs, work_ctr, nblocks, (int)(gridDim.x * WAVES), lane));
#ifdef AFX_F3_DEBUG
  if (SPEC && lane == 0 && wg < 8192) {
    g_f3_stamps[4 * wg] = stamp0; g_f3_stamps[4 * wg + 1] = stamp1; g_f3_stamps[4 * wg + 2] = wall_clock64();
    g_f3_stamps[4 * wg + 3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                              ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
  }
#endif
}

#ifdef AFX_F3_DEBUG
extern "C" __attribute__((visibility("default"))) int afx_debug_f3_stamps(unsigned long long* out, int n_waves) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_f3_stamps), sizeof(unsigned long long) * 4 * (size_t)n_waves);
}
extern "C" __attribute__((visibility("default"))) int afx_debug_f3_blocks(unsigned long long* out, int n_blocks) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_f3_blk), sizeof(unsigned long long) * (size_t)n_blocks);
}
#endif


// ---------------------------------------------------------------------------
// k_trim_decide3: the trim decision AFTER the speculative frame pass.  One workgroup per clip:
//   * librosa.effects.trim(top_db) on the sub-block sums k_frames3<SPEC> left in bsum (feature_extractor.py:72), exactly
//     as k_trim_decide does it -> [start, end), T, status; RMS rows from the same sums (:164);
//   * the clip's log-mel maximum (power_to_db's top_db reference) over the blocks the cut leaves untouched;
//   * the frames the cut does touch -- the two frames whose window crosses `start`, the two that cross `end`, and
//     the rest of their 16-frame blocks (a block maximum cannot be taken apart) -- as a list of up to-16-frame items
//     for the second k_frames3 launch.
// Frame t of the trimmed clip is absolute frame start / hop + t: later kernels read the spill at that offset.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float td3_frame_rms(const float* bs, int64_t t, int64_t nb, int half, int per, float inv_n) {
  float s = 0.f;
  for (int64_t b = (t - half) * per; b < (t + half) * per; ++b)
    if (b >= 0 && b < nb * per) s += bs[b];
  return sqrtf(s * inv_n);
}

__global__ __launch_bounds__(256) void k_trim_decide3(const ClipDesc* __restrict__ clips, ClipInfo* __restrict__ info,
                                                      const float* __restrict__ bsum, const float* __restrict__ blockmax,
                                                      BlockDesc* __restrict__ items, int* __restrict__ n_items, int max_items,
                                                      float* __restrict__ rms_rows, KParams kp) {
  __shared__ float red_f[4];
  __shared__ long long red_a[4], red_b[4];
  const int clip = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const ClipDesc cd = clips[clip];
  const int64_t N = cd.len;
  const uint32_t nonfinite = info[clip].nonfinite;
  int status = AFX_CLIP_OK;
  if (N < 2) status = AFX_CLIP_TOO_SHORT;
  else if (nonfinite) status = AFX_CLIP_NONFINITE;
  int64_t start = 0, end = N;
  const int per = kp.rms_sub;
  auto wave_maxf = [&](float v) {
    v = fmaxf(v, F3_DPP(v, 0xB1)); v = fmaxf(v, F3_DPP(v, 0x4E)); v = fmaxf(v, F3_DPP(v, 0x141)); v = fmaxf(v, F3_DPP(v, 0x140));
    const int vi = __float_as_int(v);
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(vi, 0)), __int_as_float(__builtin_amdgcn_readlane(vi, 16))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(vi, 32)), __int_as_float(__builtin_amdgcn_readlane(vi, 48))));
  };
  if ((kp.flags & AFX_FLAG_TRIM) && status == AFX_CLIP_OK) {   // uniform per workgroup
    const int th = kp.trim_hop, half = (kp.trim_frame / th) / 2;
    const int64_t nb = (N + th - 1) / th, nt = 1 + N / th;
    const float inv_n = 1.0f / (float)kp.trim_frame;
    const float* bs = bsum + cd.tblk_base * per;
    float mx = 0.f;
    for (int64_t t = tid; t < nt; t += 256) mx = fmaxf(mx, td3_frame_rms(bs, t, nb, half, per, inv_n));
    mx = wave_maxf(mx);
    if (lane == 0) red_f[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red_f[0], red_f[1]), fmaxf(red_f[2], red_f[3]));
    __syncthreads();
    const float ref_db = 10.0f * log10f(fmaxf(1e-10f, mx * mx));      // amplitude_to_db(mse, ref=np.max, amin=1e-5, top_db=None)
    long long first = (long long)1 << 62, last = -1;
    for (int64_t t = tid; t < nt; t += 256) {
      const float r = td3_frame_rms(bs, t, nb, half, per, inv_n);
      const float db = 10.0f * log10f(fmaxf(1e-10f, r * r)) - ref_db;
      if (db > -kp.trim_top_db) { if (t < first) first = t; if (t > last) last = t; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const long long f2 = __shfl_xor(first, o), l2 = __shfl_xor(last, o);
      first = f2 < first ? f2 : first; last = l2 > last ? l2 : last;
    }
    if (lane == 0) { red_a[wave] = first; red_b[wave] = last; }
    __syncthreads();
    first = red_a[0]; last = red_b[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) { first = red_a[w] < first ? red_a[w] : first; last = red_b[w] > last ? red_b[w] : last; }
    if (last >= 0) { start = first * th; end = (last + 1) * th < N ? (last + 1) * th : N; }
    else { start = 0; end = 0; }
  }
  const int hop = kp.hop;
  const int T = (int)(1 + (end - start) / hop);
  if (status == AFX_CLIP_OK && T < 9) status = AFX_CLIP_TOO_SHORT;   // librosa.feature.delta width 9
  // ---- which blocks of the speculative pass stand
  const int g0 = (int)(start / hop), glast = g0 + T - 1;
  const int nblk = cd.tpad / kFramesPerBlock;
  const bool cutL = start > 0, cutR = end < N;
  const int bLo = cutL ? (g0 + 2 + 15) >> 4 : 0;
  const int bHi = cutR ? (glast >= 17 ? (glast - 17) >> 4 : -1) : nblk - 1;
  float cm = -INFINITY;
  if (status == AFX_CLIP_OK)
    for (int b = bLo + tid; b <= bHi; b += 256) cm = fmaxf(cm, blockmax[cd.blk_base + b]);
  cm = wave_maxf(cm);
  if (lane == 0) red_f[wave] = cm;
  __syncthreads();
  cm = fmaxf(fmaxf(red_f[0], red_f[1]), fmaxf(red_f[2], red_f[3]));
  if (tid == 0) {
    ClipInfo ci;
    ci.start = start; ci.end = end; ci.T = T; ci.status = status;
    ci.lmax_ord = cm > -INFINITY ? f3_ord(cm) : 0u;
    ci.nonfinite = nonfinite;
    info[clip] = ci;
    // ---- frames to redo: [g0, 16 bLo) on a cut left side, [16 (bHi + 1), glast] on a cut right side
    if (status == AFX_CLIP_OK && (cutL || cutR)) {
      int r0[2], r1[2], nr = 0;
      if (bLo > bHi) { r0[0] = g0; r1[0] = glast + 1; nr = 1; }
      else {
        if (cutL && 16 * bLo > g0) { r0[nr] = g0; r1[nr] = 16 * bLo < glast + 1 ? 16 * bLo : glast + 1; ++nr; }
        if (cutR && 16 * (bHi + 1) <= glast) { r0[nr] = 16 * (bHi + 1) > g0 ? 16 * (bHi + 1) : g0; r1[nr] = glast + 1; ++nr; }
      }
      int cnt = 0;
      for (int r = 0; r < nr; ++r) cnt += (r1[r] - r0[r] + 15) / 16;
      if (cnt > 0) {
        const int at = atomicAdd(n_items, cnt);
        int k = 0;
        const int64_t lim = (int64_t)1 << 30;
        for (int r = 0; r < nr; ++r)
          for (int gf = r0[r]; gf < r1[r]; gf += 16, ++k) {
            if (at + k >= max_items) break;                 // cannot happen: the list holds 6 items per clip
            const int nfr = r1[r] - gf < 16 ? r1[r] - gf : 16;
            const int64_t gs = (int64_t)gf * hop - kp.n_fft / 2;
            auto rel = [&](int64_t x) { const int64_t q = x - gs; return (int32_t)(q < -lim ? -lim : (q > lim ? lim : q)); };
            BlockDesc d;
            d.sample_base = cd.off + gs; d.frame_slot = cd.frame_base + gf; d.clip_off = cd.off;
            d.keep_lo = rel(start); d.keep_hi = rel(end); d.have_lo = rel(0); d.have_hi = rel(N);
            d.clip = clip; d.t0 = 0; d.T = nfr; d.active = 1; d.pad_[0] = 0; d.pad_[1] = 0;
            items[at + k] = d;
          }
      }
    }
  }
  // ---- RMS rows from the sub-block sums (feature_extractor.py:164, librosa.feature.rms center=True), trimmed frame index
  if (rms_rows && (status == AFX_CLIP_OK || (status == AFX_CLIP_TOO_SHORT && N >= 2))) {      // extract_energy needs no delta
    const float* bs = bsum + cd.tblk_base * per;
    const int64_t s_lo = start / hop, s_hi = (end + hop - 1) / hop;
    const int nsb = kp.n_fft / hop, back = nsb / 2;
    const float inv_n = 1.0f / (float)kp.n_fft;
    for (int t = tid; t < T; t += 256) {
      float sacc = 0.f;
      for (int k = 0; k < nsb; ++k) {
        const int64_t sb = s_lo + t - back + k;
        if (sb >= s_lo && sb < s_hi) sacc += bs[sb];
      }
      rms_rows[cd.frame_base + t] = sqrtf(sacc * inv_n);
    }
  }
}

// ---------------------------------------------------------------------------
// k_build_blocks3: the speculative launch's block list, built on the device from the batch's clip records.
// reference call site: audio_feature_extraction_toolkit/core/feature_extractor.py:228-235 -- batch_process never sees the
// same clip lengths twice, so nothing per batch may be built block by block on the host: the host uploads one 48-byte
// ClipDesc per clip and this kernel writes the 64-byte record of every absolute 16-frame block (one wave per clip).  54 000 records (3.5 MB) for 1000 ten-second clips.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_build_blocks3(const ClipDesc* __restrict__ clips, int n_clips,
                                                      BlockDesc* __restrict__ blocks, int n_fft, int hop, int trim_hop, int per) {
  // one wave per clip: its record is read once (scalar), its blocks written lane by lane -- no search for the clip of a block
  const int clip = blockIdx.x;
  if (clip >= n_clips) return;
  const ClipDesc c = clips[clip];
  const int nb = c.tpad / kFramesPerBlock;
  const int64_t lim = (int64_t)1 << 30;
  const int64_t ntb = (c.len + trim_hop - 1) / trim_hop;
  for (int b = threadIdx.x; b < nb; b += 64) {
    const int64_t gs = (int64_t)b * kFramesPerBlock * hop - n_fft / 2;       // clip sample of staged index 0
    auto rel = [&](int64_t x) { const int64_t q = x - gs; return (int32_t)(q < -lim ? -lim : (q > lim ? lim : q)); };
    BlockDesc d;
    d.sample_base = c.off + gs; d.frame_slot = c.frame_base + (int64_t)b * kFramesPerBlock; d.clip_off = c.off;
    d.keep_lo = rel(0); d.keep_hi = rel(c.len); d.have_lo = d.keep_lo; d.have_hi = d.keep_hi;
    d.clip = clip; d.t0 = b * kFramesPerBlock; d.T = c.tmax; d.active = (c.len >= 2 && d.t0 < c.tmax) ? 1 : 0;
    d.pad_[0] = (int32_t)(c.tblk_base * per); d.pad_[1] = (int32_t)(ntb * per);
    blocks[c.blk_base + b] = d;
  }
}

hipError_t launch_build_blocks3(hipStream_t s, const ClipDesc* clips, int n_clips, int nblocks, BlockDesc* blocks,
                                const KParams& kp) {
  if (nblocks <= 0 || n_clips <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_build_blocks3, dim3(n_clips), dim3(64), 0, s, clips, n_clips, blocks,
                     kp.n_fft, kp.hop, kp.trim_hop, kp.rms_sub);
  return hipGetLastError();
}

bool frames3_eligible(const KParams& kp, const F3Tables& ft) {
  const int per = kp.hop > 0 ? kp.trim_hop / kp.hop : 0;
  if (!(ft.mel_rounds > 0 && kp.n_mels <= 512 && kp.hop > 0 && kp.trim_hop % kp.hop == 0 && per >= 1 && per <= 4) ||
      dev_env().no_frames3)
    return false;
  if (kp.n_fft == 1024 && kp.hop == 256) return frames3_lds_bytes(12, ft) <= 160 * 1024;
  if (kp.n_fft == 2048 && kp.hop == 512) return frames3s_lds_bytes(12, ft) <= 160 * 1024 && !dev_env().no_frames3s;
  if (kp.n_fft == 512 && kp.hop == 128) return frames3d_lds_bytes(12, ft) <= 160 * 1024 && !dev_env().no_frames3d;
  return false;
}

hipError_t launch_frames3_any(hipStream_t s, const void* samples, ClipInfo* info, const BlockDesc* blocks, int nblocks,
                              const int* nblocks_dev, const F3Tables& ft, const KParams& kp, float* logmel,
                              float* blockmax, float* bsum, bool spec, int* work_ctr, int n_cu) {
  if (kp.n_fft == 2048) return launch_frames3s(s, samples, info, blocks, nblocks, nblocks_dev, ft, kp, logmel, blockmax, bsum, spec, work_ctr, n_cu);
  if (kp.n_fft == 512) return launch_frames3d(s, samples, info, blocks, nblocks, nblocks_dev, ft, kp, logmel, blockmax, bsum, spec, work_ctr, n_cu);
  return launch_frames3(s, samples, info, blocks, nblocks, nblocks_dev, ft, kp, logmel, blockmax, bsum, spec, work_ctr, n_cu);
}

int frames3_waves(const F3Tables& ft) {
  const int forced = dev_env().f3_waves;
  if (forced == 12 || forced == 16) return frames3_lds_bytes(forced, ft) <= 160 * 1024 ? forced : 12;
  return frames3_lds_bytes(16, ft) <= 160 * 1024 ? 16 : 12;
}

template <int FMT, int WAVES, int NB0, int NB1, bool SPEC>
static hipError_t launch_frames3_t(hipStream_t s, const void* samples, ClipInfo* info, const BlockDesc* blocks,
                                   int nblocks, const int* nblocks_dev, const F3Tables& ft, const KParams& kp,
                                   float* logmel, float* blockmax, float* bsum, int* work_ctr, int n_cu) {
  static bool attr_set[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_frames3<FMT, WAVES, NB0, NB1, SPEC>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  const int grid = std::max(1, std::min(n_cu, (nblocks + WAVES - 1) / WAVES));
  hipLaunchKernelGGL((k_frames3<FMT, WAVES, NB0, NB1, SPEC>), dim3(grid), dim3(WAVES * 64), frames3_lds_bytes(WAVES, ft), s,
                     samples, info, blocks, nblocks, nblocks_dev, ft, kp, logmel, blockmax, bsum, work_ctr);
  return hipGetLastError();
}

template <int FMT, int WAVES, bool SPEC>
static hipError_t launch_frames3_w(hipStream_t s, const void* samples, ClipInfo* info, const BlockDesc* blocks,
                                   int nblocks, const int* nblocks_dev, const F3Tables& ft, const KParams& kp,
                                   float* logmel, float* blockmax, float* bsum, int* work_ctr, int n_cu) {
  // straight-line mel schedules compiled in: two rounds of width 1
  const bool two = ft.mel_rounds == 2 && ((ft.mel_rp[0] >> 4) & 15) == 1 && ((ft.mel_rp[1] >> 4) & 15) == 1 && ft.mel_all_own &&
                   ft.mel_wfloats == (int)((ft.mel_rp[0] & 15) + (ft.mel_rp[1] & 15)) * 256 && (ft.mel_rp[1] >> 8) == (ft.mel_rp[0] & 15) * 256 &&
                   !dev_env().f3_generic_mel;
  const int nb0 = ft.mel_rp[0] & 15, nb1 = ft.mel_rp[1] & 15;
  if (two && nb0 == 2 && nb1 == 7)
    return launch_frames3_t<FMT, WAVES, 2, 7, SPEC>(s, samples, info, blocks, nblocks, nblocks_dev, ft, kp, logmel, blockmax, bsum, work_ctr, n_cu);
  return launch_frames3_t<FMT, WAVES, 0, 0, SPEC>(s, samples, info, blocks, nblocks, nblocks_dev, ft, kp, logmel, blockmax, bsum, work_ctr, n_cu);
}

// spec: the speculative first launch (host-built blocks; emits bsum / blockmax); otherwise the list launch
hipError_t launch_frames3(hipStream_t s, const void* samples, ClipInfo* info, const BlockDesc* blocks, int nblocks,
                          const int* nblocks_dev, const F3Tables& ft, const KParams& kp, float* logmel,
                          float* blockmax, float* bsum, bool spec, int* work_ctr, int n_cu) {
  const int waves = frames3_waves(ft);
#define AFX_F3_GO(FMT, W)                                                                                                   \
  (spec ? launch_frames3_w<FMT, W, true>(s, samples, info, blocks, nblocks, nblocks_dev, ft, kp, logmel, blockmax, bsum, work_ctr, n_cu) \
        : launch_frames3_w<FMT, W, false>(s, samples, info, blocks, nblocks, nblocks_dev, ft, kp, logmel, blockmax, bsum, work_ctr, n_cu))
  if (kp.fmt == AFX_FMT_S16) return waves == 16 ? AFX_F3_GO(AFX_FMT_S16, 16) : AFX_F3_GO(AFX_FMT_S16, 12);
  return waves == 16 ? AFX_F3_GO(AFX_FMT_F32, 16) : AFX_F3_GO(AFX_FMT_F32, 12);
#undef AFX_F3_GO
}

hipError_t launch_trim_decide3(hipStream_t s, const ClipDesc* clips, ClipInfo* info, const float* bsum, const float* blockmax,
                               BlockDesc* items, int* n_items, int max_items, float* rms_rows, int n_clips, const KParams& kp) {
  hipLaunchKernelGGL(k_trim_decide3, dim3(n_clips), dim3(256), 0, s, clips, info, bsum, blockmax, items, n_items, max_items,
                     rms_rows, kp);
  return hipGetLastError();
}

}  // namespace afx
