// HIP kernels of libafx.so for gfx950 (MI355X, CDNA4; wave64, 160 KiB LDS/CU).
//
// Pipeline per batch of clips (reference call sites in
// audio_feature_extraction_toolkit/core/feature_extractor.py, "F:" below):
//
//   k_trim_blocks  F:69+72  pre-emphasis on the fly, sum of squares per 512-sample block
//   k_trim_decide  F:72     librosa.effects.trim(top_db=30): clip max, threshold scan -> [start,end), T
//   k_frames       F:127,164 fused: staging of the hop-strided sample block (pre-emphasis + trim mask)
//                           -> periodic window -> real FFT (N/2-point complex Stockham in LDS)
//                           -> |X|^2 -> sparse Slaney mel -> 10*log10 -> log-mel tile (+ clip max),
//                           and RMS of the same staged frame
//   k_dct          F:127    power_to_db's clip-global top_db clamp + ortho DCT-II -> MFCC rows
//   k_stats        F:137-150,171-178  Savitzky-Golay delta/delta2 (width 9, 'interp' edges) and the
//                           per-clip mean / std / ptp reductions
//
// The FFT is VALU/LDS work.  The two contractions the reference runs densely (the mel einsum and the
// DCT) go to the matrix pipe as exact-f32 v_mfma_f32_16x16x4_f32, the filterbank block-sparse (14 % of
// its 16x4 blocks are non-zero), so they run beside the VALU instead of on it.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "afx_device.h"
#include "afx_devenv.h"
#include "afx_f0.h"

namespace afx {

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
#define AFX_CBARRIER() asm volatile("" ::: "memory")
// Workgroup barrier that orders LDS only.  __syncthreads() also drains vmcnt, which would wait for
// the sample prefetch (and the log-mel stores) at every phase boundary.
#define AFX_LDS_BARRIER()                                   \
  do {                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
    __builtin_amdgcn_s_barrier();                           \
    asm volatile("" ::: "memory");                          \
  } while (0)

__device__ __forceinline__ uint32_t f2ord(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}

__device__ __forceinline__ float ld_sample(const void* samples, int fmt, int64_t idx) {
  if (fmt == AFX_FMT_S16) return (float)((const int16_t*)samples)[idx] * (1.0f / 32768.0f);
  return ((const float*)samples)[idx];
}

// out[n] = y[n] + b1*y[n-1] exactly as scipy.signal.lfilter evaluates it in
// float32: the product is rounded, then the sum (no FMA contraction).
__device__ __forceinline__ float preemph1(float y, float prev, float b1) {
#pragma clang fp contract(off)   // HIP's __fmul_rn/__fadd_rn are plain * and + and would fuse
  const float p = b1 * prev;
  return y + p;
}
// librosa's default zi = 2*y[0] - y[1]  ->  out[0] = zi + y[0]
__device__ __forceinline__ float preemph0(float y0, float y1) {
#pragma clang fp contract(off)
  const float t = 2.0f * y0;
  const float zi = t - y1;
  return zi + y0;
}

// Wave-wide reductions without LDS: DPP butterflies inside each 16-lane row, then one readlane per
// row (__shfl_xor would lower to six dependent ds_bpermute round trips).
#define AFX_DPP(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xf, 0xf, false))
template <typename Op>
__device__ __forceinline__ float row_reduce(float v, Op op) {      // every lane ends with its row's total
  v = op(v, AFX_DPP(v, 0xB1));     // quad_perm [1,0,3,2]
  v = op(v, AFX_DPP(v, 0x4E));     // quad_perm [2,3,0,1]
  v = op(v, AFX_DPP(v, 0x141));    // row_half_mirror
  v = op(v, AFX_DPP(v, 0x140));    // row_mirror
  return v;
}
template <typename Op>
__device__ __forceinline__ float wave_reduce(float v, Op op) {     // uniform result
  v = row_reduce(v, op);
  const int vi = __float_as_int(v);          // readlane is an int builtin: bit-cast, do not convert
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(vi, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(vi, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(vi, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(vi, 48));
  return op(op(r0, r1), op(r2, r3));
}
struct OpAdd { __device__ float operator()(float a, float b) const { return a + b; } };
struct OpMax { __device__ float operator()(float a, float b) const { return fmaxf(a, b); } };
struct OpMin { __device__ float operator()(float a, float b) const { return fminf(a, b); } };
__device__ __forceinline__ float wave_sum(float v) { return wave_reduce(v, OpAdd()); }
__device__ __forceinline__ float wave_max(float v) { return wave_reduce(v, OpMax()); }
__device__ __forceinline__ float wave_min(float v) { return wave_reduce(v, OpMin()); }
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---------------------------------------------------------------------------
// k_trim_blocks: sums of squares of the (pre-emphasised) samples per trim block -- the only full pass over the
// samples besides the frame kernel.  A wave owns kTrimPerWave consecutive trim blocks of one clip; when they are
// all interior, float32 and 16-byte aligned it issues every load of its span before the first use.
// ---------------------------------------------------------------------------
constexpr int kTrimPerWave = 4;

__device__ __forceinline__ float sumsq4(float v0, float v1, float v2, float v3) {
#pragma clang fp contract(off)      // one rounding sequence wherever this is inlined
  float s4 = v0 * v0; s4 += v1 * v1; s4 += v2 * v2; s4 += v3 * v3;
  return s4;
}
__device__ __forceinline__ float sq4(float y0, float y1, float y2, float y3, float prev, bool pre, float b1) {
  float v0 = y0, v1 = y1, v2 = y2, v3 = y3;
  if (pre) {
    v0 = preemph1(y0, prev, b1); v1 = preemph1(y1, y0, b1);
    v2 = preemph1(y2, y1, b1); v3 = preemph1(y3, y2, b1);
  }
  return sumsq4(v0, v1, v2, v3);
}

__global__ __launch_bounds__(256) void k_trim_blocks(const void* __restrict__ samples,
                                                     const ClipDesc* __restrict__ clips,
                                                     ClipInfo* __restrict__ info,
                                                     float* __restrict__ bsum, KParams kp) {
  const int clip = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const ClipDesc cd = clips[clip];
  const int64_t N = cd.len;
  const int th = kp.trim_hop;
  const int64_t nb = (N + th - 1) / th;
  const int64_t bfirst = ((int64_t)blockIdx.x * 4 + wave) * kTrimPerWave;
  if (bfirst >= nb) return;
  const bool pre = (kp.flags & AFX_FLAG_PREEMPH) != 0;
  const float b1 = kp.preemph_b1;
  // per > 1: a block's sum is kept as `per` sub-block sums of 256 samples (RMS rows are built from them)
  const int per = kp.rms_sub > 0 ? kp.rms_sub : 1;
  const float* base = (const float*)samples + cd.off;
  int nf = 0;

  // fast route: the wave's whole span is inside the clip, aligned, and made of 256-sample runs
  const int64_t s0 = bfirst * th, s1 = s0 + (int64_t)kTrimPerWave * th;
  const bool fast = (th == 256 || th == 512) && s0 > 0 && s1 <= N && (((cd.off + s0) & 3) == 0);
  if (fast) {
    constexpr int MAXR = kTrimPerWave * 2;
    const int nr = kTrimPerWave * (th >> 8);
    float4 q[MAXR];
    if (kp.fmt == AFX_FMT_F32) {
#pragma unroll
      for (int r = 0; r < MAXR; ++r)
        if (r < nr) q[r] = *reinterpret_cast<const float4*>(base + s0 + 256 * r + 4 * lane);
    } else {                                                    // int16: 8-byte loads, /32768 as libsndfile
      const int16_t* b16 = (const int16_t*)samples + cd.off;
      const float sc = 1.0f / 32768.0f;
#pragma unroll
      for (int r = 0; r < MAXR; ++r)
        if (r < nr) {
          const int2 w = *reinterpret_cast<const int2*>(b16 + s0 + 256 * r + 4 * lane);
          q[r] = make_float4((float)(short)(w.x & 0xffff) * sc, (float)(short)(w.x >> 16) * sc,
                             (float)(short)(w.y & 0xffff) * sc, (float)(short)(w.y >> 16) * sc);
        }
    }
    float carry = ld_sample(samples, kp.fmt, cd.off + s0 - 1);   // sample before the span (lane 0 of run 0)
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
      if (r < nr) {
        float prev = __shfl_up(q[r].w, 1);
        if (lane == 0) prev = carry;
        carry = __shfl(q[r].w, 63);
        nf |= !(isfinite(q[r].x) && isfinite(q[r].y) && isfinite(q[r].z) && isfinite(q[r].w));
        const float s4 = sq4(q[r].x, q[r].y, q[r].z, q[r].w, prev, pre, b1);
        if (per > 1 || th == 256) {                             // every run is its own sum
          const float t = wave_sum(s4);
          if (lane == 0) bsum[(cd.tblk_base + bfirst) * per + r] = t;
        } else {                                                // th == 512, one sum per block
          acc += s4;
          if (r & 1) {
            const float t = wave_sum(acc);
            if (lane == 0) bsum[cd.tblk_base + bfirst + (r >> 1)] = t;
            acc = 0.f;
          }
        }
      }
    }
    nf = __any(nf);
    if (lane == 0 && nf) atomicOr(&info[clip].nonfinite, 1u);
    return;
  }

  // general route (clip head and tail, int16 input, unaligned packing): same order of additions as above --
  // lane l takes samples 4l .. 4l+3 of every 256-sample run -- so the sums, and everything derived from them,
  // do not depend on how the clips were packed.
  for (int k = 0; k < kTrimPerWave; ++k) {
    const int64_t b = bfirst + k;
    if (b >= nb) break;
    const int64_t i0 = b * th;
    const int64_t i1 = (i0 + th < N) ? i0 + th : N;
    float* const dst = bsum + (cd.tblk_base + b) * per;
    float sum = 0.f;
    int j = 0;
    for (int64_t r0 = i0; r0 < i1; r0 += 256, ++j) {
      const int64_t i = r0 + 4 * lane;
      float y0 = 0.f, y1 = 0.f, y2 = 0.f, y3 = 0.f, prev = 0.f;
      if (i < i1) y0 = ld_sample(samples, kp.fmt, cd.off + i);
      if (i + 1 < i1) y1 = ld_sample(samples, kp.fmt, cd.off + i + 1);
      if (i + 2 < i1) y2 = ld_sample(samples, kp.fmt, cd.off + i + 2);
      if (i + 3 < i1) y3 = ld_sample(samples, kp.fmt, cd.off + i + 3);
      if (i > 0 && i < i1) prev = ld_sample(samples, kp.fmt, cd.off + i - 1);
      nf |= !(isfinite(y0) && isfinite(y1) && isfinite(y2) && isfinite(y3));
      float s4;
      if (pre && (i == 0 || i + 3 >= i1)) {                     // clip sample 0 / the clip end inside this quad
        float v0 = preemph1(y0, prev, b1), v1 = preemph1(y1, y0, b1), v2 = preemph1(y2, y1, b1), v3 = preemph1(y3, y2, b1);
        if (i == 0) v0 = (N > 1) ? preemph0(y0, y1) : y0;
        v0 = (i < i1) ? v0 : 0.f; v1 = (i + 1 < i1) ? v1 : 0.f;
        v2 = (i + 2 < i1) ? v2 : 0.f; v3 = (i + 3 < i1) ? v3 : 0.f;
        s4 = sumsq4(v0, v1, v2, v3);
      } else {
        s4 = sq4(y0, y1, y2, y3, prev, pre, b1);
      }
      if (per > 1) {                       // th == 256 * per: this run is sub-block j
        s4 = wave_sum(s4);
        if (lane == 0) dst[j] = s4;
      } else {
        sum += s4;
      }
    }
    if (per > 1) {                         // a short last block: its missing sub-blocks are empty
      for (int kk = j; kk < per; ++kk) if (lane == 0) dst[kk] = 0.f;
    } else {
      sum = wave_sum(sum);
      if (lane == 0) dst[0] = sum;
    }
  }
  nf = __any(nf);
  if (lane == 0 && nf) atomicOr(&info[clip].nonfinite, 1u);
}

// ---------------------------------------------------------------------------
// k_trim_decide: one workgroup per clip
// ---------------------------------------------------------------------------
__device__ __forceinline__ float trim_frame_rms(const float* bs, int64_t t, int64_t nb, int half, int per, float inv_n) {
  float s = 0.f;
  for (int64_t b = (t - half) * per; b < (t + half) * per; ++b)     // bs holds `per` sums per trim block
    if (b >= 0 && b < nb * per) s += bs[b];
  return sqrtf(s * inv_n);
}

__global__ __launch_bounds__(256) void k_trim_decide(const ClipDesc* __restrict__ clips,
                                                     ClipInfo* __restrict__ info,
                                                     const float* __restrict__ bsum,
                                                     BlockDesc* __restrict__ blocks,
                                                     float* __restrict__ rms_rows, KParams kp,
                                                     const void* __restrict__ samples) {
  __shared__ float red_f[4];
  __shared__ long long red_a[4], red_b[4];
  const int clip = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const ClipDesc cd = clips[clip];
  const int64_t N = cd.len;
  int status = AFX_CLIP_OK;
  if (N < 2) status = AFX_CLIP_TOO_SHORT;
  else if (info[clip].nonfinite) status = AFX_CLIP_NONFINITE;
  int64_t start = 0, end = N;
  if ((kp.flags & AFX_FLAG_TRIM) && status == AFX_CLIP_OK) {   // uniform per workgroup
    const int th = kp.trim_hop, half = (kp.trim_frame / th) / 2;
    const int64_t nb = (N + th - 1) / th, nt = 1 + N / th;
    const float inv_n = 1.0f / (float)kp.trim_frame;
    const int per = kp.rms_sub > 0 ? kp.rms_sub : 1;
    const float* bs = bsum + cd.tblk_base * per;
    float mx = 0.f;
    for (int64_t t = tid; t < nt; t += 256) mx = fmaxf(mx, trim_frame_rms(bs, t, nb, half, per, inv_n));
    mx = wave_max(mx);
    if (lane == 0) red_f[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red_f[0], red_f[1]), fmaxf(red_f[2], red_f[3]));
    // amplitude_to_db(mse, ref=np.max, amin=1e-5, top_db=None), float32
    const float ref_db = 10.0f * log10f(fmaxf(1e-10f, mx * mx));
    long long first = (long long)1 << 62, last = -1;
    for (int64_t t = tid; t < nt; t += 256) {
      const float r = trim_frame_rms(bs, t, nb, half, per, inv_n);
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
    if (last >= 0) {
      start = first * th;
      end = (last + 1) * th < N ? (last + 1) * th : N;
    } else { start = 0; end = 0; }
  }
  const int T = (int)(1 + (end - start) / kp.hop);
  if (status == AFX_CLIP_OK && T < 9) status = AFX_CLIP_TOO_SHORT;   // librosa.feature.delta width 9
  if (tid == 0) {
    ClipInfo ci;
    ci.start = start; ci.end = end; ci.T = T; ci.status = status; ci.lmax_ord = 0u;
    ci.nonfinite = info[clip].nonfinite;
    info[clip] = ci;
  }
  // RMS rows from the sub-block sums (feature_extractor.py:164, librosa.feature.rms center=True): frame t
  // covers kept samples [start + t*hop - n_fft/2, + n_fft); start is a multiple of the sub-block (= hop),
  // `end` is a multiple of it or the clip end, so the frame is a run of whole sub-blocks clipped to the kept span.
  // (also for a clip too short for the width-9 delta: extract_energy only needs librosa.feature.rms, F:164)
  if (kp.rms_sub > 0 && rms_rows && (status == AFX_CLIP_OK || (status == AFX_CLIP_TOO_SHORT && N >= 2))) {
    const float* bs = bsum + cd.tblk_base * kp.rms_sub;
    const int64_t s_lo = start / kp.hop, s_hi = (end + kp.hop - 1) / kp.hop;
    const int nsb = kp.n_fft / kp.hop, back = nsb / 2;
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
  // Shapes whose RMS rows come from the frame kernel (rms_sub == 0): a clip too short for the width-9 delta is skipped
  // by that kernel (inactive blocks), but extract_energy only calls librosa.feature.rms (F:164) and must still get its
  // statistics -- its fewer than nine frames are summed here, straight from the samples.
  if (kp.rms_sub == 0 && rms_rows && samples && status == AFX_CLIP_TOO_SHORT && N >= 2 && T >= 1) {
    const bool pre = (kp.flags & AFX_FLAG_PREEMPH) != 0;
    const float inv_n = 1.0f / (float)kp.n_fft;
    for (int t = wave; t < T; t += 4) {
      float acc = 0.f;
      for (int j = lane; j < kp.n_fft; j += 64) {
        const int64_t i = start + (int64_t)t * kp.hop - kp.n_fft / 2 + j;
        float v = 0.f;
        if (i >= start && i < end) {
          const float y = ld_sample(samples, kp.fmt, cd.off + i);
          v = y;
          if (pre) v = (i == 0) ? preemph0(y, ld_sample(samples, kp.fmt, cd.off + 1))
                                : preemph1(y, ld_sample(samples, kp.fmt, cd.off + i - 1), kp.preemph_b1);
        }
        acc = fmaf(v, v, acc);
      }
      acc = wave_sum(acc);
      if (lane == 0) rms_rows[cd.frame_base + t] = sqrtf(acc * inv_n);
    }
  }
  // block descriptors of this clip for k_frames
  const int64_t lim = (int64_t)1 << 30;
  for (int fb = tid; fb < cd.tpad / kFramesPerBlock; fb += 256) {
    BlockDesc d;
    const int t0 = fb * kFramesPerBlock;
    const int64_t g0 = start + (int64_t)t0 * kp.hop - kp.n_fft / 2;
    auto rel = [&](int64_t x) { const int64_t r = x - g0; return (int32_t)(r < -lim ? -lim : (r > lim ? lim : r)); };
    d.sample_base = cd.off + g0; d.frame_slot = cd.frame_base + t0; d.clip_off = cd.off;
    d.keep_lo = rel(start); d.keep_hi = rel(end);
    d.have_lo = rel(0); d.have_hi = rel(N);
    d.clip = clip; d.t0 = t0; d.T = T;
    d.active = (status == AFX_CLIP_OK && t0 < T) ? 1 : 0;
    d.pad_[0] = 0; d.pad_[1] = 0;
    blocks[cd.blk_base + fb] = d;
  }
}

// ---------------------------------------------------------------------------
// k_frames: the fused per-frame kernel
// ---------------------------------------------------------------------------
template <int NFFT>
struct FC {
  static constexpr int N2 = NFFT / 2;              // complex points
  static constexpr int NB = N2 + 1;                // rfft bins
  static constexpr int LPF = (N2 / 8 >= 64) ? 64 : N2 / 8;   // lanes per frame
  static constexpr int P = N2 / LPF;               // complex points per lane (8 or 16)
  static constexpr int FPW = 64 / LPF;             // frames a wave transforms at once
  static constexpr int EXN = N2 + (N2 >> 3);       // padded complex slots of one exchange buffer
  static constexpr int ITERS = kFramesPerBlock / (kWaves * FPW);
};

__host__ __device__ inline int round4(int x) { return (x + 3) & ~3; }

constexpr int kPbStride = 17;   // power-spectrum rows: 16 frames + 1 pad (conflict-free column writes)

// n_fft = 1024 (the headline configuration) runs a hand-scheduled 8x8x8 core: XOR-swizzled
// exchange image without padding, window / split twiddles in LDS, two frames in flight per wave.
__host__ __device__ inline bool fast1024(int n_fft) { return n_fft == 1024; }

struct LdsLayout { int s, ex, pb, tab, rb, total; };   // float offsets
__host__ __device__ inline LdsLayout lds_layout(int n_fft, int hop) {
  const int N2 = n_fft / 2;
  const int lpf = (N2 / 8 >= 64) ? 64 : N2 / 8;
  const int fpw = 64 / lpf;
  const int exn = fast1024(n_fft) ? N2 : N2 + (N2 >> 3);
  LdsLayout L;
  L.s = 0;
  L.ex = L.s + round4((kFramesPerBlock - 1) * hop + n_fft);
  L.pb = L.ex + kWaves * fpw * exn * 2;
  L.tab = L.pb + round4((N2 + 1 + kPbPadRows) * kPbStride);
  L.rb = L.tab + (fast1024(n_fft) ? 4 * N2 : 0);         // window[N2] float2 + post[N2] float2
  L.total = L.rb + kMelMaxSlots * 256 + (fast1024(n_fft) ? 128 : 0);   // mel partial sums; pass-2 twiddles W64^(c*r), 8x8 float2
  return L;
}

size_t frames_lds_bytes(int n_fft, int hop) {
  if (n_fft != 256 && n_fft != 512 && n_fft != 1024 && n_fft != 2048) return 0;
  return (size_t)lds_layout(n_fft, hop).total * sizeof(float);
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }   // a * (-i)

__device__ __forceinline__ void dft4(float2& x0, float2& x1, float2& x2, float2& x3) {
  const float2 a = cadd(x0, x2), b = csub(x0, x2), c = cadd(x1, x3), d = mul_mi(csub(x1, x3));
  x0 = cadd(a, c); x1 = cadd(b, d); x2 = csub(a, c); x3 = csub(b, d);
}

template <int R> __device__ __forceinline__ void dft(float2* x);
template <> __device__ __forceinline__ void dft<4>(float2* x) { dft4(x[0], x[1], x[2], x[3]); }
template <> __device__ __forceinline__ void dft<8>(float2* x) {
  float2 e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6];
  float2 o0 = x[1], o1 = x[3], o2 = x[5], o3 = x[7];
  dft4(e0, e1, e2, e3);
  dft4(o0, o1, o2, o3);
  const float h = 0.70710678118654752440f;
  o1 = make_float2((o1.x + o1.y) * h, (o1.y - o1.x) * h);      // * W8^1
  o2 = mul_mi(o2);                                             // * W8^2
  o3 = make_float2((o3.y - o3.x) * h, (-o3.x - o3.y) * h);     // * W8^3
  x[0] = cadd(e0, o0); x[1] = cadd(e1, o1); x[2] = cadd(e2, o2); x[3] = cadd(e3, o3);
  x[4] = csub(e0, o0); x[5] = csub(e1, o1); x[6] = csub(e2, o2); x[7] = csub(e3, o3);
}

__device__ __forceinline__ int expad(int a) { return a + (a >> 3); }

// One Stockham autosort pass of radix R with NS = product of earlier radices.
// A lane owns the points a_u = lif + LPF*u (u < P) in every pass; butterfly i of
// the lane takes u = i + r*(P/R).  Outputs go to the exchange buffer at
// expand(j) + r*NS; the last pass lands on the lane's own slots.
template <int N2, int LPF, int P, int R, int NS>
struct Pass {
  static constexpr int NBF = P / R;
  static constexpr int NTW = (NS > 1) ? NBF * (R - 1) : 0;
  static constexpr int STEP = N2 / (NS * R);
  static constexpr bool LAST = (NS * R == N2);

  __device__ static __forceinline__ void load_tw(const float2* __restrict__ tab, int lif, float2* tw) {
    if constexpr (NS > 1) {
#pragma unroll
      for (int i = 0; i < NBF; ++i) {
        const int jm = (lif + LPF * i) & (NS - 1);
#pragma unroll
        for (int r = 1; r < R; ++r) tw[i * (R - 1) + r - 1] = tab[jm * r * STEP];
      }
    }
  }

  __device__ static __forceinline__ void run(float2* v, const float2* tw, float2* ex, int lif) {
#pragma unroll
    for (int i = 0; i < NBF; ++i) {
      float2 x[R];
#pragma unroll
      for (int r = 0; r < R; ++r) x[r] = v[i + r * NBF];
      if constexpr (NS > 1) {
#pragma unroll
        for (int r = 1; r < R; ++r) x[r] = cmul(x[r], tw[i * (R - 1) + r - 1]);
      }
      dft<R>(x);
      const int j = lif + LPF * i;
      const int j0 = (j & ~(NS - 1)) * R + (j & (NS - 1));
#pragma unroll
      for (int r = 0; r < R; ++r) {
        ex[expad(j0 + r * NS)] = x[r];
        if constexpr (LAST) v[i + r * NBF] = x[r];
      }
    }
    AFX_CBARRIER();
    if constexpr (!LAST) {
#pragma unroll
      for (int u = 0; u < P; ++u) v[u] = ex[expad(lif + LPF * u)];
      AFX_CBARRIER();
    }
  }
};

// radix schedules
template <int NFFT> struct Sched;
template <> struct Sched<256>  { using C = FC<256>;  using P1 = Pass<C::N2, C::LPF, C::P, 8, 1>; using P2 = Pass<C::N2, C::LPF, C::P, 4, 8>;  using P3 = Pass<C::N2, C::LPF, C::P, 4, 32>;  using P4 = void; };
template <> struct Sched<512>  { using C = FC<512>;  using P1 = Pass<C::N2, C::LPF, C::P, 8, 1>; using P2 = Pass<C::N2, C::LPF, C::P, 8, 8>;  using P3 = Pass<C::N2, C::LPF, C::P, 4, 64>;  using P4 = void; };
template <> struct Sched<1024> { using C = FC<1024>; using P1 = Pass<C::N2, C::LPF, C::P, 8, 1>; using P2 = Pass<C::N2, C::LPF, C::P, 8, 8>;  using P3 = Pass<C::N2, C::LPF, C::P, 8, 64>;  using P4 = void; };
template <> struct Sched<2048> { using C = FC<2048>; using P1 = Pass<C::N2, C::LPF, C::P, 8, 1>; using P2 = Pass<C::N2, C::LPF, C::P, 8, 8>;  using P3 = Pass<C::N2, C::LPF, C::P, 4, 64>;  using P4 = Pass<C::N2, C::LPF, C::P, 4, 256>; };

template <typename PX> struct NTW { static constexpr int v = PX::NTW; };
template <> struct NTW<void> { static constexpr int v = 0; };

// Raw samples of one staged quad: the 4 samples (bit pattern of the vector load: float4, or 4 x int16
// in .x/.y) and the sample before them.  Kept as native vector registers: a struct of five floats made
// hipcc merge the two loads into an unaligned dwordx4 plus a dword and then shuffle -- and wait -- per quad.
struct RawQuad { float4 q; float prev; };

struct BlkCtx {          // uniform per workgroup; one BlockDesc resolved into scalars
  int64_t sample_base, frame_slot, clip_off;
  int keep_lo, keep_hi, have_lo, have_hi;
  int clip, t0, T;
  bool active, interior;
};

// staged index j -> raw samples j-1 .. j+3 of the block (zeros outside the clip).
// Branch-free on purpose: a per-lane "load or keep" branch makes hipcc wait vmcnt(0) inside every
// branch, which serialises the prefetch.  `interior` is uniform per block: every staged sample and its
// predecessor exist and the quads are 16-byte (F32) / 8-byte (S16) aligned -> one plain vector load per
// quad; the predecessor comes from the neighbouring lane at staging time, so only the wave's first lane
// needs it from memory: all lanes load that one (wave-uniform) address.  Edge blocks take clamped
// scalar loads + selects and carry a per-lane predecessor.
template <bool INTERIOR, int FMT>
__device__ __forceinline__ RawQuad load_raw(const void* __restrict__ samples, const BlkCtx& c, int j, int jwave) {
  RawQuad r;
  if constexpr (INTERIOR) {
    if constexpr (FMT == AFX_FMT_F32) {
      const float* base = (const float*)samples + c.sample_base;
      r.q = *reinterpret_cast<const float4*>(base + j);
      r.prev = base[jwave - 1];
    } else {
      const int16_t* base = (const int16_t*)samples + c.sample_base;
      const int2 q = *reinterpret_cast<const int2*>(base + j);
      r.q = make_float4(__int_as_float(q.x), __int_as_float(q.y), 0.f, 0.f);
      r.prev = (float)base[jwave - 1] * (1.0f / 32768.0f);
    }
  } else {
    const int lo = c.have_lo, hi = c.have_hi - 1;          // hi >= lo: clips have >= 2 samples
    auto at = [&](int jj) {
      const int jc = jj < lo ? lo : (jj > hi ? hi : jj);
      const float v = ld_sample(samples, FMT, c.sample_base + jc);
      return (jj == jc) ? v : 0.f;
    };
    r.prev = at(j - 1);
    r.q = make_float4(at(j), at(j + 1), at(j + 2), at(j + 3));
  }
  return r;
}

// pre-emphasis (as lfilter does it) + trim mask of one quad -> 4 staged samples
template <bool INTERIOR, int FMT>
__device__ __forceinline__ float4 stage_quad(const RawQuad& r, const void* __restrict__ samples,
                                             const BlkCtx& c, int j, bool pre, float b1) {
  float y0, y1, y2, y3, prev;
  if constexpr (INTERIOR && FMT == AFX_FMT_S16) {
    const int a = __float_as_int(r.q.x), b = __float_as_int(r.q.y);
    const float sc = 1.0f / 32768.0f;
    y0 = (float)(short)(a & 0xffff) * sc; y1 = (float)(short)(a >> 16) * sc;
    y2 = (float)(short)(b & 0xffff) * sc; y3 = (float)(short)(b >> 16) * sc;
  } else { y0 = r.q.x; y1 = r.q.y; y2 = r.q.z; y3 = r.q.w; }
  if constexpr (INTERIOR) {
    // predecessor = previous lane's last sample (DPP wave_shr:1); lane 0 keeps the loaded one
    prev = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(r.prev), __float_as_int(y3), 0x138, 0xf, 0xf, false));
  } else prev = r.prev;
  float v0 = y0, v1 = y1, v2 = y2, v3 = y3;
  if (pre) {
    v0 = preemph1(y0, prev, b1); v1 = preemph1(y1, y0, b1);
    v2 = preemph1(y2, y1, b1); v3 = preemph1(y3, y2, b1);
    if constexpr (!INTERIOR) {            // only edge blocks can hold the clip's sample 0
      const int e0 = c.have_lo - j;
      if (e0 >= 0 && e0 < 4) {            // librosa's zi = 2*y[0] - y[1]
        const float z = preemph0(ld_sample(samples, FMT, c.clip_off), ld_sample(samples, FMT, c.clip_off + 1));
        if (e0 == 0) v0 = z; else if (e0 == 1) v1 = z; else if (e0 == 2) v2 = z; else v3 = z;
      }
    }
  }
  const unsigned span = (unsigned)(c.keep_hi - c.keep_lo), d = (unsigned)(j - c.keep_lo);
  float4 o;
  o.x = (d < span) ? v0 : 0.f;
  o.y = (d + 1u < span) ? v1 : 0.f;
  o.z = (d + 2u < span) ? v2 : 0.f;
  o.w = (d + 3u < span) ? v3 : 0.f;
  return o;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Power-spectrum buffer PB[bin][17]: frame f of bin k at k*17 + f.  The per-frame column
// write (64 consecutive bins, fixed f) walks banks in steps of 17 -> conflict-free; the
// mel stage reads 4 consecutive bins x 16 frames per wave -> one 2-way overlap at most.

// STAMP = true is a separate diagnostic instantiation (AFX_DEBUG_STAMPS): lane 0 of every wave sums
// s_memtime deltas per phase into `stamps`; the production kernel carries none of it.
enum { ST_STAGE = 0, ST_BAR1, ST_FFT, ST_PREFETCH, ST_BAR2, ST_MEL, ST_BAR3, ST_MELFIN, ST_X0, ST_X1, ST_X2, ST_X3, ST_COUNT };

template <int NFFT, bool STAMP>
__global__ __launch_bounds__(256, (NFFT >= 2048 ? 1 : 2)) void k_frames(const void* __restrict__ samples,
                                                   ClipInfo* __restrict__ info,
                                                   const BlockDesc* __restrict__ blocks, int nblocks,
                                                   DevTables tb, KParams kp,
                                                   float* __restrict__ logmel,
                                                   float* __restrict__ rms_rows,
                                                   unsigned long long* __restrict__ stamps) {
  using C = FC<NFFT>;
  unsigned long long st_sum[ST_COUNT] = {}, st_prev = 0;
  auto stamp = [&](int ph) {
    if constexpr (STAMP) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0) only
      __builtin_amdgcn_sched_barrier(0);
      if (ph >= 0) st_sum[ph] += t - st_prev;
      st_prev = t;
    }
  };
  using S = Sched<NFFT>;
  constexpr int N2 = C::N2, NB = C::NB, LPF = C::LPF, P = C::P, FPW = C::FPW;
  constexpr int MAXCH = (NFFT * 5 + 1023) / 1024;    // 1024-sample chunks a thread prefetches (hop <= n_fft/4)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int hop = kp.hop, M = kp.n_mels;
  const LdsLayout L = lds_layout(NFFT, hop);
  float* const S_ = smem + L.s;
  float* const PB = smem + L.pb;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int lif = lane % LPF, fsub = lane / LPF;
  constexpr bool FAST = (NFFT == 1024);
  constexpr int EXSTRIDE = FAST ? N2 : C::EXN;
  float2* const EX = reinterpret_cast<float2*>(smem + L.ex) + (wave * FPW + fsub) * EXSTRIDE;
  float2* const WT = reinterpret_cast<float2*>(smem + L.tab);          // FAST only
  float2* const PT = WT + N2;
  float2* const T2 = reinterpret_cast<float2*>(smem + L.rb + kMelMaxSlots * 256);   // FAST only

  // ---- once per workgroup: zero the pad rows of PB; per-lane tables -> registers (or LDS)
  for (int i = tid; i < kPbPadRows * kPbStride; i += 256) PB[NB * kPbStride + i] = 0.f;
  float2 wreg[FAST ? 1 : P], preg[FAST ? 1 : P];
  int maddr[FAST ? 1 : P];            // exchange-buffer slot of the mirror bin Z[N2 - k] (generic path)
  {
    const float2* w2 = reinterpret_cast<const float2*>(tb.window);
    const float2* p2 = reinterpret_cast<const float2*>(tb.post);
    if constexpr (FAST) {
      // window pre-scaled by 1/2 (exact): the split below then yields X, not 2X, and |X|^2 needs no 0.25
      for (int i = tid; i < N2; i += 256) { WT[i] = make_float2(0.5f * w2[i].x, 0.5f * w2[i].y); PT[i] = p2[i]; }
      // pass-2 twiddles W_64^(c*r), c = lane & 7: 64 values, read per frame pair instead of 14 registers per lane
      if (tid < 64) T2[tid] = reinterpret_cast<const float2*>(tb.tw)[(tid >> 3) * (tid & 7) * (N2 / 64)];   // [r*8 + c], symmetric in (r, c)
    }
    if constexpr (!FAST) {
#pragma unroll
      for (int u = 0; u < P; ++u) {
        const int k = lif + LPF * u;
        wreg[u] = w2[k]; preg[u] = p2[k]; maddr[u] = expad((N2 - k) & (N2 - 1));
      }
    }
  }
  // FAST: slot swizzle sw(a) = a ^ ((a>>3)&7) ^ (((a>>6)&1)<<3) makes every exchange access of the
  // 8x8x8 schedule bank-conflict-free; per lane it collapses to four bases:
  //   pass-1 write 8j+r -> A1 ^ r;  pass-2 write -> A2 ^ 9r;  read / last write j+64r -> (r odd ? B1 : B0) + 64r
  const int swA1 = (8 * lane) ^ (lane & 7) ^ (((lane >> 3) & 1) << 3);
  const int swB0 = lane ^ ((lane >> 3) & 7), swB1 = swB0 ^ 8;
  const int swA2 = 64 * (lane >> 3) + 8 * ((lane >> 3) & 1) + (lane & 7);
  constexpr int NT2 = NTW<typename S::P2>::v, NT3 = NTW<typename S::P3>::v, NT4 = NTW<typename S::P4>::v;
  float2 tw2[NT2 > 0 ? NT2 : 1], tw3[NT3 > 0 ? NT3 : 1], tw4[NT4 > 0 ? NT4 : 1];
  {
    const float2* t2 = reinterpret_cast<const float2*>(tb.tw);
    if constexpr (!FAST) S::P2::load_tw(t2, lif, tw2);
    S::P3::load_tw(t2, lif, tw3);
    if constexpr (NT4 > 0) S::P4::load_tw(t2, lif, tw4);
  }
  // mel: this wave's first work items (all of them when n_mels <= 128) stay in registers
  const int f16k = lane & 15;
  float* const RB = smem + L.rb;
  const int mel_cnt = __builtin_amdgcn_readfirstlane(
      tb.mel_item_cnt[0] * (wave == 0) + tb.mel_item_cnt[1] * (wave == 1) +
      tb.mel_item_cnt[2] * (wave == 2) + tb.mel_item_cnt[3] * (wave == 3));
  int4 mi_a[kMelRegItems], mi_b[kMelRegItems];       // (group, b0, nb, role), (slot, nslots, kmin, -)
  float4 mi_cf[kMelRegItems];
  float mi_ko[kMelRegItems];
#pragma unroll
  for (int i = 0; i < kMelRegItems; ++i) {
    mi_a[i] = make_int4(0, 0, 0, 0); mi_b[i] = make_int4(0, 0, 0, 0);
    mi_cf[i] = make_float4(0.f, 0.f, 0.f, 0.f); mi_ko[i] = 0.f;
    if (i < mel_cnt) {
      // wave-uniform metadata -> SGPRs (the compiler cannot see that tid >> 6 is uniform)
      auto sg = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
      const int4 a = tb.mel_items[(wave * kMelMaxItems + i) * 2], bb = tb.mel_items[(wave * kMelMaxItems + i) * 2 + 1];
      mi_a[i] = make_int4(sg(a.x), sg(a.y), sg(a.z), sg(a.w));
      mi_b[i] = make_int4(sg(bb.x), sg(bb.y), sg(tb.mel_grp[a.x].x), 0);
      mi_cf[i] = tb.mel_coef[mi_a[i].x * 16 + f16k];
      mi_ko[i] = tb.mel_koff[mi_a[i].x * 16 + f16k];
    }
  }

  const bool pre = (kp.flags & AFX_FLAG_PREEMPH) != 0;
  const float b1 = kp.preemph_b1;
  const int fmt = kp.fmt;
  const int slen = (kFramesPerBlock - 1) * hop + NFFT;
  const bool hop_even = (hop & 1) == 0;

  // A block's 64-byte descriptor is fetched as one dword per lane (VMEM, so that it does not
  // share a wait counter with the LDS traffic) two blocks ahead and resolved with readlanes.
  auto fetch_desc = [&](int b) -> int {
    const int bb = b < nblocks ? b : nblocks - 1;
    return reinterpret_cast<const int*>(blocks + bb)[lane & 15];
  };
  auto resolve = [&](int w, int b) -> BlkCtx {
    BlkCtx c;
    auto rl = [&](int i) { return __builtin_amdgcn_readlane(w, i); };
    auto rl64 = [&](int i) { return (int64_t)(((uint64_t)(uint32_t)rl(i + 1) << 32) | (uint32_t)rl(i)); };
    c.sample_base = rl64(0); c.frame_slot = rl64(2); c.clip_off = rl64(4);
    c.keep_lo = rl(6); c.keep_hi = rl(7); c.have_lo = rl(8); c.have_hi = rl(9);
    c.clip = rl(10); c.t0 = rl(11); c.T = rl(12);
    c.active = (b < nblocks) && rl(13) != 0;
    c.interior = ((c.sample_base & 3) == 0) && ((slen & 3) == 0) && c.have_lo <= -1 && c.have_hi >= slen;
    return c;
  };

  // Log-mel values of the block just finished wait in registers and are stored one iteration later,
  // right after the staging wait: vmcnt retires in order and counts stores, so stores issued after the
  // sample prefetch would be drained (1-2 us) by the wait for those samples at the top of the loop.
  float lmh[kMelRegItems][4];
  bool pend = false;
  float pend_lmax = -INFINITY;
  int64_t pend_slot = 0;
  int pend_t0 = 0, pend_T = 0, pend_clip = 0;
  auto flush_logmel = [&]() {
    if (!pend) return;
    pend = false;
    int lane_f = lane;
    asm volatile("" : "+v"(lane_f));
    const int f16 = lane_f & 15, q4 = lane_f >> 4;
    const bool valid = (pend_t0 + f16) < pend_T;
    float* tile = logmel + pend_slot * (int64_t)M;
    if (!(kp.flags & 0x800)) {
#pragma unroll
      for (int i = 0; i < kMelRegItems; ++i) {
        if (i < mel_cnt && mi_a[i].w != 1) {
// tile layout [mel/4][frame][mel%4]: this lane's four filters are one 16-byte store
          const int m0 = mi_a[i].x * 16 + q4 * 4;
          if (valid && m0 < M)
            *reinterpret_cast<float4*>(tile + (m0 >> 2) * 64 + f16 * 4) = make_float4(lmh[i][0], lmh[i][1], lmh[i][2], lmh[i][3]);
        }
      }
      const float mx = wave_max(pend_lmax);
      if (lane_f == 0 && mx > -INFINITY) atomicMax(&info[pend_clip].lmax_ord, f2ord(mx));
    }
  };

  // experiment (AFX_DEBUG_SKIP bits 0x1000 / 0x2000): start half of the workgroups ~half a block late so
  // that co-resident workgroups sit in complementary phases (FFT = VALU+LDS, mel = matrix pipe)
  if (((kp.flags & 0x1000) && blockIdx.x >= gridDim.x / 2) || ((kp.flags & 0x2000) && (blockIdx.x & 1))) {
    for (int i = 0; i < 2; ++i) __builtin_amdgcn_s_sleep(127);
  }

  RawQuad pf[MAXCH];
  BlkCtx cur = resolve(fetch_desc(blockIdx.x), blockIdx.x);
  int dnext = fetch_desc(blockIdx.x + gridDim.x);
  // chunk c of a thread covers staged samples j = 4*tid + 1024*c .. +3; indices past the block are
  // clamped (loaded, never stored) so that no load sits under a per-lane branch
  const int jlast = ((slen + 3) & ~3) - 4;
  const int jwave0 = (tid & ~63) * 4;          // staged index of this wave's first lane in chunk 0
  auto prefetch = [&](const BlkCtx& bc) {
#define AFX_PF_LOOP(INTERIOR, FMT)                                                                  \
    _Pragma("unroll") for (int c = 0; c < MAXCH; ++c) {                                             \
      const int j = tid * 4 + c * 1024, jw = jwave0 + c * 1024;                                     \
      pf[c] = load_raw<INTERIOR, FMT>(samples, bc, j < jlast ? j : jlast, jw < jlast ? jw : jlast); \
    }
    if (fmt == AFX_FMT_F32) {
      if (bc.interior) { AFX_PF_LOOP(true, AFX_FMT_F32) } else { AFX_PF_LOOP(false, AFX_FMT_F32) }
    } else {
      if (bc.interior) { AFX_PF_LOOP(true, AFX_FMT_S16) } else { AFX_PF_LOOP(false, AFX_FMT_S16) }
    }
#undef AFX_PF_LOOP
  };
  auto stage_all = [&](const BlkCtx& bc) {
#define AFX_ST_LOOP(INTERIOR, FMT)                                                                  \
    _Pragma("unroll") for (int c = 0; c < MAXCH; ++c) {                                             \
      const int j = tid * 4 + c * 1024;                                                             \
      const float4 o = stage_quad<INTERIOR, FMT>(pf[c], samples, bc, j < jlast ? j : jlast, pre, b1); \
      if (j < slen) *reinterpret_cast<float4*>(S_ + j) = o;                                         \
    }                                                                                               \
    for (int j = tid * 4 + MAXCH * 1024; j < slen; j += 1024)  /* hop > n_fft/4: not prefetched */   \
      *reinterpret_cast<float4*>(S_ + j) = stage_quad<false, FMT>(load_raw<false, FMT>(samples, bc, j, j), samples, bc, j, pre, b1);
    if (fmt == AFX_FMT_F32) {
      if (bc.interior) { AFX_ST_LOOP(true, AFX_FMT_F32) } else { AFX_ST_LOOP(false, AFX_FMT_F32) }
    } else {
      if (bc.interior) { AFX_ST_LOOP(true, AFX_FMT_S16) } else { AFX_ST_LOOP(false, AFX_FMT_S16) }
    }
#undef AFX_ST_LOOP
  };
  if (cur.active && !(kp.flags & 0x100)) prefetch(cur);
  AFX_LDS_BARRIER();

  for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
    stamp(-1);
    // ---- stage block b from the prefetched registers: pre-emphasis + trim mask, once per sample
    // (0x100/0x200/0x400: timing-only ablation switches (AFX_DEBUG_SKIP), results invalid)
    if (cur.active && !(kp.flags & 0x100)) stage_all(cur);
    flush_logmel();
    stamp(ST_STAGE);
    AFX_LDS_BARRIER();
    stamp(ST_BAR1);

    const BlkCtx nxt = resolve(dnext, b + gridDim.x);
    dnext = fetch_desc(b + 2 * gridDim.x);

    // ---- per frame: window -> rFFT -> power spectrum (+ RMS of the unwindowed frame)
    if (cur.active && !(kp.flags & 0x200)) {
      if constexpr (FAST) {
        // two frames (A, B) in flight per wave; they take turns on the wave's single exchange image,
        // so each one's LDS round trip hides under the other's butterflies
#pragma unroll 1
        for (int pr = 0; pr < 2; ++pr) {
          const int flA = wave * 4 + 2 * pr;
          int a1 = swA1, a2 = swA2;              // opaque per iteration: keeps LICM from parking the 16
          asm volatile("" : "+v"(a1), "+v"(a2)); // XOR-ed exchange addresses in registers for the whole kernel
          const float* SA = S_ + flA * hop;
          const float* SB = SA + hop;
          float2 vA[8], vB[8];
          float ssA = 0.f, ssB = 0.f;
          {
            float2 xa[8], xb[8], ww[8];
            if (hop_even) {
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                xa[u] = *reinterpret_cast<const float2*>(SA + 2 * (lane + 64 * u));
                xb[u] = *reinterpret_cast<const float2*>(SB + 2 * (lane + 64 * u));
              }
            } else {
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                const int a = 2 * (lane + 64 * u);
                xa[u].x = SA[a]; xa[u].y = SA[a + 1]; xb[u].x = SB[a]; xb[u].y = SB[a + 1];
              }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) ww[u] = WT[lane + 64 * u];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              ssA += xa[u].x * xa[u].x; ssA += xa[u].y * xa[u].y;
              ssB += xb[u].x * xb[u].x; ssB += xb[u].y * xb[u].y;
              vA[u] = make_float2(xa[u].x * ww[u].x, xa[u].y * ww[u].y);
              vB[u] = make_float2(xb[u].x * ww[u].x, xb[u].y * ww[u].y);
            }
          }
          ssA = wave_sum(ssA); ssB = wave_sum(ssB);
          if (lane == 0) {
            if (cur.t0 + flA < cur.T) rms_rows[cur.frame_slot + flA] = sqrtf(ssA / (float)NFFT);
            if (cur.t0 + flA + 1 < cur.T) rms_rows[cur.frame_slot + flA + 1] = sqrtf(ssB / (float)NFFT);
          }
          // pass 1 (radix 8, no twiddles) + exchange
          dft<8>(vA); dft<8>(vB);
#pragma unroll
          for (int r = 0; r < 8; ++r) EX[a1 ^ r] = vA[r];
          AFX_CBARRIER();
#pragma unroll
          for (int r = 0; r < 8; ++r) vA[r] = EX[((r & 1) ? swB1 : swB0) + 64 * r];
          AFX_CBARRIER();
#pragma unroll
          for (int r = 0; r < 8; ++r) EX[a1 ^ r] = vB[r];
          AFX_CBARRIER();
#pragma unroll
          for (int r = 0; r < 8; ++r) vB[r] = EX[((r & 1) ? swB1 : swB0) + 64 * r];
          AFX_CBARRIER();
          // pass 2
          float2 t2v[8];
#pragma unroll
          for (int r = 1; r < 8; ++r) t2v[r] = T2[r * 8 + (lane & 7)];
#pragma unroll
          for (int r = 1; r < 8; ++r) vA[r] = cmul(vA[r], t2v[r]);
          dft<8>(vA);
#pragma unroll
          for (int r = 0; r < 8; ++r) EX[a2 ^ (9 * r)] = vA[r];
          AFX_CBARRIER();
#pragma unroll
          for (int r = 0; r < 8; ++r) vA[r] = EX[((r & 1) ? swB1 : swB0) + 64 * r];
          AFX_CBARRIER();
#pragma unroll
          for (int r = 1; r < 8; ++r) vB[r] = cmul(vB[r], t2v[r]);
          dft<8>(vB);
#pragma unroll
          for (int r = 0; r < 8; ++r) EX[a2 ^ (9 * r)] = vB[r];
          AFX_CBARRIER();
#pragma unroll
          for (int r = 0; r < 8; ++r) vB[r] = EX[((r & 1) ? swB1 : swB0) + 64 * r];
          AFX_CBARRIER();
          // pass 3: outputs land on the lane's own bins k = lane + 64 r; publish them for the mirror
          // reads Z[N2-k]; then the real-FFT split and the power spectrum.  B's pass 3 runs under A's
          // mirror-read latency.
          float* const pcol = PB + lane * kPbStride + flA;
          float2 pw[8];             // split twiddles; read once for both frames (LDS reads must not
#pragma unroll                  // sit between the PB stores: the compiler would serialise them)
          for (int u = 0; u < 8; ++u) pw[u] = PT[lane + 64 * u];
          auto split_store = [&](const float2 (&v)[8], const float2 (&m)[8], int col) {
            float pv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const float2 w = pw[u];
              const float2 z = v[u], mm = m[u];
              const float e2r = z.x + mm.x, e2i = z.y - mm.y, o2r = z.y + mm.y, o2i = mm.x - z.x;
              const float xr = e2r + w.x * o2r - w.y * o2i, xi = e2i + w.x * o2i + w.y * o2r;
              pv[u] = xr * xr + xi * xi;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) pcol[64 * u * kPbStride + col] = pv[u];
            if (lane == 0) { const float ny = 2.0f * (v[0].x - v[0].y); pcol[N2 * kPbStride + col] = ny * ny; }
          };
          const int msrc = ((64 - lane) & 63) << 2;           // ds_bpermute byte address of the mirror lane
          auto mirror = [&](const float2 (&v)[8], float2 (&m)[8]) {
            // Z[N2 - k] for k = lane + 64u sits in lane 64 - lane, register 7 - u: a crossbar permute,
            // no LDS image needed.  Lane 0 mirrors onto itself one register up ((8 - u) & 7).
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              m[u].x = __int_as_float(__builtin_amdgcn_ds_bpermute(msrc, __float_as_int(v[7 - u].x)));
              m[u].y = __int_as_float(__builtin_amdgcn_ds_bpermute(msrc, __float_as_int(v[7 - u].y)));
            }
            if (lane == 0) {
#pragma unroll
              for (int u = 0; u < 8; ++u) m[u] = v[(8 - u) & 7];
            }
          };
          {
            float2 mA[8];
#pragma unroll
            for (int r = 1; r < 8; ++r) vA[r] = cmul(vA[r], tw3[r - 1]);
            dft<8>(vA);
            mirror(vA, mA);
#pragma unroll
            for (int r = 1; r < 8; ++r) vB[r] = cmul(vB[r], tw3[r - 1]);
            dft<8>(vB);
            split_store(vA, mA, 0);
          }
          {
            float2 mB[8];
            mirror(vB, mB);
            split_store(vB, mB, 1);
          }
          AFX_CBARRIER();
        }
      } else {
#pragma unroll 1
      for (int it = 0; it < C::ITERS; ++it) {
        const int fl = (wave * C::ITERS + it) * FPW + fsub;      // frame within the block
        const float* Sf = S_ + fl * hop;
        float2 v[P];
        float ss = 0.f;
#pragma unroll
        for (int u = 0; u < P; ++u) {
          const int a = lif + LPF * u;
          float2 x;
          if (hop_even) x = *reinterpret_cast<const float2*>(Sf + 2 * a);
          else { x.x = Sf[2 * a]; x.y = Sf[2 * a + 1]; }
          ss += x.x * x.x; ss += x.y * x.y;
          v[u] = make_float2(x.x * wreg[u].x, x.y * wreg[u].y);
        }
        if constexpr (LPF == 64) ss = wave_sum(ss);
        else {
#pragma unroll
          for (int o = LPF / 2; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
        }
        if (lif == 0 && cur.t0 + fl < cur.T) rms_rows[cur.frame_slot + fl] = sqrtf(ss / (float)NFFT);

        S::P1::run(v, nullptr, EX, lif);
        S::P2::run(v, tw2, EX, lif);
        S::P3::run(v, tw3, EX, lif);
        if constexpr (NT4 > 0) S::P4::run(v, tw4, EX, lif);

        // real-FFT split: X[k] from Z[k] and Z[N2-k]; EX now holds Z in natural order
        float* const pcol = PB + lif * kPbStride + fl;
#pragma unroll
        for (int u = 0; u < P; ++u) {
          const float2 z = v[u];
          const float2 m = EX[maddr[u]];
          const float e2r = z.x + m.x, e2i = z.y - m.y;
          const float o2r = z.y + m.y, o2i = m.x - z.x;
          const float2 w = preg[u];
          const float xr = e2r + w.x * o2r - w.y * o2i;
          const float xi = e2i + w.x * o2i + w.y * o2r;
          pcol[LPF * u * kPbStride] = 0.25f * (xr * xr + xi * xi);
          if (u == 0 && lif == 0) { const float ny = z.x - z.y; pcol[N2 * kPbStride] = ny * ny; }
        }
        AFX_CBARRIER();
      }
      }
    }
    // ---- issue the next block's sample loads; they land under the mel phase (issued here rather
    // than before the FFT so that the raw quads are not live across the register-hungry FFT phase)
    stamp(ST_FFT);
    if (nxt.active && !(kp.flags & 0x100)) prefetch(nxt);
    stamp(ST_PREFETCH);
    AFX_LDS_BARRIER();
    stamp(ST_BAR2);

    // ---- mel filterbank + dB on the matrix pipe: D[16 filters][16 frames] += A[16x4] * B[4 bins x 16 frames]
    // (exact f32 MFMA over the non-zero 16x4 blocks of librosa.filters.mel; the A operand -- the
    // filter triangles -- is evaluated per lane, see MelBlocks in afx_internal.h).  The block ranges are
    // cut into work items balanced over the 4 waves; a split group's partial sums meet in an LDS slot.
    const bool mel_on = cur.active && !(kp.flags & 0x400);
    // lane index made opaque per block: otherwise LICM hoists every lane-derived address of this phase
    // out of the block loop and they sit in registers through the (register-bound) FFT phase
    int lane_m = lane;
    asm volatile("" : "+v"(lane_m));
    const int f16 = lane_m & 15, q4 = lane_m >> 4;
    const bool valid = (cur.t0 + f16) < cur.T;
    float lmax = -INFINITY;
    float* tile = logmel + cur.frame_slot * (int64_t)M;
    auto mel_item = [&](int kmin, int b0, int nb, const float4 cf, const float ko) -> f32x4 {
      const float* p0 = PB + (kmin + 4 * b0 + q4) * kPbStride + f16;
      const float* const pmax = PB + (NB + kPbPadRows - 1) * kPbStride + f16;    // a zero pad row
      float kf = (float)(q4 + 4 * b0) + ko;                       // k - kc of this lane's bin, exact
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      // 8 blocks (32 bins) per step: the 8 B-operand reads are issued together and two accumulators
      // alternate, so neither the LDS latency nor the MFMA dependency serialises the chain.  Blocks
      // past the item's range get zero weight (and rows past the Nyquist bin are the zero pad rows).
      for (int bk = 0; bk < nb; bk += 8) {
        float pb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float* p = p0 + i * 4 * kPbStride;
          pb[i] = *(p < pmax ? p : pmax);
        }
        p0 += 8 * 4 * kPbStride;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float lo = fmaf(cf.y, kf, cf.x), hi = fmaf(cf.w, kf, cf.z);
          float w = __builtin_amdgcn_fmed3f(0.f, lo, hi);           // max(0, min(lo, hi))
          w = (bk + i < nb) ? w : 0.f;                               // the next part of a split group owns those bins
          if (i & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, pb[i], acc1, 0, 0, 0);
          else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, pb[i], acc0, 0, 0, 0);
          kf += 4.0f;
        }
      }
      return acc0 + acc1;
    };
    // dst == nullptr: store the tile rows now (table-driven items); otherwise park the four values in
    // registers -- their global stores are issued next iteration, after the staging wait (see lmh)
    auto mel_finish = [&](const f32x4 acc, int g, float* dst) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = g * 16 + q4 * 4 + r;
        // 10*log10(max(amin, mel)); v_log_f32 is log2
        const float Lv = 3.01029995663981195f * __builtin_amdgcn_logf(fmaxf(kp.amin, acc[r]));
        if (dst) dst[r] = Lv;
        if (valid && m < M) {
          if (!dst && !(kp.flags & 0x800)) tile[(m >> 2) * 64 + f16 * 4 + (m & 3)] = Lv;
          lmax = fmaxf(lmax, Lv);
        }
      }
    };
    f32x4 held[kMelRegItems];
    if (mel_on) {
#pragma unroll
      for (int i = 0; i < kMelRegItems; ++i) {
        if (i < mel_cnt) {
          held[i] = mel_item(mi_b[i].z, mi_a[i].y, mi_a[i].z, mi_cf[i], mi_ko[i]);
          if (mi_a[i].w == 0) mel_finish(held[i], mi_a[i].x, lmh[i]);
          else if (mi_a[i].w == 1) *reinterpret_cast<f32x4*>(RB + mi_b[i].x * 256 + lane_m * 4) = held[i];
        }
      }
      for (int i = kMelRegItems; i < mel_cnt; ++i) {            // n_mels > 128: whole groups, tables re-read
        const int4 ia = tb.mel_items[(wave * kMelMaxItems + i) * 2];
        const f32x4 acc = mel_item(tb.mel_grp[ia.x].x, ia.y, ia.z, tb.mel_coef[ia.x * 16 + f16], tb.mel_koff[ia.x * 16 + f16]);
        mel_finish(acc, ia.x, nullptr);
      }
    }
    stamp(ST_MEL);
    if (tb.mel_n_slots > 0) {                                    // uniform for the whole grid
      AFX_LDS_BARRIER();
      stamp(ST_BAR3);
      if (mel_on) {
#pragma unroll
        for (int i = 0; i < kMelRegItems; ++i) {
          if (i < mel_cnt && mi_a[i].w == 2) {
            f32x4 acc = held[i];
            for (int sl = 0; sl < mi_b[i].y; ++sl)
              acc += *reinterpret_cast<const f32x4*>(RB + (mi_b[i].x + sl) * 256 + lane_m * 4);
            mel_finish(acc, mi_a[i].x, lmh[i]);
          }
        }
      }
    }
    stamp(ST_MELFIN);
    if (mel_on) {
      pend = true;
      pend_lmax = lmax; pend_slot = cur.frame_slot; pend_t0 = cur.t0; pend_T = cur.T; pend_clip = cur.clip;
    }
    // no barrier here: the next staging writes only S_ (dead since the barrier above) and PB is
    // rewritten only after the next iteration's first barrier.
    cur = nxt;
  }
  if constexpr (STAMP) { if (lane == 0) for (int i = 0; i < ST_COUNT; ++i) stamps[((size_t)blockIdx.x * kWaves + wave) * ST_COUNT + i] = st_sum[i]; }
  flush_logmel();
}

// Round 1's 4-wave kernel for 1024 / 256.  Since round 2 the wave-level k_frames3 serves that shape; this one is reachable only
// through AFX_NO_FRAMES3 and is compiled only into the diagnostic library (make dbg: -DAFX_WITH_FRAMES2), not into libafx.so.
#ifdef AFX_WITH_FRAMES2
// ---------------------------------------------------------------------------
// k_frames2: the n_fft = 1024 kernel.  Differences from the generic k_frames above:
//   * two real frames ride one 1024-point complex FFT (z = xA + i*xB): X_A[k], X_B[k] follow from
//     Z[k] and Z[N-k] by adds only -- no split twiddles, and no third ("mirror") exchange, because the
//     last radix-8 pass gives each lane the butterflies j and 128-j, i.e. both Z[k] and Z[N-k];
//   * schedule 16 x 8 x 8 on one wave (16 points per lane): two LDS exchanges of 8 KB per frame pair,
//     image XOR-swizzled a ^ ((a>>4)&15) -> every ds_write_b64 / ds_read_b64 conflict-free;
//   * no staging pass: a lane fetches 16-byte sample quads straight from global memory and re-cuts them into
//     rows through the wave's idle exchange image (pre-emphasis with the predecessor read at offset -1 of the
//     same image); the next block's quads fly under the second pair's FFT and the mel phase;
//   * the window (x 0.5) sits in 16 registers per lane; the exchange-image slot bases are rebuilt per pair from
//     an opaque copy of the lane id -- hoisted out of the block loop their XOR variants cost 30 registers, and
//     the kernel's register budget decides whether the other streams' small kernels fit beside it;
//   * mel: filters in octs, a lane walks one filter for the two frames of a pair (8-byte PB reads, packed FMAs).
// LDS: exchange images 32 KB + power-spectrum buffer 36.3 KB + twiddle tables 4.5 KB + mel taps 5.1 KB = 78 KB.
// ---------------------------------------------------------------------------
constexpr int kPb2Stride = 18;  // k_frames2's power-spectrum rows: 8 frame pairs + 1 pad pair.  Even, so that a pair is one
                                // aligned 8-byte access; 18 l mod 32 is a permutation of the even banks for 16 lanes
struct Lds2 { int ex, pb, t2, t3, tp, total; };         // float offsets
__host__ __device__ inline Lds2 lds2_layout(int ntaps) {
  Lds2 L;
  L.ex = 0;
  L.pb = L.ex + kWaves * 1024 * 2;
  L.t2 = L.pb + round4((513 + kPbPadRows) * kPb2Stride);    // pass-2 twiddles: 128 float2
  L.t3 = L.t2 + 256;                        // last-pass twiddles of butterfly jb: 7 x 64 float2
  L.tp = L.t3 + 7 * 64 * 2;                 // mel tap weights, oct-padded (sized by the plan's table)
  L.total = L.tp + round4(ntaps);
  return L;
}
size_t frames2_lds_bytes(int ntaps) { return (size_t)lds2_layout(ntaps).total * sizeof(float); }

// radix-16 DFT in registers as 4 x 4 with the W16 twiddles as constants
__device__ __forceinline__ void dft16(float2* x) {
  const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
  float2 t[4][4];                       // t[k1][n2]
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) {
    float2 a = x[n2], b = x[n2 + 4], c = x[n2 + 8], d = x[n2 + 12];
    dft4(a, b, c, d);
    t[0][n2] = a; t[1][n2] = b; t[2][n2] = c; t[3][n2] = d;
  }
  // t[k1][n2] *= W16^(n2*k1)
  auto mulw = [&](float2 v, float wr, float wi) { return make_float2(v.x * wr - v.y * wi, v.x * wi + v.y * wr); };
  t[1][1] = mulw(t[1][1], c1, -s1);                                   // W16^1
  t[1][2] = make_float2((t[1][2].x + t[1][2].y) * h, (t[1][2].y - t[1][2].x) * h);   // W16^2 = W8^1
  t[1][3] = mulw(t[1][3], s1, -c1);                                   // W16^3
  t[2][1] = make_float2((t[2][1].x + t[2][1].y) * h, (t[2][1].y - t[2][1].x) * h);   // W16^2
  t[2][2] = mul_mi(t[2][2]);                                          // W16^4 = -i
  t[2][3] = make_float2((t[2][3].y - t[2][3].x) * h, (-t[2][3].x - t[2][3].y) * h);  // W16^6 = W8^3
  t[3][1] = mulw(t[3][1], s1, -c1);                                   // W16^3
  t[3][2] = make_float2((t[3][2].y - t[3][2].x) * h, (-t[3][2].x - t[3][2].y) * h);  // W16^6
  t[3][3] = mulw(t[3][3], -c1, s1);                                   // W16^9 = -W16^1
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    float2 a = t[k1][0], b = t[k1][1], c = t[k1][2], d = t[k1][3];
    dft4(a, b, c, d);
    x[k1] = a; x[k1 + 4] = b; x[k1 + 8] = c; x[k1 + 12] = d;
  }
}

// DBG: the timing-only ablation switches (AFX_DEBUG_SKIP) are compiled in; the production instantiation does not
// carry the flag word or its branches
template <int FMT, bool STAMP, bool DBG>
__global__ __launch_bounds__(256, 2) void k_frames2(const void* __restrict__ samples,
                                                    ClipInfo* __restrict__ info,
                                                    const BlockDesc* __restrict__ blocks, int nblocks,
                                                    DevTables tb, KParams kp,
                                                    float* __restrict__ logmel,
                                                    float* __restrict__ rms_rows,
                                                    unsigned long long* __restrict__ stamps) {
  constexpr int N = 1024, NB = 513;
  unsigned long long st_sum[ST_COUNT] = {}, st_prev = 0;
  auto stamp = [&](int ph) {
    if constexpr (STAMP) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_sched_barrier(0);
      if (ph >= 0) st_sum[ph] += t - st_prev;
      st_prev = t;
    }
  };
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Lds2 L = lds2_layout(tb.mel_ntaps);
  const int hop = kp.hop, M = kp.n_mels;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float2* const EX = reinterpret_cast<float2*>(smem + L.ex) + wave * 1024;
  float* const PB = smem + L.pb;
  float2* const T2 = reinterpret_cast<float2*>(smem + L.t2);
  float2* const T3 = reinterpret_cast<float2*>(smem + L.t3);
  float* const TP = smem + L.tp;

  // ---- once per workgroup: tables -> LDS, per-lane twiddles -> registers
  for (int i = tid; i < kPbPadRows * kPb2Stride; i += 256) PB[NB * kPb2Stride + i] = 0.f;
  const float2* w1024 = reinterpret_cast<const float2*>(tb.post);      // exp(-2 pi i k / 1024), k < 512
  auto W = [&](int m) {                                                  // W_1024^m, 0 <= m < 1024
    const float2 v = w1024[m & 511];
    return (m & 512) ? make_float2(-v.x, -v.y) : v;
  };
  if (tid < 128) T2[tid] = W(8 * (tid >> 4) * (tid & 15));              // pass-2 twiddles W_128^(c*r) at [r*16 + c]: a row per r,
                                                                         // so the 16 distinct c of a wave read 128 contiguous bytes
  float wreg[16];                          // this lane's 16 window values (w[n] = w[N - n]) x 0.5: the A/B split then needs no 1/2
#pragma unroll
  for (int u = 0; u < 16; ++u) wreg[u] = 0.5f * tb.window[u < 8 ? lane + 64 * u : (64 - lane) + 64 * (15 - u)];
  const int ja = lane, jb = lane ? 128 - lane : 64;                      // last-pass butterflies of this lane
  float2 tw3a[7];                                                        // butterfly ja: registers; jb: LDS table
#pragma unroll
  for (int r = 1; r < 8; ++r) tw3a[r - 1] = W(ja * r);
  if (tid < 64) {
#pragma unroll
    for (int r = 1; r < 8; ++r) T3[(r - 1) * 64 + tid] = W(jb * r);
  }
  for (int i = tid; i < tb.mel_ntaps; i += 256) TP[i] = tb.mel_taps[i];
  // exchange-image slots (see header): all per-lane bases

  // mel octs (8 filters) of this wave: oct NO-1 - (4*it + (it odd ? 3 - wave : wave)), it = 0..3 -- a snake over the
  // octs from the widest down, which balances the four waves because tap counts grow with the filter index.
  // Computed where used from `wave` (never kept per oct: that would cost scalar registers).
  constexpr int kOctsPerWave = kMelMaxOcts / 4;
  const int NO = (M + 7) >> 3;
  auto oct_of = [&](int wv, int it) { return NO - 1 - (4 * it + ((it & 1) ? 3 - wv : wv)); };
  const bool pre = (kp.flags & AFX_FLAG_PREEMPH) != 0;
  const float b1 = kp.preemph_b1;

  auto fetch_desc = [&](int b) -> int {
    const int bb = b < nblocks ? b : nblocks - 1;
    return reinterpret_cast<const int*>(blocks + bb)[lane & 15];
  };
  auto resolve = [&](int w, int b) -> BlkCtx {
    BlkCtx c;
    auto rl = [&](int i) { return __builtin_amdgcn_readlane(w, i); };
    auto rl64 = [&](int i) { return (int64_t)(((uint64_t)(uint32_t)rl(i + 1) << 32) | (uint32_t)rl(i)); };
    c.sample_base = rl64(0); c.frame_slot = rl64(2); c.clip_off = rl64(4);
    c.keep_lo = rl(6); c.keep_hi = rl(7); c.have_lo = rl(8); c.have_hi = rl(9);
    c.clip = rl(10); c.t0 = 0; c.T = rl(12) - rl(11);     // only T - t0 (frames left from this block on) is used here
    c.active = (b < nblocks) && rl(13) != 0;
    c.interior = false;
    return c;
  };

  // deferred log-mel stores (see k_frames)
  float2 lmh[kOctsPerWave];                 // log-mel of this lane's filter for its two frames, per oct
  bool pend = false;
  float pend_lmax = -INFINITY;
  int64_t pend_slot = 0;
  int pend_t0 = 0, pend_T = 0, pend_clip = 0;
  auto flush_logmel = [&]() {
    if (!pend) return;
    pend = false;
    int lane_f = lane;
    asm volatile("" : "+v"(lane_f));
    const int fp2 = (lane_f & 7) * 2, j8 = lane_f >> 3;               // frames fp2, fp2 + 1; filter j8 of the oct
    const bool v0 = (pend_t0 + fp2) < pend_T, v1 = (pend_t0 + fp2 + 1) < pend_T;
    // tile layout [mel/4][frame][mel%4]
    float* tile = logmel + pend_slot * (int64_t)M + (j8 >> 2) * 64 + fp2 * 4 + (j8 & 3);
    int wv = wave;
    asm volatile("" : "+s"(wv));
#pragma unroll
    for (int i = 0; i < kOctsPerWave; ++i) {
      const int o = oct_of(wv, i);
      if (o >= 0 && o * 8 + j8 < M) {
        if (v0) tile[o * 128] = lmh[i].x;
        if (v1) tile[o * 128 + 4] = lmh[i].y;
      }
    }
    const float mx = wave_max(pend_lmax);
    if (lane_f == 0 && mx > -INFINITY) atomicMax(&info[pend_clip].lmax_ord, f2ord(mx));
  };

  // ---- sample rows.  A lane holds samples l + 64u ("row" u).  hop = 256 = 4 rows, so frame B of a pair is
  // frame A shifted by 4 rows and the wave's second pair starts 8 rows after the first: one window of
  // 20 rows serves pair 0, 12 of them plus 8 new rows serve pair 1.  Rows are fetched raw (next block's
  // 20 rows under the second pair's FFT and the mel phase; the 8 new rows under the first pair's FFT),
  // pre-emphasised once in place, and shared by the two frames of a pair.
  // Fetch: 16-byte loads (the texture-address unit charges per instruction, ~16 cycles per wave, whatever
  // the width): lane l takes samples 4l..4l+3 of each 256-sample chunk -- 5 chunks for a 20-row window,
  // 2 for the 8 new rows -- and the rows are re-cut through this wave's idle exchange image:
  // written as 16-byte quads at float offset 4, read back as rows (offset 4) and as their left
  // neighbours (offset 3; slot 3 holds the sample before the window).  S16 clips fetch 8 bytes per quad.
  float rows[20], inc[8];
  float4 qrows[5], qinc[2];
  float np_rows = 0.f, np_inc = 0.f;
  bool rows_raw = false, inc_raw = false;              // false: the pair is an edge pair (clamped path)
  auto raw_ld = [&](int64_t idx) -> float {            // one converted sample (edge path, predecessors)
    if constexpr (FMT == AFX_FMT_S16) return (float)((const int16_t*)samples)[idx] * (1.0f / 32768.0f);
    else return ((const float*)samples)[idx];
  };
  auto quad_ld = [&](int64_t idx) -> float4 {          // 4 consecutive samples, idx % 4 == 0
    if constexpr (FMT == AFX_FMT_S16) {
      const int2 q = *reinterpret_cast<const int2*>((const int16_t*)samples + idx);
      const float sc = 1.0f / 32768.0f;
      return make_float4((float)(short)(q.x & 0xffff) * sc, (float)(short)(q.x >> 16) * sc,
                         (float)(short)(q.y & 0xffff) * sc, (float)(short)(q.y >> 16) * sc);
    } else return *reinterpret_cast<const float4*>((const float*)samples + idx);
  };
  auto pair_is_interior = [&](const BlkCtx& c, int fl) -> bool {       // both frames: all samples and their
    const int j0 = fl * hop, j1 = j0 + hop + N;                        // predecessors exist and are kept,
    return (j0 - 1 >= c.have_lo) && (j1 <= c.have_hi) && (j0 >= c.keep_lo) && (j1 <= c.keep_hi) &&
           ((c.sample_base & 3) == 0);                                 // and the quads are aligned
  };
  auto issue_rows = [&](const BlkCtx& c) {             // first pair of block c: rows 0..19
    rows_raw = c.active && pair_is_interior(c, wave * 4);
    if (rows_raw && !(DBG && (kp.flags & 0x100))) {             // 0x100: timing-only ablation (stale registers)
      const int64_t ba = c.sample_base + (int64_t)(wave * 4) * hop;
#pragma unroll
      for (int ch = 0; ch < 5; ++ch) qrows[ch] = quad_ld(ba + 4 * lane + 256 * ch);
      np_rows = raw_ld(ba - 1);                        // wave-uniform address
    }
  };
  auto issue_inc = [&](const BlkCtx& c) {              // second pair: its 8 new rows (20..27 of the window)
    inc_raw = pair_is_interior(c, wave * 4 + 2);
    if (inc_raw && !(DBG && (kp.flags & 0x100))) {
      const int64_t ba = c.sample_base + (int64_t)(wave * 4) * hop + 64 * 20;
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) qinc[ch] = quad_ld(ba + 4 * lane + 256 * ch);
      np_inc = raw_ld(ba - 1);
    }
  };
  float* const XB = reinterpret_cast<float*>(EX);
  // quads q[0..NCH) + predecessor p0 -> rows r[0..4*NCH), pre-emphasised when `pre`
  auto cut_rows = [&](const float4* q, float p0, float* r, auto NCHt) {
    constexpr int NCH = decltype(NCHt)::value, NR = 4 * NCH;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) *reinterpret_cast<float4*>(XB + 4 + 4 * lane + 256 * ch) = q[ch];
    XB[3] = p0;
    AFX_CBARRIER();
    float prev[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) { r[u] = XB[4 + 64 * u + lane]; prev[u] = XB[3 + 64 * u + lane]; }
    AFX_CBARRIER();
    if (pre) {
#pragma unroll
      for (int u = 0; u < NR; ++u) r[u] = preemph1(r[u], prev[u], b1);
    }
  };
  auto edge_sample = [&](const BlkCtx& c, int j) -> float {            // pre-emphasised, trim-masked sample j
    const int lo = c.have_lo, hi = c.have_hi - 1;
    const int jc = j < lo ? lo : (j > hi ? hi : j), jp = (j - 1) < lo ? lo : ((j - 1) > hi ? hi : (j - 1));
    const float y = (jc == j) ? raw_ld(c.sample_base + jc) : 0.f;
    const float yp = (jp == j - 1) ? raw_ld(c.sample_base + jp) : 0.f;
    float v = y;
    if (pre) {
      v = preemph1(y, yp, b1);
      if (j == lo) v = preemph0(raw_ld(c.clip_off), raw_ld(c.clip_off + 1));   // clip sample 0
    }
    return (j >= c.keep_lo && j < c.keep_hi) ? v : 0.f;
  };

  // ---- one pair: z = w*yA + i*w*yB -> 1024-point FFT -> |X_A|^2, |X_B|^2 into PB columns flA, flA+1
  auto fft_pair = [&](float2 (&v)[16], int flA) {
    stamp(ST_STAGE);
    // exchange-image slots, rebuilt per pair from an opaque copy of the lane id: kept across the block loop they
    // would hold five registers that the frame kernel's 224-register budget does not have
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int sA1 = 16 * ln + (ln & 15);
    const int sR1 = ln ^ (ln >> 4);
    const int sW2a = 128 * (ln >> 4) + ((ln & 15) ^ (((ln >> 4) & 1) << 3));
    const int jbl = ln ? 128 - ln : 64;
    const int sRa = ln ^ ((ln >> 4) & 7), sRb = jbl ^ ((jbl >> 4) & 7);
    // pass 1: radix 16 (no twiddles), exchange
    dft16(v);
#pragma unroll
    for (int r = 0; r < 16; ++r) EX[sA1 ^ r] = v[r];
    AFX_CBARRIER();
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = EX[(sR1 ^ ((u & 3) << 2)) + 64 * u];
    AFX_CBARRIER();
    stamp(ST_BAR1);
    // pass 2: radix 8 x 2 butterflies (j = lane, lane + 64): inputs u = i + 2r, twiddle W_128^((lane&15) r)
    {
      float2 t2v[8];
#pragma unroll
      for (int r = 1; r < 8; ++r) t2v[r] = T2[r * 16 + (lane & 15)];
      float2 xa[8], xb[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) { xa[r] = v[2 * r]; xb[r] = v[2 * r + 1]; }
#pragma unroll
      for (int r = 1; r < 8; ++r) { xa[r] = cmul(xa[r], t2v[r]); xb[r] = cmul(xb[r], t2v[r]); }
      dft<8>(xa); dft<8>(xb);
#pragma unroll
      // butterfly j = lane + 64 sits 512 slots further (x = (lane >> 4) + 4, same parity; 17 r and sW2a are < 512)
      for (int r = 0; r < 8; ++r) { EX[sW2a ^ (17 * r)] = xa[r]; EX[(sW2a ^ (17 * r)) + 512] = xb[r]; }
    }
    AFX_CBARRIER();
    stamp(ST_PREFETCH);
    // pass 3: radix 8, butterflies ja = lane and jb = 128 - lane (lane 0: 0 and 64)
    float2 A[8], B[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      A[r] = EX[(sRa ^ ((r & 1) << 3)) + 128 * r];
      B[r] = EX[(sRb ^ ((r & 1) << 3)) + 128 * r];
    }
    AFX_CBARRIER();
#pragma unroll
    for (int r = 1; r < 8; ++r) { A[r] = cmul(A[r], tw3a[r - 1]); B[r] = cmul(B[r], T3[(r - 1) * 64 + lane]); }
    dft<8>(A); dft<8>(B);
    // A[r] = Z[lane + 128 r], B[r] = Z[128 - lane + 128 r]: pair s holds Z[k], Z[N-k] with k = lane + 128 s.
    // Lane 0 owns the self-mirrored butterflies 0 and 64 and pairs inside them.
    const bool l0 = lane == 0;
    float* const pcol = PB + flA;
    auto power = [&](float2 za, float2 zb, int bin) {
      const float ar = za.x + zb.x, ai = za.y - zb.y, br = za.y + zb.y, bi = za.x - zb.x;
      *reinterpret_cast<float2*>(pcol + bin * kPb2Stride) =        // |X_A[bin]|^2, |X_B[bin]|^2: flA is even
          make_float2(ar * ar + ai * ai, br * br + bi * bi);
    };
    auto sel = [&](float2 a, float2 b) { return make_float2(l0 ? b.x : a.x, l0 ? b.y : a.y); };
    power(sel(A[0], A[1]), sel(B[7], A[7]), l0 ? 128 : lane);
    power(sel(A[1], A[2]), sel(B[6], A[6]), l0 ? 256 : lane + 128);
    power(sel(A[2], A[3]), sel(B[5], A[5]), l0 ? 384 : lane + 256);
    power(sel(A[3], B[0]), sel(B[4], B[7]), l0 ? 64 : lane + 384);
    power(sel(A[4], B[1]), sel(B[3], B[6]), l0 ? 192 : 512 - lane);
    power(sel(A[5], B[2]), sel(B[2], B[5]), l0 ? 320 : 384 - lane);
    power(sel(A[6], B[3]), sel(B[1], B[4]), l0 ? 448 : 256 - lane);
    power(sel(A[7], A[0]), sel(B[0], A[0]), l0 ? 0 : 128 - lane);
    if (l0) {                                              // Nyquist bin from Z[512] = A[4]
      *reinterpret_cast<float2*>(pcol + 512 * kPb2Stride) = make_float2(4.f * A[4].x * A[4].x, 4.f * A[4].y * A[4].y);
    }
    AFX_CBARRIER();
    stamp(ST_FFT);
  };
  // window the pair whose frame A is rows y[0..15] and frame B rows y[4..19] (RMS rows come from k_trim_decide)
  auto make_z = [&](const float (&y)[20], float2 (&v)[16], const BlkCtx& c, int flA) {
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = make_float2(wreg[u] * y[u], wreg[u] * y[u + 4]);
  };

  // experiment (AFX_DEBUG_SKIP bits 0x1000 / 0x2000): start half of the workgroups ~half a block late
  if (DBG && (((kp.flags & 0x1000) && blockIdx.x >= gridDim.x / 2) || ((kp.flags & 0x2000) && (blockIdx.x & 1)))) {
    for (int i = 0; i < 2; ++i) __builtin_amdgcn_s_sleep(127);
  }
  BlkCtx cur = resolve(fetch_desc(blockIdx.x), blockIdx.x);
  int dnext = fetch_desc(blockIdx.x + gridDim.x);
  issue_rows(cur);
  AFX_LDS_BARRIER();

  for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
    stamp(-1);
    const BlkCtx nxt = resolve(dnext, b + gridDim.x);
    dnext = fetch_desc(b + 2 * gridDim.x);

    if (cur.active && !(DBG && (kp.flags & 0x200))) {
      const int fl0 = wave * 4;
      float2 v[16];
      // ---- pair 0 (frames fl0, fl0+1): rows 0..19
      if (rows_raw) {
        cut_rows(qrows, np_rows, rows, std::integral_constant<int, 5>());
      } else {
        // edge pair (clip start/end, trimmed span): clamped loads in a rolled loop (low register
        // pressure, rare), parked in the exchange image and read back with static row indices
#pragma unroll 1
        for (int u = 0; u < 20; ++u) XB[64 * u + lane] = edge_sample(cur, fl0 * hop + lane + 64 * u);
        AFX_CBARRIER();
#pragma unroll
        for (int u = 0; u < 20; ++u) rows[u] = XB[64 * u + lane];
        AFX_CBARRIER();
      }
      stamp(ST_X0);
      make_z(rows, v, cur, fl0);
      stamp(ST_X1);
      flush_logmel();                        // older than every load issued from here on
      stamp(ST_X2);
      issue_inc(cur);
      fft_pair(v, fl0);
      // ---- pair 1 (frames fl0+2, fl0+3): rows 8..27 = 12 kept rows + the 8 fetched ones
      float r1[20];
#pragma unroll
      for (int u = 0; u < 12; ++u) r1[u] = rows[u + 8];
      if (inc_raw) {
        cut_rows(qinc, np_inc, inc, std::integral_constant<int, 2>());
#pragma unroll
        for (int u = 0; u < 8; ++u) r1[12 + u] = inc[u];
      } else {
#pragma unroll 1
        for (int u = 0; u < 20; ++u) XB[64 * u + lane] = edge_sample(cur, (fl0 + 2) * hop + lane + 64 * u);
        AFX_CBARRIER();
#pragma unroll
        for (int u = 0; u < 20; ++u) r1[u] = XB[64 * u + lane];
        AFX_CBARRIER();
      }
      stamp(ST_X0);
      make_z(r1, v, cur, fl0 + 2);
      stamp(ST_X1);
      issue_rows(nxt);                       // next block's first pair lands under this FFT and the mel phase
      fft_pair(v, fl0 + 2);
    } else {
      flush_logmel();
      issue_rows(nxt);
    }
    stamp(ST_FFT);
    AFX_LDS_BARRIER();
    stamp(ST_BAR2);

    // ---- mel filterbank + dB on the vector pipe.  A mel row touches only its own ~2..60 bins, so as a matrix
    // product it is >85 % zeros even block-sparse; here every multiply is a real tap.  Filters are taken eight
    // at a time (an oct): lane (pair fp, j) walks the taps of filter 8*oct + j for the frames 2fp and 2fp+1 -- one
    // 8-byte P read (both frames), a quarter of a 16-byte weight read (shared by the oct's 8 lanes of that filter)
    // and one packed FMA per tap.
    const bool mel_on = cur.active && !(DBG && (kp.flags & 0x400));
    if (mel_on) {
      int lane_m = lane;
      asm volatile("" : "+v"(lane_m));
      const int fp2 = (lane_m & 7) * 2, j8 = lane_m >> 3;
      const bool v0 = (cur.t0 + fp2) < cur.T, v1 = (cur.t0 + fp2 + 1) < cur.T;
      float lmax = -INFINITY;
      int wv = wave;
      asm volatile("" : "+s"(wv));
      int mw[kOctsPerWave];
#pragma unroll
      for (int it = 0; it < kOctsPerWave; ++it) { const int o = oct_of(wv, it); mw[it] = o >= 0 ? tb.mel_meta[8 * o + j8] : 0; }
#pragma unroll
      for (int it = 0; it < kOctsPerWave; ++it) {
        const int o = oct_of(wv, it);
        if (o >= 0) {
          const int n4 = __builtin_amdgcn_readfirstlane(mw[it] >> 10) & 31;   // same for the eight filters of an oct, >= 1
          const float* p = PB + (mw[it] & 1023) * kPb2Stride + fp2;
          const float4* w = reinterpret_cast<const float4*>(TP + (mw[it] >> 15));
          float2 a0 = make_float2(0.f, 0.f), a1 = make_float2(0.f, 0.f);
          auto ld = [&](float2 (&xx)[4], float4& c, int bi) {
#pragma unroll
            for (int t = 0; t < 4; ++t) xx[t] = *reinterpret_cast<const float2*>(p + (4 * bi + t) * kPb2Stride);
            c = w[bi];
          };
          auto fm = [&](const float2 (&xx)[4], const float4& c) {
            a0.x = fmaf(c.x, xx[0].x, a0.x); a0.y = fmaf(c.x, xx[0].y, a0.y);
            a1.x = fmaf(c.y, xx[1].x, a1.x); a1.y = fmaf(c.y, xx[1].y, a1.y);
            a0.x = fmaf(c.z, xx[2].x, a0.x); a0.y = fmaf(c.z, xx[2].y, a0.y);
            a1.x = fmaf(c.w, xx[3].x, a1.x); a1.y = fmaf(c.w, xx[3].y, a1.y);
          };
          float2 x0[4], x1[4];
          float4 c0, c1;
          ld(x0, c0, 0);
          int bi = 1;
          for (; bi + 1 < n4; bi += 2) {                    // two batches per turn, the next pair in flight
            ld(x1, c1, bi);
            fm(x0, c0);
            ld(x0, c0, bi + 1);
            fm(x1, c1);
          }
          if (bi < n4) { ld(x1, c1, bi); fm(x0, c0); fm(x1, c1); }
          else fm(x0, c0);
          const float L0 = 3.01029995663981195f * __builtin_amdgcn_logf(fmaxf(kp.amin, a0.x + a1.x));
          const float L1 = 3.01029995663981195f * __builtin_amdgcn_logf(fmaxf(kp.amin, a0.y + a1.y));
          lmh[it] = make_float2(L0, L1);
          if (o * 8 + j8 < M) {
            if (v0) lmax = fmaxf(lmax, L0);
            if (v1) lmax = fmaxf(lmax, L1);
          }
        }
      }
      pend = true;
      pend_lmax = lmax; pend_slot = cur.frame_slot; pend_t0 = cur.t0; pend_T = cur.T; pend_clip = cur.clip;
    }
    stamp(ST_MEL);
    AFX_LDS_BARRIER();      // PB is free for the next block's spectra
    stamp(ST_BAR3);
    stamp(ST_MELFIN);
    cur = nxt;
  }
  if constexpr (STAMP) { if (lane == 0) for (int i = 0; i < ST_COUNT; ++i) stamps[((size_t)blockIdx.x * kWaves + wave) * ST_COUNT + i] = st_sum[i]; }
  flush_logmel();
}
#endif  // AFX_WITH_FRAMES2

// ---------------------------------------------------------------------------
// k_dct: clamp at (clip max - top_db), ortho DCT-II on the matrix pipe.
// One wave per 16-frame log-mel tile: the tile's [mel][16 frames] layout is exactly the
// MFMA B-operand order, so each k-step is one coalesced 256-byte load.
// ---------------------------------------------------------------------------
template <int NCG, bool FM>
__global__ __launch_bounds__(256) void k_dct(const ClipDesc* __restrict__ clips,
                                             const ClipInfo* __restrict__ info,
                                             const float* __restrict__ dctA, KParams kp,
                                             const float* __restrict__ logmel,
                                             float* __restrict__ mfcc, int spec) {
  const int clip = blockIdx.y;
  const ClipInfo ci = info[clip];
  if (ci.status != AFX_CLIP_OK) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t0 = (blockIdx.x * 4 + wave) * 16;
  if (t0 >= ci.T) return;
  const ClipDesc cd = clips[clip];
  const int M = kp.n_mels, K = kp.n_mfcc, NI = M >> 2;
  const float theta = ord2f(ci.lmax_ord) - kp.top_db;
  const int f = lane & 15, q = lane >> 4;
  // tile layout [mel/4][frame][mel%4]: B[k = q][j = f] of k-step i is at 64 i + 4 f + q (one 256-B row per step)
  // FM (k_frames3's spill): frame-major [frame][mel], B[k = q][j = f] of k-step i is mel 4 i + q of frame t0 + f
  // spec: the spill holds absolute frames (k_frames3 ran before the trim decision); trimmed frame t is frame start / hop + t
  const int g0 = spec ? (int)(ci.start / kp.hop) : 0;
  const float* tile = FM ? logmel + (cd.frame_base + g0 + t0 + f) * (int64_t)M + q
                         : logmel + (cd.frame_base + t0) * (int64_t)M + f * 4 + q;
  f32x4 acc[NCG];
#pragma unroll
  for (int c = 0; c < NCG; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int i = 0; i < NI; ++i) {
    const float Lc = fmaxf(tile[FM ? i * 4 : i * 64], theta);
#pragma unroll
    for (int c = 0; c < NCG; ++c)
      acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(dctA[((int64_t)c * NI + i) * 64 + lane], Lc, acc[c], 0, 0, 0);
  }
  if (t0 + f < ci.T) {
    float* out = mfcc + cd.frame_base * (int64_t)K + t0 + f;
#pragma unroll
    for (int c = 0; c < NCG; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = c * 16 + q * 4 + r;
        if (k < K) out[(int64_t)k * cd.tpad] = acc[c][r];
      }
  }
}

// k_dct16<NCG>: the same for n_mels % 16 == 0 and n_mfcc <= 16 NCG <= 48 (the reference's 128 / 13).  A wave takes kDctTiles
// tiles; lane (f, q) fetches filters 16 s + 4 q + {0..3} of frame f with one 16-byte load (a wave-load is 1 KB
// contiguous), all loads of its tiles issued before the first use; the DCT matrix sits in registers.
constexpr int kDctTiles = 1;
template <int NCG, bool FM>
__global__ __launch_bounds__(256) void k_dct16(const ClipDesc* __restrict__ clips,
                                               const ClipInfo* __restrict__ info,
                                               const float* __restrict__ dctP, KParams kp,
                                               const float* __restrict__ logmel,
                                               float* __restrict__ mfcc, int spec) {
  const int clip = blockIdx.y;
  const ClipInfo ci = info[clip];
  if (ci.status != AFX_CLIP_OK) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile0 = (blockIdx.x * 4 + wave) * kDctTiles;
  if (tile0 * 16 >= ci.T) return;
  const ClipDesc cd = clips[clip];
  const int M = kp.n_mels, K = kp.n_mfcc, S = M >> 4;         // S <= 8
  const float theta = ord2f(ci.lmax_ord) - kp.top_db;
  const int f = lane & 15, q = lane >> 4;
  float4 x[kDctTiles][8];
#pragma unroll
  for (int j = 0; j < kDctTiles; ++j) {
    const int t0 = (tile0 + j) * 16;
    // tiles past the clip's last frame are not read (t0 is wave-uniform); [mel/4][frame][mel%4]: quad row 4 s + q
    // FM: frame-major [frame][mel] -- the same four filters 16 s + 4 q + {0..3} of frame f, 16 bytes at mel offset 16 s + 4 q
    const float* tile = FM ? logmel + (cd.frame_base + (spec ? (int)(ci.start / kp.hop) : 0) + t0 + f) * (int64_t)M + q * 4
                           : logmel + (cd.frame_base + t0) * (int64_t)M + (q * 16 + f) * 4;
#pragma unroll
    for (int s = 0; s < 8; ++s)
      x[j][s] = (s < S && t0 < ci.T) ? *reinterpret_cast<const float4*>(tile + s * (FM ? 16 : 256)) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float a[NCG][8][4];
#pragma unroll
  for (int g = 0; g < NCG; ++g)
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int c = 0; c < 4; ++c) a[g][s][c] = s < S ? dctP[((g * S + s) * 4 + c) * 64 + lane] : 0.f;
#pragma unroll
  for (int j = 0; j < kDctTiles; ++j) {
    const int t0 = (tile0 + j) * 16;
    if (t0 >= ci.T) break;
    f32x4 acc[NCG][2];
#pragma unroll
    for (int g = 0; g < NCG; ++g) { acc[g][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[g][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < S) {
        const float b0 = fmaxf(x[j][s].x, theta), b1 = fmaxf(x[j][s].y, theta);
        const float b2 = fmaxf(x[j][s].z, theta), b3 = fmaxf(x[j][s].w, theta);
#pragma unroll
        for (int g = 0; g < NCG; ++g) {
          acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][s][0], b0, acc[g][0], 0, 0, 0);
          acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][s][1], b1, acc[g][1], 0, 0, 0);
          acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][s][2], b2, acc[g][0], 0, 0, 0);
          acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][s][3], b3, acc[g][1], 0, 0, 0);
        }
      }
    }
    if (t0 + f < ci.T) {
      float* out = mfcc + cd.frame_base * (int64_t)K + t0 + f;
#pragma unroll
      for (int g = 0; g < NCG; ++g) {
        const f32x4 r4 = acc[g][0] + acc[g][1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = g * 16 + q * 4 + r;
          if (k < K) out[(int64_t)k * cd.tpad] = r4[r];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// k_stats: one wave per (clip, row); rows 0..K-1 = MFCC coefficients, row K = RMS
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stats(const ClipDesc* __restrict__ clips,
                                               const ClipInfo* __restrict__ info, KParams kp,
                                               const float* __restrict__ mfcc,
                                               const float* __restrict__ rms_rows,
                                               float* __restrict__ stats,
                                               float* __restrict__ frames_out,
                                               const int64_t* __restrict__ frame_offsets,
                                               ClipInfo* __restrict__ info_out) {
  const int clip = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int K = kp.n_mfcc;
  const int row = blockIdx.x * 4 + wave;
  if (row > K) return;
  const ClipInfo ci = info[clip];
  if (info_out && row == K && lane == 0) info_out[clip] = ci;      // the caller's copy (host memory the device can write)
  float* st = stats + (int64_t)clip * (4 * K + 3);
  const ClipDesc cd = clips[clip];
  // a clip with fewer than 9 frames fails the MFCC rows (librosa.feature.delta raises) but still has an RMS row:
  // extract_energy (F:153-179) only calls librosa.feature.rms
  const bool energy_only = ci.status == AFX_CLIP_TOO_SHORT && cd.len >= 2 && ci.T >= 1;   // RMS rows from the sub-block sums, or from k_trim_decide
  if (ci.status != AFX_CLIP_OK && !(energy_only && row == K)) {
    if (lane == 0) {
      if (row < K) { st[row] = 0.f; st[K + row] = 0.f; st[2 * K + row] = 0.f; st[3 * K + row] = 0.f; }
      else { st[4 * K] = 0.f; st[4 * K + 1] = 0.f; st[4 * K + 2] = 0.f; }
    }
    return;
  }
  const int T = ci.T;
  const double invT = 1.0 / (double)T;
  float* fo = frames_out ? frames_out + frame_offsets[clip] : nullptr;
  const int64_t fstride = cd.tmax;
  if (row < K) {
    const float* x = mfcc + cd.frame_base * (int64_t)K + (int64_t)row * cd.tpad;
    // the nine frames at either end, for the delta means below: fetched now, one per lane, handed to lane 0 later
    const float ev = x[lane < 9 ? lane : (lane < 18 ? T - 18 + lane : 0)];
    // one pass: sum and sum of squares in float64 (values up to ~1e3, T ~ 1e3: the squares' sum is exact to 1e-8 of
    // a variance term of 1e4 or more); sum (x - c)^2 = ss - 2 c s + T c^2 about the float32 mean c, as numpy's std
    double s = 0.0, ss = 0.0;
#pragma unroll 8
    for (int t = lane; t < T; t += 64) { const double v = (double)x[t]; s += v; ss = fma(v, v, ss); }
    s = wave_sum_d(s); ss = wave_sum_d(ss);
    const double mean = s * invT;
    const float meanf = (float)mean;
    const double c = (double)meanf;
    const double s2 = fmax(ss - 2.0 * c * s + (double)T * c * c, 0.0);
    if (fo) {
#pragma unroll 2
      for (int t = lane; t < T; t += 64) {
        // savgol_filter(width 9, polyorder=deriv=order, mode='interp'): interior taps; the
        // fitted edge polynomial has a constant derivative, so frames 0..3 / T-4..T-1 repeat
        // frame 4 / frame T-5.
        const int tc = t < 4 ? 4 : (t > T - 5 ? T - 5 : t);
        const float* cc = x + tc;
        const double d1 = (4.0 * ((double)cc[4] - (double)cc[-4]) + 3.0 * ((double)cc[3] - (double)cc[-3]) +
                           2.0 * ((double)cc[2] - (double)cc[-2]) + ((double)cc[1] - (double)cc[-1])) * (1.0 / 60.0);
        const double d2 = (28.0 * ((double)cc[4] + (double)cc[-4]) + 7.0 * ((double)cc[3] + (double)cc[-3]) -
                           8.0 * ((double)cc[2] + (double)cc[-2]) - 17.0 * ((double)cc[1] + (double)cc[-1]) -
                           20.0 * (double)cc[0]) * (1.0 / 462.0);
        fo[(int64_t)row * fstride + t] = x[t];
        fo[(int64_t)(K + row) * fstride + t] = (float)d1;
        fo[(int64_t)(2 * K + row) * fstride + t] = (float)d2;
      }
    }
    // Means of the two delta rows without the rows: both filters are differences, so their sum over the frames
    // telescopes to the nine frames at either end (the first filter) and the second one's weights cancel the bulk of
    // the row exactly (28 + 7 - 8 - 17 = 10 on either side of -20).  Frames 0..3 and T-4..T-1 repeat frame 4 / T-5.
    double h[9], g[9];                                    // h[i] = x[i], g[i] = x[T - 9 + i]
#pragma unroll
    for (int i = 0; i < 9; ++i) { h[i] = (double)__shfl(ev, i); g[i] = (double)__shfl(ev, 9 + i); }
    if (lane == 0) {
      auto D1 = [](const double* c) {
        return (4.0 * (c[4] - c[-4]) + 3.0 * (c[3] - c[-3]) + 2.0 * (c[2] - c[-2]) + (c[1] - c[-1])) * (1.0 / 60.0);
      };
      auto D2 = [](const double* c) {
        return (28.0 * (c[4] + c[-4]) + 7.0 * (c[3] + c[-3]) - 8.0 * (c[2] + c[-2]) - 17.0 * (c[1] + c[-1]) - 20.0 * c[0]) * (1.0 / 462.0);
      };
      const double w2[5] = {0.0, -17.0, -8.0, 7.0, 28.0};
      double sd1 = 0.0, sd2 = 0.0;
#pragma unroll
      for (int k = 1; k <= 4; ++k) {
        double head = 0.0, tail = 0.0, hl = 0.0, hr = 0.0, tl = 0.0, tr = 0.0;
#pragma unroll
        for (int i = 4 - k; i <= 3 + k; ++i) head += h[i];          // x[4 - k .. 3 + k]
#pragma unroll
        for (int i = 5 - k; i <= 4 + k; ++i) tail += g[i];          // x[T - 4 - k .. T - 5 + k]
#pragma unroll
        for (int i = 4 - k; i <= 3; ++i) hl += h[i];
#pragma unroll
        for (int i = 4; i <= 3 + k; ++i) hr += h[i];
#pragma unroll
        for (int i = 5 - k; i <= 4; ++i) tl += g[i];
#pragma unroll
        for (int i = 5; i <= 4 + k; ++i) tr += g[i];
        sd1 += (double)k * (tail - head);
        sd2 += w2[k] * ((hl - hr) + (tr - tl));
      }
      sd1 = sd1 * (1.0 / 60.0) + 4.0 * ((double)(float)D1(h + 4) + (double)(float)D1(g + 4));
      sd2 = sd2 * (1.0 / 462.0) + 4.0 * ((double)(float)D2(h + 4) + (double)(float)D2(g + 4));
      st[row] = meanf;
      st[K + row] = (float)sqrt(s2 * invT);
      st[2 * K + row] = (float)(sd1 * invT);
      st[3 * K + row] = (float)(sd2 * invT);
    }
  } else {
    const float* r = rms_rows + cd.frame_base;
    double s = 0.0;
    float mx = -INFINITY, mn = INFINITY;
    for (int t = lane; t < T; t += 64) {
      const float v = r[t];
      s += (double)v; mx = fmaxf(mx, v); mn = fminf(mn, v);
      if (fo) fo[(int64_t)(3 * K) * fstride + t] = v;
    }
    const double mean = wave_sum_d(s) * invT;
    const float meanf = (float)mean;
    mx = wave_max(mx); mn = wave_min(mn);
    double s2 = 0.0;
    for (int t = lane; t < T; t += 64) { const float d = r[t] - meanf; s2 += (double)d * (double)d; }
    s2 = wave_sum_d(s2);
    if (lane == 0) {
      st[4 * K] = meanf;
      st[4 * K + 1] = (float)sqrt(s2 * invT);
      st[4 * K + 2] = mx - mn;
    }
  }
}

// preprocess_audio(y): the pre-emphasised signal itself (F:69), one clip
__global__ __launch_bounds__(256) void k_preemph(const float* __restrict__ y, float* __restrict__ out,
                                                 int64_t n, float b1) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  out[i] = (i == 0) ? ((n > 1) ? preemph0(y[0], y[1]) : y[0]) : preemph1(y[i], y[i - 1], b1);
}

constexpr int kF0PrepPerThread = 8;
// extract_f0 staging: the preprocessed signal itself (pre-emphasised, trimmed span moved to the clip's
// offset) as float32 -- what the reference hands to librosa.pyin (feature_extractor.py:195).
__global__ __launch_bounds__(256) void k_f0_prep(const void* __restrict__ samples,
                                                 const ClipDesc* __restrict__ clips,
                                                 const ClipInfo* __restrict__ info,
                                                 float* __restrict__ ysig, KParams kp) {
  const int clip = blockIdx.y;
  const ClipInfo ci = info[clip];
  const ClipDesc cd = clips[clip];
  const int64_t np = ci.end - ci.start;
  const bool pre = (kp.flags & AFX_FLAG_PREEMPH) != 0;
  // a workgroup takes kF0PrepPerThread runs of 256 consecutive samples (a wave per sample and a workgroup per 256 of them was
  // bound by the number of workgroups: 3.4 M waves for 1000 ten-second clips, 0.68 ms for 1.8 GB of traffic)
  for (int64_t n0 = (int64_t)blockIdx.x * (256 * kF0PrepPerThread); n0 < np; n0 += (int64_t)gridDim.x * (256 * kF0PrepPerThread)) {
#pragma unroll
    for (int u = 0; u < kF0PrepPerThread; ++u) {
      const int64_t n = n0 + u * 256 + threadIdx.x;
      if (n >= np) break;
      const int64_t i = ci.start + n;
      const float y = ld_sample(samples, kp.fmt, cd.off + i);
      float v = y;
      if (pre) {
        if (i == 0) v = (cd.len > 1) ? preemph0(y, ld_sample(samples, kp.fmt, cd.off + 1)) : y;
        else v = preemph1(y, ld_sample(samples, kp.fmt, cd.off + i - 1), kp.preemph_b1);
      }
      ysig[cd.off + n] = v;
    }
  }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
hipError_t launch_trim_blocks(hipStream_t s, const void* samples, const ClipDesc* clips, ClipInfo* info,
                              float* bsum, int n_clips, int max_tblocks, const KParams& kp) {
  dim3 grid((max_tblocks + 4 * kTrimPerWave - 1) / (4 * kTrimPerWave), n_clips);
  hipLaunchKernelGGL(k_trim_blocks, grid, dim3(256), 0, s, samples, clips, info, bsum, kp);
  return hipGetLastError();
}

hipError_t launch_trim_decide(hipStream_t s, const ClipDesc* clips, ClipInfo* info, const float* bsum,
                              BlockDesc* blocks, float* rms_rows, int n_clips, const KParams& kp, const void* samples) {
  hipLaunchKernelGGL(k_trim_decide, dim3(n_clips), dim3(256), 0, s, clips, info, bsum, blocks, rms_rows, kp, samples);
  return hipGetLastError();
}

bool frames2_eligible(const KParams& kp, const DevTables& tb) {
#ifndef AFX_WITH_FRAMES2
  (void)kp; (void)tb;
  return false;
#else
  const int per = kp.hop > 0 ? kp.trim_hop / kp.hop : 0;
  return kp.n_fft == 1024 && kp.hop == 256 && tb.mel_ntaps > 0 && frames2_lds_bytes(tb.mel_ntaps) <= 80 * 1024 && kp.n_mels <= 128 &&
         kp.trim_hop % kp.hop == 0 && per >= 1 && per <= 4 && !dev_env().generic_1024;
#endif
}

template <int NFFT>
static hipError_t launch_frames_t(hipStream_t s, const void* samples, ClipInfo* info,
                                  const BlockDesc* blocks, int nblocks, const DevTables& tb, const KParams& kp,
                                  float* logmel, float* rms_rows, int grid, unsigned long long* stamps) {
  const size_t lds = frames_lds_bytes(NFFT, kp.hop);
  static bool attr_set[64][2] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const int st = stamps ? 1 : 0;
  if (dev >= 0 && dev < 64 && !attr_set[dev][st]) {
    e = st ? hipFuncSetAttribute(reinterpret_cast<const void*>(&k_frames<NFFT, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
           : hipFuncSetAttribute(reinterpret_cast<const void*>(&k_frames<NFFT, false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set[dev][st] = true;
  }
  if (stamps)
    hipLaunchKernelGGL((k_frames<NFFT, true>), dim3(grid), dim3(256), lds, s, samples, info, blocks, nblocks,
                       tb, kp, logmel, rms_rows, stamps);
  else
    hipLaunchKernelGGL((k_frames<NFFT, false>), dim3(grid), dim3(256), lds, s, samples, info, blocks, nblocks,
                       tb, kp, logmel, rms_rows, stamps);
  return hipGetLastError();
}

#ifdef AFX_WITH_FRAMES2
template <int FMT, bool STAMP, bool DBG>
static hipError_t launch_frames2_t(hipStream_t s, const void* samples, ClipInfo* info, const BlockDesc* blocks,
                                   int nblocks, const DevTables& tb, const KParams& kp, float* logmel,
                                   float* rms_rows, int grid, unsigned long long* stamps) {
  static bool attr_set[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_frames2<FMT, STAMP, DBG>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((k_frames2<FMT, STAMP, DBG>), dim3(grid), dim3(256), frames2_lds_bytes(tb.mel_ntaps), s, samples, info,
                     blocks, nblocks, tb, kp, logmel, rms_rows, stamps);
  return hipGetLastError();
}

#endif  // AFX_WITH_FRAMES2

hipError_t launch_frames(hipStream_t s, const void* samples, ClipInfo* info,
                         const BlockDesc* blocks, int nblocks, const DevTables& tb, const KParams& kp,
                         float* logmel, float* rms_rows, int grid, unsigned long long* stamps) {
#ifdef AFX_WITH_FRAMES2
  if (frames2_eligible(kp, tb) && kp.rms_sub > 0) {
    const bool dbg = (kp.flags & 0x7f00) != 0 || stamps != nullptr;     // ablation switches / stamps: diagnostic instantiations
#define AFX_F2(FMT)                                                                                              \
    (stamps ? launch_frames2_t<FMT, true, true>(s, samples, info, blocks, nblocks, tb, kp, logmel, rms_rows, grid, stamps)   \
            : dbg ? launch_frames2_t<FMT, false, true>(s, samples, info, blocks, nblocks, tb, kp, logmel, rms_rows, grid, stamps)  \
                  : launch_frames2_t<FMT, false, false>(s, samples, info, blocks, nblocks, tb, kp, logmel, rms_rows, grid, stamps))
    if (kp.fmt == AFX_FMT_S16) return AFX_F2(AFX_FMT_S16);
    return AFX_F2(AFX_FMT_F32);
#undef AFX_F2
  }
#endif
  switch (kp.n_fft) {
    case 256:  return launch_frames_t<256>(s, samples, info, blocks, nblocks, tb, kp, logmel, rms_rows, grid, stamps);
    case 512:  return launch_frames_t<512>(s, samples, info, blocks, nblocks, tb, kp, logmel, rms_rows, grid, stamps);
    case 1024: return launch_frames_t<1024>(s, samples, info, blocks, nblocks, tb, kp, logmel, rms_rows, grid, stamps);
    case 2048: return launch_frames_t<2048>(s, samples, info, blocks, nblocks, tb, kp, logmel, rms_rows, grid, stamps);
    default: return hipErrorInvalidValue;
  }
}

// k_dct16l<NCG>: k_dct16 for more than 16 coefficients (NCG = 2, 3).  There the coefficient images are 64 / 96 registers
// per lane, re-read from memory by every wave for every 16-frame tile (96 loads in front of 96 MFMAs: 0.30 ms for the
// 40 coefficients of the 16 kHz configuration, against 0.13 ms of spill traffic).  Here a workgroup copies the images
// to LDS once (NCG x 8 KB), walks many tiles of its clip, and fetches each MFMA's A operand with one ds_read_b32;
// the next tile's frames are in flight while the current one is multiplied.
template <int NCG, bool FM>
__global__ __launch_bounds__(256) void k_dct16l(const ClipDesc* __restrict__ clips,
                                                const ClipInfo* __restrict__ info,
                                                const float* __restrict__ dctP, KParams kp,
                                                const float* __restrict__ logmel,
                                                float* __restrict__ mfcc, int spec) {
  extern __shared__ float dct_tab[];                     // [(g S + s) 4 + c][64 lanes]
  const int clip = blockIdx.y;
  const ClipInfo ci = info[clip];
  if (ci.status != AFX_CLIP_OK) return;                  // uniform per workgroup
  const int M = kp.n_mels, K = kp.n_mfcc, S = M >> 4;    // S <= 8
  for (int i = threadIdx.x; i < NCG * S * 4 * 64; i += 256) dct_tab[i] = dctP[i];
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const ClipDesc cd = clips[clip];
  const float theta = ord2f(ci.lmax_ord) - kp.top_db;
  const int f = lane & 15, q = lane >> 4;
  const int ntiles = (ci.T + 15) >> 4, tstep = gridDim.x * 4;
  const int g0 = spec ? (int)(ci.start / kp.hop) : 0;
  auto load_tile = [&](int tile, float4 (&x)[8]) {
    const int t0 = tile * 16;
    const float* src = FM ? logmel + (cd.frame_base + g0 + t0 + f) * (int64_t)M + q * 4
                          : logmel + (cd.frame_base + t0) * (int64_t)M + (q * 16 + f) * 4;
#pragma unroll
    for (int s = 0; s < 8; ++s)
      x[s] = (s < S && tile < ntiles) ? *reinterpret_cast<const float4*>(src + s * (FM ? 16 : 256)) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  float4 xc[8], xn[8];
  int tile = blockIdx.x * 4 + wave;
  load_tile(tile, xc);
  for (; tile < ntiles; tile += tstep) {
    load_tile(tile + tstep, xn);
    f32x4 acc[NCG][2];
#pragma unroll
    for (int g = 0; g < NCG; ++g) { acc[g][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[g][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < S) {
        const float b0 = fmaxf(xc[s].x, theta), b1 = fmaxf(xc[s].y, theta);
        const float b2 = fmaxf(xc[s].z, theta), b3 = fmaxf(xc[s].w, theta);
#pragma unroll
        for (int g = 0; g < NCG; ++g) {
          const float* a = dct_tab + ((g * S + s) * 4) * 64 + lane;
          acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b0, acc[g][0], 0, 0, 0);
          acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[64], b1, acc[g][1], 0, 0, 0);
          acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[128], b2, acc[g][0], 0, 0, 0);
          acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[192], b3, acc[g][1], 0, 0, 0);
        }
      }
    }
    const int t0 = tile * 16;
    if (t0 + f < ci.T) {
      float* out = mfcc + cd.frame_base * (int64_t)K + t0 + f;
#pragma unroll
      for (int g = 0; g < NCG; ++g) {
        const f32x4 r4 = acc[g][0] + acc[g][1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = g * 16 + q * 4 + r;
          if (k < K) out[(int64_t)k * cd.tpad] = r4[r];
        }
      }
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) xc[s] = xn[s];
  }
}

template <bool FM>
static hipError_t launch_dct_t(hipStream_t s, const ClipDesc* clips, const ClipInfo* info, const DevTables& tb,
                               const KParams& kp, const float* logmel, float* mfcc, int n_clips, int max_tmax, int spec) {
  if (tb.dctP && kp.n_mels % 16 == 0 && kp.n_mels <= 128 && kp.n_mfcc <= 48) {
    dim3 g16(((max_tmax + 15) / 16 + 4 * kDctTiles - 1) / (4 * kDctTiles), n_clips);
    const int ncg = (kp.n_mfcc + 15) / 16;
    if (ncg == 1) hipLaunchKernelGGL((k_dct16<1, FM>), g16, dim3(256), 0, s, clips, info, tb.dctP, kp, logmel, mfcc, spec);
    else {
      // a workgroup per 64 tiles of a clip (16 per wave): the table copy is paid once per 512 KB of frames
      const int tiles = (max_tmax + 15) / 16;
      dim3 gl(std::max(1, (tiles + 63) / 64), n_clips);
      const size_t lds = (size_t)ncg * (kp.n_mels / 16) * 4 * 64 * sizeof(float);
      if (dev_env().no_dct16l) {
        if (ncg == 2) hipLaunchKernelGGL((k_dct16<2, FM>), g16, dim3(256), 0, s, clips, info, tb.dctP, kp, logmel, mfcc, spec);
        else hipLaunchKernelGGL((k_dct16<3, FM>), g16, dim3(256), 0, s, clips, info, tb.dctP, kp, logmel, mfcc, spec);
      } else if (ncg == 2) hipLaunchKernelGGL((k_dct16l<2, FM>), gl, dim3(256), lds, s, clips, info, tb.dctP, kp, logmel, mfcc, spec);
      else hipLaunchKernelGGL((k_dct16l<3, FM>), gl, dim3(256), lds, s, clips, info, tb.dctP, kp, logmel, mfcc, spec);
    }
    return hipGetLastError();
  }
  dim3 grid(((max_tmax + 15) / 16 + 3) / 4, n_clips);
  switch (tb.n_cgroups) {
    case 1: hipLaunchKernelGGL((k_dct<1, FM>), grid, dim3(256), 0, s, clips, info, tb.dctA, kp, logmel, mfcc, spec); break;
    case 2: hipLaunchKernelGGL((k_dct<2, FM>), grid, dim3(256), 0, s, clips, info, tb.dctA, kp, logmel, mfcc, spec); break;
    case 3: hipLaunchKernelGGL((k_dct<3, FM>), grid, dim3(256), 0, s, clips, info, tb.dctA, kp, logmel, mfcc, spec); break;
    case 4: hipLaunchKernelGGL((k_dct<4, FM>), grid, dim3(256), 0, s, clips, info, tb.dctA, kp, logmel, mfcc, spec); break;
    case 5: hipLaunchKernelGGL((k_dct<5, FM>), grid, dim3(256), 0, s, clips, info, tb.dctA, kp, logmel, mfcc, spec); break;
    case 6: hipLaunchKernelGGL((k_dct<6, FM>), grid, dim3(256), 0, s, clips, info, tb.dctA, kp, logmel, mfcc, spec); break;
    case 7: hipLaunchKernelGGL((k_dct<7, FM>), grid, dim3(256), 0, s, clips, info, tb.dctA, kp, logmel, mfcc, spec); break;
    case 8: hipLaunchKernelGGL((k_dct<8, FM>), grid, dim3(256), 0, s, clips, info, tb.dctA, kp, logmel, mfcc, spec); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_dct(hipStream_t s, const ClipDesc* clips, const ClipInfo* info, const DevTables& tb,
                      const KParams& kp, const float* logmel, float* mfcc, int n_clips, int max_tmax, bool frame_major,
                      bool spec) {
  return frame_major ? launch_dct_t<true>(s, clips, info, tb, kp, logmel, mfcc, n_clips, max_tmax, spec ? 1 : 0)
                     : launch_dct_t<false>(s, clips, info, tb, kp, logmel, mfcc, n_clips, max_tmax, 0);
}

hipError_t launch_stats(hipStream_t s, const ClipDesc* clips, const ClipInfo* info, const KParams& kp,
                        const float* mfcc, const float* rms_rows, float* stats, float* frames_out,
                        const int64_t* frame_offsets, int n_clips, ClipInfo* info_out) {
  dim3 grid((kp.n_mfcc + 1 + 3) / 4, n_clips);
  hipLaunchKernelGGL(k_stats, grid, dim3(256), 0, s, clips, info, kp, mfcc, rms_rows, stats, frames_out,
                     frame_offsets, info_out);
  return hipGetLastError();
}

// k_finish: the last kernel of a batch.  Clears what the next batch expects cleared (the clip records and the list /
// ticket counters: two fill commands less in front of every batch), then stores the batch's sequence number to a
// completion flag in host memory the device can write -- the host can spin on that word instead of asking the
// runtime (hipEventSynchronize on a stream that already holds the next batch's commands was measured to return late).
__global__ __launch_bounds__(1024) void k_finish(uint4* info, int n_info16, int* counters, unsigned* flag, unsigned seq) {
  for (int i = threadIdx.x; i < n_info16; i += 1024) info[i] = uint4{0u, 0u, 0u, 0u};
  if (counters && threadIdx.x < 4) counters[threadIdx.x] = 0;
  __syncthreads();
  if (flag && threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_finish(hipStream_t s, ClipInfo* info, int n_clips, int* counters, unsigned* flag_dev, unsigned seq) {
  static_assert(sizeof(ClipInfo) == 32, "k_finish clears ClipInfo as two 16-byte words");
  hipLaunchKernelGGL(k_finish, dim3(1), dim3(1024), 0, s, (uint4*)info, n_clips * 2, counters, flag_dev, seq);
  return hipGetLastError();
}

hipError_t launch_preemph(hipStream_t s, const float* y, float* out, int64_t n, float b1) {
  hipLaunchKernelGGL(k_preemph, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y, out, n, b1);
  return hipGetLastError();
}

hipError_t launch_f0_prep(hipStream_t s, const void* samples, const ClipDesc* clips, const ClipInfo* info,
                          float* ysig, int n_clips, int64_t max_len, const KParams& kp) {
  const int64_t per_wg = 256 * kF0PrepPerThread;
  const int gx = (int)std::min<int64_t>(std::max<int64_t>((max_len + per_wg - 1) / per_wg, 1), 1024);
  hipLaunchKernelGGL(k_f0_prep, dim3(gx, n_clips), dim3(256), 0, s, samples, clips, info, ysig, kp);
  return hipGetLastError();
}

}  // namespace afx
