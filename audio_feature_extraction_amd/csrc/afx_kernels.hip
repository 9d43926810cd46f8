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
// No MFMA: no stage is a dense contraction (butterflies, a 1.5 %-dense filterbank, a 13x128 DCT).
#include <hip/hip_runtime.h>

#include "afx_device.h"

namespace afx {

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
#define AFX_CBARRIER() asm volatile("" ::: "memory")

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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---------------------------------------------------------------------------
// k_trim_blocks: one wave per trim block (trim_hop samples) of one clip
// ---------------------------------------------------------------------------
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
  const int64_t b = (int64_t)blockIdx.x * 4 + wave;
  if (b >= nb) return;
  const int64_t i0 = b * th;
  const int64_t i1 = (i0 + th < N) ? i0 + th : N;
  const bool pre = (kp.flags & AFX_FLAG_PREEMPH) != 0;
  const float b1 = kp.preemph_b1;
  float sum = 0.f;
  int nf = 0;
  const bool vec = kp.fmt == AFX_FMT_F32 && (((cd.off + i0) & 3) == 0) && ((th & 255) == 0) &&
                   (i1 - i0 == th) && i0 > 0;
  if (vec) {
    const float* base = (const float*)samples + cd.off;
    for (int64_t i = i0 + 4 * lane; i < i1; i += 256) {
      const float4 q = *reinterpret_cast<const float4*>(base + i);
      float prev = __shfl_up(q.w, 1);
      if (lane == 0) prev = base[i - 1];
      nf |= !(isfinite(q.x) && isfinite(q.y) && isfinite(q.z) && isfinite(q.w));
      float v0 = q.x, v1 = q.y, v2 = q.z, v3 = q.w;
      if (pre) {
        v0 = preemph1(q.x, prev, b1); v1 = preemph1(q.y, q.x, b1);
        v2 = preemph1(q.z, q.y, b1); v3 = preemph1(q.w, q.z, b1);
      }
      sum += v0 * v0; sum += v1 * v1; sum += v2 * v2; sum += v3 * v3;
    }
  } else {
    for (int64_t i = i0 + lane; i < i1; i += 64) {
      const float y = ld_sample(samples, kp.fmt, cd.off + i);
      nf |= !isfinite(y);
      float v = y;
      if (pre) {
        if (i == 0) v = (N > 1) ? preemph0(y, ld_sample(samples, kp.fmt, cd.off + 1)) : y;
        else v = preemph1(y, ld_sample(samples, kp.fmt, cd.off + i - 1), b1);
      }
      sum += v * v;
    }
  }
  sum = wave_sum(sum);
  nf = __any(nf);
  if (lane == 0) {
    bsum[cd.tblk_base + b] = sum;
    if (nf) atomicOr(&info[clip].nonfinite, 1u);
  }
}

// ---------------------------------------------------------------------------
// k_trim_decide: one workgroup per clip
// ---------------------------------------------------------------------------
__device__ __forceinline__ float trim_frame_rms(const float* bs, int64_t t, int64_t nb, int half, float inv_n) {
  float s = 0.f;
  for (int64_t b = t - half; b < t + half; ++b)
    if (b >= 0 && b < nb) s += bs[b];
  return sqrtf(s * inv_n);
}

__global__ __launch_bounds__(256) void k_trim_decide(const ClipDesc* __restrict__ clips,
                                                     ClipInfo* __restrict__ info,
                                                     const float* __restrict__ bsum, KParams kp) {
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
    const float* bs = bsum + cd.tblk_base;
    float mx = 0.f;
    for (int64_t t = tid; t < nt; t += 256) mx = fmaxf(mx, trim_frame_rms(bs, t, nb, half, inv_n));
    mx = wave_max(mx);
    if (lane == 0) red_f[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red_f[0], red_f[1]), fmaxf(red_f[2], red_f[3]));
    // amplitude_to_db(mse, ref=np.max, amin=1e-5, top_db=None), float32
    const float ref_db = 10.0f * log10f(fmaxf(1e-10f, mx * mx));
    long long first = (long long)1 << 62, last = -1;
    for (int64_t t = tid; t < nt; t += 256) {
      const float r = trim_frame_rms(bs, t, nb, half, inv_n);
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
  if (tid == 0) {
    const int64_t np = end - start;
    const int T = (int)(1 + np / kp.hop);
    if (status == AFX_CLIP_OK && T < 9) status = AFX_CLIP_TOO_SHORT;   // librosa.feature.delta width 9
    ClipInfo ci;
    ci.start = start; ci.end = end; ci.T = T; ci.status = status; ci.lmax_ord = 0u;
    ci.nonfinite = info[clip].nonfinite;
    info[clip] = ci;
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

struct LdsLayout { int s, ex, pb, mw, mt, total; };   // float offsets
__host__ __device__ inline LdsLayout lds_layout(int n_fft, int hop, int ntaps, int n_mels) {
  const int N2 = n_fft / 2;
  const int lpf = (N2 / 8 >= 64) ? 64 : N2 / 8;
  const int fpw = 64 / lpf;
  const int exn = N2 + (N2 >> 3);
  LdsLayout L;
  L.s = 0;
  L.ex = L.s + round4((kFramesPerBlock - 1) * hop + n_fft);
  L.pb = L.ex + kWaves * fpw * exn * 2;
  L.mw = L.pb + round4((N2 + 1) * 16);
  L.mt = L.mw + round4(ntaps);
  L.total = L.mt + round4(3 * n_mels);
  return L;
}

size_t frames_lds_bytes(int n_fft, int hop, int ntaps, int n_mels) {
  if (n_fft != 256 && n_fft != 512 && n_fft != 1024 && n_fft != 2048) return 0;
  return (size_t)lds_layout(n_fft, hop, ntaps, n_mels).total * sizeof(float);
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

// power-spectrum buffer: bin-major, 16 frames per row, frame index XOR-swizzled
// by the bin so that both the per-frame column writes (32 consecutive bins) and
// the per-bin row reads (16 frames, two bins of opposite parity per 32 lanes)
// are bank-conflict-free with ds_*_b32.
__device__ __forceinline__ int pbidx(int k, int f) { return k * 16 + (f ^ ((k >> 1) & 15)); }

template <int NFFT>
__global__ __launch_bounds__(256, (NFFT >= 2048 ? 1 : 2)) void k_frames(const void* __restrict__ samples,
                                                   const ClipDesc* __restrict__ clips,
                                                   ClipInfo* __restrict__ info,
                                                   const int2* __restrict__ blocks, int nblocks,
                                                   DevTables tb, KParams kp,
                                                   float* __restrict__ logmel,
                                                   float* __restrict__ rms_rows) {
  using C = FC<NFFT>;
  using S = Sched<NFFT>;
  constexpr int N2 = C::N2, NB = C::NB, LPF = C::LPF, P = C::P, FPW = C::FPW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int hop = kp.hop, M = kp.n_mels;
  const LdsLayout L = lds_layout(NFFT, hop, tb.ntaps, M);
  float* const S_ = smem + L.s;
  float* const PB = smem + L.pb;
  float* const MW = smem + L.mw;
  int* const MT = reinterpret_cast<int*>(smem + L.mt);

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int lif = lane % LPF, fsub = lane / LPF;
  float2* const EX = reinterpret_cast<float2*>(smem + L.ex) + (wave * FPW + fsub) * C::EXN;

  // ---- once per workgroup: sparse-mel tables -> LDS; per-lane tables -> registers
  for (int i = tid; i < tb.ntaps; i += 256) MW[i] = tb.taps[i];
  for (int i = tid; i < M; i += 256) {
    MT[i] = tb.mel_k0[i]; MT[M + i] = tb.mel_n4[i]; MT[2 * M + i] = tb.mel_wo[i];
  }
  float2 wreg[P], preg[P];
  {
    const float2* w2 = reinterpret_cast<const float2*>(tb.window);
    const float2* p2 = reinterpret_cast<const float2*>(tb.post);
#pragma unroll
    for (int u = 0; u < P; ++u) { wreg[u] = w2[lif + LPF * u]; preg[u] = p2[lif + LPF * u]; }
  }
  constexpr int NT2 = NTW<typename S::P2>::v, NT3 = NTW<typename S::P3>::v, NT4 = NTW<typename S::P4>::v;
  float2 tw2[NT2 > 0 ? NT2 : 1], tw3[NT3 > 0 ? NT3 : 1], tw4[NT4 > 0 ? NT4 : 1];
  {
    const float2* t2 = reinterpret_cast<const float2*>(tb.tw);
    S::P2::load_tw(t2, lif, tw2);
    S::P3::load_tw(t2, lif, tw3);
    if constexpr (NT4 > 0) S::P4::load_tw(t2, lif, tw4);
  }
  const bool pre = (kp.flags & AFX_FLAG_PREEMPH) != 0;
  const float b1 = kp.preemph_b1;
  const int slen = (kFramesPerBlock - 1) * hop + NFFT;
  const bool hop_even = (hop & 1) == 0;
  __syncthreads();

  for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
    const int2 bd = blocks[b];
    const int clip = bd.x, t0 = bd.y * kFramesPerBlock;
    const int T = info[clip].T;
    if (info[clip].status != AFX_CLIP_OK || t0 >= T) continue;     // uniform per workgroup
    const int64_t cstart = info[clip].start, cend = info[clip].end;
    const ClipDesc cd = clips[clip];
    const int64_t N = cd.len;

    // ---- stage the hop-strided sample block: pre-emphasis + trim mask, once per sample
    const int64_t g0 = cstart + (int64_t)t0 * hop - NFFT / 2;
    for (int j = tid * 4; j < slen; j += 1024) {
      const int64_t i = g0 + j;
      float y[4], prev;
      const bool fast = (i >= 1) && (i + 3 < N) && (((cd.off + i) & 3) == 0);
      if (fast) {
        if (kp.fmt == AFX_FMT_F32) {
          const float* base = (const float*)samples + cd.off;
          const float4 q = *reinterpret_cast<const float4*>(base + i);
          y[0] = q.x; y[1] = q.y; y[2] = q.z; y[3] = q.w;
          prev = base[i - 1];
        } else {
          const int16_t* base = (const int16_t*)samples + cd.off;
          const short4 q = *reinterpret_cast<const short4*>(base + i);
          const float sc = 1.0f / 32768.0f;
          y[0] = q.x * sc; y[1] = q.y * sc; y[2] = q.z * sc; y[3] = q.w * sc;
          prev = base[i - 1] * sc;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t ii = i + e;
          y[e] = (ii >= 0 && ii < N) ? ld_sample(samples, kp.fmt, cd.off + ii) : 0.f;
        }
        prev = (i - 1 >= 0 && i - 1 < N) ? ld_sample(samples, kp.fmt, cd.off + i - 1) : 0.f;
      }
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t ii = i + e;
        float v = y[e];
        if (pre) {
          v = preemph1(y[e], e == 0 ? prev : y[e - 1], b1);
          if (ii == 0) v = preemph0(y[e], ld_sample(samples, kp.fmt, cd.off + 1));
        }
        o[e] = (ii >= cstart && ii < cend) ? v : 0.f;
      }
      *reinterpret_cast<float4*>(S_ + j) = make_float4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();

    // ---- per frame: window -> rFFT -> power spectrum (+ RMS of the unwindowed frame)
#pragma unroll 1
    for (int it = 0; it < C::ITERS; ++it) {
      const int fl = (it * kWaves + wave) * FPW + fsub;      // frame within the block
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
#pragma unroll
      for (int o = LPF / 2; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
      if (lif == 0 && t0 + fl < T) rms_rows[cd.frame_base + t0 + fl] = sqrtf(ss / (float)NFFT);

      S::P1::run(v, nullptr, EX, lif);
      S::P2::run(v, tw2, EX, lif);
      S::P3::run(v, tw3, EX, lif);
      if constexpr (NT4 > 0) S::P4::run(v, tw4, EX, lif);

      // real-FFT split: X[k] from Z[k] and Z[N2-k]; EX now holds Z in natural order
#pragma unroll
      for (int u = 0; u < P; ++u) {
        const int k = lif + LPF * u;
        const float2 z = v[u];
        const float2 m = EX[expad((N2 - k) & (N2 - 1))];
        const float e2r = z.x + m.x, e2i = z.y - m.y;
        const float o2r = z.y + m.y, o2i = m.x - z.x;
        const float2 w = preg[u];
        const float xr = e2r + w.x * o2r - w.y * o2i;
        const float xi = e2i + w.x * o2i + w.y * o2r;
        PB[pbidx(k, fl)] = 0.25f * (xr * xr + xi * xi);
        if (u == 0 && lif == 0) { const float ny = z.x - z.y; PB[pbidx(N2, fl)] = ny * ny; }
      }
      AFX_CBARRIER();
    }
    __syncthreads();

    // ---- sparse mel + dB: lane = (frame f, tap quarter q); four filters per step
    {
      const int f = lane & 15, q = lane >> 4;
      const bool valid = (t0 + f) < T;
      float lmax = -INFINITY;
      float* tile = logmel + (cd.frame_base + t0) * (int64_t)M;
      const int nq = M >> 2;
      for (int it = 0; it * kWaves < nq; ++it) {
        const int qd = it * kWaves + ((it & 1) ? (kWaves - 1 - wave) : wave);
        if (qd >= nq) continue;
        float acc[4];
#pragma unroll
        for (int jf = 0; jf < 4; ++jf) {
          const int m = qd * 4 + jf;
          const int k0 = __builtin_amdgcn_readfirstlane(MT[m]);
          const int n4 = __builtin_amdgcn_readfirstlane(MT[M + m]);
          const int wo = __builtin_amdgcn_readfirstlane(MT[2 * M + m]);
          float a = 0.f;
          for (int i = 0; i < n4; ++i) {
            int kk = k0 + 4 * i + q;
            kk = kk < NB ? kk : NB - 1;
            a += MW[wo + 4 * i + q] * PB[pbidx(kk, f)];
          }
          acc[jf] = a;
        }
        // reduce-scatter over the four quarters: quarter q ends with filter 4*qd + q
        const bool hi2 = (q & 2) != 0, hi1 = (q & 1) != 0;
        const float s0 = hi2 ? acc[0] : acc[2], k0_ = hi2 ? acc[2] : acc[0];
        const float s1 = hi2 ? acc[1] : acc[3], k1_ = hi2 ? acc[3] : acc[1];
        const float a0 = k0_ + __shfl_xor(s0, 32);
        const float a1 = k1_ + __shfl_xor(s1, 32);
        const float sn = hi1 ? a0 : a1, kn = hi1 ? a1 : a0;
        const float tot = kn + __shfl_xor(sn, 16);
        // 10*log10(max(amin, mel)) ; v_log_f32 is log2
        const float Lv = 3.01029995663981195f * __builtin_amdgcn_logf(fmaxf(kp.amin, tot));
        if (valid) {
          tile[(qd * 4 + q) * 16 + f] = Lv;
          lmax = fmaxf(lmax, Lv);
        }
      }
      lmax = wave_max(lmax);
      if (lane == 0 && lmax > -INFINITY) atomicMax(&info[clip].lmax_ord, f2ord(lmax));
    }
    // no barrier needed here: the next staging only writes S_ (dead since the barrier above),
    // and PB is rewritten only after the next iteration's first barrier.
  }
}

// ---------------------------------------------------------------------------
// k_dct: clamp at (clip max - top_db), ortho DCT-II; thread = frame
// ---------------------------------------------------------------------------
template <int KMAX>
__global__ __launch_bounds__(256) void k_dct(const ClipDesc* __restrict__ clips,
                                             const ClipInfo* __restrict__ info,
                                             const float* __restrict__ dct, KParams kp,
                                             const float* __restrict__ logmel,
                                             float* __restrict__ mfcc) {
  const int clip = blockIdx.y;
  const ClipInfo ci = info[clip];
  if (ci.status != AFX_CLIP_OK) return;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x * 256 >= ci.T) return;
  const ClipDesc cd = clips[clip];
  const int M = kp.n_mels, K = kp.n_mfcc;
  const float theta = ord2f(ci.lmax_ord) - kp.top_db;
  const bool valid = t < ci.T;
  const int tt = valid ? t : ci.T - 1;
  const float* tile = logmel + (cd.frame_base + (tt & ~15)) * (int64_t)M + (tt & 15);
  float acc[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) acc[k] = 0.f;
  for (int m = 0; m < M; ++m) {
    const float Lc = fmaxf(tile[m * 16], theta);
    // dct is zero-padded to KMAX rows on the device, so no k < K test in the hot loop
#pragma unroll
    for (int k = 0; k < KMAX; ++k) acc[k] += dct[k * M + m] * Lc;
  }
  if (valid) {
    float* out = mfcc + cd.frame_base * (int64_t)K + t;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) out[(int64_t)k * cd.tpad] = acc[k];
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
                                               const int64_t* __restrict__ frame_offsets) {
  const int clip = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int K = kp.n_mfcc;
  const int row = blockIdx.x * 4 + wave;
  if (row > K) return;
  const ClipInfo ci = info[clip];
  float* st = stats + (int64_t)clip * (4 * K + 3);
  if (ci.status != AFX_CLIP_OK) {
    if (lane == 0) {
      if (row < K) { st[row] = 0.f; st[K + row] = 0.f; st[2 * K + row] = 0.f; st[3 * K + row] = 0.f; }
      else { st[4 * K] = 0.f; st[4 * K + 1] = 0.f; st[4 * K + 2] = 0.f; }
    }
    return;
  }
  const ClipDesc cd = clips[clip];
  const int T = ci.T;
  const double invT = 1.0 / (double)T;
  float* fo = frames_out ? frames_out + frame_offsets[clip] : nullptr;
  const int64_t fstride = cd.tmax;
  if (row < K) {
    const float* x = mfcc + cd.frame_base * (int64_t)K + (int64_t)row * cd.tpad;
    double s = 0.0;
    for (int t = lane; t < T; t += 64) s += (double)x[t];
    const double mean = wave_sum_d(s) * invT;
    const float meanf = (float)mean;
    double s2 = 0.0, sd1 = 0.0, sd2 = 0.0;
    for (int t = lane; t < T; t += 64) {
      const float d = x[t] - meanf;
      s2 += (double)d * (double)d;
      // savgol_filter(width 9, polyorder=deriv=order, mode='interp'): interior taps; the
      // fitted edge polynomial has a constant derivative, so frames 0..3 / T-4..T-1 repeat
      // frame 4 / frame T-5.
      const int tc = t < 4 ? 4 : (t > T - 5 ? T - 5 : t);
      const float* c = x + tc;
      const double d1 = (4.0 * ((double)c[4] - (double)c[-4]) + 3.0 * ((double)c[3] - (double)c[-3]) +
                         2.0 * ((double)c[2] - (double)c[-2]) + ((double)c[1] - (double)c[-1])) / 60.0;
      const double d2 = (28.0 * ((double)c[4] + (double)c[-4]) + 7.0 * ((double)c[3] + (double)c[-3]) -
                         8.0 * ((double)c[2] + (double)c[-2]) - 17.0 * ((double)c[1] + (double)c[-1]) -
                         20.0 * (double)c[0]) / 462.0;
      const float d1f = (float)d1, d2f = (float)d2;
      sd1 += (double)d1f; sd2 += (double)d2f;
      if (fo) {
        fo[(int64_t)row * fstride + t] = x[t];
        fo[(int64_t)(K + row) * fstride + t] = d1f;
        fo[(int64_t)(2 * K + row) * fstride + t] = d2f;
      }
    }
    s2 = wave_sum_d(s2); sd1 = wave_sum_d(sd1); sd2 = wave_sum_d(sd2);
    if (lane == 0) {
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

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
hipError_t launch_trim_blocks(hipStream_t s, const void* samples, const ClipDesc* clips, ClipInfo* info,
                              float* bsum, int n_clips, int max_tblocks, const KParams& kp) {
  dim3 grid((max_tblocks + 3) / 4, n_clips);
  hipLaunchKernelGGL(k_trim_blocks, grid, dim3(256), 0, s, samples, clips, info, bsum, kp);
  return hipGetLastError();
}

hipError_t launch_trim_decide(hipStream_t s, const ClipDesc* clips, ClipInfo* info, const float* bsum,
                              int n_clips, const KParams& kp) {
  hipLaunchKernelGGL(k_trim_decide, dim3(n_clips), dim3(256), 0, s, clips, info, bsum, kp);
  return hipGetLastError();
}

template <int NFFT>
static hipError_t launch_frames_t(hipStream_t s, const void* samples, const ClipDesc* clips, ClipInfo* info,
                                  const int2* blocks, int nblocks, const DevTables& tb, const KParams& kp,
                                  float* logmel, float* rms_rows, int grid) {
  const size_t lds = frames_lds_bytes(NFFT, kp.hop, tb.ntaps, kp.n_mels);
  static bool attr_set[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_frames<NFFT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(k_frames<NFFT>, dim3(grid), dim3(256), lds, s, samples, clips, info, blocks, nblocks,
                     tb, kp, logmel, rms_rows);
  return hipGetLastError();
}

hipError_t launch_frames(hipStream_t s, const void* samples, const ClipDesc* clips, ClipInfo* info,
                         const int2* blocks, int nblocks, const DevTables& tb, const KParams& kp,
                         float* logmel, float* rms_rows, int grid) {
  switch (kp.n_fft) {
    case 256:  return launch_frames_t<256>(s, samples, clips, info, blocks, nblocks, tb, kp, logmel, rms_rows, grid);
    case 512:  return launch_frames_t<512>(s, samples, clips, info, blocks, nblocks, tb, kp, logmel, rms_rows, grid);
    case 1024: return launch_frames_t<1024>(s, samples, clips, info, blocks, nblocks, tb, kp, logmel, rms_rows, grid);
    case 2048: return launch_frames_t<2048>(s, samples, clips, info, blocks, nblocks, tb, kp, logmel, rms_rows, grid);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_dct(hipStream_t s, const ClipDesc* clips, const ClipInfo* info, const DevTables& tb,
                      const KParams& kp, const float* logmel, float* mfcc, int n_clips, int max_tmax) {
  dim3 grid((max_tmax + 255) / 256, n_clips);
  const int K = kp.n_mfcc;
  if (K <= 16) hipLaunchKernelGGL(k_dct<16>, grid, dim3(256), 0, s, clips, info, tb.dct, kp, logmel, mfcc);
  else if (K <= 32) hipLaunchKernelGGL(k_dct<32>, grid, dim3(256), 0, s, clips, info, tb.dct, kp, logmel, mfcc);
  else if (K <= 64) hipLaunchKernelGGL(k_dct<64>, grid, dim3(256), 0, s, clips, info, tb.dct, kp, logmel, mfcc);
  else hipLaunchKernelGGL(k_dct<128>, grid, dim3(256), 0, s, clips, info, tb.dct, kp, logmel, mfcc);
  return hipGetLastError();
}

hipError_t launch_stats(hipStream_t s, const ClipDesc* clips, const ClipInfo* info, const KParams& kp,
                        const float* mfcc, const float* rms_rows, float* stats, float* frames_out,
                        const int64_t* frame_offsets, int n_clips) {
  dim3 grid((kp.n_mfcc + 1 + 3) / 4, n_clips);
  hipLaunchKernelGGL(k_stats, grid, dim3(256), 0, s, clips, info, kp, mfcc, rms_rows, stats, frames_out,
                     frame_offsets);
  return hipGetLastError();
}

hipError_t launch_preemph(hipStream_t s, const float* y, float* out, int64_t n, float b1) {
  hipLaunchKernelGGL(k_preemph, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y, out, n, b1);
  return hipGetLastError();
}

}  // namespace afx
