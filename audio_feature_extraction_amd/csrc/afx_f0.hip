// extract_f0 on MI355X: librosa.pyin at the reference's call site
// (audio_feature_extraction_toolkit/core/feature_extractor.py:87-94) as four kernels over the
// preprocessed signal.  Everything that decides a pitch is float64, as in the pinned numpy stack;
// the one float32 piece -- the running frame energy, a sequential np.cumsum of a float32 array --
// is reproduced add for add, because near a trough the difference function is smaller than its
// rounding.
//   k_f0_energy   one lane per frame: e[W + tau] - e[tau] of the float32 running sum of squares
//   k_f0_yin      one wave per frame: autocorrelation (direct, float64), difference function,
//                 cumulative-mean normalisation, troughs, threshold/Boltzmann/Beta probabilities,
//                 parabolic refinement, pitch bins -> a sparse observation column per frame, stored as the
//                 logs the Viterbi pass scatters
//   k_f0_viterbi  one workgroup per clip: log-domain Viterbi over 2 x n_bins states with the banded
//                 transition matrix -- value-only forward pass, one barrier per step, every value
//                 column kept
//   k_f0_backtrack one wave per clip: the arg-max recomputed along the path through the kept columns
//                 (an LDS-DMA ring runs five steps ahead of it), then the f0 statistics
//                 (feature_extractor.py:97-114)
// A wave issues one instruction per ~9 ticks on this part however many waves share its SIMD
// (tools/micro/f64_rate.hip): these kernels are bound by the instruction count of their longest wave,
// so their loops are split by regime and carry no per-step tests.
#include <hip/hip_runtime.h>

#include <cmath>
#include <type_traits>

#include "afx_f0.h"

namespace afx {

// The timing-only ablation switches and per-phase cycle stamps of AFX_F0_DEBUG exist in libafx_dbg.so only (make dbg:
// -DAFX_F0_DEBUG_BUILD=1); the product library ignores the variable and carries neither the code nor its registers.
#ifndef AFX_F0_DEBUG_BUILD
#define AFX_F0_DEBUG_BUILD 0
#endif
constexpr bool kF0Dbg = AFX_F0_DEBUG_BUILD != 0;

#define F0_WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

__device__ __forceinline__ double shfl_d(double v, int src) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl(lo, src); hi = __shfl(hi, src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_up_d(double v, int delta) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_up(lo, delta); hi = __shfl_up(hi, delta);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_xor_d(double v, int m) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, m); hi = __shfl_xor(hi, m);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += shfl_xor_d(v, o);
  return v;
}
__device__ __forceinline__ double wave_min_d(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmin(v, shfl_xor_d(v, o));
  return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int lanes_below(unsigned long long mask) {       // set bits of mask below this lane
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

// quad_perm / row_mirror / row_half_mirror only: every lane has a source, so bound_ctrl changes nothing but spares the
// v_mov_b32 that would otherwise initialise the destination with the `old` value
#define F0_DPP_I(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xf, 0xf, true)
template <int CTRL> __device__ __forceinline__ double dpp_dd(double v) {
  return __hiloint2double(F0_DPP_I(__double2hiint(v), CTRL), F0_DPP_I(__double2loint(v), CTRL));
}
// max of two doubles that are never NaN, as the one instruction it is: fmax() makes the compiler quiet a possible signalling
// NaN in every operand whose origin it cannot see (a v_max_f64 x, x, x in front of the real one -- for values read from LDS,
// moved by DPP, or merged from two branches)
__device__ __forceinline__ double max_nn(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ double wave_max_dpp(double v) {         // uniform result
  v = max_nn(v, dpp_dd<0xB1>(v));    // quad_perm [1,0,3,2]      (callers: Viterbi values, -inf or finite)
  v = max_nn(v, dpp_dd<0x4E>(v));    // quad_perm [2,3,0,1]
  v = max_nn(v, dpp_dd<0x141>(v));   // row_half_mirror
  v = max_nn(v, dpp_dd<0x140>(v));   // row_mirror
  const int lo = __double2loint(v), hi = __double2hiint(v);
  auto rl = [&](int l) { return __hiloint2double(__builtin_amdgcn_readlane(hi, l), __builtin_amdgcn_readlane(lo, l)); };
  return fmax(fmax(rl(0), rl(16)), fmax(rl(32), rl(48)));
}
__device__ __forceinline__ int wave_min_dpp(int v) {               // uniform result
  v = min(v, F0_DPP_I(v, 0xB1));
  v = min(v, F0_DPP_I(v, 0x4E));
  v = min(v, F0_DPP_I(v, 0x141));
  v = min(v, F0_DPP_I(v, 0x140));
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

__device__ __forceinline__ void lds_fmax(double* p, double v) {      // *p = max(*p, v) in LDS, no return value
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  __hip_atomic_fetch_max((__attribute__((address_space(3))) double*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma clang diagnostic pop
}

typedef const double __attribute__((address_space(4))) cdouble_k;   // constant address space: uniform reads become s_load
__device__ __forceinline__ cdouble_k* as_constant(const double* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  return (cdouble_k*)p;
#pragma clang diagnostic pop
}

// ---------------------------------------------------------------------------------------------
// k_f0_energy: energy_frames[tau] = e[W + tau] - e[tau], e = np.cumsum(frame ** 2) in float32.
// The chain is serial per frame, so a lane takes a frame; the block's samples sit in LDS once
// (index padded by one word per hop so that the lanes' strided walks hit distinct banks), and the
// tau <= max_period history of each lane in a [tau][lane] array that is overwritten by the result
// and written out transposed (coalesced).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sq_acc(float e, float y) {
#pragma clang fp contract(off)       // y**2 is rounded before the running sum takes it
  const float q = y * y;
  return e + q;
}

size_t f0_energy_lds_bytes(const F0Params& fp) {
  const size_t span = (size_t)(fp.epb - 1) * fp.hop + fp.W + fp.n_tau;
  return (span + span / fp.hop + 8) * 4 + (size_t)fp.n_tau * (fp.epb + 1) * 4;
}

constexpr int kEnergyWaves = 4;       // waves per workgroup of k_f0_energy

__global__ __launch_bounds__(64 * kEnergyWaves) void k_f0_energy(const float* __restrict__ ysig,
                                                                 const ClipDesc* __restrict__ clips,
                                                                 const ClipInfo* __restrict__ info,
                                                                 float* __restrict__ energy, F0Params fp) {
  extern __shared__ float sme[];
  const int clip = blockIdx.y;
  const ClipInfo ci = info[clip];
  if (ci.status == AFX_CLIP_NONFINITE) return;
  const int T = ci.T, E = fp.epb, hop = fp.hop;
  const int t0 = blockIdx.x * E;
  if (t0 >= T) return;
  const ClipDesc cd = clips[clip];
  const int64_t np = ci.end - ci.start;
  constexpr int NT = 64 * kEnergyWaves;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int span = (E - 1) * hop + fp.W + fp.n_tau;
  float* S = sme;
  float* H = sme + (span + span / hop + 8);
  const int hs = E + 1;
  const int64_t g0 = (int64_t)t0 * hop - fp.n_fft / 2;
  const float* y = ysig + cd.off;
  for (int i = tid; i < span; i += NT) {
    const int64_t g = g0 + i;
    S[i + i / hop] = (g >= 0 && g < np) ? y[g] : 0.f;
  }
  __syncthreads();
  // The chain is serial per frame, one lane each; a wave's time is its instruction count whatever the number of
  // active lanes, so the chain is split into its three regimes (history kept / nothing stored / difference formed)
  // and into runs between the index pads, each an unrolled loop with immediate offsets and no per-step tests.
  // Four waves share a block's frames: staging and the transposed write-out go four times faster, the chains run
  // on the four SIMDs side by side (sixteen waves of four lanes were slower: the chain is issued per wave).
  const int fw = E / kEnergyWaves > 0 ? E / kEnergyWaves : 1;      // frames per wave
  const int fr = wave * fw + lane;                                 // this lane's frame of the block
  if (lane < fw && fr < E && t0 + fr < T) {
    float e = 0.f;
    const int W = fp.W, n_tau = fp.n_tau;                          // n_tau <= W (max_period <= n_fft - W - 1)
    // padded index of sample fr * hop + n is fr * (hop + 1) + n + n / hop
    auto seg = [&](int nb, int ne, auto phase) {                   // steps [nb, ne) inside one hop-sized run
      constexpr int PH = decltype(phase)::value;
      const float* sp = S + fr * (hop + 1) + nb / hop;
      float* hk = H + fr;                                          // history / result column of this frame
      int n = nb;
      for (; n + 8 <= ne; n += 8) {
        float yv[8], hv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          yv[u] = sp[n + u];
          if constexpr (PH == 3) hv[u] = hk[(n + u - W) * hs];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          e = sq_acc(e, yv[u]);
          if constexpr (PH == 1) hk[(n + u) * hs] = e;
          if constexpr (PH == 3) {
            float d = e - hv[u];
            if (fabsf(d) < 1e-6f) d = 0.f;
            hk[(n + u - W) * hs] = d;
          }
        }
      }
      for (; n < ne; ++n) {
        e = sq_acc(e, sp[n]);
        if constexpr (PH == 1) hk[n * hs] = e;
        if constexpr (PH == 3) {
          float d = e - hk[(n - W) * hs];
          if (fabsf(d) < 1e-6f) d = 0.f;
          hk[(n - W) * hs] = d;
        }
      }
    };
    auto range = [&](int a, int b, auto phase) {
      while (a < b) {
        const int stop = (a / hop + 1) * hop;
        const int ne = stop < b ? stop : b;
        seg(a, ne, phase);
        a = ne;
      }
    };
    range(0, n_tau, std::integral_constant<int, 1>());             // e[n] kept for n < n_tau
    range(n_tau, W, std::integral_constant<int, 2>());
    range(W, W + n_tau, std::integral_constant<int, 3>());         // e[W + tau] - e[tau]
  }
  __syncthreads();
  for (int f = wave; f < E && t0 + f < T; f += kEnergyWaves) {
    float* row = energy + (cd.frame_base + t0 + f) * (int64_t)fp.n_tau_pad;
    for (int tau = lane; tau < fp.n_tau; tau += 64) row[tau] = H[tau * hs + f];
  }
}

// ---------------------------------------------------------------------------------------------
// k_f0_energy2: the same rows without the tau-history.  e[tau] is needed W steps after the chain has passed it, and
// keeping it cost 1.35 KB of LDS per frame -- 157 KB per workgroup of 64 frames: one workgroup, one wave per SIMD, on a
// chain whose every step waits for the one before.  Instead the lane runs the chain a second time, W steps behind:
// a[tau] = e[tau] is re-formed from the same samples in the same order (bit for bit the same float32 sums) while the
// first chain is at e[W + tau], 40 % more additions on lanes that were waiting anyway, and two independent chains per
// lane where they overlap.  LDS then holds the samples only (1 KB per frame), so LPW lanes per wave (LPW 8: 32 frames,
// 37 KB) leave room for four workgroups per CU, four waves per SIMD.  The differences leave through a wave-local
// [frame][32 lags] tile: every 32 lags the wave's 64 lanes write its frames' pieces of the rows, 128 bytes each.
// Needs W % hop == 0 (the pads of the two index streams then coincide); anything else takes k_f0_energy.
// ---------------------------------------------------------------------------------------------
constexpr int kE2Tile = 32;
__host__ __device__ inline size_t f0_energy2_span(const F0Params& fp, int lpw) {
  return (size_t)(kEnergyWaves * lpw - 1) * fp.hop + fp.W + fp.n_tau;
}
size_t f0_energy2_lds_bytes(const F0Params& fp, int lpw) {
  const size_t span = f0_energy2_span(fp, lpw);
  return (span + span / fp.hop + 8 + (size_t)kEnergyWaves * lpw * (kE2Tile + 1)) * 4;
}

template <int LPW>
__global__ __launch_bounds__(64 * kEnergyWaves) void k_f0_energy2(const float* __restrict__ ysig,
                                                                  const ClipDesc* __restrict__ clips,
                                                                  const ClipInfo* __restrict__ info,
                                                                  float* __restrict__ energy, F0Params fp) {
  extern __shared__ float sme[];
  const int clip = blockIdx.y;
  const ClipInfo ci = info[clip];
  if (ci.status == AFX_CLIP_NONFINITE) return;
  constexpr int E = kEnergyWaves * LPW;
  const int T = ci.T, hop = fp.hop, W = fp.W, n_tau = fp.n_tau;
  const int t0 = blockIdx.x * E;
  if (t0 >= T) return;
  const ClipDesc cd = clips[clip];
  const int64_t np = ci.end - ci.start;
  constexpr int NT = 64 * kEnergyWaves;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int span = (int)f0_energy2_span(fp, LPW);
  float* S = sme;
  float* OB = sme + (span + span / hop + 8) + wave * LPW * (kE2Tile + 1);     // this wave's [frame][lag] tile
  const int64_t g0 = (int64_t)t0 * hop - fp.n_fft / 2;
  const float* y = ysig + cd.off;
  for (int i = tid; i < span; i += NT) {
    const int64_t g = g0 + i;
    S[i + i / hop] = (g >= 0 && g < np) ? y[g] : 0.f;
  }
  __syncthreads();
  const int fr = wave * LPW + lane;                                 // this lane's frame of the block
  const bool chain = lane < LPW && t0 + fr < T;
  // padded index of sample fr * hop + n is fr * (hop + 1) + n + n / hop; a run stays inside one hop-sized piece
  const float* base = S + (chain ? fr : 0) * (hop + 1);
  float e = 0.f, a = 0.f;
  if (chain) {                                                      // e[0 .. W - 1]: nothing leaves
    for (int r0 = 0; r0 < W; r0 += hop) {
      const float* sp = base + r0 + r0 / hop;
      int n = 0;
      for (; n + 8 <= hop; n += 8) {
        float yv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) yv[u] = sp[n + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) e = sq_acc(e, yv[u]);
      }
      for (; n < hop; ++n) e = sq_acc(e, sp[n]);
    }
  }
  const int wq = W / hop;                                           // pads between sample tau and sample W + tau
  for (int tb = 0; tb < n_tau; tb += kE2Tile) {
    const int te = tb + kE2Tile < n_tau ? tb + kE2Tile : n_tau;
    if (chain) {
      int tau = tb;
      while (tau < te) {                                            // runs between the pads (a tile may straddle one)
        const int stop = (tau / hop + 1) * hop;
        const int re = stop < te ? stop : te;
        const float* pa = base + tau / hop;                        // + tau
        const float* pe = pa + W + wq;                              // sample W + tau
        float* ob = OB + lane * (kE2Tile + 1) - tb;
        int n = tau;
        for (; n + 8 <= re; n += 8) {
          float ya[8], ye[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) { ya[u] = pa[n + u]; ye[u] = pe[n + u]; }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            a = sq_acc(a, ya[u]);
            e = sq_acc(e, ye[u]);
            float d = e - a;
            if (fabsf(d) < 1e-6f) d = 0.f;
            ob[n + u] = d;
          }
        }
        for (; n < re; ++n) {
          a = sq_acc(a, pa[n]);
          e = sq_acc(e, pe[n]);
          float d = e - a;
          if (fabsf(d) < 1e-6f) d = 0.f;
          ob[n] = d;
        }
        tau = re;
      }
    }
    F0_WAVE_SYNC();
    // the wave's LPW rows of this tile: lanes (row, lag) = (i / 32, i % 32), 128 bytes of a row per 32 lanes
    for (int i = lane; i < LPW * kE2Tile; i += 64) {
      const int r = i / kE2Tile, c = i % kE2Tile;
      const int f = wave * LPW + r;
      if (t0 + f < T && tb + c < te)
        energy[(cd.frame_base + t0 + f) * (int64_t)fp.n_tau_pad + tb + c] = OB[r * (kE2Tile + 1) + c];
    }
    F0_WAVE_SYNC();
  }
}

// ---------------------------------------------------------------------------------------------
// k_f0_yin
// ---------------------------------------------------------------------------------------------

struct YinLds { size_t span, per_wave, tables, total; };
// compact (the REF instantiation): the trough / candidate arrays CP, CB live in the D row behind its 64 event flags -- D is
// idle once the normalised difference is formed -- and the two tables that are only read with uniform indices (beta,
// cumbeta) come through the scalar cache: 39.7 KB per workgroup, a fourth workgroup per CU
__host__ __device__ inline YinLds yin_lds_i(int hop_, int n_fft_, int n_tau_pad_, int slots_, int cap_, int fpb, bool yf, bool compact = false) {
  YinLds L;
  L.span = (size_t)(fpb - 1) * hop_ + n_fft_ + 64;
  // per wave (doubles): D[n_tau_pad] | X[slots*64 + 2] | CP[cap] | CB[cap] (ints, cap/2 doubles)
  L.per_wave = (size_t)n_tau_pad_ + (size_t)slots_ * 64 + 2 + cap_ + (cap_ + 1) / 2;
  // shared tables: thr[101] | beta[100] | cumbeta[101] | bfact[cap+1] | bexp[cap+1]
  L.tables = 101 + 100 + 101 + 2 * ((size_t)cap_ + 1);
  // the staged signal is kept as the float32 it is (converted on read: half the bytes of the autocorrelation's LDS reads and
  // 20 KB less per workgroup); span is rounded up to an even count so that the double arrays behind it stay 8-byte aligned
  // (yf: the instantiations with at most 6 lags per lane; the lag-heavy ones re-read the signal 11..16 times per step and keep it
  // as float64 -- a conversion per read costs them more than the bytes)
  L.span = (L.span + 1) & ~(size_t)1;
  if (compact) {
    L.per_wave = (size_t)n_tau_pad_ + (size_t)slots_ * 64 + 2;
    L.tables = 101 + 2 * ((size_t)cap_ + 1);
  }
  L.total = L.span * (yf ? sizeof(float) : sizeof(double)) + (4 * L.per_wave + L.tables) * sizeof(double);
  return L;
}
__host__ __device__ inline YinLds yin_lds(const F0Params& fp, int fpb, bool yf) {
  return yin_lds_i(fp.hop, fp.n_fft, fp.n_tau_pad, fp.slots, fp.cap, fpb, yf);
}
// frames one workgroup owns: 16, or 8 where that (and only that) lets a third workgroup onto the CU -- the kernel is bound by
// how often a wave gets to issue, and the lag-heavy instantiations (more than 6 lags per lane) cannot use a third wave anyway
int f0_yin_frames_per_block(const F0Params& fp) {
  const int need = fp.R > fp.slots ? fp.R : fp.slots;
  const size_t third = 160 * 1024 / 3;
  return (need <= 6 && yin_lds(fp, 16, true).total > third && yin_lds(fp, 8, true).total <= third) ? 8 : 16;
}
size_t f0_yin_lds_bytes(const F0Params& fp) {
  const int need = fp.R > fp.slots ? fp.R : fp.slots;
  return yin_lds(fp, f0_yin_frames_per_block(fp), need <= 6).total;
}

// RR >= lags per lane (fp.R), SS >= trough slots per lane (fp.slots): per-lane arrays are sized by them
// REF: the reference's shape (22050 Hz, frame_length 1024, C2..C7) compiled in -- periods 10..338, 339 lags in rows of 384,
// 329 kept, 168 candidates at most, 601 pitch bins: the kernel is short of scalar registers, and offsets become immediates.
// SH: 0 any shape (from the parameters); 1 the reference's (above; compact LDS layout); 2 the same pitch range at 16 kHz,
// frame_length 512 (BASELINE configs[2]): periods 7..245, 246 lags in rows of 256, 239 kept, 128 candidates; 3 the same at
// 44.1 kHz, frame_length 2048 (configs[4]): periods 21..675, 676 lags in rows of 704, 655 kept, 336 candidates.
template <int SH> struct YinShape { static constexpr int hop = 0, W = 0, n_fft = 0, R = 0, slots = 0, n_lag = 0, n_tau = 0, n_tau_pad = 0, min_period = 0, max_period = 0, cap = 0, n_bins = 0; };
template <> struct YinShape<1> { static constexpr int hop = 256, W = 512, n_fft = 1024, R = 6, slots = 6, n_lag = 329, n_tau = 339, n_tau_pad = 384, min_period = 10, max_period = 338, cap = 168, n_bins = 601; };
template <> struct YinShape<2> { static constexpr int hop = 128, W = 256, n_fft = 512, R = 4, slots = 4, n_lag = 239, n_tau = 246, n_tau_pad = 256, min_period = 7, max_period = 245, cap = 128, n_bins = 601; };
template <> struct YinShape<3> { static constexpr int hop = 512, W = 1024, n_fft = 2048, R = 11, slots = 11, n_lag = 655, n_tau = 676, n_tau_pad = 704, min_period = 21, max_period = 675, cap = 336, n_bins = 601; };
template <int SH>
static bool yin_shape_is(const F0Params& fp) {
  typedef YinShape<SH> Y;
  return fp.hop == Y::hop && fp.W == Y::W && fp.n_fft == Y::n_fft && fp.R == Y::R && fp.slots == Y::slots && fp.n_lag == Y::n_lag &&
         fp.n_tau == Y::n_tau && fp.n_tau_pad == Y::n_tau_pad && fp.min_period == Y::min_period && fp.max_period == Y::max_period &&
         fp.cap == Y::cap && fp.n_bins == Y::n_bins;
}
template <int RR, int SS, int FPB, int SH>
__global__ __launch_bounds__(256, ((SH == 1 || SH == 2) ? 4 : RR <= 6 ? 3 : 1)) void k_f0_yin(const float* __restrict__ ysig,
                                                const ClipDesc* __restrict__ clips,
                                                const ClipInfo* __restrict__ info,
                                                const float* __restrict__ energy,
                                                F0Tables tb, F0Params fp,
                                                int32_t* __restrict__ cand_cnt,
                                                double* __restrict__ cand_vp,
                                                int16_t* __restrict__ cand_bin,
                                                double* __restrict__ cand_prob,
                                                double* __restrict__ cand_lp,
                                                double* __restrict__ cand_lu) {
  extern __shared__ double smy[];
  const int clip = blockIdx.y;
  const ClipInfo ci = info[clip];
  if (ci.status == AFX_CLIP_NONFINITE) return;
  const int T = ci.T;
  const int t0 = blockIdx.x * FPB;
  if (t0 >= T) return;
  const ClipDesc cd = clips[clip];
  const int64_t np = ci.end - ci.start;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr bool YF = RR <= 6;
  typedef typename std::conditional<YF, float, double>::type ysig_t;
  typedef YinShape<SH> YS;
  constexpr bool REF = SH == 1, KN = SH != 0;
  const int hop = KN ? YS::hop : fp.hop, W = KN ? YS::W : fp.W, R = KN ? YS::R : fp.R, slots = KN ? YS::slots : fp.slots, n_lag = KN ? YS::n_lag : fp.n_lag;
  const int n_fft = KN ? YS::n_fft : fp.n_fft, n_tau = KN ? YS::n_tau : fp.n_tau, n_tau_pad = KN ? YS::n_tau_pad : fp.n_tau_pad;
  const int min_period = KN ? YS::min_period : fp.min_period, max_period = KN ? YS::max_period : fp.max_period, cap = KN ? YS::cap : fp.cap;
  const int n_bins = KN ? YS::n_bins : fp.n_bins;
  const YinLds L = yin_lds_i(hop, n_fft, n_tau_pad, slots, cap, FPB, YF, REF);
  static_assert(!REF || 64 + 168 + 84 <= 384, "CP / CB behind the event flags of the D row");
  ysig_t* Y = reinterpret_cast<ysig_t*>(smy);
  double* const smd = smy + (YF ? L.span / 2 : L.span);       // the double arrays behind the staged signal
  double* D = smd + (size_t)wave * L.per_wave;
  double* X = D + n_tau_pad;
  double* CP = REF ? D + 64 : X + slots * 64 + 2;
  int* CB = reinterpret_cast<int*>(CP + cap);
  // the probability tables are read inside the threshold loop with data-dependent indices: keep them in LDS
  double* Tthr = smd + 4 * L.per_wave;
  double* Tfact = Tthr + 101;
  double* Texp = Tfact + cap + 1;
  double* TbetaL = Texp + cap + 1;                              // (not REF) the two uniformly indexed tables
  double* TcumL = TbetaL + 100;
  for (int i = tid; i < 101; i += 256) Tthr[i] = tb.thr[i];
  for (int i = tid; i <= cap; i += 256) { Tfact[i] = tb.bfact[i]; Texp[i] = tb.bexp[i]; }
  if constexpr (!REF) {
    for (int i = tid; i < 101; i += 256) TcumL[i] = tb.cumbeta[i];
    for (int i = tid; i < 100; i += 256) TbetaL[i] = tb.beta[i];
  }
  cdouble_k* const beta_k = as_constant(tb.beta);
  cdouble_k* const cum_k = as_constant(tb.cumbeta);
  auto Tbeta_at = [&](int i) -> double { if constexpr (REF) return beta_k[i]; else return TbetaL[i]; };
  auto Tcum_at = [&](int i) -> double { if constexpr (REF) return cum_k[i]; else return TcumL[i]; };

  {
    const int64_t g0 = (int64_t)t0 * hop - n_fft / 2;
    const float* y = ysig + cd.off;
    for (int i = tid; i < (int)L.span; i += 256) {
      const int64_t g = g0 + i;
      Y[i] = (g >= 0 && g < np) ? (ysig_t)y[g] : (ysig_t)0;
    }
  }
  __syncthreads();

  const bool shared_chunks = W == 2 * hop;
  // AFX_F0_DEBUG & 32: cycles per phase of workgroup (0, 0), printed per wave (developer aid)
  const int ydbg = kF0Dbg ? fp.debug : 0;
  const bool stamping = kF0Dbg && (ydbg & 32) != 0;
  unsigned long long ph[8] = {}, ph_t = 0;
  auto stamp = [&](int i) {
    if (stamping) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_sched_barrier(0);
      if (i >= 0) ph[i] += now - ph_t;
      ph_t = now;
    }
  };
  double carry[RR];
#pragma unroll
  for (int r = 0; r < RR; ++r) carry[r] = 0.0;
  for (int fi = 0; fi < FPB / 4; ++fi) {
    const int f = wave * (FPB / 4) + fi;
    const int t = t0 + f;
    if (t >= T) break;                                   // wave-uniform
    const int64_t slot = cd.frame_base + t;
    const ysig_t* F = Y + (size_t)f * hop;

    stamp(-1);
    // ---- autocorrelation acf[tau] = sum_{i=1..W} y[i] y[i + tau]  (what irfft(rfft(y) rfft(y[W:0:-1])) [W:] is).
    // With W = 2 hop a frame is two hop-sized chunks and neighbouring frames share one: the wave keeps the
    // previous chunk's partial sums and adds one new chunk per frame (5 chunks for its 4 frames instead of 8).
    double acc[RR];
    // The window is walked 64 steps apart together: for i = i0 + 64 g the operand y[i + lane + 64 r] is the one of
    // (i0, r + g), so G steps share G + RR - 1 window reads instead of G * RR -- the loop is bound by LDS bandwidth
    // (eight waves per CU reading 512 bytes per operand), and this cuts its traffic by ~2.5.
    auto chunk_g = [&](const ysig_t* base, double (&out)[RR], auto Gt) {
      constexpr int G = decltype(Gt)::value;
#pragma unroll 2
      for (int i0 = 1; i0 <= 64; ++i0) {
        const ysig_t* q = base + i0 + lane;
        double v[G + RR - 1], yg[G];
#pragma unroll
        for (int m = 0; m < G + RR - 1; ++m) v[m] = (double)q[64 * m];
#pragma unroll
        for (int g = 0; g < G; ++g) yg[g] = (double)base[i0 + 64 * g];
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
          for (int r = 0; r < RR; ++r) out[r] = fma(yg[g], v[g + r], out[r]);
        }
      }
    };
    auto chunk = [&](const ysig_t* base, int len, double (&out)[RR]) {     // out[r] = sum_{i=1..len} base[i] base[i + lane + 64 r]
#pragma unroll
      for (int r = 0; r < RR; ++r) out[r] = 0.0;
      if (len == 256) { chunk_g(base, out, std::integral_constant<int, 4>()); return; }
      if (len == 512) { chunk_g(base, out, std::integral_constant<int, 8>()); return; }
      if (len == 128) { chunk_g(base, out, std::integral_constant<int, 2>()); return; }
#pragma unroll 4
      for (int i = 1; i <= len; ++i) {
        const double yi = (double)base[i];
        const ysig_t* q = base + i + lane;
#pragma unroll
        for (int r = 0; r < RR; ++r) out[r] = fma(yi, (double)q[64 * r], out[r]);
      }
    };
    if (shared_chunks) {
      double cur[RR];
      if (fi == 0) chunk(F, hop, carry);
      chunk(F + hop, (ydbg & 4) ? 8 : hop, cur);
#pragma unroll
      for (int r = 0; r < RR; ++r) { acc[r] = carry[r] + cur[r]; carry[r] = cur[r]; }
    } else {
      chunk(F, (ydbg & 4) ? 8 : W, acc);
    }
    stamp(0);
    // ---- difference function d = (e[0] + e[tau]) [float32] - 2 acf [float64]
    const float* Erow = energy + slot * (int64_t)n_tau_pad;
    const float e0 = Erow[0];
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const int tau = lane + 64 * r;
      if (r < R && tau < n_tau) {
        double a = acc[r];
        if (fabs(a) < 1e-6) a = 0.0;
        const float s32 = e0 + Erow[tau];
        D[tau] = (double)s32 - 2.0 * a;
      }
    }
    F0_WAVE_SYNC();
    // ---- cumulative mean over tau = 1 .. max_period, kept for tau >= min_period
    {
      const int C = (max_period + 63) / 64;
      const int lo = 1 + lane * C;
      double s = 0.0;
      for (int k = 0; k < C; ++k) { const int tau = lo + k; if (tau <= max_period) s += D[tau]; }
      double incl = s;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const double v = shfl_up_d(incl, o); if (lane >= o) incl += v; }
      double run = incl - s;
      for (int k = 0; k < C; ++k) {
        const int tau = lo + k;
        if (tau <= max_period) {
          run += D[tau];
          if (tau >= min_period) X[tau - min_period] = D[tau] / (run / (double)tau + fp.tiny);
        }
      }
    }
    F0_WAVE_SYNC();
    stamp(1);
    // ---- troughs (librosa.util.localmin, with the pyin rule for index 0), compacted in increasing lag:
    // at most every second lag is a trough, so the threshold loop below runs over CS = ceil(n_troughs / 64)
    // dense slots instead of one slot per 64 lags
    constexpr int CSM = (SS + 1) / 2;                    // dense trough slots per lane
    int n_tr = 0;
    {
#pragma unroll
      for (int s = 0; s < SS; ++s) {
        if (s < slots) {
          const int p = lane + 64 * s;
          bool t_here = false;
          double x = 0.0;
          if (p < n_lag) {
            x = X[p];
            const double xm = p > 0 ? X[p - 1] : 0.0, xp = p + 1 < n_lag ? X[p + 1] : 0.0;
            t_here = p == 0 ? (x < xp) : (p == n_lag - 1 ? (x < xm) : (x < xm && x <= xp));
          }
          const unsigned long long m = __ballot(t_here);
          if (t_here) { const int j = n_tr + lanes_below(m); CP[j] = x; CB[j] = p; }   // heights / lags, reused below
          n_tr += __popcll(m);
        }
      }
    }
    F0_WAVE_SYNC();
    stamp(2);
    int cnt = 0;
    double vp = 0.0;
    if (n_tr > 0) {
      const int CS = (n_tr + 63) >> 6;
      double h[CSM], pr[CSM];
      int lagp[CSM];
      bool tr[CSM];
#pragma unroll
      for (int c = 0; c < CSM; ++c) {
        const int j = lane + 64 * c;
        tr[c] = c < CS && j < n_tr;
        h[c] = tr[c] ? CP[j] : 0.0;
        lagp[c] = tr[c] ? CB[j] : 0;
        pr[c] = 0.0;
      }
      F0_WAVE_SYNC();
      // ---- probabilities: for every threshold, a Boltzmann prior over the troughs below it.  The thresholds rise,
      // so the set of troughs below one only grows, and it changes at no more than n_troughs of the 100 thresholds:
      // a trough enters at kin = the first threshold above its height; between two entry events ranks, count and
      // prior are constant, so they are formed once per run and the thresholds of the run only add prior * beta[k],
      // one after the other as before (the sums are bit-identical to re-ranking at every threshold; librosa takes
      // them as a BLAS dot, whose order is not pinned, and a strongly voiced frame's total decides between an
      // unvoiced observation of 0 and of 2e-19 -- see DESIGN.md 7 -- so the order is not touched).
      const int kmax = (ydbg & 8) ? 4 : kF0Thresholds;
      int kin[CSM];
#pragma unroll
      for (int c = 0; c < CSM; ++c) {
        kin[c] = 1 << 20;
        if (tr[c]) {
          int kk = (int)(fmin(fmax(h[c], 0.0), 2.0) * (double)kF0Thresholds) + 1;
          kk = kk < 1 ? 1 : (kk > kF0Thresholds ? kF0Thresholds : kk);
          while (kk > 1 && h[c] < Tthr[kk - 1]) --kk;                       // first k with h < thr[k]
          while (kk <= kF0Thresholds && !(h[c] < Tthr[kk])) ++kk;
          if (kk <= kF0Thresholds) kin[c] = kk;
        }
      }
      // the entry events as two 64-bit masks over k (a flag per threshold in the wave's idle D row, one ballot per
      // half): stepping from run to run is then scalar bit scanning, not a wave reduction
      int* EV = reinterpret_cast<int*>(D);
      EV[lane] = 0; EV[lane + 64] = 0;
      F0_WAVE_SYNC();
#pragma unroll
      for (int c = 0; c < CSM; ++c)
        if (c < CS && kin[c] <= kF0Thresholds) EV[kin[c]] = 1;
      F0_WAVE_SYNC();
      unsigned long long ev0 = __ballot(EV[lane] != 0), ev1 = __ballot(EV[lane + 64] != 0);
      F0_WAVE_SYNC();
      auto take_event = [&]() -> int {                                       // smallest remaining event, removed
        int k = 1 << 20;
        if (ev0) { k = (int)__builtin_ctzll(ev0); ev0 &= ev0 - 1; }
        else if (ev1) { k = 64 + (int)__builtin_ctzll(ev1); ev1 &= ev1 - 1; }
        return k;
      };
      int ka = take_event();
      while (ka <= kmax) {
        const int kn = take_event();                                         // the run is ka .. kn - 1
        const int kb = kn > kmax + 1 ? kmax + 1 : kn;
        int pos[CSM];
        int n = 0;
#pragma unroll
        for (int c = 0; c < CSM; ++c) {
          pos[c] = -1;
          if (c < CS) {
            const bool below = kin[c] <= ka;
            const unsigned long long m = __ballot(below);
            pos[c] = below ? n + lanes_below(m) : -1;
            n += __popcll(m);
          }
        }
        const double fact = Tfact[n];
        double fe[CSM];                                                      // prior of this run; 0 for a trough not below
#pragma unroll
        for (int c = 0; c < CSM; ++c) fe[c] = (c < CS && pos[c] >= 0) ? fact * Texp[pos[c]] : 0.0;
#pragma unroll 4
        for (int k = ka; k < kb; ++k) {
          const double bk = Tbeta_at(k - 1);
#pragma unroll
          for (int c = 0; c < CSM; ++c)
            if (c < CS) pr[c] += fe[c] * bk;
        }
        ka = kn;
      }
      stamp(3);
      // global minimum (first occurrence) collects the mass of the thresholds it does not undercut
      double hm = INFINITY;
#pragma unroll
      for (int c = 0; c < CSM; ++c)
        if (tr[c]) hm = fmin(hm, h[c]);
      hm = wave_min_d(hm);
      int jm = 1 << 30;
#pragma unroll
      for (int c = 0; c < CSM; ++c)
        if (tr[c] && h[c] == hm) jm = min(jm, lane + 64 * c);
      jm = wave_min_i(jm);
      int nbelow = 0;
      for (int k0 = 1; k0 <= kF0Thresholds; k0 += 64) {
        const int k = k0 + lane;
        nbelow += __popcll(__ballot(k <= kF0Thresholds && !(hm < Tthr[k <= kF0Thresholds ? k : kF0Thresholds])));
      }
      const double extra = fp.no_trough_prob * Tcum_at(nbelow);
#pragma unroll
      for (int c = 0; c < CSM; ++c)
        if (tr[c] && lane + 64 * c == jm) pr[c] += extra;
      stamp(4);
      // ---- candidates in increasing period: refine, map to a pitch bin
#pragma unroll
      for (int c = 0; c < CSM; ++c) {
        if (c < CS) {
          const bool nz = tr[c] && pr[c] != 0.0;
          const unsigned long long m = __ballot(nz);
          if (nz) {
            const int p = lagp[c];
            double shift = 0.0;
            if (p > 0 && p < n_lag - 1) {
              const double xm = X[p - 1], xp = X[p + 1];
              const double a = xp + xm - 2.0 * h[c];
              const double b = (xp - xm) / 2.0;
              shift = fabs(b) >= fabs(a) ? 0.0 : -b / a;
            }
            const double period = (double)(min_period + p) + shift;
            const double f0 = fp.sr / period;
            double bf = rint(fp.bins_per_octave * log2(f0 / fp.fmin));
            bf = bf < 0.0 ? 0.0 : (bf > (double)n_bins ? (double)n_bins : bf);
            const int j = cnt + lanes_below(m);
            CB[j] = (int)bf;
            CP[j] = pr[c];
          }
          cnt += __popcll(m);
        }
      }
      F0_WAVE_SYNC();
      stamp(5);
      // several candidates in one bin: the last (longest period) wins, as numpy's indexed assignment
      double val[CSM];                     // this lane's kept in-range candidates (0 otherwise), j = lane + 64 m
#pragma unroll
      for (int m = 0; m < CSM; ++m) {
        val[m] = 0.0;
        const int j = lane + 64 * m;
        if (j < cnt) {
          const int b = CB[j];
          const bool keep = (j == cnt - 1) || CB[j + 1] != b;
          const double pj = CP[j];
          cand_bin[slot * cap + j] = (int16_t)(keep ? b : -1);
          cand_lp[slot * cap + j] = log(pj + fp.tiny);              // what the Viterbi pass scatters (round 2: a pass of its own)
          if (cand_prob) cand_prob[slot * cap + j] = pj;            // diagnostics only (AFX_F0_DUMP)
          if (keep && b >= 0 && b < n_bins) val[m] = pj;
        }
      }
      // voiced_prob = np.sum(observation[:n_bins], axis=0): numpy adds the rows one after another, i.e. the kept
      // candidates in ascending bin order (descending j), sequentially.  The order is reproduced because the sum is
      // mathematically 1 for a strongly voiced frame and the unvoiced observation is (1 - sum) / n_bins: 0 or 1e-19.
      // The terms come out of the lanes' registers by readlane (a dropped candidate adds an exact +0.0): a loop over
      // LDS cost two dependent round trips per term.
#pragma unroll
      for (int m = CSM - 1; m >= 0; --m) {
        const int lo = 64 * m;
        const int hi = (cnt < lo + 64 ? cnt : lo + 64) - 1;
        const int vlo = __double2loint(val[m]), vhi = __double2hiint(val[m]);
        for (int j = hi; j >= lo; --j)
          vp += __hiloint2double(__builtin_amdgcn_readlane(vhi, j - lo), __builtin_amdgcn_readlane(vlo, j - lo));
      }
      F0_WAVE_SYNC();
    }
    {
      const double vpc = vp < 0.0 ? 0.0 : (vp > 1.0 ? 1.0 : vp);
      const double lu = log((1.0 - vpc) / (double)n_bins + fp.tiny);   // the unvoiced bins' common log observation
      if (lane == 0) {
        cand_cnt[slot] = cnt;
        cand_vp[slot] = vpc;
        cand_lu[slot] = lu;
      }
    }
    stamp(6);
  }
  if (stamping && blockIdx.x == 1 && blockIdx.y == 0 && lane == 0)
    printf("yin wave %d: acf %llu diff+cmean %llu troughs %llu thresholds %llu minimum %llu candidates %llu write+vp %llu\n", wave,
           ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6]);
}

// ---------------------------------------------------------------------------------------------
// k_f0_viterbi: librosa.sequence.viterbi on the (sparse) observation columns, then the statistics.
// States 0 .. n_bins-1 voiced, n_bins .. 2 n_bins-1 unvoiced.  Transition kron([[.99,.01],[.01,.99]],
// local): a source row is a triangle of half-width `band` around it, cut at the range ends and
// normalised, so log(A + tiny) has 2 x (2 band + 1) distinct rows of (2 band + 1) entries -- held in
// LDS -- and log(tiny) everywhere else.  Out-of-band moves therefore all cost the same: the best of them
// is the previous column's global maximum, which is computed once per step.
//
// Back-pointers are not formed in the forward pass.  Back-tracking visits one state per step, so only T of
// the T x 2 n_bins arg-maxima are ever used: the forward pass keeps the *values* (add + max per band entry,
// half the instructions of a running arg-max and no index registers, which is what lets two workgroups
// share a CU) and writes every value column to HBM (2 n_bins doubles per frame); the backward pass (k_f0_backtrack) then
// recomputes, for the one state on the path, the same sums in the same order and takes their first maximum
// (numpy's argmax rule) -- bit-identical to forming all pointers up front.
// ---------------------------------------------------------------------------------------------
constexpr int kVitThreads = 640;
__device__ __forceinline__ int vit_role(int hw) {       // hardware wave -> the block of 64 targets it owns (a permutation)
  if (kVitThreads != 640) return hw;
  const unsigned long long roles = 0x1876549032ull;    // hw 0..9 -> 2 3 0 9 4 5 6 7 8 1 (nibble per wave, low first)
  return (int)((roles >> (4 * hw)) & 15);
}

struct VitLds { size_t v, olp, lt, red, edge, total; };
__host__ __device__ inline VitLds vit_lds(int n_bins, int band_) {
  const size_t S = 2 * (size_t)n_bins, width = 2 * (size_t)band_ + 1;
  VitLds L;
  L.v = 0;                                   // two value columns of 2 (n_bins + 2 band) + 4 band doubles
  L.olp = 2 * (S + 8 * band_);               // 3 n_bins doubles
  L.lt = L.olp + 3 * n_bins;                 // width * width doubles: the `stay` rows (k_f0_backtrack holds both tables)
  L.red = L.lt + width * width;              // 32 doubles + 32 ints (16 doubles): two sets of per-wave partials
  L.edge = L.red + 48;                       // per wave 4 x 2 band doubles (its best edge-class move per range-end target); 2 counters
  L.total = (L.edge + (size_t)(kVitThreads / 64) * 8 * band_ + 2) * sizeof(double);
  return L;
}
size_t f0_viterbi_lds_bytes(const F0Params& fp) { return vit_lds(fp.n_bins, fp.band).total; }

// MODE 0: production; 1: the timing-only ablation bits of AFX_F0_DEBUG honoured; 2: per-phase cycle stamps as well.
// NBT / BANDT: n_bins and band compiled in (0: taken from the parameters) -- the step loop is short of scalar registers,
// and with the reference's shape (601 bins, band 25) as constants the column strides and table offsets are immediates.
template <int MODE, int NBT, int BANDT>
// (eight waves per SIMD asked for, i.e. at most 64 VGPRs: two 10-wave workgroups per CU need six, but above 64 registers the
// kernel runs as if it had the CU to itself -- 68 VGPRs: 23.1 -> 26.2 ms, profiles/r03_ab_runs.txt)
__global__ __launch_bounds__(kVitThreads, 8) void k_f0_viterbi(const ClipDesc* __restrict__ clips,
                                                               const ClipInfo* __restrict__ info,
                                                               F0Tables tb, F0Params fp,
                                                               const int32_t* __restrict__ cand_cnt,
                                                               const int16_t* __restrict__ cand_bin,
                                                               const double* __restrict__ cand_lp,
                                                               const double* __restrict__ cand_lu,
                                                               double* __restrict__ vrows,
                                                               VitBest* __restrict__ vbest) {
  extern __shared__ double smv[];
  const int clip = blockIdx.x;
  const ClipInfo ci = info[clip];
  // AFX_F0_DEBUG & 16: per-phase cycle counts of workgroup 0, printed per wave (developer aid)
  unsigned long long ph[8] = {}, ph_t = 0;
  auto stamp = [&](int i) {
    if constexpr (MODE == 2) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_sched_barrier(0);
      if (i >= 0) ph[i] += now - ph_t;
      ph_t = now;
    }
  };
  constexpr int NW = kVitThreads / 64;
  static_assert(NW <= 16, "the per-wave partials are combined inside one DPP row");
  // A workgroup's ten waves land 3 / 3 / 2 / 2 on the CU's four SIMDs (wave w on SIMD w % 4), and so do those of the
  // second workgroup of the CU.  The two waves that own the targets next to the range ends issue about 1.6 times the
  // instructions of the others (the edge-class sources, below): they take the two SIMDs that hold two waves, not three.
  const int lane = threadIdx.x & 63;
  const int wave = vit_role(threadIdx.x >> 6);
  const int tid = wave * 64 + lane;
  if (ci.status == AFX_CLIP_NONFINITE || ci.T < 1) return;       // k_f0_backtrack writes the clip's (empty) results
  const ClipDesc cd = clips[clip];
  const int T = ci.T, nb = NBT ? NBT : fp.n_bins, S = 2 * nb, band = BANDT ? BANDT : fp.band, width = 2 * band + 1;
  const int dbg = MODE ? fp.debug : 0;
  const VitLds L = vit_lds(nb, band);
  const int VM = nb + 2 * band;                                 // main cells per voicing (guards included)
  const int VS = 2 * VM + 4 * band;                             // doubles per value column: main[2], edge[2]
  double* const vbuf = smv + L.v;
  double* vprev = vbuf;
  double* vcur = vbuf + VS;
  double* olp3 = smv + L.olp;                                   // three observation columns in rotation
  double* LT = smv + L.lt;
  double* redv = smv + L.red;
  int* redi = reinterpret_cast<int*>(redv + 32);
  double* const ebuf = smv + L.edge;                            // [wave][low | high][voiced | unvoiced][2 band]
  int* const ecnt = reinterpret_cast<int*>(ebuf + NW * 8 * band);   // waves that have delivered their share
  const double c0 = fp.c0;
  double* const vclip = vrows + (size_t)cd.frame_base * S;      // this clip's value columns, row t at t * S
  VitBest* const bclip = vbest + cd.frame_base;                 // (max, first argmax) of column t - 1 at [t]

  for (int i = tid; i < width * width; i += kVitThreads) LT[i] = tb.lt[i];
  for (int i = tid; i < 3 * nb; i += kVitThreads) olp3[i] = c0;
  for (int i = tid; i < 2 * VS; i += kVitThreads) vbuf[i] = -INFINITY;      // guards and edge-class cells stay -inf
  if (tid < 2) ecnt[tid] = 0;
  const double lpi_u = log(1.0 / (double)nb + fp.tiny);

  // (max value, lowest index) over the wave of a per-lane (value, index): DPP butterflies, no LDS round trips
  auto wave_best = [&](double& bv, int& bi) {
    const double m = wave_max_dpp(bv);
    const unsigned long long at = __ballot(bv == m);                       // lanes that hold the maximum (never none)
    if (__popcll(at) == 1) bi = __builtin_amdgcn_readlane(bi, (int)__builtin_ctzll(at));   // the usual case: one holder
    else bi = wave_min_dpp(bv == m ? bi : (1 << 30));                     // a tie: the lowest state index among them
    bv = m;
  };
  // the same over the first 16 lanes only (the per-wave partials): one DPP row, one readlane
  auto row0_best = [&](double& bv, int& bi) {
    double v = bv;
    v = max_nn(v, dpp_dd<0xB1>(v)); v = max_nn(v, dpp_dd<0x4E>(v)); v = max_nn(v, dpp_dd<0x141>(v)); v = max_nn(v, dpp_dd<0x140>(v));
    const double m = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 0), __builtin_amdgcn_readlane(__double2loint(v), 0));
    const unsigned long long at = __ballot(bv == m) & 0xffffull;
    if (__popcll(at) == 1) bi = __builtin_amdgcn_readlane(bi, (int)__builtin_ctzll(at));
    else {
      int x = bv == m ? bi : (1 << 30);
      x = min(x, F0_DPP_I(x, 0xB1)); x = min(x, F0_DPP_I(x, 0x4E)); x = min(x, F0_DPP_I(x, 0x141)); x = min(x, F0_DPP_I(x, 0x140));
      bi = __builtin_amdgcn_readlane(x, 0);
    }
    bv = m;
  };
  // The block-wide (max, first arg-max) of a column is assembled from per-wave partials that the producing step
  // leaves in LDS (two sets, alternating), so that a step needs a single barrier.
  auto put_partial = [&](int set, double bv, int bi) {
    wave_best(bv, bi);
    if (lane == 0) { redv[set * 16 + wave] = bv; redi[set * 16 + wave] = bi; }
  };
  auto get_best = [&](int set, double& gmax, int& garg) {
    double bv = -INFINITY; int bi = 1 << 30;
    if (lane < NW) { bv = redv[set * 16 + lane]; bi = redi[set * 16 + lane]; }
    row0_best(bv, bi);
    gmax = bv; garg = bi;
  };

  const double* LTs = LT;                               // stay (the switch table is k_f0_backtrack's business)
  // the interior row class (every source at least `band` bins from both range ends) is read through the scalar
  // cache: its index is wave-uniform, so the band walk's weights cost no LDS traffic and no vector registers
  cdouble_k* const kk = as_constant(static_cast<const double*>(__builtin_assume_aligned(tb.ltw, 64)));   // the stay weights in walk order (hipMalloc'ed)

  // Value columns in LDS, two in rotation.  A column is kept as
  //   main[v][band + b]  b = -band .. nb - 1 + band: the value of (voicing v, bin b) when b is an interior-class
  //                      source, -inf in the guard cells and for the edge-class sources (b < band, b >= nb - band),
  //   edge[v][k]         the true values of the 2 band edge-class sources (low ones first),
  // so that *every* target walks its band with the interior row's weights and no masking (a missing source reads
  // -inf), and only the two waves that own targets within 2 band of a range end add the edge-class sources, one
  // uniform table row per source.
  // What the columns hold is already folded over the source's voicing: with d = log(.99) - log(.01) (taken from the
  // table: stay minus switch entry), a move into a voiced target costs max(v_voiced, v_unvoiced - d) + stay, into an
  // unvoiced one max(v_voiced - d, v_unvoiced) + stay, so a band entry is one add and one max per target instead of
  // two each.  (v_unvoiced - d) + stay is (v_unvoiced + switch) up to the rounding of d -- at most an ulp of the
  // value; the backward pass, which decides the path, takes the arg-max of the recurrence's own sums.)
  const double dsw = tb.lt[band] - tb.lt[(size_t)width * width + band];
  auto put_value = [&](double* col, int jb, double xv, double xu) {
    const double cv = fmax(xv, xu - dsw), cu = fmax(xv - dsw, xu);
    if (jb < band) { col[2 * VM + jb] = cv; col[2 * VM + 2 * band + jb] = cu; }
    else if (jb >= nb - band) { const int k = band + jb - (nb - band); col[2 * VM + k] = cv; col[2 * VM + 2 * band + k] = cu; }
    else { col[band + jb] = cv; col[VM + band + jb] = cu; }
  };

  // Observation columns: three LDS columns in rotation, all log(0) except where a step's candidates were
  // scattered.  During step t a thread scatters its candidate of step t + 1 (loaded a step earlier) and takes
  // back the one it scattered for step t - 1; the three touch different columns, so the step's one barrier
  // orders everything.  The logs come from k_f0_yin.
  auto load_cand = [&](int t, int& bin, double& lp, double& lu_) {
    const int64_t ns = cd.frame_base + t;
    const int cnt = cand_cnt[ns];
    lu_ = cand_lu[ns];
    bin = -1; lp = 0.0;
    if (tid < cnt && !(dbg & 128)) { bin = cand_bin[ns * fp.cap + tid]; lp = cand_lp[ns * fp.cap + tid]; }   // 128: timing only
  };
  int nx_bin; double nx_lp, nx_lu;
  load_cand(0, nx_bin, nx_lp, nx_lu);
  __syncthreads();
  int bin_prev = -1, bin_cur = (nx_bin >= 0 && nx_bin < nb) ? nx_bin : -1;
  if (bin_cur >= 0) olp3[bin_cur] = nx_lp;
  double lu = nx_lu;
  if (T > 1) load_cand(1, nx_bin, nx_lp, nx_lu);
  __syncthreads();

  // targets of this thread: bin tid (+ a multiple of the block size); when one pass covers the range, the last
  // wave takes the top 64 bins so that the high-edge targets share one wave
  const bool single = nb <= kVitThreads && nb >= 128;
  // the weights of this wave's share of the edge-class moves (below): lane j's entry of the row of source w + i NW
  constexpr int kShare = BANDT > 0 ? (2 * BANDT + NW - 1) / NW : 1;
  double wsh[kShare];
  if constexpr (BANDT > 0) {
#pragma unroll
    for (int i = 0; i < kShare; ++i) {
      const int p = wave + i * NW;
      const bool low = p < band;
      const int k = low ? p : p - band;
      const int d = low ? lane - k + band : lane - k;            // jb - b + band, jb = lane | nb - 2 band + lane
      const int rc = low ? 1 + k : 2 * band - k;
      wsh[i] = p < 2 * band ? LTs[(unsigned)d <= (unsigned)(2 * band) ? rc * width + d : width] : 0.0;
    }
  }
  for (int t = 0; t < T; ++t) {
    stamp(-1);
    const double* olp = olp3 + (t % 3) * nb;
    // ---- the next step's observation column, the one before's taken back
    if (bin_prev >= 0) olp3[((t + 2) % 3) * nb + bin_prev] = c0;
    int bin_next = -1; double lu_next = 0.0;
    if (t + 1 < T) {
      bin_next = (nx_bin >= 0 && nx_bin < nb) ? nx_bin : -1;
      if (bin_next >= 0) olp3[((t + 1) % 3) * nb + bin_next] = nx_lp;
      lu_next = nx_lu;
      if (t + 2 < T) load_cand(t + 2, nx_bin, nx_lp, nx_lu);
    }
    stamp(0);
    double* const vout = vclip + (size_t)t * S;
    double pbv = -INFINITY; int pbi = 1 << 30;          // this thread's (max, lowest index) of the new column
    auto note = [&](double xv, double xu, int jb, bool first) {
      if (first) { pbv = xv; pbi = jb; }                                 // a thread's first target: nothing to compare with
      else if (xv > pbv || (xv == pbv && jb < pbi)) { pbv = xv; pbi = jb; }
      if (xu > pbv || (xu == pbv && nb + jb < pbi)) { pbv = xu; pbi = nb + jb; }
    };
    double gmax = 0.0; int garg = 0;
    if (t > 0) {
      get_best(t & 1, gmax, garg);
      if (tid == 0) { VitBest vb; vb.value = gmax; vb.arg = garg; vb.pad = 0; bclip[t] = vb; }
    }
    stamp(1);
    const int gb = garg >= nb ? garg - nb : garg;
    const double* m0 = vprev;                            // folded for voiced targets, main cells
    const double* m1 = vprev + VM;                       // folded for unvoiced targets
    const double* e0 = vprev + 2 * VM;                   // edge-class sources
    const double* e1 = e0 + 2 * band;
    // ---- the edge-class sources' moves, every wave a share (one pass over the range: `single`).  The 2 band sources
    // next to a range end have a transition row of their own each, and the 2 band targets they reach were two waves'
    // business: 2 band extra passes of ten instructions on top of a walk of 2 band + 1 entries, with eight waves
    // waiting at the barrier.  Now wave w takes sources w, w + NW, ...: lane j forms the move into range-end target j
    // (the row is uniform; a lane out of the source's band reads a log(0) cell), the wave's best goes into its row of
    // an LDS array, and the wave that owns the targets takes the maximum over the rows behind its own walk -- by then
    // the others have long delivered (it checks a counter).  max is exact and order-free: the same values as before.
    // (ds_max_f64 into one shared row was tried first: 3.4 ms slower than no sharing, the atomics serialise.)
    const bool share_edges = single && t > 0 && !(dbg & 4) && !(dbg & 256);      // 256: the owners' own passes (A/B)
    if (share_edges) {
      double lv = -INFINITY, lu_ = -INFINITY, hv = -INFINITY, hu = -INFINITY;
      if constexpr (BANDT > 0) {                                   // this wave's sources and their weights never change
#pragma unroll
        for (int i = 0; i < kShare; ++i) {
          const int p = wave + i * NW;
          if (p < 2 * band) {
            const double sv = e0[p] + wsh[i], su = e1[p] + wsh[i];
            if (p < band) { lv = max_nn(lv, sv); lu_ = max_nn(lu_, su); } else { hv = max_nn(hv, sv); hu = max_nn(hu, su); }
          }
        }
      } else {
      for (int p = wave; p < 2 * band; p += NW) {
        const bool low = p < band;
        const int k = low ? p : p - band;
        const int d = low ? lane - k + band : lane - k;          // jb - b + band, jb = lane | nb - 2 band + lane
        const int rc = low ? 1 + k : 2 * band - k;
        const double ws = LTs[(unsigned)d <= (unsigned)(2 * band) ? rc * width + d : width];
        const double sv = e0[p] + ws, su = e1[p] + ws;
        if (low) { lv = fmax(lv, sv); lu_ = fmax(lu_, su); } else { hv = fmax(hv, sv); hu = fmax(hu, su); }
      }
      }
      if (lane < 2 * band) {
        double* const mine = ebuf + wave * 8 * band + lane;
        mine[0] = lv; mine[2 * band] = lu_; mine[4 * band] = hv; mine[6 * band] = hu;
      }
      if (lane == 0) __hip_atomic_fetch_add(ecnt, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    for (int base = 0; base < nb; base += kVitThreads) {
      int jb = base + tid;
      bool live = jb < nb;
      if (single) {
        if (wave == NW - 1) { jb = nb - 64 + lane; live = true; }
        else live = tid < nb - 64;
      }
      if (!__any(live)) continue;
      const int jc = live ? jb : band;                   // idle lanes walk somewhere harmless
      double xv, xu;
      if (t == 0) {
        xv = olp[jc] + c0; xu = lu + lpi_u;
      } else {
        double bv = -INFINITY, bu = -INFINITY;           // best move into (voiced jb), (unvoiced jb)
        if (!(dbg & 1)) {
          const double* p0 = m0 + jc;                    // cells of the sources jc - band .. jc + band
          const double* p1 = m1 + jc;
#pragma unroll 4
          for (int e = 0; e < width; ++e) {
            const double ws = kk[e];
            bv = fmax(bv, p0[e] + ws);
            bu = fmax(bu, p1[e] + ws);
          }
          // edge-class source b -> target jb: LT[rc][jb - b + band], rc = 1 + b (low), 1 + band + (nb - 1 - b)
          // (high).  A lane whose target is out of that source's band reads a log(0) cell instead (row 1,
          // entry 0): harmless, the best out-of-band move below is at least as good.
          auto edge_pass = [&](int b, int k, int rc) {
            const int d = jc - b + band;
            const int idx = (unsigned)d <= (unsigned)(2 * band) ? rc * width + d : width;
            const double ws = LTs[idx];
            bv = fmax(bv, e0[k] + ws);
            bu = fmax(bu, e1[k] + ws);
          };
          if (share_edges) {
            const bool lo_owner = wave == 0, hi_owner = wave == NW - 1;
            if (lo_owner || hi_owner) {
              // the counter runs on: NW more per step (t = 1 is the first step with shares)
              while (__hip_atomic_load(ecnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < NW * t) __builtin_amdgcn_s_sleep(1);
              const int jr = lo_owner ? lane : lane - (64 - 2 * band);      // target index within the range end
              const int j = jr < 0 ? 0 : (jr >= 2 * band ? 2 * band - 1 : jr);   // (the wave's other lanes read along)
              const double* eb = ebuf + (hi_owner ? 4 * band : 0) + j;
              double ev = -INFINITY, eu = -INFINITY;
#pragma unroll
              for (int w = 0; w < NW; ++w) { ev = max_nn(ev, eb[w * 8 * band]); eu = max_nn(eu, eb[w * 8 * band + 2 * band]); }
              if (jr == j) { bv = max_nn(bv, ev); bu = max_nn(bu, eu); }
            }
          } else if (!(dbg & 4)) {
            if (__any(live && jb < 2 * band)) {
#pragma unroll 5
              for (int b = 0; b < band; ++b) edge_pass(b, b, 1 + b);
            }
            if (__any(live && jb > nb - 1 - 2 * band)) {
#pragma unroll 5
              for (int k = 0; k < band; ++k) edge_pass(nb - band + k, band + k, 2 * band - k);
            }
          }
        }
        const int blo = jc - band < 0 ? 0 : jc - band, bhi = jc + band > nb - 1 ? nb - 1 : jc + band;
        if (gb < blo || gb > bhi) {                      // the best out-of-band source
          const double cand = gmax + c0;
          bv = fmax(bv, cand); bu = fmax(bu, cand);
        }
        xv = olp[jc] + bv; xu = lu + bu;
      }
      stamp(2);
      if (live) {
        put_value(vcur, jb, xv, xu);
        if (!(dbg & 64)) { vout[jb] = xv; vout[nb + jb] = xu; }      // 64: timing only, no column stores
        note(xv, xu, jb, base == 0);
      }
    }
    stamp(3);
    put_partial((t + 1) & 1, pbv, pbi);
    bin_prev = bin_cur; bin_cur = bin_next; lu = lu_next;
    double* tmp = vprev; vprev = vcur; vcur = tmp;
    stamp(4);
    __syncthreads();
    stamp(5);
  }
  // ---- the last column's (max, first arg-max): where k_f0_backtrack starts
  double gmax; int garg;
  get_best(T & 1, gmax, garg);
  if (tid == 0) { VitBest vb; vb.value = gmax; vb.arg = garg; vb.pad = 0; bclip[0] = vb; }
  stamp(6);
  if constexpr (MODE == 2) {
    if (clip == 0 && lane == 0)
      printf("vit wave %d T %d: top %llu best %llu walk %llu tail %llu partial %llu barrier %llu end %llu\n", wave, T,
             ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6]);
  }
}

// ---------------------------------------------------------------------------------------------
// k_f0_backtrack: the path through the value columns k_f0_viterbi left behind, then the statistics.
// One wave per clip, four clips per workgroup.  A step is short (the first maximum over the 2 (2 band + 1) moves into
// the one state on the path), but it cannot start before the step after it has named that state, and its operands --
// 2 band + 1 cells of each half of one column -- are 9.6 KB apart from the previous step's, in HBM: inside
// k_f0_viterbi's workgroup the walk cost one memory round trip per step (2.2 us, 1.9 ms per clip, with every other wave
// of the workgroup and the CU's LDS waiting).  Here the columns come through an LDS-DMA ring D - 1 steps ahead of their
// use: a move inside the band shifts the bin by at most `band`, so the window of +- (D - 1) band bins around the bin of
// the step just decided holds whatever band the step D - 1 later will need -- unless the path takes an out-of-band
// move in between, in which case that step reads its cells directly (and waits for them).  The ring is read with
// opaque ds_read instructions: the compiler orders every LDS access it can see behind *all* outstanding LDS-DMA
// (s_waitcnt vmcnt(0)), which would make the ring one slot deep.  For the same reason the loop has no other LDS
// access and no other memory instruction: the per-column (max, arg-max) records and the states travel 64 steps at a
// time through the lanes' registers.
// ---------------------------------------------------------------------------------------------
constexpr int kBtWaves = 4;
constexpr int kBtWin = 256;                       // doubles per half column in a ring slot: two 1 KB DMA instructions
__host__ __device__ inline int f0_bt_depth(int band) { return 2 * band * 5 + 2 <= kBtWin ? 6 : 5; }
size_t f0_backtrack_lds_bytes(const F0Params& fp) {
  const size_t width = 2 * (size_t)fp.band + 1;
  return (2 * width * width + 2 + (size_t)kBtWaves * (f0_bt_depth(fp.band) + 1) * 2 * kBtWin) * sizeof(double);
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
#pragma clang diagnostic pop
}
__device__ __forceinline__ double lds_read_opaque(unsigned addr) {      // complete after lds_wait_opaque()
  double v;
  asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void lds_wait_opaque(double& a, double& b, double& c, double& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}

template <int D>
__global__ __launch_bounds__(64 * kBtWaves) void k_f0_backtrack(const ClipDesc* __restrict__ clips,
                                                                const ClipInfo* __restrict__ info, F0Tables tb,
                                                                F0Params fp, const double* __restrict__ vrows,
                                                                const VitBest* __restrict__ vbest,
                                                                uint16_t* __restrict__ states,
                                                                double* __restrict__ out_stats,
                                                                double* __restrict__ out_f0,
                                                                const int64_t* __restrict__ f0_offsets, int n_clips) {
  extern __shared__ double smb[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nb = fp.n_bins, S = 2 * nb, band = fp.band, width = 2 * band + 1;
  const int ltn = 2 * width * width;
  for (int i = threadIdx.x; i < ltn; i += 64 * kBtWaves) smb[i] = tb.lt[i];
  __syncthreads();                                                  // the only barrier: from here on the waves are on their own
  const int clip = blockIdx.x * kBtWaves + wave;
  if (clip >= n_clips) return;
  const ClipInfo ci = info[clip];
  const ClipDesc cd = clips[clip];
  double* st = out_stats + (size_t)clip * 4;
  if (ci.status == AFX_CLIP_NONFINITE || ci.T < 1) {
    if (lane == 0) { st[0] = 0.0; st[1] = 0.0; st[2] = 1.0; st[3] = 0.0; }
    if (out_f0) for (int t = lane; t < cd.tmax; t += 64) out_f0[f0_offsets[clip] + t] = (double)NAN;
    return;
  }
  const int T = ci.T;
  const double c0 = fp.c0;
  const double* const vclip = vrows + (size_t)cd.frame_base * S;
  const VitBest* const bclip = vbest + cd.frame_base;
  uint16_t* const sts = states + cd.frame_base;
  double* const ring = smb + ((ltn + 1) & ~1) + (size_t)wave * (D + 1) * 2 * kBtWin;
  const unsigned lt_a = lds_addr(smb), ring_a = lds_addr(ring);
  const int HW = band * (D - 1);

  // the window of a column around bin `cen`: each half starts at an even element of the column (16-byte DMA pieces)
  auto win_lo = [&](int cen, int v) { int b0 = cen - HW; b0 = b0 < 0 ? 0 : b0; return (v * nb + b0) & ~1; };
  auto prefetch = [&](int c, int cen) {                              // four DMA instructions, always
    const int slot = c < 0 ? D : c % D;                              // past the first column: into the spare slot
    const double* R = vclip + (size_t)(c < 0 ? 0 : c) * S;
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const double* src = R + win_lo(cen, v) + 2 * lane;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 128 * i),
                                         (__attribute__((address_space(3))) void*)(ring + (slot * 2 + v) * kBtWin + 128 * i),
                                         16, 0, 0);
    }
  };

  const VitBest last = bclip[0];                                     // (max, first arg-max) of column T - 1
  int cur = last.arg;
  int cens = 0;                                                      // lane s: the centre slot s was fetched around
  {
    const int j0 = cur < nb ? cur : cur - nb;
    for (int k = 0; k < D - 1; ++k) {                                // columns T - 2 .. T - D
      const int c = T - 2 - k;
      prefetch(c, j0);
      if (c >= 0) cens = (lane == (c % D) ? (j0) : cens);
    }
  }
  int pend = 0;                                                      // lane i: the state of step t_hi - i, not yet stored
  int t_hi = T - 1;
  pend = (lane == (0) ? (cur) : pend);
  double s1 = 0.0, c1 = 0.0;                                         // sum and count of the voiced frames' frequencies
  auto tally = [&](int q) { if (q < nb) { s1 += tb.freqs[q]; c1 += 1.0; } };

  for (int tb0 = T - 1; tb0 >= 1; tb0 -= 64) {                       // steps tb0, tb0 - 1, ... (64 at most)
    const int ti = tb0 - lane;
    double gv = 0.0; int ga = 0;
    if (ti >= 1) { const VitBest vb = bclip[ti]; gv = vb.value; ga = vb.arg; }
    int gvl = __double2loint(gv), gvh = __double2hiint(gv);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(gvl), "+v"(gvh), "+v"(ga)::"memory");     // here, not at the first use inside the loop
    const int nsteps = tb0 < 64 ? tb0 : 64;
    for (int k = 0; k < nsteps; ++k) {
      const int t = tb0 - k, c = t - 1;                              // the move into state `cur` of frame t, out of column c
      const double g_value = __hiloint2double(__builtin_amdgcn_readlane(gvh, k), __builtin_amdgcn_readlane(gvl, k));
      const int g_arg = __builtin_amdgcn_readlane(ga, k);
      const bool tv = cur < nb;
      const int jb = tv ? cur : cur - nb;
      const int b = jb - band + lane;                                // this lane's source bin
      const bool ok = lane < width && b >= 0 && b < nb;
      const int blo = jb - band < 0 ? 0 : jb - band, bhi = jb + band > nb - 1 ? nb - 1 : jb + band;
      const int slot = c % D;
      const int cen = __builtin_amdgcn_readlane(cens, slot);
      const int w0 = win_lo(cen, 0), w1 = win_lo(cen, 1);
      const bool hit = blo >= w0 && bhi - w0 < kBtWin && nb + blo >= w1 && nb + bhi - w1 < kBtWin;
      const int bc = ok ? b : blo;                                   // idle lanes read somewhere harmless
      const int rc = bc < band ? 1 + bc : (bc > nb - 1 - band ? 1 + band + (nb - 1 - bc) : 0);
      const int idx = rc * width + (jb - bc + band);                 // entry jb - b + band of that row
      double ws = lds_read_opaque(lt_a + 8u * (unsigned)idx);
      double ww = lds_read_opaque(lt_a + 8u * (unsigned)(width * width + idx));
      double a0, a1;
      if (hit) {
        // this column's four DMAs have landed once at most those of the D - 2 columns behind it are outstanding
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (D - 2)) : "memory");
        a0 = lds_read_opaque(ring_a + 8u * (unsigned)((slot * 2 + 0) * kBtWin + (bc - w0)));
        a1 = lds_read_opaque(ring_a + 8u * (unsigned)((slot * 2 + 1) * kBtWin + (nb + bc - w1)));
      } else {
        const double* R = vclip + (size_t)c * S;
        a0 = R[bc]; a1 = R[nb + bc];
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1)::"memory");   // on this path only
      }
      lds_wait_opaque(ws, ww, a0, a1);
      double best = -INFINITY; int bi = 1 << 30;
      if (ok) {
        const double cv = a0 + (tv ? ws : ww), cu = a1 + (tv ? ww : ws);
        best = cv; bi = b;                                           // voiced sources come first (lower index)
        if (cu > best) { best = cu; bi = nb + b; }
      }
      {                                                              // (max, lowest index) over the wave
        const double m = wave_max_dpp(best);
        const unsigned long long at = __ballot(best == m);
        if (__popcll(at) == 1) bi = __builtin_amdgcn_readlane(bi, (int)__builtin_ctzll(at));
        else bi = wave_min_dpp(best == m ? bi : (1 << 30));
        best = m;
      }
      const int gb = g_arg >= nb ? g_arg - nb : g_arg;
      if (gb < blo || gb > bhi) {                                    // the best out-of-band source
        const double cand = g_value + c0;
        if (cand > best || (cand == best && g_arg < bi)) { best = cand; bi = g_arg; }
      }
      cur = __builtin_amdgcn_readfirstlane(bi);
      // the column D - 1 steps on, around the bin just decided
      {
        const int jn = cur < nb ? cur : cur - nb;
        const int cn = c - (D - 1);
        prefetch(cn, jn);
        if (cn >= 0) cens = (lane == (cn % D) ? (jn) : cens);
      }
      // state of frame t - 1: into the pending register, flushed 64 at a time
      const int pos = t_hi - (t - 1);
      if (pos == 64) {
        const int q = pend;                                          // lane i: frame t_hi - i
        sts[t_hi - lane] = (uint16_t)q;
        tally(q);
        t_hi -= 64;
        pend = (lane == (0) ? (cur) : pend);
      } else {
        pend = (lane == (pos) ? (cur) : pend);
      }
    }
  }
  {                                                                  // frames t_hi .. 0 are still pending
    const int q = pend;
    if (lane <= t_hi) { sts[t_hi - lane] = (uint16_t)q; tally(q); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __threadfence_block();

  // ---- statistics over voiced frames (feature_extractor.py:97-107)
  if (out_f0) {
    for (int t = lane; t < T; t += 64) { const int q = sts[t]; out_f0[f0_offsets[clip] + t] = q < nb ? tb.freqs[q] : (double)NAN; }
    for (int t = T + lane; t < cd.tmax; t += 64) out_f0[f0_offsets[clip] + t] = (double)NAN;      // trimmed away
  }
  const double cntv = wave_sum_d(c1);
  const double sum = wave_sum_d(s1);
  if (cntv > 0.0) {
    const double mean = sum / cntv;
    double s2 = 0.0;
    for (int t = lane; t < T; t += 64) {
      const int q = sts[t];
      if (q < nb) { const double d = tb.freqs[q] - mean; s2 += d * d; }
    }
    const double var = wave_sum_d(s2) / cntv;
    if (lane == 0) {
      const double missing = ((double)T - cntv) / (double)T;
      st[0] = mean; st[1] = sqrt(var); st[2] = missing; st[3] = 1.0 - missing;
    }
  } else if (lane == 0) {
    st[0] = 0.0; st[1] = 0.0; st[2] = 1.0; st[3] = 0.0;
  }
}

// ---------------------------------------------------------------------------------------------
// k_zcr: librosa.feature.zero_crossing_rate of the preprocessed signal, one wave per frame.  Centre padding is
// 'edge' (the clip's first / last sample), samples with |y| <= 1e-10 count as +0, a crossing is a change of sign
// bit between neighbours, the first sample of a frame never is one; rate = crossings / frame_length (float64).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_zcr(const float* __restrict__ ysig, const ClipDesc* __restrict__ clips,
                                             const ClipInfo* __restrict__ info, int n_fft, int hop,
                                             double* __restrict__ out, const int64_t* __restrict__ out_offsets) {
  const int clip = blockIdx.y;
  const ClipInfo ci = info[clip];
  if (ci.status == AFX_CLIP_NONFINITE) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.x * 4 + wave;
  if (t >= ci.T) return;
  const ClipDesc cd = clips[clip];
  const int64_t np = ci.end - ci.start;
  const float* y = ysig + cd.off;
  const int64_t g0 = (int64_t)t * hop - n_fft / 2;
  auto neg = [&](int64_t g) -> bool {                    // sign bit of padded sample g after the 1e-10 clip
    if (np <= 0) return false;
    const int64_t c = g < 0 ? 0 : (g >= np ? np - 1 : g);
    const float v = y[c];
    return fabsf(v) <= 1e-10f ? false : (v < 0.f);       // NaN cannot occur (non-finite clips are flagged)
  };
  int cnt = 0;
  for (int n = lane; n < n_fft; n += 64) {
    if (n > 0) cnt += (neg(g0 + n) != neg(g0 + n - 1)) ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o);
  if (lane == 0) out[out_offsets[clip] + t] = (double)cnt / (double)n_fft;
}

hipError_t launch_zcr(hipStream_t s, const float* ysig, const ClipDesc* clips, const ClipInfo* info, int n_fft, int hop,
                      double* out, const int64_t* out_offsets, int n_clips, int max_tmax) {
  dim3 grid((max_tmax + 3) / 4, n_clips);
  hipLaunchKernelGGL(k_zcr, grid, dim3(256), 0, s, ysig, clips, info, n_fft, hop, out, out_offsets);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
template <typename K>
static hipError_t allow_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

hipError_t launch_f0_energy(hipStream_t s, const float* ysig, const ClipDesc* clips, const ClipInfo* info,
                            float* energy, int n_clips, int max_tmax, const F0Params& fp) {
  hipError_t e;
  if (fp.W % fp.hop == 0 && fp.n_tau <= fp.W) {
    // lanes per wave: 8, or 4 for the long hops, if that keeps the workgroup's samples within a quarter of the CU's LDS
    // (four workgroups per CU)
#define AFX_E2_LAUNCH(LPW)                                                                                               \
    do {                                                                                                                 \
      const size_t lds2 = f0_energy2_lds_bytes(fp, LPW);                                                                 \
      if (lds2 <= 40 * 1024) {                                                                                           \
        dim3 grid2((max_tmax + kEnergyWaves * LPW - 1) / (kEnergyWaves * LPW), n_clips);                                 \
        hipLaunchKernelGGL(k_f0_energy2<LPW>, grid2, dim3(64 * kEnergyWaves), lds2, s, ysig, clips, info, energy, fp);   \
        return hipGetLastError();                                                                                        \
      }                                                                                                                  \
    } while (0)
    AFX_E2_LAUNCH(8);
    AFX_E2_LAUNCH(4);
#undef AFX_E2_LAUNCH
  }
  const size_t lds = f0_energy_lds_bytes(fp);
  if ((e = allow_lds(k_f0_energy, lds)) != hipSuccess) return e;
  dim3 grid((max_tmax + fp.epb - 1) / fp.epb, n_clips);
  hipLaunchKernelGGL(k_f0_energy, grid, dim3(64 * kEnergyWaves), lds, s, ysig, clips, info, energy, fp);
  return hipGetLastError();
}

hipError_t launch_f0_yin(hipStream_t s, const float* ysig, const ClipDesc* clips, const ClipInfo* info,
                         const float* energy, const F0Tables& tb, const F0Params& fp,
                         int32_t* cand_cnt, double* cand_vp, int16_t* cand_bin, double* cand_prob,
                         double* cand_lp, double* cand_lu, int n_clips, int max_tmax) {
  const size_t lds = f0_yin_lds_bytes(fp);
  const int fpb = f0_yin_frames_per_block(fp);
  dim3 grid((max_tmax + fpb - 1) / fpb, n_clips);
  const int need = fp.R > fp.slots ? fp.R : fp.slots;
#define AFX_YIN_LAUNCH_FR(N, F, SH)                                                                             \
  do {                                                                                                         \
    hipError_t e2 = allow_lds(k_f0_yin<N, N, F, SH>, lds);                                                     \
    if (e2 != hipSuccess) return e2;                                                                           \
    hipLaunchKernelGGL((k_f0_yin<N, N, F, SH>), grid, dim3(256), lds, s, ysig, clips, info, energy, tb, fp, cand_cnt, \
                       cand_vp, cand_bin, cand_prob, cand_lp, cand_lu);                                        \
  } while (0)
#define AFX_YIN_LAUNCH_F(N, F) AFX_YIN_LAUNCH_FR(N, F, 0)
  if (yin_shape_is<1>(fp) && fpb == 8) {
    const size_t lds_ref = yin_lds_i(fp.hop, fp.n_fft, fp.n_tau_pad, fp.slots, fp.cap, 8, true, true).total;
    hipLaunchKernelGGL((k_f0_yin<6, 6, 8, 1>), grid, dim3(256), lds_ref, s, ysig, clips, info, energy, tb, fp, cand_cnt,
                       cand_vp, cand_bin, cand_prob, cand_lp, cand_lu);
    return hipGetLastError();
  }
  if (yin_shape_is<2>(fp) && fpb == 16) { AFX_YIN_LAUNCH_FR(4, 16, 2); return hipGetLastError(); }
  if (yin_shape_is<3>(fp) && fpb == 16) { AFX_YIN_LAUNCH_FR(11, 16, 3); return hipGetLastError(); }
#define AFX_YIN_LAUNCH(N) do { if (fpb == 8) AFX_YIN_LAUNCH_F(N, 8); else AFX_YIN_LAUNCH_F(N, 16); } while (0)
  if (need <= 4) AFX_YIN_LAUNCH(4);
  else if (need <= 6) AFX_YIN_LAUNCH(6);
  else if (need <= 8) AFX_YIN_LAUNCH_F(8, 16);
  else if (need <= 11) AFX_YIN_LAUNCH_F(11, 16);
  else AFX_YIN_LAUNCH_F(16, 16);
#undef AFX_YIN_LAUNCH
#undef AFX_YIN_LAUNCH_F
#undef AFX_YIN_LAUNCH_FR
  return hipGetLastError();
}

static hipError_t launch_f0_backtrack(hipStream_t s, const ClipDesc* clips, const ClipInfo* info, const F0Tables& tb,
                                      const F0Params& fp, const double* vrows, const VitBest* vbest, uint16_t* states,
                                      double* out_stats, double* out_f0, const int64_t* f0_offsets, int n_clips) {
  const size_t lds = f0_backtrack_lds_bytes(fp);
  const dim3 grid((n_clips + kBtWaves - 1) / kBtWaves), block(64 * kBtWaves);
  hipError_t e;
  if (f0_bt_depth(fp.band) == 6) {
    if ((e = allow_lds(k_f0_backtrack<6>, lds)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_f0_backtrack<6>, grid, block, lds, s, clips, info, tb, fp, vrows, vbest, states, out_stats, out_f0,
                       f0_offsets, n_clips);
  } else {
    if ((e = allow_lds(k_f0_backtrack<5>, lds)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_f0_backtrack<5>, grid, block, lds, s, clips, info, tb, fp, vrows, vbest, states, out_stats, out_f0,
                       f0_offsets, n_clips);
  }
  return hipGetLastError();
}

hipError_t launch_f0_viterbi(hipStream_t s, const ClipDesc* clips, const ClipInfo* info, const F0Tables& tb,
                             const F0Params& fp, const int32_t* cand_cnt,
                             const int16_t* cand_bin, const double* cand_lp, const double* cand_lu,
                             double* vrows, VitBest* vbest,
                             uint16_t* states, double* out_stats, double* out_f0, const int64_t* f0_offsets,
                             int n_clips) {
  const size_t lds = f0_viterbi_lds_bytes(fp);
  hipError_t e;
#define AFX_VIT_LAUNCH(MODE, NBT, BANDT)                                                                              \
  do {                                                                                                               \
    if ((e = allow_lds(k_f0_viterbi<MODE, NBT, BANDT>, lds)) != hipSuccess) return e;                                \
    hipLaunchKernelGGL((k_f0_viterbi<MODE, NBT, BANDT>), dim3(n_clips), dim3(kVitThreads), lds, s, clips, info, tb, fp, \
                       cand_cnt, cand_bin, cand_lp, cand_lu, vrows, vbest);                                          \
  } while (0)
  const bool ref_shape = fp.n_bins == 601 && fp.band == 25;       // fmin / fmax of the reference at hop / sr = 256 / 22050, 512 / 44100
#if AFX_F0_DEBUG_BUILD
  if (fp.debug & 16) AFX_VIT_LAUNCH(2, 0, 0);
  else if (fp.debug) { if (ref_shape) AFX_VIT_LAUNCH(1, 601, 25); else AFX_VIT_LAUNCH(1, 0, 0); }
  else
#endif
  if (ref_shape) AFX_VIT_LAUNCH(0, 601, 25);
  else if (fp.n_bins == 601 && fp.band == 15) AFX_VIT_LAUNCH(0, 601, 15);       // the same pitch range at hop / sr = 128 / 16000
  else AFX_VIT_LAUNCH(0, 0, 0);
#undef AFX_VIT_LAUNCH
  if ((e = hipGetLastError()) != hipSuccess) return e;
  return launch_f0_backtrack(s, clips, info, tb, fp, vrows, vbest, states, out_stats, out_f0, f0_offsets, n_clips);
}

}  // namespace afx
