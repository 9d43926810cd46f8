// k_tail: everything behind the frame kernels for one clip in one workgroup -- power_to_db's clip-global clamp, the ortho
// DCT-II, and the per-clip statistics -- with the MFCC rows never leaving the chip.
// (reference call sites: audio_feature_extraction_toolkit/core/feature_extractor.py:127-134 librosa.feature.mfcc ->
//  power_to_db(top_db=80) + scipy.fft.dct; :137-138 librosa.feature.delta; :141-150 mean / std / delta means;
//  :164-178 rms statistics.  Shape that motivates it: 04_feature_extraction_experiment/feature_extraction.py:35-41,
//  n_mfcc = 40 at 16 kHz / 512 / 128.)
//
// Why.  The two-kernel tail (k_dct16* then k_stats) writes the MFCC rows [clip][k][Tpad] to HBM and reads them back.  For the
// speech configuration (40 coefficients, hop 128) that is 216 MB out + 213 MB in per 1000 clips -- more than the samples
// themselves (640 MB) -- and a second launch.  Only 4K + 3 numbers per clip are wanted: sum and sum of squares per
// coefficient (float64 accumulators, as k_stats), and the nine frames at either end of every row (the means of the two
// Savitzky-Golay delta rows telescope to them, see k_stats).  So: a workgroup per clip walks the clip's log-mel spill
// once in 16-frame tiles, multiplies each tile by the DCT images (exact-f32 MFMA, images in LDS: k_dct16l's loop), and
// every lane adds the four coefficients it receives of its frame to its own float64 sums; the tiles that hold the
// first / last nine frames also drop their values into a 18-column LDS table.  One reduction at the end.  The spill is
// read once at HBM rate; nothing else moves.  Per-frame export (out_frames) keeps the two-kernel path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "afx_device.h"
#include "afx_devenv.h"

namespace afx {

typedef float tl_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float tl_ord2f(uint32_t o) {
  const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}
// sum over the 16 lanes of a row (all lanes of the row end with the total).  The two dwords of a double go through DPP moves
// (__shfl_xor lowers to ds_bpermute round trips: 24 sums x 4 steps of those were 10 us per clip and wave)
#define TL_DPP(x, ctrl) __builtin_amdgcn_update_dpp(0, (x), (ctrl), 0xf, 0xf, false)
__device__ __forceinline__ double tl_dpp_d(double v, const int ctrl_sel) {
  const long long b = __double_as_longlong(v);
  int lo = (int)b, hi = (int)(b >> 32);
  switch (ctrl_sel) {
    case 0: lo = TL_DPP(lo, 0xB1); hi = TL_DPP(hi, 0xB1); break;       // quad_perm [1,0,3,2]
    case 1: lo = TL_DPP(lo, 0x4E); hi = TL_DPP(hi, 0x4E); break;       // quad_perm [2,3,0,1]
    case 2: lo = TL_DPP(lo, 0x141); hi = TL_DPP(hi, 0x141); break;     // row_half_mirror
    default: lo = TL_DPP(lo, 0x140); hi = TL_DPP(hi, 0x140); break;    // row_mirror
  }
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double tl_row_sum(double v) {
  v += tl_dpp_d(v, 0); v += tl_dpp_d(v, 1); v += tl_dpp_d(v, 2); v += tl_dpp_d(v, 3);
  return v;
}
__device__ __forceinline__ double tl_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}


// SYM: the DCT-II rows are symmetric (even k) or antisymmetric (odd k) about the middle of the mel axis,
// d_k(M - 1 - m) = (-1)^k d_k(m), so the contraction folds to M / 2 terms over L_m + L_{M-1-m} (even rows) or L_m - L_{M-1-m}
// (odd rows): NG = 2 GH row groups -- GH of even coefficients, then GH of odd ones -- at half the k-steps each.  For 17..32 /
// 33..48 coefficients that is 32 / 64 MFMAs per 16-frame tile instead of 64 / 96 (the kernel is bound by the matrix pipe
// there, not by the spill's bytes).  Lane (f, q) then also fetches the mirrored filters 16 (S - 1 - s) + 4 (3 - q) + {3..0}.
// Plain (SYM = false): NG = ceil(K / 16) groups of consecutive coefficients over all M terms.
//
// RING > 0 (n_mels = 128 only): the tiles do not pass through registers on their way in.  A tile is 16 consecutive 512-byte
// rows of the spill = 8 KB of contiguous memory; each wave keeps a ring of RING such tiles in LDS, filled by LDS-DMA
// (global_load_lds_dwordx4: 8 wave-instructions of 1 KB per tile, no destination registers) RING - 1 tiles ahead, and reads
// its MFMA operands from the ring with one ds_read_b128 per 16-filter segment.  Whole 128-byte lines are fetched (the
// register path's loads are 16 rows x 64 bytes per instruction, half lines), and the bytes in flight no longer compete with
// the float64 sums for registers -- with 33..48 coefficients the register path keeps ~4 KB per wave in flight at two waves
// per SIMD and reads the spill at 2.9 TB/s.  The LDS image is lane-linear (the DMA's rule), so the bank swizzle is applied on
// the SOURCE side: 16-byte chunk j of row f is stored at chunk position j ^ f of row f, and lane (f, q) reads chunk 4 s + q
// of its row conflict-free.  RING = 0: the register path above (any n_mels that is a multiple of 16).
template <int NG, bool SYM, int WAVES, int RING>
__global__ __launch_bounds__(WAVES * 64, ((NG == 1 && WAVES == 4 && RING == 0) ? 4 : 1)) void k_tail(const ClipDesc* __restrict__ clips,
                                                          const ClipInfo* __restrict__ info,
                                                          const float* __restrict__ dctP, KParams kp,
                                                          const float* __restrict__ logmel,
                                                          const float* __restrict__ rms_rows,
                                                          float* __restrict__ stats,
                                                          ClipInfo* __restrict__ info_out, int spec, int n_clips) {
  extern __shared__ float tl_smem[];
  constexpr int NCG = NG;                                  // rows of coefficients handled: 16 NG
  constexpr int GH = SYM ? NG / 2 : NG;
  const int M = kp.n_mels, K = kp.n_mfcc, S = M >> 4;     // S <= 8
  const int KS = SYM ? S >> 1 : S;                         // 16-filter segments the contraction runs over
  float* const dct_tab = tl_smem;                          // [(g KS + s) 4 + c][64 lanes]
  float* const edge = dct_tab + NCG * KS * 4 * 64;         // [k < 16 NG][18]: frames 0..8, T-9..T-1 of row k
  double* const part = reinterpret_cast<double*>(edge + NCG * 16 * 18 + (((NCG * 16 * 18) & 1) ? 1 : 0));   // [wave][k][2]
  double* const red = part + WAVES * NCG * 16 * 2;        // block reductions of the RMS row: [wave][4]
  float* const ring = reinterpret_cast<float*>(red + WAVES * 4);      // RING > 0: [wave][RING][16 rows x 128 floats], 16-byte aligned
  constexpr int kTailWaves = WAVES;
  // coefficient of row i of group g: consecutive, or (SYM) even coefficients in the first GH groups, odd ones in the rest
  auto coef = [&](int g, int i) -> int { return SYM ? 2 * (16 * (g % GH) + i) + g / GH : 16 * g + i; };
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // a workgroup walks clips blockIdx.x, + gridDim.x, ...: the DCT images are copied to LDS once per workgroup, not per clip
  for (int i = tid * 4; i < NCG * KS * 4 * 64; i += kTailWaves * 64 * 4)
    *reinterpret_cast<float4*>(dct_tab + i) = *reinterpret_cast<const float4*>(dctP + i);
  for (int clip = blockIdx.x; clip < n_clips; clip += gridDim.x) {
  __syncthreads();                                         // the previous clip's readers of edge / part / red are done (and the table is there)
  const ClipInfo ci = info[clip];
  const ClipDesc cd = clips[clip];
  if (info_out && tid == 0) info_out[clip] = ci;           // the caller's copy (host memory the device can write)
  float* const st = stats + (int64_t)clip * (4 * K + 3);
  // a clip with fewer than 9 frames fails the MFCC rows (librosa.feature.delta raises) but still has an RMS row:
  // extract_energy (F:153-179) only calls librosa.feature.rms
  const bool energy_only = ci.status == AFX_CLIP_TOO_SHORT && cd.len >= 2 && ci.T >= 1;
  if (ci.status != AFX_CLIP_OK) {                          // uniform per workgroup
    for (int i = tid; i < 4 * K + (energy_only ? 0 : 3); i += kTailWaves * 64) st[i] = 0.f;
    if (!energy_only) continue;
  }
  const int T = ci.T;
  const double invT = 1.0 / (double)T;

  if (ci.status == AFX_CLIP_OK) {
    const float theta = tl_ord2f(ci.lmax_ord) - kp.top_db;
    const int f = lane & 15, q = lane >> 4;
    const int ntiles = (T + 15) >> 4;
    // spec: the spill holds absolute frames (the frame kernel ran before the trim decision); trimmed frame t is frame start / hop + t
    const int g0 = spec ? (int)(ci.start / kp.hop) : 0;
    // frame-major spill [frame][mel]: lane (f, q) fetches filters 16 s + 4 q + {0..3} of frame f with one 16-byte load
    // (a wave-load is a run of whole 64-byte pieces of 16 rows); rows past the clip's last frame are read but never used
    // (the spill is allocated 16 frames beyond the batch's last row)
    // One 16-filter segment of a tile: x[s] = filters 16 s + 4 q + {0..3} of frame f; SYM also x[4 + s] = their mirrors
    // 16 (S - 1 - s) + 4 (3 - q) + {0..3}.  A segment's registers are refilled with the NEXT tile's values as soon as its
    // MFMAs have read them -- the prefetch buffer is the operand buffer (a second one would cost 32 registers and, for 33..48
    // coefficients, the second wave per SIMD).
    auto load_seg = [&](int tile, int s, float4 (&x)[8]) {
      const bool on = tile < ntiles;
      const float* row = logmel + (cd.frame_base + g0 + tile * 16 + f) * (int64_t)M;
      x[s] = on ? *reinterpret_cast<const float4*>(row + q * 4 + s * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (SYM) x[4 + s] = on ? *reinterpret_cast<const float4*>(row + (3 - q) * 4 + (S - 1 - s) * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    // ---- LDS-DMA ring (RING > 0)
    float* const myring = ring + wave * (RING > 0 ? RING : 1) * 2048;
    auto dma_tile = [&](int tile, int slot) {              // 8 KB: position p = 64 i + lane of the image <- chunk (p & 31) ^ f of row f = p >> 5
      const float* tb = logmel + (cd.frame_base + g0 + tile * 16) * (int64_t)M;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int fr = 2 * i + (lane >> 5), j = (lane & 31) ^ fr;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tb + fr * 128 + j * 4),
                                         (__attribute__((address_space(3))) void*)(myring + slot * 2048 + i * 256), 16, 0, 0);
      }
    };
    auto ring_seg = [&](int slot, int chunk) -> float4 {   // chunk `chunk` (four filters) of this lane's frame row
      return *reinterpret_cast<const float4*>(myring + slot * 2048 + (f * 32 + (chunk ^ f)) * 4);
    };
    double sm[NCG][4], sq[NCG][4];
#pragma unroll
    for (int g = 0; g < NCG; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) { sm[g][r] = 0.0; sq[g][r] = 0.0; }
    float4 xc[8];
    int tile = wave;
    if constexpr (RING > 0) {
#pragma unroll
      for (int r = 0; r < RING; ++r) if (tile + r * kTailWaves < ntiles) dma_tile(tile + r * kTailWaves, r);
    } else {
#pragma unroll
    for (int s = 0; s < 8; ++s) if (s < KS) load_seg(tile, s, xc);
    }
    int slot = 0;
    for (; tile < ntiles; tile += kTailWaves) {
      if constexpr (RING > 0) {
        // this tile's eight DMAs have landed once at most those of the RING - 1 tiles behind it are outstanding
        if (tile + (RING - 1) * kTailWaves < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (RING - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      tl_f32x4 acc[NCG][2];
#pragma unroll
      for (int g = 0; g < NCG; ++g) { acc[g][0] = tl_f32x4{0.f, 0.f, 0.f, 0.f}; acc[g][1] = tl_f32x4{0.f, 0.f, 0.f, 0.f}; }
      if constexpr (SYM) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          if (s < KS) {
            if constexpr (RING > 0) { xc[s] = ring_seg(slot, 4 * s + q); xc[4 + s] = ring_seg(slot, 4 * (S - 1 - s) + (3 - q)); }
            // clamp first (power_to_db's top_db), then fold: filter m with its mirror M - 1 - m (component c with 3 - c)
            const float u0 = fmaxf(xc[s].x, theta), u1 = fmaxf(xc[s].y, theta), u2 = fmaxf(xc[s].z, theta), u3 = fmaxf(xc[s].w, theta);
            const float w0 = fmaxf(xc[4 + s].w, theta), w1 = fmaxf(xc[4 + s].z, theta), w2 = fmaxf(xc[4 + s].y, theta), w3 = fmaxf(xc[4 + s].x, theta);
            const float e0 = u0 + w0, e1 = u1 + w1, e2 = u2 + w2, e3 = u3 + w3;
            const float o0 = u0 - w0, o1 = u1 - w1, o2 = u2 - w2, o3 = u3 - w3;
#pragma unroll
            for (int g = 0; g < NCG; ++g) {
              const float* a = dct_tab + ((g * KS + s) * 4) * 64 + lane;
              const bool ev = g < GH;
              acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], ev ? e0 : o0, acc[g][0], 0, 0, 0);
              acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[64], ev ? e1 : o1, acc[g][1], 0, 0, 0);
              acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[128], ev ? e2 : o2, acc[g][0], 0, 0, 0);
              acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[192], ev ? e3 : o3, acc[g][1], 0, 0, 0);
            }
            if constexpr (RING == 0) load_seg(tile + kTailWaves, s, xc);
          }
        }
      } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (s < S) {
          if constexpr (RING > 0) xc[s] = ring_seg(slot, 4 * s + q);
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
          if constexpr (RING == 0) load_seg(tile + kTailWaves, s, xc);
        }
      }
      }
      if constexpr (RING > 0) {
        // every operand read of this slot has returned (its MFMAs were issued): refill it RING tiles on
        if (tile + RING * kTailWaves < ntiles) dma_tile(tile + RING * kTailWaves, slot);
        slot = slot + 1 == RING ? 0 : slot + 1;
      }
      const int t = tile * 16 + f;
      const bool live = t < T;
      const bool edge_tile = tile * 16 < 9 || tile * 16 + 16 > T - 9;      // wave-uniform: holds one of the 18 end frames
#pragma unroll
      for (int g = 0; g < NCG; ++g) {
        const tl_f32x4 r4 = acc[g][0] + acc[g][1];       // the same two-accumulator sum as k_dct16 / k_dct16l: identical rows
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double v = live ? (double)r4[r] : 0.0;
          sm[g][r] += v; sq[g][r] = fma(v, v, sq[g][r]);
        }
        if (edge_tile && live) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int k = coef(g, q * 4 + r);
            if (t < 9) edge[k * 18 + t] = r4[r];
            if (t >= T - 9) edge[k * 18 + 9 + (t - (T - 9))] = r4[r];
          }
        }
      }
    }
    // ---- per wave: totals of the 16 frame lanes of each coefficient row
#pragma unroll
    for (int g = 0; g < NCG; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double a = tl_row_sum(sm[g][r]), b = tl_row_sum(sq[g][r]);
        if (f == 0) {
          const int k = coef(g, q * 4 + r);
          part[(wave * NCG * 16 + k) * 2] = a;
          part[(wave * NCG * 16 + k) * 2 + 1] = b;
        }
      }
    __syncthreads();
    // ---- one thread per coefficient: mean, population std, and the means of the two delta rows from the row's ends
    if (tid < K) {
      const int row = tid;
      double s = 0.0, ss = 0.0;
#pragma unroll
      for (int w = 0; w < kTailWaves; ++w) { s += part[(w * NCG * 16 + row) * 2]; ss += part[(w * NCG * 16 + row) * 2 + 1]; }
      const double mean = s * invT;
      const float meanf = (float)mean;
      const double c = (double)meanf;
      // sum (x - c)^2 = ss - 2 c s + T c^2 about the float32 mean c, as numpy's std (k_stats)
      const double s2 = fmax(ss - 2.0 * c * s + (double)T * c * c, 0.0);
      double h[9], g9[9];                                   // h[i] = x[i], g9[i] = x[T - 9 + i]
#pragma unroll
      for (int i = 0; i < 9; ++i) { h[i] = (double)edge[row * 18 + i]; g9[i] = (double)edge[row * 18 + 9 + i]; }
      // Both savgol filters (width 9, polyorder = deriv = order, mode 'interp') are differences: their sums over the frames
      // telescope to the nine frames at either end; frames 0..3 / T-4..T-1 repeat frame 4 / T-5 (k_stats, same arithmetic)
      auto D1 = [](const double* cc) {
        return (4.0 * (cc[4] - cc[-4]) + 3.0 * (cc[3] - cc[-3]) + 2.0 * (cc[2] - cc[-2]) + (cc[1] - cc[-1])) * (1.0 / 60.0);
      };
      auto D2 = [](const double* cc) {
        return (28.0 * (cc[4] + cc[-4]) + 7.0 * (cc[3] + cc[-3]) - 8.0 * (cc[2] + cc[-2]) - 17.0 * (cc[1] + cc[-1]) - 20.0 * cc[0]) * (1.0 / 462.0);
      };
      const double w2[5] = {0.0, -17.0, -8.0, 7.0, 28.0};
      double sd1 = 0.0, sd2 = 0.0;
#pragma unroll
      for (int k = 1; k <= 4; ++k) {
        double head = 0.0, tail = 0.0, hl = 0.0, hr = 0.0, tl = 0.0, tr = 0.0;
#pragma unroll
        for (int i = 4 - k; i <= 3 + k; ++i) head += h[i];          // x[4 - k .. 3 + k]
#pragma unroll
        for (int i = 5 - k; i <= 4 + k; ++i) tail += g9[i];         // x[T - 4 - k .. T - 5 + k]
#pragma unroll
        for (int i = 4 - k; i <= 3; ++i) hl += h[i];
#pragma unroll
        for (int i = 4; i <= 3 + k; ++i) hr += h[i];
#pragma unroll
        for (int i = 5 - k; i <= 4; ++i) tl += g9[i];
#pragma unroll
        for (int i = 5; i <= 4 + k; ++i) tr += g9[i];
        sd1 += (double)k * (tail - head);
        sd2 += w2[k] * ((hl - hr) + (tr - tl));
      }
      sd1 = sd1 * (1.0 / 60.0) + 4.0 * ((double)(float)D1(h + 4) + (double)(float)D1(g9 + 4));
      sd2 = sd2 * (1.0 / 462.0) + 4.0 * ((double)(float)D2(h + 4) + (double)(float)D2(g9 + 4));
      st[row] = meanf;
      st[K + row] = (float)sqrt(s2 * invT);
      st[2 * K + row] = (float)(sd1 * invT);
      st[3 * K + row] = (float)(sd2 * invT);
    }
  }

  // ---- RMS row: mean, population std about the float32 mean, peak to peak (F:171-178)
  {
    const float* r = rms_rows + cd.frame_base;
    double s = 0.0;
    float mx = -INFINITY, mn = INFINITY;
    for (int t = tid; t < T; t += kTailWaves * 64) {
      const float v = r[t];
      s += (double)v; mx = fmaxf(mx, v); mn = fminf(mn, v);
    }
    s = tl_wave_sum(s);
    double dmx = (double)mx, dmn = (double)mn;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { dmx = fmax(dmx, __shfl_xor(dmx, o)); dmn = fmin(dmn, __shfl_xor(dmn, o)); }
    if (lane == 0) { red[wave * 4] = s; red[wave * 4 + 1] = dmx; red[wave * 4 + 2] = dmn; }
    __syncthreads();
    double tot = 0.0, gmx = -INFINITY, gmn = INFINITY;
#pragma unroll
    for (int w = 0; w < kTailWaves; ++w) { tot += red[w * 4]; gmx = fmax(gmx, red[w * 4 + 1]); gmn = fmin(gmn, red[w * 4 + 2]); }
    const float meanf = (float)(tot * invT);
    double s2 = 0.0;
    for (int t = tid; t < T; t += kTailWaves * 64) { const float d = r[t] - meanf; s2 += (double)d * (double)d; }
    s2 = tl_wave_sum(s2);
    if (lane == 0) red[wave * 4 + 3] = s2;
    __syncthreads();
    if (tid == 0) {
      double v2 = 0.0;
#pragma unroll
      for (int w = 0; w < kTailWaves; ++w) v2 += red[w * 4 + 3];
      st[4 * K] = meanf;
      st[4 * K + 1] = (float)sqrt(v2 * invT);
      st[4 * K + 2] = (float)gmx - (float)gmn;
    }
  }
  }
}

bool tail_eligible(const KParams& kp, const DevTables& tb) {
  return tb.dctP != nullptr && kp.n_mels % 16 == 0 && kp.n_mels <= 128 && kp.n_mfcc <= 48;
}

// the folded contraction pays when it needs fewer row groups per k-step pair: GH = ceil(ceil(K / 2) / 16) < ceil(K / 16)
int tail_sym_groups(const KParams& kp, const DevTables& tb) {
  if (!tb.dctS || kp.n_mels % 32 != 0) return 0;
  const int gh = ((kp.n_mfcc + 1) / 2 + 15) / 16, ncg = (kp.n_mfcc + 15) / 16;
  return gh < ncg ? gh : 0;
}

static size_t tail_lds_bytes(int ng, int ks, int waves, int ring) {
  const size_t tab = (size_t)ng * ks * 4 * 64, edge = (size_t)ng * 16 * 18 + (((ng * 16 * 18) & 1) ? 1 : 0);
  return (tab + edge) * sizeof(float) + ((size_t)waves * ng * 16 * 2 + waves * 4) * sizeof(double) +
         (size_t)waves * ring * 2048 * sizeof(float) + 16;
}

template <int NG, bool SYM, int WAVES, int RING>
static hipError_t launch_tail_t(hipStream_t s, const ClipDesc* clips, const ClipInfo* info, const float* table, const KParams& kp,
                                const float* logmel, const float* rms_rows, float* stats, ClipInfo* info_out, int n_clips,
                                int spec, int n_cu) {
  const size_t lds = tail_lds_bytes(NG, SYM ? kp.n_mels / 32 : kp.n_mels / 16, WAVES, RING);
  // as many workgroups as the chip holds at once (asked of the runtime once per instantiation and device -- the dynamic-LDS
  // attribute is per device too: batch_process drives every visible GPU from one process), each walking its share of the clips
  static int per_cu_dev[64] = {};
  int dev = 0;
  hipError_t e0 = hipGetDevice(&dev);
  if (e0 != hipSuccess) return e0;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (per_cu_dev[dev] == 0) {
    const void* fn = reinterpret_cast<const void*>(&k_tail<NG, SYM, WAVES, RING>);
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
    }
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, WAVES * 64, lds) != hipSuccess || nb < 1) nb = 1;
    per_cu_dev[dev] = std::min(nb, 8);
  }
  const int per_cu = per_cu_dev[dev];
  dim3 grid(std::max(1, std::min(n_clips, n_cu * per_cu))), block(WAVES * 64);
  hipLaunchKernelGGL((k_tail<NG, SYM, WAVES, RING>), grid, block, lds, s, clips, info, table, kp, logmel, rms_rows, stats, info_out, spec, n_clips);
  return hipGetLastError();
}

hipError_t launch_tail(hipStream_t s, const ClipDesc* clips, const ClipInfo* info, const DevTables& tb, const KParams& kp,
                       const float* logmel, const float* rms_rows, float* stats, ClipInfo* info_out, int n_clips, int spec,
                       int n_cu) {
  const int gh = tail_sym_groups(kp, tb);
  // mode: 0 the shipped choice; 1 registers; LDS-DMA ring of 2: 4 waves x 4 tiles, 3: 8 waves x 2 tiles, 4: 4 waves x 2 tiles with two
  // workgroups per CU (A/B: AFX_TAIL_MODE).  Shipped: 3, and 4 for up to 16 coefficients -- a clip is only 54 tiles, and with two
  // clips per CU the reductions and the RMS pass at the end of one run beside the tiles of the other (cfg 2 step 0.686 -> 0.679 ms;
  // with 40 coefficients the wider kernel loses 7 % that way, with 20 it makes no difference)
  int mode = dev_env().tail_mode;
  if (kp.n_mels != 128) mode = 1;                          // the ring's image is laid out for 512-byte rows
  else if (mode == 0) mode = (gh == 0 && kp.n_mfcc <= 16) ? 4 : 3;
#define AFX_TAIL_W(NG, SYM, TAB, W, R) launch_tail_t<NG, SYM, W, R>(s, clips, info, TAB, kp, logmel, rms_rows, stats, info_out, n_clips, spec, n_cu)
#define AFX_TAIL(NG, SYM, TAB) (mode == 2 ? AFX_TAIL_W(NG, SYM, TAB, 4, 4) : mode == 3 ? AFX_TAIL_W(NG, SYM, TAB, 8, 2) : mode == 4 ? AFX_TAIL_W(NG, SYM, TAB, 4, 2) : AFX_TAIL_W(NG, SYM, TAB, 4, 0))
  if (gh == 1) return AFX_TAIL(2, true, tb.dctS);
  if (gh == 2) return AFX_TAIL(4, true, tb.dctS);
  switch ((kp.n_mfcc + 15) / 16) {
    case 1: return AFX_TAIL(1, false, tb.dctP);
    case 2: return AFX_TAIL(2, false, tb.dctP);
    case 3: return AFX_TAIL(3, false, tb.dctP);
    default: return hipErrorInvalidValue;
  }
#undef AFX_TAIL
#undef AFX_TAIL_W
}

}  // namespace afx
