// Device-side building blocks shared by the wave-level frame kernels (afx_frames3*.hip): packed-f32 helpers with
// VOP3P source modifiers, the radix-4 / 8 / 16 butterflies, pre-emphasis as scipy.signal.lfilter rounds it.
// Include from .hip files only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace afx {

typedef float v2 __attribute__((ext_vector_type(2)));

// ---- packed-f32 helpers (VOP3P source modifiers do the swaps and sign flips) ------------------------------------
// a + (-i) b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ v2 add_mi(v2 a, v2 b) {
  v2 d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d;
}
// a + i b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ v2 add_pi(v2 a, v2 b) {
  v2 d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d;
}
// complex product v * w
__device__ __forceinline__ v2 cmul(v2 v, v2 w) {
  v2 t, d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(v), "v"(w));                       // (v.x w.x, v.y w.x)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(v), "v"(w), "v"(t));
  return d;
}
// two independent products, interleaved: the dependent multiply / fma of one product would otherwise sit back to
// back, and the compiler pads an inline-asm VALU dependence with an s_nop
__device__ __forceinline__ void cmul2(v2& a, v2 wa, v2& b, v2 wb) {
  v2 ta, tb, da, db;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(ta) : "v"(a), "v"(wa));
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(tb) : "v"(b), "v"(wb));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(da) : "v"(a), "v"(wa), "v"(ta));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(db) : "v"(b), "v"(wb), "v"(tb));
  a = da; b = db;
}
// the same with the (constant) twiddle in scalar registers
__device__ __forceinline__ v2 cmul_s(v2 v, v2 w) {
  v2 t, d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(v), "s"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(v), "s"(w), "v"(t));
  return d;
}
__device__ __forceinline__ void cmul2_s(v2& a, v2 wa, v2& b, v2 wb) {
  v2 ta, tb, da, db;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(ta) : "v"(a), "s"(wa));
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(tb) : "v"(b), "s"(wb));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(da) : "v"(a), "s"(wa), "v"(ta));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(db) : "v"(b), "s"(wb), "v"(tb));
  a = da; b = db;
}
// b + (-i) h e  and  b + i h e   (h = H.x)
__device__ __forceinline__ v2 fma_mi(v2 e, v2 H, v2 b) {
  v2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(d) : "v"(e), "s"(H), "v"(b)); return d;
}
__device__ __forceinline__ v2 fma_pi(v2 e, v2 H, v2 b) {
  v2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(e), "s"(H), "v"(b)); return d;
}
// (s.x^2 + d.y^2, s.y^2 + d.x^2)
__device__ __forceinline__ v2 sqsum(v2 s, v2 d) {
  v2 t, r;
  asm("v_pk_mul_f32 %0, %1, %1" : "=v"(t) : "v"(s));
  asm("v_pk_fma_f32 %0, %1, %1, %2 op_sel:[1,1,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(d), "v"(t));
  return r;
}

__device__ __forceinline__ void sqsum2(v2 s0, v2 d0, v2 s1, v2 d1, v2& r0, v2& r1) {
  v2 t0, t1;
  asm("v_pk_mul_f32 %0, %1, %1" : "=v"(t0) : "v"(s0));
  asm("v_pk_mul_f32 %0, %1, %1" : "=v"(t1) : "v"(s1));
  asm("v_pk_fma_f32 %0, %1, %1, %2 op_sel:[1,1,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(d0), "v"(t0));
  asm("v_pk_fma_f32 %0, %1, %1, %2 op_sel:[1,1,0] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(d1), "v"(t1));
}
// fmaxf without the canonicalising v_max x, x pair hipcc puts in front of it (the operands here are never signalling NaNs)
__device__ __forceinline__ float f3_max(float a, float b) {
  float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
}

// LDS read that the load/store optimizer leaves alone: merged into ds_read2_b64 two 8-byte reads take 8 LDS
// cycles instead of 2 + 2 (MI355X_MICROARCH.md, LDS table; SQ_LDS_IDX_ACTIVE confirmed it on this kernel)
typedef const volatile v2 __attribute__((address_space(3))) * f3_lds_cv2;
__device__ __forceinline__ v2 ldv(const v2* p) { return *(f3_lds_cv2)(p); }
// LDS store the load/store optimizer leaves alone (two ds_write_b64 take 2 x 6 LDS cycles, the merged ds_write2_b64 13)
typedef volatile v2 __attribute__((address_space(3))) * f3_lds_v2;
__device__ __forceinline__ void stv(v2* p, v2 v) { *(f3_lds_v2)(p) = v; }
// the 16-byte form (one ds_read_b128: table entries that are always wanted together)
typedef const volatile float4 __attribute__((address_space(3))) * f3_lds_cv4;
__device__ __forceinline__ float4 ldv4(const float4* p) {
  const float __attribute__((ext_vector_type(4))) v = *(const volatile float __attribute__((ext_vector_type(4))) __attribute__((address_space(3)))*)(p);
  return float4{v.x, v.y, v.z, v.w};
}

// mel taps: an (A, B) pair of one bin times a weight that sits in the low / high half of a register pair (two of the four
// weights of a 16-byte read).  The half is an op_sel modifier; hipcc's own code copies the high half into an even register.
__device__ __forceinline__ v2 f3_mel_mul_lo(v2 q, v2 c) {
  v2 d; asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(q), "v"(c)); return d;                          // q * c.x
}
__device__ __forceinline__ v2 f3_mel_mul_hi(v2 q, v2 c) {
  v2 d; asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(q), "v"(c)); return d;             // q * c.y
}
__device__ __forceinline__ v2 f3_mel_fma_lo(v2 q, v2 c, v2 a) {
  v2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(q), "v"(c), "v"(a)); return d;            // q * c.x + a
}
__device__ __forceinline__ v2 f3_mel_fma_hi(v2 q, v2 c, v2 a) {
  v2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(q), "v"(c), "v"(a)); return d;   // q * c.y + a
}

__device__ __forceinline__ void f3_dft4(v2& x0, v2& x1, v2& x2, v2& x3) {
  const v2 a = x0 + x2, b = x0 - x2, c = x1 + x3, e = x1 - x3;
  x0 = a + c; x2 = a - c; x1 = add_mi(b, e); x3 = add_pi(b, e);
}

// radix-8 butterfly, natural order in and out (inputs already twiddled)
__device__ __forceinline__ void f3_dft8(v2* x, const v2 H) {
  v2 e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6];
  v2 o0 = x[1], o1 = x[3], o2 = x[5], o3 = x[7];
  f3_dft4(e0, e1, e2, e3);
  f3_dft4(o0, o1, o2, o3);
  const v2 q1 = add_mi(o1, o1);          // o1 * W8^1 = h q1
  const v2 q3 = add_pi(o3, o3);          // o3 * W8^3 = -h q3
  x[0] = e0 + o0; x[4] = e0 - o0;
  x[1] = q1 * H + e1; x[5] = e1 - q1 * H;
  x[2] = add_mi(e2, o2); x[6] = add_pi(e2, o2);
  x[3] = e3 - q3 * H; x[7] = q3 * H + e3;
}

// radix-16 butterfly as 4 x 4, W16 twiddles folded into the second layer's adds where they are h (1 -+ i) or -i
__device__ __forceinline__ void f3_dft16(v2* x, const v2 H, const v2 W1, const v2 W3) {
  v2 t0[4], t1[4], t2[4], t3[4];
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) {
    v2 a = x[n2], b = x[n2 + 4], c = x[n2 + 8], d = x[n2 + 12];
    f3_dft4(a, b, c, d);
    t0[n2] = a; t1[n2] = b; t2[n2] = c; t3[n2] = d;
  }
  {   // k1 = 0
    f3_dft4(t0[0], t0[1], t0[2], t0[3]);
    x[0] = t0[0]; x[4] = t0[1]; x[8] = t0[2]; x[12] = t0[3];
  }
  {   // k1 = 1: twiddles W16^1, W16^2 = h (1 - i), W16^3
    v2 p1 = t1[1], p3 = t1[3];
    cmul2_s(p1, W1, p3, W3);
    const v2 q = add_mi(t1[2], t1[2]);
    const v2 a = q * H + t1[0], b = t1[0] - q * H, c = p1 + p3, e = p1 - p3;
    x[1] = a + c; x[9] = a - c; x[5] = add_mi(b, e); x[13] = add_pi(b, e);
  }
  {   // k1 = 2: twiddles W16^2, W16^4 = -i, W16^6 = -h (1 + i)
    const v2 q1 = add_mi(t2[1], t2[1]), q3 = add_pi(t2[3], t2[3]);
    const v2 a = add_mi(t2[0], t2[2]), b = add_pi(t2[0], t2[2]);
    const v2 c = q1 - q3, e = q1 + q3;                       // both still to be scaled by h
    x[2] = c * H + a; x[10] = a - c * H; x[6] = fma_mi(e, H, b); x[14] = fma_pi(e, H, b);
  }
  {   // k1 = 3: twiddles W16^3, W16^6, W16^9 = -W16^1
    v2 p1 = t3[1], m3 = t3[3];
    cmul2_s(p1, W3, m3, W1);
    const v2 q = add_pi(t3[2], t3[2]);
    const v2 a = t3[0] - q * H, b = q * H + t3[0], c = p1 - m3, e = p1 + m3;
    x[3] = a + c; x[11] = a - c; x[7] = add_mi(b, e); x[15] = add_pi(b, e);
  }
}

__device__ __forceinline__ float f3_pre1(float y, float prev, float b1) {     // as scipy.signal.lfilter rounds it
#pragma clang fp contract(off)
  const float p = b1 * prev;
  return y + p;
}
__device__ __forceinline__ float f3_pre0(float y0, float y1) {                // librosa's zi = 2 y0 - y1
#pragma clang fp contract(off)
  const float t = 2.0f * y0;
  const float zi = t - y1;
  return zi + y0;
}
__device__ __forceinline__ uint32_t f3_ord(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// a wave-uniform value the optimizer may not trace back to its definition (no strength reduction across loop trips)
__device__ __forceinline__ int f3_opaque(int x) { asm volatile("" : "+s"(x)); return x; }
// timing-only ablation switches (libafx built with -DAFX_F3_DEBUG, AFX_DEBUG_SKIP bits << 8 in kp.flags); results invalid
#ifdef AFX_F3_DEBUG
#define F3_SKIP(bit) ((kp.flags & (bit)) != 0)
#else
#define F3_SKIP(bit) false
#endif
// Which blocks a wave works on.  With a work counter (a zeroed device int), about half of the block list is dealt out up
// front, C0 consecutive blocks per wave, and the rest is taken by ticket in runs that shrink (C0 / 2, C0 / 3, then single
// blocks): the four waves of a SIMD do not advance at one speed (the issue arbiter favours the oldest; 615 .. 900 us were
// measured for equal shares of k_frames3), so equal shares leave the SIMDs under-occupied for the last third of a
// launch, while one ticket per block makes every wave queue on one L2 atomic (79 000 of them slowed the 512 / 128
// kernel by 60 %).  Without a counter (the short list launch): blocks wg, wg + waves, ...
struct F3Runs {
  int b_lo, b_hi, stride;
};
// the ticket geometry is recomputed from (nblocks, total_waves) where a ticket is drawn -- a few scalar instructions once per
// run -- instead of living in four more scalar registers across the pair loop (the kernels sit at the scalar file's limit;
// a spilled scalar register costs a vector register)
__device__ __forceinline__ F3Runs f3_runs_init(const int* work_ctr, int nblocks, int total_waves, int wg, bool contiguous) {
  F3Runs r;
  r.stride = 1;
  if (work_ctr) {
    const int C0 = nblocks / (2 * total_waves);
    r.b_lo = wg * C0; r.b_hi = r.b_lo + C0;
  } else if (contiguous) {
    r.b_lo = (int)((long long)wg * nblocks / total_waves); r.b_hi = (int)((long long)(wg + 1) * nblocks / total_waves);
  } else {
    r.b_lo = wg; r.b_hi = nblocks; r.stride = total_waves;
  }
  return r;
}
// the wave's next run; false when the list is exhausted (wave-uniform)
__device__ __forceinline__ bool f3_runs_next(F3Runs& r, int* work_ctr, int nblocks, int total_waves, int lane) {
  if (!work_ctr) return false;
  int t = 0;
  if (lane == 0) t = atomicAdd(work_ctr, 1);
  t = __builtin_amdgcn_readfirstlane(t);
  const int C0 = nblocks / (2 * total_waves), dyn0 = total_waves * C0;
  const int cA = C0 / 2 > 1 ? C0 / 2 : 1, cB = C0 / 3 > 1 ? C0 / 3 : 1;
  int c = 1;
  if (t < total_waves) { r.b_lo = dyn0 + cA * t; c = cA; }
  else if (t < 2 * total_waves) { r.b_lo = dyn0 + cA * total_waves + cB * (t - total_waves); c = cB; }
  else r.b_lo = dyn0 + (cA + cB) * total_waves + (t - 2 * total_waves);
  if (r.b_lo >= nblocks) return false;
  r.b_hi = r.b_lo + c < nblocks ? r.b_lo + c : nblocks;
  return true;
}

#define F3_DPP(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xf, 0xf, false))

// v[lane] + v[lane ^ 16] and v[lane] + v[lane ^ 32] by gfx950's row / half swaps (one swap + one add each)
__device__ __forceinline__ float f3_add_xor16(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float f3_add_xor32(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// d = mask[lane] ? b : a with a wave-uniform 64-bit lane mask (one v_cndmask; the compiler's own code for a lane-indexed
// bit test is a 64-bit shift, an and and a compare)
__device__ __forceinline__ float f3_sel(float a, float b, unsigned long long mask) {
  float d; asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(mask)); return d;
}
// Two wave-wide sums at once: returns a register whose odd lanes all hold sum(x) and whose even lanes all hold sum(y)
// (x, y: one addend per lane).  Pairs first, then one register carries both parities through the remaining steps.
__device__ __forceinline__ float f3_sum2(float x, float y) {
  x += F3_DPP(x, 0xB1); y += F3_DPP(y, 0xB1);                   // lane ^ 1
  float m = f3_sel(y, x, 0xAAAAAAAAAAAAAAAAull);
  m += F3_DPP(m, 0x4E);                                          // lane ^ 2
  m += F3_DPP(m, 0x124); m += F3_DPP(m, 0x128);                  // row_ror 4, 8: the four quads of a row
  return f3_add_xor32(f3_add_xor16(m));
}


}  // namespace afx
