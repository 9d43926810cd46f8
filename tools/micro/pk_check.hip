#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2 add_mi(v2 a, v2 b) { v2 d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ v2 add_pi(v2 a, v2 b) { v2 d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ v2 cmul(v2 v, v2 w) {
  v2 t, d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(v), "v"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(v), "v"(w), "v"(t));
  return d;
}
__device__ __forceinline__ v2 cmulc(v2 v, v2 w) {   // v * conj(w)
  v2 t, d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(v), "v"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(d) : "v"(v), "v"(w), "v"(t));
  return d;
}
__device__ __forceinline__ v2 sqsum(v2 s, v2 d) {   // (s.x^2 + d.y^2, s.y^2 + d.x^2)
  v2 t, r;
  asm("v_pk_mul_f32 %0, %1, %1" : "=v"(t) : "v"(s));
  asm("v_pk_fma_f32 %0, %1, %1, %2 op_sel:[1,1,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(d), "v"(t));
  return r;
}
__global__ void k(const v2* in, v2* out) {
  v2 a = in[0], b = in[1];
  out[0] = add_mi(a, b); out[1] = add_pi(a, b); out[2] = cmul(a, b); out[3] = cmulc(a, b); out[4] = sqsum(a, b);
  v2 H = {0.70710678f, 0.70710678f};
  out[5] = a * H + b; out[6] = b - a * H; out[7] = a + b; out[8] = a - b;
}
int main() {
  v2 h[2] = {{1.f, 2.f}, {3.f, 5.f}}; v2 *di, *dout; v2 o[9];
  hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o)); hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 1>>>(di, dout); hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  const char* n[9] = {"a+(-i)b (6,-1)", "a+ib (-4,5)", "a*b (-7,11)", "a*conj(b) (13,1)", "sqsum (1+25, 4+9)=(26,13)", "a*H+b", "b-a*H", "a+b", "a-b"};
  for (int i = 0; i < 9; ++i) printf("%s -> (%g, %g)\n", n[i], o[i].x, o[i].y);
}
