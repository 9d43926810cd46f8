// microbenchmark: issue cost per wave-instruction of the float64 VALU ops the pYIN Viterbi walk uses (gfx950).
// One wave per SIMD (4 per workgroup, one workgroup per CU), 8 independent chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void k(unsigned long long* out, double* sink, int iters, double seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = seed * (lane + i + 1);
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)seed * (lane + i + 1);
  const double c = seed * 0.5;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[u]) : "v"(c));
      if (OP == 1) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[u]) : "v"(c));
      if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[u]) : "v"(c));
      if (OP == 3) asm volatile("v_cmp_gt_f64 vcc, %0, %1" :: "v"(a[u]), "v"(c) : "vcc");
      if (OP == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[u]) : "v"((float)c));
      if (OP == 5) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[u]) : "v"(c));
      if (OP == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[u]) : "v"((float)c) : "vcc");
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
  double s = 0; for (int i = 0; i < 8; ++i) s += a[i] + f[i];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, int waves) {
  const int iters = 2000, blocks = 256;
  unsigned long long* d; double* s;
  hipMalloc(&d, blocks * waves * 8); hipMalloc(&s, blocks * waves * 64 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64 * waves), 0, 0, d, s, iters, 1e-3);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64 * waves), 0, 0, d, s, iters, 1e-3);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  static unsigned long long h[256 * 16]; hipMemcpy(h, d, blocks * waves * 8, hipMemcpyDeviceToHost);
  double sum = 0; for (int i = 0; i < blocks * waves; ++i) sum += h[i];
  const double n = iters * 8.0;
  // s_memtime counts at 100 MHz on this part: also report wall-clock ns per instruction per SIMD
  printf("%-14s waves/CU=%2d  memtime ticks/instr/wave %7.3f   wall: %6.2f ns per wave-instr per SIMD\n", name, waves,
         sum / (blocks * waves) / n, ms * 1e6 / (n * (waves / 4.0)));
  hipFree(d); hipFree(s);
}
int main() {
  for (int w : {4, 8}) {
    if (w == 4) { run<4>("v_add_f32", 4); run<0>("v_add_f64", 4); run<1>("v_max_f64", 4); run<2>("v_fma_f64", 4);
                  run<3>("v_cmp_gt_f64", 4); run<5>("v_mul_f64", 4); run<6>("v_cndmask_b32", 4); }
    else { run<4>("v_add_f32", 8); run<0>("v_add_f64", 8); run<1>("v_max_f64", 8); run<2>("v_fma_f64", 8); run<3>("v_cmp_gt_f64", 8); }
  }
  // more waves per SIMD: a wave64 VALU instruction occupies the SIMD for four cycles, so once the SIMD saturates the
  // ticks per instruction per wave are 4 x (waves per SIMD) -- which also calibrates the tick against the cycle
  run<4>("v_add_f32", 12); run<4>("v_add_f32", 16); run<0>("v_add_f64", 12); run<0>("v_add_f64", 16);
  return 0;
}
