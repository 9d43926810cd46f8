// microbenchmark: issue cost of v_mfma_f32_16x16x4_f32 on gfx950, alone and with the VALU work the mel
// stage puts between two MFMAs.  hipcc --offload-arch=gfx950 -O3 mfma_f32.hip -o mfma_f32 && ./mfma_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(unsigned long long* out, float* sink, int iters, float a0) {
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0,0,0,0}, acc3 = {0,0,0,0};
  float kf = threadIdx.x, b = threadIdx.x * 0.5f + a0;
  const float c0 = a0, c1 = a0 * 0.5f, c2 = 1.f - a0, c3 = -a0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float w = b;
      if (MODE >= 1) { const float lo = fmaf(c1, kf, c0), hi = fmaf(c3, kf, c2); w = __builtin_amdgcn_fmed3f(0.f, lo, hi); kf += 4.f; }
      if (MODE == 2 || MODE == 0) { if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b, acc1, 0, 0, 0); else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b, acc0, 0, 0, 0); }
      if (MODE == 3) { if ((u & 3) == 0) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b, acc0, 0, 0, 0); else if ((u&3)==1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b, acc1, 0, 0, 0); else if ((u&3)==2) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b, acc2, 0, 0, 0); else acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b, acc3, 0, 0, 0); }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3] + kf;
}
template <int MODE> void run(const char* name, int waves_per_cu) {
  const int iters = 2000, blocks = 256, threads = 64 * waves_per_cu;
  unsigned long long* d; float* s;
  hipMalloc(&d, blocks * waves_per_cu * 8); hipMalloc(&s, blocks * threads * 4);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, s, iters, 0.25f);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, s, iters, 0.25f);
  hipDeviceSynchronize();
  unsigned long long h[256 * 16]; hipMemcpy(h, d, blocks * waves_per_cu * 8, hipMemcpyDeviceToHost);
  double sum = 0; for (int i = 0; i < blocks * waves_per_cu; ++i) sum += h[i];
  printf("%-34s waves/CU=%2d  cycles per MFMA slot = %.1f\n", name, waves_per_cu, sum / (blocks * waves_per_cu) / (iters * 8.0));
  hipFree(d); hipFree(s);
}
int main() {
  for (int w : {1, 4, 8}) {
    if (w == 1) { run<0>("mfma only, 2 accumulators", 1); run<3>("mfma only, 4 accumulators", 1); run<2>("mfma + 4 VALU, 2 acc", 1); run<1>("4 VALU only", 1); }
    if (w == 4) { run<0>("mfma only, 2 accumulators", 4); run<3>("mfma only, 4 accumulators", 4); run<2>("mfma + 4 VALU, 2 acc", 4); run<1>("4 VALU only", 4); }
    if (w == 8) { run<0>("mfma only, 2 accumulators", 8); run<3>("mfma only, 4 accumulators", 8); run<2>("mfma + 4 VALU, 2 acc", 8); run<1>("4 VALU only", 8); }
  }
  return 0;
}
