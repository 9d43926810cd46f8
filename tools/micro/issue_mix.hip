// microbenchmark (gfx950): does a SIMD issue LDS / scalar instructions beside its vector instructions, or do they
// share issue slots?  Per iteration 8 v_pk_fma_f32 plus a variable number of ds_read_b64 / ds_write_b64 / s_add,
// LDS results consumed one iteration later (no s_waitcnt in the loop except lgkmcnt(4)), at 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int NR, int NW, int NS>
__global__ void k(unsigned long long* out, float* sink, int iters, float seed) {
  __shared__ f2 lds[16 * 1024 / 8 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f2 a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = f2{seed * (lane + i + 1), seed * (lane + 2 * i + 1)};
  const f2 c = f2{seed * 0.5f, seed * 0.25f};
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = f2{seed, seed};
  __syncthreads();
  const unsigned base = (unsigned)(size_t)(lds) + (wave * 512 + lane) * 8;      // LDS byte address (low 32 bits of the generic pointer)
  f2 r[4] = {c, c, c, c};
  int sacc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[u]) : "v"(c));
      if (u < NR) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[u & 3]) : "v"(base), "n"(u * 512));
      if (u < NW) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(base), "v"(a[u]), "n"(u * 512 + 4096));
      if (u < NS) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sacc));
    }
    if (NR + NW > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
  float s = sacc;
  for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
  for (int i = 0; i < 4; ++i) s += r[i].x;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NR, int NW, int NS> void run(int waves) {
  const int iters = 2000, blocks = 256;
  unsigned long long* d; float* s;
  hipMalloc(&d, blocks * waves * 8); hipMalloc(&s, blocks * waves * 64 * 4);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<NR, NW, NS>), dim3(blocks), dim3(64 * waves), 0, 0, d, s, iters, 1e-3f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * waves);
  hipMemcpy(h.data(), d, blocks * waves * 8, hipMemcpyDeviceToHost);
  double st = 0; for (auto v : h) st += v;
  const double per_iter = st / (blocks * waves) / iters;
  printf("waves/SIMD=%d  8 pk_fma + %d ds_read_b64 + %d ds_write_b64 + %d s_add: %7.1f cycles per iteration per wave = %5.1f per SIMD\n",
         waves / 4, NR, NW, NS, per_iter, per_iter / (waves / 4.0));
  hipFree(d); hipFree(s);
}
int main() {
  for (int w : {4, 8, 12, 16}) {
    run<0, 0, 0>(w); run<2, 0, 0>(w); run<4, 0, 0>(w); run<0, 2, 0>(w); run<0, 4, 0>(w); run<2, 2, 0>(w); run<0, 0, 4>(w); run<2, 2, 4>(w); run<8, 0, 0>(w); run<0, 8, 0>(w);
    printf("\n");
  }
  return 0;
}
