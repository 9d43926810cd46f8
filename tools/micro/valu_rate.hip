// microbenchmark (gfx950): VALU issue rate of the float32 instructions the frame kernel is made of, at 1..4 waves
// per SIMD, from wall time (HIP events) and from s_memtime / s_memrealtime (clock under load).
//   OP 0 v_fma_f32   1 v_pk_fma_f32   2 v_pk_add_f32   3 v_pk_mul_f32   4 v_add_f32   5 v_mov_b32
//   6 pk_fma + ds_read_b64 (1 read per 4 VALU)   7 pk_fma + ds_read_b64 + ds_write_b64 (1+1 per 8 VALU)
//   8 v_log_f32   9 v_pk_fma_f32 with 16 chains   10 v_fma_f32 with a dependent chain of 2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP>
__global__ void k(unsigned long long* out, float* sink, int iters, float seed) {
  __shared__ float2 lds[4096];
  const int lane = threadIdx.x & 63;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = f2{seed * (lane + i + 1), seed * (lane + 2 * i + 1)};
  const f2 c = f2{seed * 0.5f, seed * 0.25f};
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = make_float2(seed, seed);
  __syncthreads();
  float2* lp = lds + ((threadIdx.x * 1) & 4095);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[u].x) : "v"(c.x));
      if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[u]) : "v"(c));
      if (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[u]) : "v"(c));
      if (OP == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[u]) : "v"(c));
      if (OP == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[u].x) : "v"(c.x));
      if (OP == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(a[u].x) : "v"(c.x));
      if (OP == 6) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[u]) : "v"(c));
        if ((u & 3) == 3) { float2 t = lp[u * 64]; asm volatile("" :: "v"(t.x), "v"(t.y)); }
      }
      if (OP == 7) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[u]) : "v"(c));
        if (u == 3) { float2 t = lp[u * 64]; asm volatile("" :: "v"(t.x), "v"(t.y)); }
        if (u == 7) { lp[u * 64] = make_float2(a[0].x, a[1].y); }
      }
      if (OP == 8) asm volatile("v_log_f32 %0, %0" : "+v"(a[u].x));
      if (OP == 9) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[u]) : "v"(c));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[u + 8]) : "v"(c));
      }
      if (OP == 10) {
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[u & 1].x) : "v"(c.x));
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  if (lane == 0) {
    out[(blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)) * 2] = t1 - t0;
    out[(blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0;
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, int waves, int per_iter = 8) {
  const int iters = 4000, blocks = 256;
  unsigned long long* d; float* s;
  hipMalloc(&d, blocks * waves * 16); hipMalloc(&s, blocks * waves * 64 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64 * waves), 0, 0, d, s, iters, 1e-3f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64 * waves), 0, 0, d, s, iters, 1e-3f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * waves * 2);
  hipMemcpy(h.data(), d, blocks * waves * 16, hipMemcpyDeviceToHost);
  double st = 0, sr = 0; for (int i = 0; i < blocks * waves; ++i) { st += h[2 * i]; sr += h[2 * i + 1]; }
  const double n = (double)iters * per_iter;             // VALU instructions per wave
  const double ghz = st / sr * 0.1;                      // s_memrealtime = 100 MHz
  const double cyc_per_instr_wave = st / (blocks * waves) / n;
  printf("%-22s waves/SIMD=%d  clock %.2f GHz  cycles/instr/wave %6.2f  -> cycles per instr per SIMD %5.2f   wall %.3f ms\n",
         name, waves / 4, ghz, cyc_per_instr_wave, cyc_per_instr_wave / (waves / 4.0), ms);
  hipFree(d); hipFree(s);
}
int main() {
  for (int w : {4, 8, 12, 16}) {
    run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<2>("v_pk_add_f32", w); run<3>("v_pk_mul_f32", w);
    run<4>("v_add_f32", w); run<5>("v_mov_b32", w); run<6>("pk_fma+ds_read/4", w); run<7>("pk_fma+rd+wr/8", w);
    run<8>("v_log_f32", w); run<9>("v_pk_fma_f32 x16", w, 16); run<10>("v_fma_f32 dep2", w);
    printf("\n");
  }
  return 0;
}
