// microbenchmark: LDS-pipe cost per wave-instruction of the ops the FFT exchange uses (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void k(unsigned long long* out, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8 * 1024 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* base = lds + wave * 4096;
  float2 v2 = make_float2(lane, lane * 2.f); float4 v4 = make_float4(lane, 1, 2, 3); float v1 = lane;
  float acc = 0.f;
  for (int i = lane; i < 4096; i += 64) base[i] = i;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (OP == 0) reinterpret_cast<float2*>(base)[lane + 64 * (u & 7)] = v2;                       // ds_write_b64
      if (OP == 1) { float2 t = reinterpret_cast<float2*>(base)[lane + 64 * (u & 7)]; acc += t.x + t.y; } // ds_read_b64
      if (OP == 2) base[lane + 64 * u] = v1;                                                            // ds_write_b32
      if (OP == 3) acc += base[lane + 64 * u];                                                          // ds_read_b32
      if (OP == 4) acc += __int_as_float(__builtin_amdgcn_ds_bpermute(((63 - lane) << 2), __float_as_int(v1 + u))); // bpermute
      if (OP == 5) reinterpret_cast<float4*>(base)[lane + 64 * (u & 3)] = v4;                       // ds_write_b128
      if (OP == 6) { float4 t = reinterpret_cast<float4*>(base)[lane + 64 * (u & 3)]; acc += t.x + t.w; }  // ds_read_b128
    }
    asm volatile("" ::: "memory");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc + v2.x + v4.x;
}
template <int OP> void run(const char* name, int waves) {
  const int iters = 1000, blocks = 256;
  unsigned long long* d; float* s;
  hipMalloc(&d, blocks * waves * 8); hipMalloc(&s, blocks * waves * 64 * 4);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64 * waves), 0, 0, d, s, iters);
  hipDeviceSynchronize();
  unsigned long long h[256 * 8]; hipMemcpy(h, d, blocks * waves * 8, hipMemcpyDeviceToHost);
  double sum = 0; for (int i = 0; i < blocks * waves; ++i) sum += h[i];
  const double per_wave = sum / (blocks * waves) / (iters * 16.0);
  printf("%-16s waves/CU=%d  cycles per instr per wave = %6.1f  => LDS pipe cycles per instr = %5.1f\n", name, waves, per_wave, per_wave / waves);
  hipFree(d); hipFree(s);
}
int main() {
  run<0>("ds_write_b64", 8); run<1>("ds_read_b64", 8); run<2>("ds_write_b32", 8); run<3>("ds_read_b32", 8);
  run<4>("ds_bpermute_b32", 8); run<5>("ds_write_b128", 8); run<6>("ds_read_b128", 8);
  run<0>("ds_write_b64", 4); run<1>("ds_read_b64", 4); run<4>("ds_bpermute_b32", 4);
  return 0;
}
