// Micro-probe: what does s_memtime count?  A back-to-back MFMA stream of known length (32 matrix cycles per instruction per SIMD) is
// timed with s_memtime and with hipEvents, on trivial and on random operands (DVFS).
//   hipcc -O3 --offload-arch=gfx950 clock_probe.hip -o clock_probe && ./clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define MFMA(j) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b))
__global__ __launch_bounds__(256) void probe(int iters, int rnd, long long* out, float* sink) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  f32x16 acc[6];
  for (int j = 0; j < 6; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  f16x8 a, b;
  unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  for (int r = 0; r < 8; ++r) {
    s = s * 1664525u + 1013904223u; const float u = ((s >> 8) & 0xffff) / 65536.0f - 0.5f;
    s = s * 1664525u + 1013904223u; const float v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f;
    a[r] = (_Float16)(rnd ? u : 1.0f); b[r] = (_Float16)(rnd ? v : 0.0f);
  }
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 6; ++j) MFMA(j);
  }
  asm volatile("s_nop 15\n\ts_nop 15");
  const long long t1 = __builtin_amdgcn_s_memtime();
  float x = 0.f;
  for (int j = 0; j < 6; ++j) x += acc[j][0] + acc[j][15];
  if (x == 12345.678f) sink[0] = x;
  if (lane == 0 && blockIdx.x == 0 && wid == 0) out[0] = t1 - t0;
}
int main() {
  long long* out; float* sink;
  CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rnd = 0; rnd < 2; ++rnd)
    for (int rep = 0; rep < 3; ++rep) {
      const int iters = 200000;
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(probe, dim3(256 * 2), dim3(256), 0, 0, iters, rnd, out, sink);  // 2 blocks of 4 waves per CU: 2 waves per SIMD
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      long long h; CHECK(hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost));
      const double mfma_per_simd = 2.0 * 6.0 * iters;  // two waves per SIMD
      printf("%s operands: %7.2f ms wall, s_memtime ticks %lld -> tick rate %.3f GHz; matrix cycles per SIMD %.0f -> clock >= %.3f GHz; ticks per MFMA (per SIMD) %.2f\n",
             rnd ? "random " : "trivial", ms, h, h / (ms * 1e6), mfma_per_simd * 32, mfma_per_simd * 32 / (ms * 1e6), h / mfma_per_simd);
    }
  return 0;
}
