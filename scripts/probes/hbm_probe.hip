// Micro-probe: achievable HBM rates for streaming reads, streaming writes and a copy, 16 bytes per lane, all CUs.
//   hipcc -O3 --offload-arch=gfx950 hbm_probe.hip -o hbm_probe && ./hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void rd(const uint4* __restrict__ src, size_t n, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = src[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void wr(uint4* __restrict__ dst, size_t n) {
  const uint4 v = {1u, 2u, 3u, (unsigned)threadIdx.x};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}
__global__ __launch_bounds__(256) void cp(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// 1 read : 3 writes (the q | k | v projection's mix)
__global__ __launch_bounds__(256) void r1w3(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = src[i];
    dst[i] = v; dst[i + n] = v; dst[i + 2 * n] = v;
  }
}

template <typename F>
static float time_ms(F f) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CHECK(hipEventRecord(a)); f(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    if (r && ms < best) best = ms;
  }
  return best;
}

int main() {
  for (size_t mb : {100, 400, 2000}) {   // 100 MB fits the 256 MB Infinity Cache, 2 GB does not
    const size_t bytes = mb << 20, n = bytes / 16;
    uint4 *a, *b; unsigned* sink;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, 3 * bytes)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(a, 1, bytes)); CHECK(hipMemset(b, 1, 3 * bytes));
    const int grid = 256 * 8;
    const float tr = time_ms([&] { hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, a, n, sink); });
    const float tw = time_ms([&] { hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, b, n); });
    const float tc = time_ms([&] { hipLaunchKernelGGL(cp, dim3(grid), dim3(256), 0, 0, a, b, n); });
    const float t13 = time_ms([&] { hipLaunchKernelGGL(r1w3, dim3(grid), dim3(256), 0, 0, a, b, n); });
    printf("%5zu MB: read %6.2f TB/s   write %6.2f TB/s   copy %6.2f TB/s (read + written bytes)   1 read : 3 writes %6.2f TB/s\n", mb,
           bytes / (tr * 1e-3) / 1e12, bytes / (tw * 1e-3) / 1e12, 2.0 * bytes / (tc * 1e-3) / 1e12, 4.0 * bytes / (t13 * 1e-3) / 1e12);
    CHECK(hipFree(a)); CHECK(hipFree(b)); CHECK(hipFree(sink));
  }
  // small footprints, rewritten / re-read 32 times per launch: the per-CU path into L2 without HBM behind it
  for (size_t mb : {8, 16, 64}) {
    const size_t bytes = mb << 20, n = bytes / 16;
    uint4 *a; unsigned* sink;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(a, 1, bytes));
    const int grid = 256 * 8, reps = 32;
    const float tw = time_ms([&] { for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, a, n); });
    const float tr = time_ms([&] { for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, a, n, sink); });
    printf("%5zu MB x %d launches: read %6.2f TB/s   write %6.2f TB/s\n", mb, reps, reps * (double)bytes / (tr * 1e-3) / 1e12, reps * (double)bytes / (tw * 1e-3) / 1e12);
    CHECK(hipFree(a)); CHECK(hipFree(sink));
  }
  return 0;
}
