// Micro-probe: what does ONE CU sustain on vector-memory stores / loads (16 bytes per lane, fully coalesced), and how does it scale
// with the number of CUs active?  (Is the ~10 B/clk/CU seen with every CU storing a per-CU limit or the chip's write limit?)
//   hipcc -O3 --offload-arch=gfx950 cu_store_probe.hip -o cu_store_probe && ./cu_store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(512) void wr(uint4* __restrict__ dst, size_t per_wg) {
  uint4* p = dst + (size_t)blockIdx.x * per_wg;
  const uint4 v = {1u, 2u, 3u, (unsigned)threadIdx.x};
  for (size_t i = threadIdx.x; i < per_wg; i += 512) p[i] = v;
}
__global__ __launch_bounds__(512) void rd(const uint4* __restrict__ src, size_t per_wg, unsigned* sink) {
  const uint4* p = src + (size_t)blockIdx.x * per_wg;
  unsigned acc = 0;
#pragma unroll 8
  for (size_t i = threadIdx.x; i < per_wg; i += 512) { const uint4 v = p[i]; acc += v.x ^ v.w; }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const size_t per_wg_bytes = 64u << 20, per_wg = per_wg_bytes / 16;
  uint4* buf; unsigned* sink;
  CHECK(hipMalloc(&buf, per_wg_bytes * 256)); CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(buf, 1, per_wg_bytes * 256));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  const double ghz = 2.1;
  for (int ncu : {1, 2, 8, 32, 64, 128, 256}) {
    float best_w = 1e30f, best_r = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      float ms;
      CHECK(hipEventRecord(a)); hipLaunchKernelGGL(wr, dim3(ncu), dim3(512), 0, 0, buf, per_wg); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
      CHECK(hipEventElapsedTime(&ms, a, b)); if (rep && ms < best_w) best_w = ms;
      CHECK(hipEventRecord(a)); hipLaunchKernelGGL(rd, dim3(ncu), dim3(512), 0, 0, buf, per_wg, sink); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
      CHECK(hipEventElapsedTime(&ms, a, b)); if (rep && ms < best_r) best_r = ms;
    }
    const double gw = per_wg_bytes / (best_w * 1e-3) / 1e9, gr = per_wg_bytes / (best_r * 1e-3) / 1e9;
    printf("%3d workgroups (one per CU, 64 MiB each): store %6.1f GB/s per CU = %5.1f B/clk @%.1f GHz (chip %5.2f TB/s)   load %6.1f GB/s per CU = %5.1f B/clk (chip %5.2f TB/s)\n",
           ncu, gw, gw / ghz, ghz, gw * ncu / 1e3, gr, gr / ghz, gr * ncu / 1e3);
  }
  return 0;
}
