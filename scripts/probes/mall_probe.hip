// Micro-probe (round 4): does the 256 MiB Infinity Cache keep what a kernel just WROTE, i.e. would a consumer kernel that walks its
// input in the REVERSE of the producer's order (newest bytes first) read part of it from the cache?  Producer: streaming 16-byte
// stores over S MiB (optionally preceded by reading 3 S MiB, like the attention kernel that reads q | k | v and writes ctx).
// Consumer: streaming 16-byte loads over the same S MiB, forward or reversed (block-granular), timed.
//   hipcc -O3 --offload-arch=gfx950 mall_probe.hip -o mall_probe && ./mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void writer(uint4* dst, size_t n16, unsigned v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = uint4{v, v + 1, v + 2, (unsigned)i};
}
// chunk-granular order: chunk c of `nchunk` (64 KiB each) is visited at position c (forward) or nchunk - 1 - c (reverse)
__global__ __launch_bounds__(256) void reader(const uint4* src, size_t nchunk, int reverse, unsigned* sink) {
  unsigned acc = 0;
  for (size_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
    const size_t cc = reverse ? nchunk - 1 - c : c;
    const uint4* p = src + cc * 4096;  // 64 KiB = 4096 x 16 B
#pragma unroll 4
    for (int i = threadIdx.x; i < 4096; i += 256) {
      const uint4 v = p[i];
      acc += v.x ^ v.w;
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  hipDeviceProp_t pr;
  CHECK(hipGetDeviceProperties(&pr, 0));
  const int grid = pr.multiProcessorCount * 8;
  printf("%s: %d CUs\n", pr.gcnArchName, pr.multiProcessorCount);
  unsigned* sink;
  CHECK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int mib : {32, 64, 100, 200, 400}) {
    const size_t bytes = (size_t)mib << 20, n16 = bytes / 16, nchunk = bytes / 65536;
    uint4 *buf, *other;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&other, 3 * bytes));
    CHECK(hipMemset(other, 1, 3 * bytes));
    for (int pre = 0; pre < 2; ++pre)       // pre = 1: the producer kernel is preceded by 3 S MiB of reads (other traffic through the cache)
      for (int rev = 0; rev < 2; ++rev) {
        float best = 1e30f, wbest = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
          if (pre) hipLaunchKernelGGL(reader, dim3(grid), dim3(256), 0, 0, other, 3 * nchunk, 0, sink);
          CHECK(hipEventRecord(e0));
          hipLaunchKernelGGL(writer, dim3(grid), dim3(256), 0, 0, buf, n16, (unsigned)rep);
          CHECK(hipEventRecord(e1));
          CHECK(hipEventSynchronize(e1));
          float wms;
          CHECK(hipEventElapsedTime(&wms, e0, e1));
          CHECK(hipEventRecord(e0));
          hipLaunchKernelGGL(reader, dim3(grid), dim3(256), 0, 0, buf, nchunk, rev, sink);
          CHECK(hipEventRecord(e1));
          CHECK(hipEventSynchronize(e1));
          float ms;
          CHECK(hipEventElapsedTime(&ms, e0, e1));
          if (rep && ms < best) best = ms;
          if (rep && wms < wbest) wbest = wms;
        }
        printf("%4d MiB written %s then read %-8s: read %7.3f ms = %6.2f TB/s   (write %6.2f TB/s)\n", mib,
               pre ? "(after 3x reads of other data)" : "                              ", rev ? "REVERSED" : "forward", best,
               bytes / (best * 1e-3) / 1e12, bytes / (wbest * 1e-3) / 1e12);
      }
    CHECK(hipFree(buf));
    CHECK(hipFree(other));
  }
  return 0;
}
