// Micro-probe: how fast can ONE CU pull bytes into LDS with buffer_load ... lds (1 KiB per wave-instruction),
// by source (L2-resident region shared by every workgroup vs a streaming region unique per workgroup) and by the
// number of issuing waves / pieces kept in flight?  Evidence for DESIGN.md 4.1 (the GEMM k-loops' ingest bound).
//   hipcc -O3 --offload-arch=gfx950 dma_probe.hip -o dma_probe && ./dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) unsigned char* lds_ptr_t;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ constexpr int vm_imm(int n) { return ((n >> 4) << 14) | 0x0F70 | (n & 15); }

// mode: 0 = hot only (HOT_KB per iteration, the same region for every workgroup and iteration -> L2 hits)
//       1 = stream only (STR_KB per iteration, unique per workgroup and iteration -> HBM / MALL)
//       2 = both per iteration (HOT_KB hot + STR_KB streaming)  = the GEMM k-tile mix (48 + 16)
// NW issuing waves share the pieces of an iteration; each wave waits until at most KEEP of its pieces are in flight
// before it starts the next iteration (KEEP = 0: drain).
template <int NW, int KEEP, bool PLAIN>
__global__ __launch_bounds__(64 * NW) void probe(const unsigned char* hot, const unsigned char* stream, int mode, int hot_kb,
                                                 int str_kb, int iters, size_t stream_stride, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rh = make_rsrc(hot);
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(stream + (size_t)blockIdx.x * stream_stride);
  const int nh = (mode == 1) ? 0 : hot_kb, ns = (mode == 0) ? 0 : str_kb;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    for (int pc = wid; pc < nh; pc += NW) {
      if constexpr (PLAIN) {
        const uint4 v = *reinterpret_cast<const uint4*>(hot + pc * 1024 + lane * 16);
        acc += v.x ^ v.w;
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (lds_ptr_t)(smem + (pc & 127) * 1024), 16, lane * 16, pc * 1024, 0, 0);
      }
    }
    for (int pc = wid; pc < ns; pc += NW) {
      const int off = (it * str_kb + pc) * 1024;
      if constexpr (PLAIN) {
        const uint4 v = *reinterpret_cast<const uint4*>(stream + (size_t)blockIdx.x * stream_stride + off + lane * 16);
        acc += v.x ^ v.w;
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + ((nh + pc) & 127) * 1024), 16, lane * 16, off, 0, 0);
      }
    }
    if constexpr (!PLAIN) __builtin_amdgcn_s_waitcnt(vm_imm(KEEP));
  }
  if constexpr (!PLAIN) {
    __builtin_amdgcn_s_waitcnt(vm_imm(0));
    acc = reinterpret_cast<unsigned*>(smem)[threadIdx.x];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int NW, int KEEP, bool PLAIN>
static void run(const char* tag, const unsigned char* hot, const unsigned char* stream, unsigned* sink, int n_cu, int mode,
                int hot_kb, int str_kb, int iters, size_t stride, double ghz) {
  auto k = probe<NW, KEEP, PLAIN>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k, dim3(n_cu), dim3(64 * NW), 128 * 1024, 0, hot, stream, mode, hot_kb, str_kb, iters, stride, sink);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    if (rep && ms < best) best = ms;
  }
  const double kb = ((mode == 1 ? 0 : hot_kb) + (mode == 0 ? 0 : str_kb)) * (double)iters;
  const double gbs_cu = kb * 1024 / (best * 1e-3) / 1e9;
  printf("%-8s waves=%d keep=%2d mode=%d (%2d KiB hot + %2d KiB stream per iteration): %8.3f ms  %6.1f GB/s per CU  %5.1f B/clk/CU @%.2f GHz  %6.2f TB/s chip\n",
         tag, NW, KEEP, mode, mode == 1 ? 0 : hot_kb, mode == 0 ? 0 : str_kb, best, gbs_cu, gbs_cu / ghz, ghz, gbs_cu * n_cu / 1e3);
}

int main(int argc, char** argv) {
  hipDeviceProp_t pr;
  CHECK(hipGetDeviceProperties(&pr, 0));
  const int n_cu = pr.multiProcessorCount;
  const double ghz = argc > 1 ? atof(argv[1]) : 1.9;
  printf("%s: %d CUs, clockRate %.2f GHz (B/clk uses %.2f GHz)\n", pr.gcnArchName, n_cu, pr.clockRate / 1e6, ghz);
  const int iters = 400, hot_kb = 48, str_kb = 16;
  const size_t stride = (size_t)iters * 64 * 1024;  // room for 64 KiB per iteration per workgroup
  unsigned char *hot, *stream;
  unsigned* sink;
  CHECK(hipMalloc(&hot, 1 << 20));
  CHECK(hipMalloc(&stream, stride * n_cu));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(hot, 1, 1 << 20));
  CHECK(hipMemset(stream, 1, stride * n_cu));
  for (int mode = 0; mode < 3; ++mode) {
    run<1, 16, false>("lds-dma", hot, stream, sink, n_cu, mode, hot_kb, str_kb, iters, stride, ghz);
    run<1, 48, false>("lds-dma", hot, stream, sink, n_cu, mode, hot_kb, str_kb, iters, stride, ghz);
    run<2, 16, false>("lds-dma", hot, stream, sink, n_cu, mode, hot_kb, str_kb, iters, stride, ghz);
    run<4, 8, false>("lds-dma", hot, stream, sink, n_cu, mode, hot_kb, str_kb, iters, stride, ghz);
    run<8, 4, false>("lds-dma", hot, stream, sink, n_cu, mode, hot_kb, str_kb, iters, stride, ghz);
    run<8, 0, false>("lds-dma", hot, stream, sink, n_cu, mode, hot_kb, str_kb, iters, stride, ghz);
    run<8, 0, true>("plain", hot, stream, sink, n_cu, mode, hot_kb, str_kb, iters, stride, ghz);
  }
  // stream only, 64 KiB per iteration (all bytes unique): the HBM-side ceiling of the same loop
  run<8, 4, false>("lds-dma", hot, stream, sink, n_cu, 1, 0, 64, iters, stride, ghz);
  // hot only, 64 KiB per iteration
  run<8, 4, false>("lds-dma", hot, stream, sink, n_cu, 0, 64, 0, iters, stride, ghz);
  run<1, 16, false>("lds-dma", hot, stream, sink, n_cu, 0, 64, 0, iters, stride, ghz);
  return 0;
}
