// Micro-probe (round 6): is "buffer_load ... lds" followed at once by a write of M0 a hazard when the vector-memory queue is deep?
// ffn16.hip's first builds fetched a wave's 6 / 12 KiB of operand rows with LDS-DMA pieces interleaved with ordinary loads in the
// prologue (18-36 vector-memory instructions back to back, M0 rewritten right behind every piece); now and then a piece's LDS rows
// kept the previous launch's contents (profiles/r06_ffn16_notes.log).  Here: every wave issues NP pieces back to back, each as
//   s_mov_b32 m0, dst_i ; s_nop 0 ; buffer_load_dwordx4 ... lds ; [GAP x s_nop 0] ; (next) s_mov_b32 m0, dst_{i+1} ...
// optionally with FILL ordinary loads in front (a deep queue) and scattered per-lane offsets (the row-image pattern), from a cold
// buffer; LDS is filled with a sentinel first; after vmcnt(0) + barrier every piece must hold ITS source bytes.
//   hipcc -O3 --offload-arch=gfx950 m0_war_probe.hip -o m0_war_probe && ./m0_war_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NP, int GAP, int FILL, bool SCATTER>
__global__ __launch_bounds__(512) void probe(const unsigned* src, unsigned src_bytes, const unsigned* cold, int* bad, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wq = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned* lds = reinterpret_cast<unsigned*>(smem);
  for (int i = threadIdx.x; i < 8 * NP * 256; i += 512) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(cold), 0, 0x7fffffff, 0x00020000);
  // per-lane source offset inside a piece: linear, or the row-image pattern (16 rows x 4 units: row c at c * 16, unit g at g * 512)
  const int voff = SCATTER ? (lane & 15) * 16 + (lane >> 4) * 512 : lane * 16;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem + (unsigned)(wq * NP * 1024);
  const int base = ((int)blockIdx.x * 8 + wq) * NP * 4096;   // this wave's source region (a piece spans 2 KiB when scattered)
  u32x4 f[FILL > 0 ? FILL : 1];
#pragma unroll
  for (int k = 0; k < FILL; ++k)
    f[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rc, lane * 16, (((int)blockIdx.x * 8 + wq) * FILL + k) * 4096, 0));
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(i * 1024));
    const int so = __builtin_amdgcn_readfirstlane(base + i * 4096);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(voff), "s"(rs), "s"(so) : "memory");
#pragma unroll
    for (int g = 0; g < GAP; ++g) asm volatile("s_nop 0");
    asm volatile("s_mov_b32 m0, -1" ::: "memory");   // the write of M0 right behind the piece (what the next piece's s_mov does in a kernel)
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();
  unsigned acc = 0;
#pragma unroll
  for (int k = 0; k < FILL; ++k) acc += f[k][0];
  int wrong = 0;
  for (int i = 0; i < NP; ++i) {
    const unsigned got = lds[(wq * NP + i) * 256 + lane * 4];
    const unsigned want = src[(base + i * 4096 + voff) / 4];
    wrong += got != want;
  }
  if (wrong) atomicAdd(bad, wrong);
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int NP, int GAP, int FILL, bool SCATTER>
static void run(const unsigned* src, unsigned bytes, const unsigned* cold, int* bad, unsigned* sink, int reps) {
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<NP, GAP, FILL, SCATTER>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * NP * 1024));
  int total = 0;
  for (int r = 0; r < reps; ++r) {
    CHECK(hipMemset(bad, 0, 4));
    hipLaunchKernelGGL((probe<NP, GAP, FILL, SCATTER>), dim3(256), dim3(512), 8 * NP * 1024, 0, src, bytes, cold, bad, sink);
    CHECK(hipDeviceSynchronize());
    int h = 0;
    CHECK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
    total += h;
  }
  printf("pieces per wave %2d  gap %2d s_nop  %2d ordinary loads in front  %-9s: %d wrong lane-pieces in %d launches of 256 x 8 waves\n", NP, GAP, FILL,
         SCATTER ? "scattered" : "linear", total, reps);
}

int main() {
  const size_t n = (size_t)256 * 8 * 12 * 4096 + 65536;
  std::vector<unsigned> h(n / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u) | 1u;
  unsigned *src, *cold, *sink;
  int* bad;
  CHECK(hipMalloc(&src, n)); CHECK(hipMalloc(&cold, (size_t)256 * 8 * 24 * 4096 + 65536)); CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&bad, 4));
  CHECK(hipMemcpy(src, h.data(), n, hipMemcpyHostToDevice));
  CHECK(hipMemset(cold, 1, (size_t)256 * 8 * 24 * 4096 + 65536));
  const int reps = 200;
  run<12, 0, 0, false>(src, (unsigned)n, cold, bad, sink, reps);
  run<12, 0, 0, true>(src, (unsigned)n, cold, bad, sink, reps);
  run<12, 0, 24, false>(src, (unsigned)n, cold, bad, sink, reps);
  run<12, 0, 24, true>(src, (unsigned)n, cold, bad, sink, reps);
  run<6, 0, 12, true>(src, (unsigned)n, cold, bad, sink, reps);
  run<12, 2, 24, true>(src, (unsigned)n, cold, bad, sink, reps);
  run<12, 8, 24, true>(src, (unsigned)n, cold, bad, sink, reps);
  return 0;
}
