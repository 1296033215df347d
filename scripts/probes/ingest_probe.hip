// Micro-probe (round 4): why does the GEMM's copy stream need 2200-2900 cycles per 64 KiB k-tile when
// dma_probe lands 48 KiB hot + 16 KiB stream per iteration in ~1700?  This probe rebuilds the product's loader protocol
// (gemm_img.hip: two loader waves, W ring of 2 x 48 KiB, A ring of 3 x 16 KiB, counted vmcnt, ONE raw s_barrier per stage,
// eight consumer waves) and switches the product's ingredients on one at a time:
//     hot set      48 KiB re-read every stage (dma_probe: fits mostly in the 32 KiB L1 + always the same L2 lines)
//                  vs the real thing: a 576 KiB weight tile walked stage by stage (12 x 48 KiB), three tiles per XCD
//     barrier      free-running loaders (keep the ring's worth in flight) vs the per-stage barrier with 8 waiting consumer waves
//     reads        the consumers' fragment reads (20 ds_read_b128 per wave and stage = 160 KiB of LDS reads per stage)
//     mfma         36 MFMAs per consumer wave and stage (the k-tile's matrix work)
//     stores       192 KiB of 16-byte stores per 12 stages from the consumer waves (a tile's output)
// Output: average cycles per stage (s_memtime of workgroup 0, and wall time x clock) per variant.
//   hipcc -O3 --offload-arch=gfx950 ingest_probe.hip -o ingest_probe && ./ingest_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) unsigned char* lds_ptr_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ constexpr int vm_imm(int n) { return ((n >> 4) << 14) | 0x0F70 | (n & 15); }
#define WAIT_VM(n) __builtin_amdgcn_s_waitcnt(vm_imm(n))
__device__ __forceinline__ void barrier_keep_vm() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int W_STAGE = 48 * 1024, A_STAGE = 16 * 1024, OFF_A = 2 * W_STAGE, SMEM = OFF_A + 3 * A_STAGE;

struct Args {
  const unsigned char* w;   // weight image: n_wtiles x (nk x 48 KiB)
  const unsigned char* a;   // activation panels: one per workgroup group, nk x 16 KiB each, walked tile after tile
  unsigned char* out;       // store target: 192 KiB per workgroup and tile
  unsigned long long* stamps;
  int nk;                   // stages per tile (12)
  int stages;               // total stages per workgroup
  int hot_small;            // 1: every stage re-reads the SAME 48 KiB (dma_probe's hot set)
  int n_wtiles;             // weight tiles cycled over by blockIdx (3 = q | k | v)
  int a_share;              // workgroups sharing one A panel (3)
  size_t a_stride;          // bytes between A panels
  int delay_cycles;         // workgroup j of an XCD starts (j % delay_mod) * delay_cycles late (de-phasing / staggering experiments)
  int delay_mod;
  int store_aux;            // cache policy bits of the output stores (0 plain, 2 nt, 16 sc1, 17 sc0 sc1)
  int store_spread;         // 1: the tile's 24 stores per lane are issued two per stage over the NEXT tile's stages instead of in one burst
  int store_same;           // 1: every tile stores to the same 192 KiB (48 MiB footprint chip-wide instead of 196 MiB)
  int store_excl;           // 1: no load is in flight while the tile's stores are: the loaders drain (vmcnt 0) and wait at two extra barriers per tile,
                            //    the consumers issue the burst between them and drain it (vmcnt 0) before the second
};

// FLAGS: 1 barrier per stage (else free-running loaders), 2 consumer fragment reads, 4 MFMAs, 8 stores per tile
template <int FLAGS>
__global__ __launch_bounds__(640) void probe(Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int wt = jx % p.n_wtiles;                     // neighbouring workgroups of one XCD take the column tiles of one panel
  const int grp = xcd + 8 * (jx / p.a_share);         // ... and share its A panel
  constexpr bool BAR = FLAGS & 1, RD = FLAGS & 2, MM = FLAGS & 4, ST = FLAGS & 8;
  if (p.delay_cycles > 0) {
    const unsigned long long until = __builtin_amdgcn_s_memtime() + (unsigned long long)(jx % p.delay_mod) * p.delay_cycles;
    while (__builtin_amdgcn_s_memtime() < until) __builtin_amdgcn_s_sleep(8);
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wid >= 8) {  // ---- loaders
    const int li = wid - 8;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(p.w) + (size_t)wt * p.nk * W_STAGE, 0, p.nk * W_STAGE, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(p.a) + (size_t)grp * p.a_stride, 0, 0x7fffffff, 0x00020000);
    int w_g = 0, a_g = 0;
    auto issue_w = [&]() {
      const int so = p.hot_small ? 0 : (w_g % p.nk) * W_STAGE;
      lds_ptr_t dst = (lds_ptr_t)(smem) + (w_g & 1) * W_STAGE;
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        const int j = li + 2 * i;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, dst + j * 1024, 16, lane * 16, so + j * 1024, 0, 0);
      }
      ++w_g;
    };
    auto issue_a = [&]() {
      lds_ptr_t dst = (lds_ptr_t)(smem) + OFF_A + (a_g % 3) * A_STAGE;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int j = li + 2 * i;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, dst + j * 1024, 16, lane * 16, a_g * A_STAGE + j * 1024, 0, 0);
      }
      ++a_g;
    };
    issue_a();
    issue_w();
    issue_a();
    int lkt = 0;
    for (int g = 0; g < p.stages; ++g) {
      WAIT_VM(8);
      if constexpr (BAR) barrier_keep_vm();
      if (ST && p.store_excl && g > 0 && lkt == 0) {  // the consumers are about to store the tile that ended with stage g - 1
        WAIT_VM(0);
        barrier_keep_vm();  // loads drained -> stores may start
        barrier_keep_vm();  // stores drained -> loads resume
      }
      issue_w();
      issue_a();
      if (++lkt == p.nk) lkt = 0;
    }
    WAIT_VM(0);
  } else {  // ---- consumers
    f32x16 acc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int wn = wid & 3, wm = wid >> 2;
    int kt = 0, tile = 0;
    for (int g = 0; g < p.stages; ++g) {
      if constexpr (BAR) barrier_keep_vm();
      if (ST && p.store_excl && g > 0 && kt == 0) {
        barrier_keep_vm();
        {
          unsigned char* dst = p.out + ((size_t)blockIdx.x * 4 + (p.store_same ? 0 : ((tile - 1) & 3))) * 192 * 1024 + wid * 24 * 1024;
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 24 * 1024, 0x00020000);
          for (int i = 0; i < 24; ++i) {
            const u32x4 v = {__builtin_bit_cast(unsigned, acc[0][0]), (unsigned)i, (unsigned)g, (unsigned)lane};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16 + i * 1024, 0, 16);
          }
        }
        WAIT_VM(0);
        barrier_keep_vm();
      }
      if constexpr (RD || MM) {
        const unsigned char* wb = smem + (g & 1) * W_STAGE + wn * 96 * 128;
        const unsigned char* ab = smem + OFF_A + (g % 3) * A_STAGE + wm * 64 * 128;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {  // 4 fragment sets of (3 W + 2 A) reads, 9 MFMAs behind each: 36 per stage
          f16x8 wf[3], af[2];
          // the product's unit-major pieces: row l31 -> piece l31 / 8, unit u at position u ^ (piece & 1): conflict-free 16-byte reads
          const int l31 = lane & 31, u = (s4 * 2 + (lane >> 5)) & 7;
          const int off = (l31 >> 3) * 1024 + (((u ^ ((l31 >> 3) & 1)) * 8 + (l31 & 7)) << 4);
          if constexpr (RD) {
#pragma unroll
            for (int j = 0; j < 3; ++j) wf[j] = *reinterpret_cast<const f16x8*>(wb + j * 4096 + off);
#pragma unroll
            for (int j = 0; j < 2; ++j) af[j] = *reinterpret_cast<const f16x8*>(ab + j * 4096 + off);
          } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) wf[j] = f16x8{(_Float16)1, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 2; ++j) af[j] = f16x8{(_Float16)1, 0, 0, 0, 0, 0, 0, 0};
          }
          if constexpr (MM) {
#pragma unroll
            for (int rep = 0; rep < 3; ++rep)
#pragma unroll
              for (int j = 0; j < 3; ++j) {
                acc[j * 2 + (rep & 1)] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], af[rep & 1], acc[j * 2 + (rep & 1)], 0, 0, 0);
              }
          } else if constexpr (RD) {
            asm volatile("" ::"v"(wf[0]), "v"(wf[1]), "v"(wf[2]), "v"(af[0]), "v"(af[1]));
          }
        }
      }
      auto store_some = [&](int first, int count, int tl_) {
        unsigned char* dst = p.out + ((size_t)blockIdx.x * 4 + (p.store_same ? 0 : (tl_ & 3))) * 192 * 1024 + wid * 24 * 1024;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 24 * 1024, 0x00020000);
        for (int i = first; i < first + count; ++i) {
          const u32x4 v = {__builtin_bit_cast(unsigned, acc[0][0]), (unsigned)i, (unsigned)g, (unsigned)lane};
          switch (p.store_aux) {
            case 0: __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16 + i * 1024, 0, 0); break;
            case 2: __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16 + i * 1024, 0, 2); break;
            case 17: __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16 + i * 1024, 0, 17); break;
            default: __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16 + i * 1024, 0, 16); break;
          }
        }
      };
      if constexpr (ST) {
        if (p.store_spread && tile > 0) store_some(kt * 24 / p.nk, 24 / p.nk, tile - 1);  // two per stage (nk = 12)
      }
      if (++kt == p.nk) {
        kt = 0;
        if constexpr (ST) {  // the tile's output: 192 KiB per workgroup = 24 KiB per consumer wave = 24 x 16-byte stores per lane
          if (!p.store_spread && !p.store_excl) store_some(0, 24, tile);
        }
        ++tile;
      }
    }
    if (acc[0][0] == 12345.f && p.stamps) p.stamps[100] = 1;
  }
  if (blockIdx.x == 0 && tid == 512 && p.stamps) p.stamps[0] = __builtin_amdgcn_s_memtime() - t0;  // (a loader: the last to finish)
}

template <int FLAGS>
static double run(const char* tag, Args a, int n_cu, double ghz) {
  auto k = probe<FLAGS>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  unsigned long long cyc = 0;
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(n_cu), dim3(640), SMEM, 0, a);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) {
      best = ms;
      CHECK(hipMemcpy(&cyc, a.stamps, 8, hipMemcpyDeviceToHost));
    }
  }
  const double per_stage_wall = best * 1e-3 * ghz * 1e9 / a.stages;
  printf("%-58s %8.3f ms  %7.0f cyc/stage (s_memtime, WG 0)  %7.0f (wall x %.2f GHz)  %5.1f B/clk/CU\n", tag, best,
         (double)cyc / a.stages, per_stage_wall, ghz, 65536.0 / ((double)cyc / a.stages));
  return (double)cyc / a.stages;
}

int main(int argc, char** argv) {
  hipDeviceProp_t pr;
  CHECK(hipGetDeviceProperties(&pr, 0));
  const int n_cu = pr.multiProcessorCount / 8 * 8;
  const double ghz = argc > 1 ? atof(argv[1]) : 1.9;
  printf("%s: %d CUs (grid %d), B/clk and wall cycles use %.2f GHz; a stage = 48 KiB W + 16 KiB A = 65536 B\n", pr.gcnArchName,
         pr.multiProcessorCount, n_cu, ghz);
  Args a;
  a.nk = 12;
  a.stages = 12 * 24;
  a.n_wtiles = 3;
  a.a_share = 3;
  a.a_stride = (size_t)(a.stages + 4) * A_STAGE;
  a.delay_cycles = 0; a.delay_mod = 1; a.store_aux = 16; a.store_spread = 0; a.store_same = 0; a.store_excl = 0;
  unsigned char *w, *act, *out;
  CHECK(hipMalloc(&w, (size_t)3 * 12 * W_STAGE));
  CHECK(hipMalloc(&act, a.a_stride * n_cu));
  CHECK(hipMalloc(&out, (size_t)n_cu * 4 * 192 * 1024));
  CHECK(hipMalloc(&a.stamps, 1024));
  CHECK(hipMemset(w, 1, (size_t)3 * 12 * W_STAGE));
  CHECK(hipMemset(act, 1, a.a_stride * n_cu));
  CHECK(hipMemset(a.stamps, 0, 1024));
  a.w = w; a.a = act; a.out = out;
  printf("-- every stage re-reads the SAME 48 KiB of W (what dma_probe measured)\n");
  a.hot_small = 1;
  run<0>("free-running loaders", a, n_cu, ghz);
  run<1>("+ barrier per stage (8 consumer waves wait)", a, n_cu, ghz);
  printf("-- W = three 576 KiB tiles walked stage by stage (the product's stream), A panels shared by 3 workgroups\n");
  a.hot_small = 0;
  run<0>("free-running loaders", a, n_cu, ghz);
  run<1>("+ barrier per stage (8 consumer waves wait)", a, n_cu, ghz);
  run<3>("+ barrier + consumers' fragment reads (160 KiB LDS / stage)", a, n_cu, ghz);
  run<5>("+ barrier + 36 MFMAs per consumer wave and stage", a, n_cu, ghz);
  run<7>("+ barrier + fragment reads + MFMAs", a, n_cu, ghz);
  run<9>("+ barrier + 192 KiB of stores per 12 stages", a, n_cu, ghz);
  run<15>("+ barrier + reads + MFMAs + stores (the whole k-loop)", a, n_cu, ghz);
  printf("-- the store burst, varied (barrier + reads + MFMAs + stores)\n");
  a.store_aux = 0;  run<15>("plain stores", a, n_cu, ghz);
  a.store_aux = 2;  run<15>("nt stores", a, n_cu, ghz);
  a.store_aux = 17; run<15>("sc0 sc1 stores", a, n_cu, ghz);
  a.store_aux = 16;
  a.store_same = 1; run<15>("sc1 stores, every tile to the same 192 KiB (48 MiB chip-wide)", a, n_cu, ghz);
  a.store_aux = 0;  run<15>("plain stores, every tile to the same 192 KiB", a, n_cu, ghz);
  a.store_aux = 16; a.store_same = 0;
  a.store_spread = 1; run<15>("sc1 stores spread over the next tile's stages (2 per lane and stage)", a, n_cu, ghz);
  a.store_spread = 0;
  a.delay_cycles = 41000 / 32; a.delay_mod = 32;
  run<15>("burst stores, workgroups of an XCD de-phased over one tile period", a, n_cu, ghz);
  a.delay_cycles = 0; a.delay_mod = 1;
  a.store_excl = 1;
  run<15>("EXCLUSIVE store burst (no load in flight while the stores are), lockstep", a, n_cu, ghz);
  a.delay_cycles = 30000 / 32; a.delay_mod = 32;
  run<15>("EXCLUSIVE store burst, workgroups of an XCD de-phased over one tile period", a, n_cu, ghz);
  a.delay_cycles = 30000 / 8; a.delay_mod = 8;
  run<15>("EXCLUSIVE store burst, 8 phases", a, n_cu, ghz);
  a.delay_cycles = 0; a.delay_mod = 1; a.store_excl = 0;
  printf("-- which costs more: every workgroup on ONE weight tile (the N = 384 GEMMs) or private A panels (3x the HBM reads)?\n");
  a.n_wtiles = 1; a.a_share = 1;
  run<1>("+ barrier: 1 weight tile, private A (attn-out, FFN-down, head)", a, n_cu, ghz);
  a.n_wtiles = 3; a.a_share = 1;
  run<1>("+ barrier: 3 weight tiles, private A", a, n_cu, ghz);
  a.n_wtiles = 1; a.a_share = 3;
  run<1>("+ barrier: 1 weight tile, A shared by 3", a, n_cu, ghz);
  a.n_wtiles = 1; a.a_share = 1;
  a.delay_cycles = 450; a.delay_mod = 4;
  run<1>("+ barrier: 1 weight tile, private A, workgroups staggered by 0 / 450 / 900 / 1350 cycles", a, n_cu, ghz);
  a.delay_cycles = 0; a.delay_mod = 1;
  a.n_wtiles = 3; a.a_share = 3;
  printf("-- K = 768 (24 stages per tile, 1.15 MiB weight tile)\n");
  a.nk = 24;
  CHECK(hipFree(w));
  CHECK(hipMalloc(&w, (size_t)3 * 24 * W_STAGE));
  CHECK(hipMemset(w, 1, (size_t)3 * 24 * W_STAGE));
  a.w = w;
  run<1>("+ barrier per stage", a, n_cu, ghz);
  return 0;
}
