// Compile / run probe for the lead of DESIGN.md section 8 item 2 (round 4, written when the round's GPU minutes were spent: compiled and
// inspected here, NOT yet run): the projection half of a fused q|k|v + attention kernel with the hidden state STATIONARY in registers.
//   * a workgroup = one 128-token sequence, four waves of one SIMD each (512 registers), wave w owns token rows 32 w .. 32 w + 31;
//   * the wave's rows of h (K = 384, hi + lo fragments) stay in 192 VGPRs as MFMA B operands for the whole kernel;
//   * per head: the 96 weight rows (q_h | k_h | v_h) stream through an LDS ring k-tile by k-tile (12 KiB each, copied by the compute
//     waves themselves), three 32 x 32 accumulators per wave, epilogue: scale + bias, hi/lo split, quad -> operand exchange; q_h stays
//     in registers, k_h and v_h go to LDS (here: all three are folded into a checksum so that nothing is optimised away).
// What it answers without a GPU: does hipcc hold 192 stationary operand registers + 48 accumulator registers + the epilogue in a
// 512-register wave without spilling or routing tiles through AGPR copies?  (hipcc -O3 --offload-arch=gfx950 -S: see the numbers
// in DESIGN.md.)  What it will answer on a GPU: matrix-pipe utilisation of one wave per SIMD that issues its own copies.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 seq_proj_probe.hip -o seq_proj_probe && ./seq_proj_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) unsigned char* lds_ptr_t;

constexpr int NKT = 12;                 // K = 384
constexpr int HEADS = 12;
constexpr int KT_BYTES = 96 * 128;      // one k-tile of a head's 96 weight rows: 12 KiB (hi | lo, 128 B per row)
constexpr int NST = 4;                  // ring depth (stages)

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, lds_ptr_t dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, voff, soff, 0, 0);
}

// W: [head][k-tile][8 units: hi 0-3, lo 4-7][96 rows][16 B], H: [sequence][4 row blocks][12 k-tiles][8 units][32 rows][16 B]
template <int KPS, bool SCHED, bool LATE = false>  // k-tiles per ring stage (= per barrier); SCHED: operand reads interleaved with the MFMAs by decree; LATE: the copies of stage + 3 are issued behind the stage's MFMAs instead of in front
__global__ __launch_bounds__(256) void seq_proj_kernel(const unsigned char* W, const unsigned char* Himg, float* out, int nseq) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE = KPS * KT_BYTES;
  const int lane = threadIdx.x & 63, wq = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  float check = 0.f;
  for (int seq = blockIdx.x; seq < nseq; seq += gridDim.x) {
    // the wave's rows of h as B-operand fragments: 12 k-tiles x 2 k16 steps x (hi, lo) x 4 registers = 192
    f16x8 hh[NKT][2], hl[NKT][2];
    const unsigned char* hb = Himg + ((size_t)seq * 4 + wq) * NKT * 4096 + l31 * 16;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        hh[kt][c] = *reinterpret_cast<const f16x8*>(hb + kt * 4096 + (2 * c + half) * 512);
        hl[kt][c] = *reinterpret_cast<const f16x8*>(hb + kt * 4096 + (4 + 2 * c + half) * 512);
      }
    const int total = HEADS * NKT / KPS;  // stream positions (head, stage)
    auto issue = [&](int pos) {           // the four waves copy a stage: 12 KPS pieces of 1 KiB, 3 KPS per wave
      if (pos >= total) return;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(W) + (size_t)pos * STAGE, 0, STAGE, 0x00020000);
#pragma unroll
      for (int k = 0; k < 3 * KPS; ++k) dma16(rs, (lds_ptr_t)(smem) + (pos % NST) * STAGE + (wq + 4 * k) * 1024, lane * 16, (wq + 4 * k) * 1024);
    };
    issue(0);
    issue(1);
    issue(2);
    for (int head = 0; head < HEADS; ++head) {
      f32x16 acc[3];
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
      for (int sg = 0; sg < NKT / KPS; ++sg) {
        const int pos = head * (NKT / KPS) + sg;
        __builtin_amdgcn_s_waitcnt(0x0F70 | ((6 * KPS) & 15) | (((6 * KPS) >> 4) << 14));  // vmcnt(6 KPS): this stage landed (two younger in flight)
        __builtin_amdgcn_s_barrier();
        if (!LATE) issue(pos + 3);
#pragma unroll
        for (int kk = 0; kk < KPS; ++kk) {
          const int kt = sg * KPS + kk;
          // stage layout UNIT-MAJOR: [unit 0-7][96 rows][16 B] -- a half-wave reads 32 consecutive 16-byte slots (row-major rows of
          // 128 B put every other row on the same banks: the first version of this probe measured 8-way conflicts, not the design)
          const unsigned char* st = smem + (pos % NST) * STAGE + kk * KT_BYTES + l31 * 16;
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              const f16x8 wh = *reinterpret_cast<const f16x8*>(st + (2 * c + half) * 1536 + j * 512);
              const f16x8 wl = *reinterpret_cast<const f16x8*>(st + (4 + 2 * c + half) * 1536 + j * 512);
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, hh[kt][c], acc[j], 0, 0, 0);
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, hl[kt][c], acc[j], 0, 0, 0);
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, hh[kt][c], acc[j], 0, 0, 0);
            }
        }
        if (LATE) issue(pos + 3);
        if constexpr (SCHED) {  // 12 KPS operand reads, 18 KPS MFMAs: four reads up front, then two reads behind every three MFMAs
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
          for (int g = 0; g < 6 * KPS; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            if (g < 6 * KPS - 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          }
        }
      }
      // epilogue stand-in: scale, split into fp16 hi / lo pairs (what the operand conversion costs), fold
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float a = acc[j][r] * 0.125f + 0.5f, b = acc[j][r + 1] * 0.125f + 0.5f;
          const _Float16 ah = (_Float16)a, bh = (_Float16)b;
          const _Float16 al = (_Float16)(a - (float)ah), bl = (_Float16)(b - (float)bh);
          check += (float)ah + (float)al + (float)bh + (float)bl;
        }
    }
  }
  if (out) out[blockIdx.x * 256 + threadIdx.x] = check;
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main() {
  hipDeviceProp_t pr;
  CHECK(hipGetDeviceProperties(&pr, 0));
  const int nseq = 512, grid = pr.multiProcessorCount;
  unsigned char *W, *H;
  float* out;
  const size_t wbytes = (size_t)HEADS * NKT * KT_BYTES, hbytes = (size_t)nseq * 4 * NKT * 4096;
  CHECK(hipMalloc(&W, wbytes));
  CHECK(hipMalloc(&H, hbytes));
  CHECK(hipMalloc(&out, grid * 256 * 4));
  std::vector<unsigned short> hw(wbytes / 2), hh(hbytes / 2);
  unsigned s = 12345u;
  for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = 0x2C00 + ((s >> 10) & 0x3FF); }   // fp16 in [2^-4, 2^-3)
  for (auto& v : hh) { s = s * 1664525u + 1013904223u; v = 0x3800 + ((s >> 10) & 0x3FF); }   // fp16 in [0.5, 1)
  CHECK(hipMemcpy(W, hw.data(), wbytes, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(H, hh.data(), hbytes, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const double mfma = (double)nseq * 4 * HEADS * NKT * 18;   // MFMAs issued
  printf("%s: q|k|v projection of %d sequences, h stationary; the matrix pipe alone: %.1f us at 2.1 GHz\n", pr.gcnArchName, nseq,
         mfma / (grid * 4) * 32 / 2.1e3);
  auto run = [&](auto kernel, int kps, int sched) {
    const int smem = NST * kps * KT_BYTES;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), smem, 0, W, H, out, nseq);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep && ms < best) best = ms;
    }
    printf("  k-tiles per barrier %d, dictated schedule %d: %.1f us\n", kps, sched, best * 1e3);
  };
  run(seq_proj_kernel<1, false>, 1, 0);
  run(seq_proj_kernel<1, true>, 1, 1);
  run(seq_proj_kernel<2, false>, 2, 0);
  run(seq_proj_kernel<2, true>, 2, 1);
  run(seq_proj_kernel<3, true>, 3, 1);
  printf("  copies issued behind the stage's MFMAs:\n");
  run(seq_proj_kernel<1, true, true>, 1, 1);
  run(seq_proj_kernel<2, true, true>, 2, 1);
  return 0;
}
