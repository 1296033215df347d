// Probe (round 4, end of round): BertIntermediate + BertOutput (FFN-up, GELU, FFN-down) in ONE kernel with the layer input stationary
// in registers, the 768-wide intermediate never leaving the chip.  One workgroup of four 512-register waves per 128-token sequence;
// wave w owns token rows 32 w .. 32 w + 31:
//   * a (K = 384, hi + lo operand fragments): 192 registers, stationary (also the residual of the final LayerNorm);
//   * the 384 output columns of the down projection: twelve 32 x 32 accumulators = 192 registers (AGPRs);
//   * the intermediate in chunks of 64 columns: two accumulators (32), GELU + hi/lo split + quad -> operand exchange -> 16 registers
//     of B operand for the chunk's two k-tiles of the down projection;
//   * weights through LDS, copied by the compute waves: W1 chunk = 12 k-tiles x 8 KiB, W2 chunk = 2 k-tiles x 48 KiB.
// Arithmetic of the epilogues is a stand-in (scale, erf-free "gelu", split); the checksum keeps everything alive.  Answers: does it
// fit 512 registers without spills (compile), and what fraction of the matrix pipe does one wave per SIMD reach (run).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 ffn_fused_probe.hip -o ffn_fused_probe.bin && ./ffn_fused_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char* lds_ptr_t;

constexpr int NKT = 12;                  // K = 384
constexpr int NCH = 12;                  // 768 / 64 chunks
constexpr int S1 = 64 * 128;             // W1 stage: one k-tile of a chunk's 64 rows, unit-major [8 units][64 rows][16 B]: 8 KiB
constexpr int S2 = 384 * 128;            // W2 stage: one k-tile (32 k) of all 384 output rows, unit-major: 48 KiB
constexpr int N1 = 4, N2 = 2;            // ring depths
constexpr int OFF2 = N1 * S1;
constexpr int SMEM = OFF2 + N2 * S2;     // 32 + 96 = 128 KiB

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, lds_ptr_t dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void swap32(unsigned& a, unsigned& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}

__global__ __launch_bounds__(256) void ffn_fused_kernel(const unsigned char* W1, const unsigned char* W2, const unsigned char* Aimg, float* out, int nseq) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wq = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  float check = 0.f;
  for (int seq = blockIdx.x; seq < nseq; seq += gridDim.x) {
    f16x8 ah[NKT][2], al[NKT][2];
    const unsigned char* ab = Aimg + ((size_t)seq * 4 + wq) * NKT * 4096 + l31 * 16;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        ah[kt][c] = *reinterpret_cast<const f16x8*>(ab + kt * 4096 + (2 * c + half) * 512);
        al[kt][c] = *reinterpret_cast<const f16x8*>(ab + kt * 4096 + (4 + 2 * c + half) * 512);
      }
    f32x16 oacc[12];
#pragma unroll
    for (int j = 0; j < 12; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[j][r] = 0.f;
    // stream 1: W1 stages (chunk, k-tile), 8 pieces each = 2 per wave; stream 2: W2 stages (chunk, k-tile 0 / 1), 48 pieces = 12 per wave
    auto issue1 = [&](int pos) {
      if (pos >= NCH * NKT) return;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(W1) + (size_t)pos * S1, 0, S1, 0x00020000);
#pragma unroll
      for (int k = 0; k < 2; ++k) dma16(rs, (lds_ptr_t)(smem) + (pos % N1) * S1 + (wq + 4 * k) * 1024, lane * 16, (wq + 4 * k) * 1024);
    };
    auto issue2 = [&](int pos) {
      if (pos >= NCH * 2) return;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(W2) + (size_t)pos * S2, 0, S2, 0x00020000);
#pragma unroll
      for (int k = 0; k < 12; ++k) dma16(rs, (lds_ptr_t)(smem) + OFF2 + (pos % N2) * S2 + (wq + 4 * k) * 1024, lane * 16, (wq + 4 * k) * 1024);
    };
    issue1(0);
    issue1(1);
    issue1(2);
    issue2(0);
    issue2(1);
    for (int ch = 0; ch < NCH; ++ch) {
      // ---- up projection of the chunk: 64 columns = two accumulators
      f32x16 u[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) u[j][r] = 0.f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const int pos = ch * NKT + kt;
        __builtin_amdgcn_s_waitcnt(0x0F70 | 0);  // vmcnt(0) (probe: no counted waits -- the W2 stages are in flight too)
        __builtin_amdgcn_s_barrier();
        issue1(pos + 3);
        const unsigned char* st = smem + (pos % N1) * S1 + l31 * 16;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const f16x8 wh = *reinterpret_cast<const f16x8*>(st + (2 * c + half) * 1024 + j * 512);
            const f16x8 wl = *reinterpret_cast<const f16x8*>(st + (4 + 2 * c + half) * 1024 + j * 512);
            u[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, ah[kt][c], u[j], 0, 0, 0);
            u[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, al[kt][c], u[j], 0, 0, 0);
            u[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, ah[kt][c], u[j], 0, 0, 0);
          }
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          if (g < 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
      }
      // ---- "GELU", hi / lo split, quad -> operand exchange: the chunk becomes the B operand of two k-tiles of the down projection
      f16x8 gh[2][2], gl[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        unsigned H[8], L[8];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int dd = 0; dd < 2; ++dd) {
            float x0 = u[j][4 * q + 2 * dd] * 0.125f, x1 = u[j][4 * q + 2 * dd + 1] * 0.125f;
            x0 = x0 * (0.5f + 0.25f * x0 * __builtin_amdgcn_rcpf(1.0f + x0 * x0));
            x1 = x1 * (0.5f + 0.25f * x1 * __builtin_amdgcn_rcpf(1.0f + x1 * x1));
            const f32x2 xv = {x0, x1};
            const f16x2 hv = __builtin_convertvector(xv, f16x2);
            const f32x2 rv = {x0 - (float)hv[0], x1 - (float)hv[1]};
            const f16x2 lv = __builtin_convertvector(rv, f16x2);
            H[2 * q + dd] = __builtin_bit_cast(unsigned, hv);
            L[2 * q + dd] = __builtin_bit_cast(unsigned, lv);
          }
        // half 0 wants quads 0 and 2 complete (k16 steps c = 0, 1 of its 8-value slice), half 1 quads 1 and 3
        swap32(H[2], H[0]);
        swap32(H[3], H[1]);
        swap32(H[6], H[4]);
        swap32(H[7], H[5]);
        swap32(L[2], L[0]);
        swap32(L[3], L[1]);
        swap32(L[6], L[4]);
        swap32(L[7], L[5]);
        gh[j][0] = __builtin_bit_cast(f16x8, u32x4{H[0], H[1], H[2], H[3]});
        gh[j][1] = __builtin_bit_cast(f16x8, u32x4{H[4], H[5], H[6], H[7]});
        gl[j][0] = __builtin_bit_cast(f16x8, u32x4{L[0], L[1], L[2], L[3]});
        gl[j][1] = __builtin_bit_cast(f16x8, u32x4{L[4], L[5], L[6], L[7]});
      }
      // ---- down projection, the chunk's two k-tiles: all twelve accumulators
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int pos = ch * 2 + j;
        __builtin_amdgcn_s_waitcnt(0x0F70 | 0);
        __builtin_amdgcn_s_barrier();
        const unsigned char* st = smem + OFF2 + (pos % N2) * S2 + l31 * 16;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int t = 0; t < 12; ++t) {
            const f16x8 wh = *reinterpret_cast<const f16x8*>(st + (2 * c + half) * 6144 + t * 512);
            const f16x8 wl = *reinterpret_cast<const f16x8*>(st + (4 + 2 * c + half) * 6144 + t * 512);
            oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, gh[j][c], oacc[t], 0, 0, 0);
            oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, gl[j][c], oacc[t], 0, 0, 0);
            oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, gh[j][c], oacc[t], 0, 0, 0);
          }
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int g = 0; g < 24; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          if (g < 22) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_s_barrier();   // every wave is done with the stage before it is overwritten
        issue2(pos + 2);
      }
    }
    // ---- stand-in for bias + residual + LayerNorm: row sums in-lane, fold
#pragma unroll
    for (int t = 0; t < 12; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) check += oacc[t][r];
    check += (float)ah[0][0][0] + (float)al[11][1][7];
  }
  if (out) out[blockIdx.x * 256 + threadIdx.x] = check;
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main() {
  hipDeviceProp_t pr;
  CHECK(hipGetDeviceProperties(&pr, 0));
  const int nseq = 512, grid = pr.multiProcessorCount;
  unsigned char *W1, *W2, *A;
  float* out;
  const size_t w1b = (size_t)NCH * NKT * S1, w2b = (size_t)NCH * 2 * S2, ab = (size_t)nseq * 4 * NKT * 4096;
  CHECK(hipMalloc(&W1, w1b));
  CHECK(hipMalloc(&W2, w2b));
  CHECK(hipMalloc(&A, ab));
  CHECK(hipMalloc(&out, grid * 256 * 4));
  unsigned s = 12345u;
  auto fill = [&](unsigned char* d, size_t bytes, unsigned short base) {
    std::vector<unsigned short> h(bytes / 2);
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = base + ((s >> 10) & 0x3FF); }
    CHECK(hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice));
  };
  fill(W1, w1b, 0x2C00);
  fill(W2, w2b, 0x2C00);
  fill(A, ab, 0x3800);
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const double mfma = (double)nseq * 4 * NCH * (NKT * 12 + 2 * 72);
  printf("%s: FFN-up + GELU + FFN-down of %d sequences in one kernel, layer input stationary; the matrix pipe alone: %.1f us at 2.1 GHz\n",
         pr.gcnArchName, nseq, mfma / (grid * 4) * 32 / 2.1e3);
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(ffn_fused_kernel, dim3(grid), dim3(256), SMEM, 0, W1, W2, A, out, nseq);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %.1f us\n", ms * 1e3);
  }
  return 0;
}
