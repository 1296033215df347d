// Micro-probe 2b (round 6): coissue2_probe.hip with v_mfma_f32_16x16x32_f16 (8 passes, 16 cycles nominal): where does VALU work hide beside it on one SIMD?
//   A  one wave per SIMD:  6 x [MFMA + N fillers] per iteration, N = 0..12              -> cycles per iteration (192 = hidden)
//   B  two waves per SIMD, both run A's stream                                          -> per-wave cycles (192 = hidden)
//   C  cross-wave: waves 0-3 run 6 x [MFMA + s_nop pad P], waves 4-7 a pure VALU loop sized to the same time alone
//   D  cross-wave with the roles on the YOUNGER waves swapped (waves 4-7 MFMA, 0-3 VALU)
//   E  cross-wave, MFMA stream back to back, VALU kinds: fma, mul, add, mov, cvt_pk_f16, fma_mix, rcp
// Every instruction is an asm volatile statement, so the issue order is the program order.
//   hipcc -O3 --offload-arch=gfx950 coissue2_probe.hip -o coissue2_probe && ./coissue2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define MFMA(j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b))
// filler kinds
#define F_FMA(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[r]) : "v"(c1), "v"(c2))
#define F_MUL(r) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[r]) : "v"(c1))
#define F_ADD(r) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[r]) : "v"(c2))
#define F_MOV(r) asm volatile("v_mov_b32 %0, %1" : "+v"(x[r]) : "v"(c2))
#define F_CVT(r) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x[r]) : "v"(c2))
#define F_MIX(r) asm volatile("v_fma_mix_f32 %0, %0, %1, %2" : "+v"(x[r]) : "v"(c1), "v"(c2))
#define F_RCP(r) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[r]))
#define F_PKF(r) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[(r) >> 1]) : "v"(d1), "v"(d2))
#define F_SWP(r) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x[r]), "+v"(x[((r) + 8) & 15]))

template <int KIND>
__device__ __forceinline__ void filler(float (&x)[16], double (&y)[8], float c1, float c2, double d1, double d2, int r) {
  if constexpr (KIND == 0) F_FMA(r);
  else if constexpr (KIND == 1) F_MUL(r);
  else if constexpr (KIND == 2) F_ADD(r);
  else if constexpr (KIND == 3) F_MOV(r);
  else if constexpr (KIND == 4) F_CVT(r);
  else if constexpr (KIND == 5) F_MIX(r);
  else if constexpr (KIND == 6) F_RCP(r);
  else if constexpr (KIND == 7) F_PKF(r);
  else F_SWP(r);
}

// role 1: 6 x [MFMA + N fillers of KIND + PAD s_nop states]; role 2: 16 fillers of KIND per iteration; role 0: idle
template <int N, int KIND, int PAD>
__global__ __launch_bounds__(512) void probe(int role_old, int role_young, int it_mfma, int it_valu, long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int role = wid < 4 ? role_old : role_young;
  float x[16];
  double y[8];
  for (int r = 0; r < 16; ++r) x[r] = lane * 0.001f + r;
  for (int r = 0; r < 8; ++r) y[r] = (double)lane + r;
  const float c1 = 1.0001f + lane * 1e-9f, c2 = 0.5f + lane * 1e-9f;
  const double d1 = __builtin_bit_cast(double, make_float2(c1, c1)), d2 = __builtin_bit_cast(double, make_float2(c2, c2));
  f32x16 acc[6];
  for (int j = 0; j < 6; ++j) for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
  f16x8 a, b;
  for (int r = 0; r < 8; ++r) { a[r] = (_Float16)(lane * 0.01f + r); b[r] = (_Float16)(r * 0.5f); }
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime(), t1;
  if (role == 1) {
    for (int it = 0; it < it_mfma; ++it) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        MFMA(j);
#pragma unroll
        for (int f = 0; f < N; ++f) filler<KIND>(x, y, c1, c2, d1, d2, (j * N + f) & 15);
        if constexpr (PAD > 0) {
#pragma unroll
          for (int q = 0; q < PAD; ++q) asm volatile("s_nop 0");
        }
      }
    }
    asm volatile("s_nop 15\n\ts_nop 15");
  } else if (role == 2) {
    for (int it = 0; it < it_valu; ++it) {
#pragma unroll
      for (int r = 0; r < 16; ++r) filler<KIND>(x, y, c1, c2, d1, d2, r);
    }
  }
  t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 6; ++j) s += acc[j][0] + acc[j][3];
  for (int r = 0; r < 16; ++r) s += x[r];
  for (int r = 0; r < 8; ++r) s += (float)y[r];
  if (s == 12345.678f) sink[0] = s;
  if (lane == 0 && blockIdx.x == 0) out[wid] = t1 - t0;
}

static long long* g_out; static float* g_sink;
template <int N, int KIND, int PAD>
static void run(int threads, int ro, int ry, int im, int iv, double* old_cyc, double* young_cyc) {
  long long h[8] = {0};
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<N, KIND, PAD>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  hipLaunchKernelGGL((probe<N, KIND, PAD>), dim3(256), dim3(threads), 100 * 1024, 0, ro, ry, im, iv, g_out, g_sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(h, g_out, sizeof(h), hipMemcpyDeviceToHost));
  *old_cyc = (double)h[0];
  *young_cyc = (double)h[4];
}

static const char* kname[] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_mov_b32", "v_cvt_pk_f16_f32", "v_fma_mix_f32", "v_rcp_f32", "v_pk_fma_f32", "v_permlane32_swap"};

template <int N, int KIND>
static void caseAB() {
  const int it = 2000;
  double o, y;
  run<N, KIND, 0>(256, 1, 0, it, 0, &o, &y);
  const double a = o / it;
  run<N, KIND, 0>(512, 1, 1, it, 0, &o, &y);
  printf("A/B %-18s N=%2d fillers per MFMA: one wave/SIMD %6.1f cyc/iter (6 MFMA of 16 cycles: 96 = hidden) | two waves/SIMD old %6.1f young %6.1f (192 = hidden)\n",
         kname[KIND], N, a, o / it, y / it);
}

template <int KIND, int PAD>
static void caseC() {
  const int im = 2000;
  double o, y, m_alone, v_alone;
  run<0, KIND, PAD>(512, 1, 0, im, 0, &o, &y); m_alone = o;
  run<0, KIND, PAD>(512, 0, 2, 0, 1000, &o, &y); v_alone = y / 1000;
  const int iv = (int)(m_alone / v_alone);  // the VALU wave alone takes as long as the MFMA wave alone
  run<0, KIND, PAD>(512, 0, 2, 0, iv, &o, &y); const double v_al = y;
  run<0, KIND, PAD>(512, 1, 2, im, iv, &o, &y);
  printf("C   %-18s pad %2d: MFMA(old) alone %8.0f  VALU(young) alone %8.0f | together MFMA %8.0f VALU %8.0f  (VALU per 16: %5.1f cyc alone)\n",
         kname[KIND], PAD, m_alone, v_al, o, y, v_alone);
  run<0, KIND, PAD>(512, 2, 1, im, iv, &o, &y);
  printf("D   %-18s pad %2d: roles swapped (MFMA on the younger waves)             | together MFMA %8.0f VALU %8.0f\n", kname[KIND], PAD, y, o);
}

int main() {
  CHECK(hipMalloc(&g_out, 64)); CHECK(hipMalloc(&g_sink, 64));
  caseAB<0, 0>(); caseAB<1, 0>(); caseAB<2, 0>(); caseAB<3, 0>(); caseAB<4, 0>(); caseAB<6, 0>(); caseAB<8, 0>();
  caseAB<2, 6>(); caseAB<2, 5>(); caseAB<2, 4>(); caseAB<1, 7>(); caseAB<2, 7>();
  caseC<0, 0>(); caseC<0, 2>(); caseC<0, 4>(); caseC<6, 0>(); caseC<5, 0>(); caseC<4, 0>(); caseC<7, 0>();
  return 0;
}
