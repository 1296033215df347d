// Micro-probe: do VALU instructions of one wave overlap with the MFMAs of ANOTHER wave on the same SIMD?
// One workgroup of 8 waves per CU (wave w -> SIMD w % 4): waves 0-3 run a chain-free MFMA loop, waves 4-7 a VALU loop.
// Each role is timed (s_memtime of lane 0) alone and together.  sum-like "together" times = no overlap.
//   hipcc -O3 --offload-arch=gfx950 coissue_probe.hip -o coissue_probe && ./coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// mode bit 0: MFMA waves active, bit 1: VALU waves active.  kind: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_cvt_pk_f16_f32 + v_cvt_f32_f16,
// 3 v_permlane32_swap, 4 ds_read_b128 (LDS pipe)
template <int KIND, int PRIO>
__global__ __launch_bounds__(512) void probe(int mode, int iters, long long* out, float* sink) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = (float)i;
  __syncthreads();
  long long t0 = 0, t1 = 0;
  if (wid < 4) {
    if (mode & 1) {
      if (PRIO == 1) __builtin_amdgcn_s_setprio(3);
      f32x16 acc[6];
      for (int j = 0; j < 6; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      f16x8 a, b;
      for (int r = 0; r < 8; ++r) { a[r] = (_Float16)(lane * 0.01f + r); b[r] = (_Float16)(r * 0.5f); }
      t0 = (long long)__builtin_amdgcn_s_memtime();
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
      }
      float s = 0.f;
      for (int j = 0; j < 6; ++j) s += acc[j][0] + acc[j][15];
      t1 = (long long)__builtin_amdgcn_s_memtime();
      if (s == 12345.f) sink[0] = s;
    }
  } else {
    if (mode & 2) {
      if (PRIO == 2) __builtin_amdgcn_s_setprio(3);
      float x[16];
      for (int r = 0; r < 16; ++r) x[r] = lane * 0.001f + r;
      t0 = (long long)__builtin_amdgcn_s_memtime();
      for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) x[r] = __builtin_fmaf(x[r], 1.0001f, 0.5f);
        } else if constexpr (KIND == 1) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            f32x2 v = {x[r], x[r + 1]};
            v = __builtin_elementwise_fma(v, f32x2{1.0001f, 1.0001f}, f32x2{0.5f, 0.5f});
            x[r] = v[0]; x[r + 1] = v[1];
          }
        } else if constexpr (KIND == 2) {
#pragma unroll
          for (int r = 0; r < 16; ++r) x[r] = (float)(_Float16)x[r] + 1.0f;
        } else if constexpr (KIND == 3) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            unsigned u = __builtin_bit_cast(unsigned, x[r]), v = __builtin_bit_cast(unsigned, x[r + 1]);
            const auto q = __builtin_amdgcn_permlane32_swap(u, v, false, false);
            x[r] = __builtin_bit_cast(float, q[0]); x[r + 1] = __builtin_bit_cast(float, q[1]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; r += 4) {
            const float4 v = *reinterpret_cast<const float4*>(&lds[((lane + it + r) & 1023) * 4]);
            x[r] += v.x; x[r + 1] += v.y; x[r + 2] += v.z; x[r + 3] += v.w;
          }
        }
      }
      float s = 0.f;
      for (int r = 0; r < 16; ++r) s += x[r];
      t1 = (long long)__builtin_amdgcn_s_memtime();
      if (s == 12345.f) sink[1] = s;
    }
  }
  if (lane == 0 && blockIdx.x == 0) out[wid] = t1 - t0;
}

template <int KIND, int PRIO>
static void run(const char* name, long long* out, float* sink) {
  const int iters = 2000;
  long long h[3][8];
  for (int mode = 1; mode <= 3; ++mode) {
    hipLaunchKernelGGL((probe<KIND, PRIO>), dim3(256), dim3(512), 0, 0, mode, iters, out, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h[mode - 1], out, sizeof(long long) * 8, hipMemcpyDeviceToHost));
  }
  printf("%-28s prio=%d  per iteration (6 MFMA 32x32x16 | 16 VALU-class ops):  MFMA alone %6.1f  VALU alone %6.1f  together: MFMA %6.1f  VALU %6.1f\n",
         name, PRIO, h[0][0] / (double)iters, h[1][4] / (double)iters, h[2][0] / (double)iters, h[2][4] / (double)iters);
}

int main() {
  long long* out; float* sink;
  CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&sink, 64));
  // prio 0: default; 1: the MFMA waves run at s_setprio 3; 2: the VALU waves run at s_setprio 3
  run<0, 0>("v_fma_f32", out, sink);        run<0, 1>("v_fma_f32", out, sink);        run<0, 2>("v_fma_f32", out, sink);
  run<1, 0>("v_pk_fma_f32", out, sink);     run<1, 2>("v_pk_fma_f32", out, sink);
  run<2, 0>("v_cvt f16<->f32 + add", out, sink); run<2, 2>("v_cvt f16<->f32 + add", out, sink);
  run<3, 0>("v_permlane32_swap", out, sink); run<3, 2>("v_permlane32_swap", out, sink);
  run<4, 0>("ds_read_b128 + 4 add", out, sink); run<4, 2>("ds_read_b128 + 4 add", out, sink);
  return 0;
}
