// Token GEMMs of the BertForDiffusion forward on the CDNA4 matrix cores, exact fp32.
//
//   C[M,N] = A[M,K] * W[N,K]^T + bias[N]   (+ GELU | + residual | + residual + LayerNorm)
//
// replaces torch.nn.Linear inside HF BertSelfAttention.{query,key,value},
// BertSelfOutput.dense, BertIntermediate.dense, BertOutput.dense (transformers
// 4.11.3, called from foldingdiff/modelling.py:473-480) and
// AnglesPredictor.dense1 (modelling.py:203-205).  nn.Linear stores W as
// [out, in] = [N, K], i.e. both operands are K-contiguous ("B^T" form).
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- fp32 products, fp32 accumulate, bitwise
// an fmaf chain (MI355X guide, section 3).  Peak 157.3 TFLOP/s.
//
// Tiling (wave = 64 lanes, 4 waves / workgroup):
//   * generic:   128 x 128 block, waves 2 x 2, each 64 x 64 (2 x 2 MFMA tiles), BK = 32
//   * LN-fused:  128 x N   block (N = 384 | 192 = whole rows), waves 4 x 1, each
//                32 x N (1 x N/32 MFMA tiles), BK = 16; LayerNorm reduces inside the wave.
//   LDS rows are padded by 4 floats so the ds_read_b128 operand fetches are
//   conflict free (row stride 36 / 20 dwords -> 16 distinct 16-byte slots).
//   One MFMA consumes k = {k0+s, k0+4+s} (half-wave 0 / 1): each lane feeds four
//   consecutive MFMAs from ONE 16-byte LDS read; the k permutation is applied to
//   A and W alike so every product a[m,k]*w[n,k] is formed exactly once.
#include "fdmi_kernels.h"

namespace fdmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { EPI_LN_INTERNAL = 3 };

struct GemmArgs {
  const float* A;
  const float* W;
  const float* bias;
  const float* resid;
  const float* gamma;
  const float* beta;
  float* C;
  int M, N, K;
  float eps;
};

// Bijective XCD-aware remap (guide T1): workgroup b runs on XCD b % 8; give each
// XCD a contiguous run of tiles so the n-tiles sharing one A row panel hit one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int WAVES_M, int WAVES_N, int MT, int NT, int BK, int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  constexpr int BM = WAVES_M * MT * 32;
  constexpr int BN = WAVES_N * NT * 32;
  constexpr int LDK = BK + 4;
  constexpr int QPR = BK / 4;               // float4 per tile row
  constexpr int A_IT = BM * QPR / 256;      // float4 per thread, A tile
  constexpr int W_IT = BN * QPR / 256;
  static_assert(BM * QPR % 256 == 0 && BN * QPR % 256 == 0, "tile/threads");
  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDK];
  float* As = smem;
  float* Ws = smem + BM * LDK;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  const int K = p.K;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Unconditional loads (indices clamped, not predicated): rows >= M / >= N accumulate copies
  // of the last valid row and are never stored; a load inside a branch would defeat hipcc's
  // s_waitcnt vmcnt accounting.
  f32x4 ra[A_IT], rw[W_IT];  // ext_vector registers (HIP's float4 struct copies can end up in scratch)
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int idx = tid + 256 * i, row = idx / QPR, c4 = idx % QPR;
      const int gm = m0 + row < p.M ? m0 + row : p.M - 1;
      ra[i] = *reinterpret_cast<const f32x4*>(p.A + (size_t)gm * K + k0 + c4 * 4);
    }
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int idx = tid + 256 * i, row = idx / QPR, c4 = idx % QPR;
      const int gn = n0 + row < p.N ? n0 + row : p.N - 1;
      rw[i] = *reinterpret_cast<const f32x4*>(p.W + (size_t)gn * K + k0 + c4 * 4);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int idx = tid + 256 * i, row = idx / QPR, c4 = idx % QPR;
      *reinterpret_cast<f32x4*>(&As[row * LDK + c4 * 4]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int idx = tid + 256 * i, row = idx / QPR, c4 = idx % QPR;
      *reinterpret_cast<f32x4*>(&Ws[row * LDK + c4 * 4]) = rw[i];
    }
  };

  gload(0);
  lstore();
  __syncthreads();
  const int nk = K / BK;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload((kt + 1) * BK);  // in flight under the MFMAs below
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      f32x4 a[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i)
        a[i] = *reinterpret_cast<const f32x4*>(&As[((wm * MT + i) * 32 + l31) * LDK + kk * 8 + half * 4]);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(&Ws[((wn * NT + j) * 32 + l31) * LDK + kk * 8 + half * 4]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[s], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
    if (kt + 1 < nk) {
      lstore();
      __syncthreads();
    }
  }

  // ---- epilogue.  C/D layout of the 32x32 tile: col = lane & 31,
  //      row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
  if constexpr (EPI == EPI_LN_INTERNAL) {
    static_assert(WAVES_N == 1, "LayerNorm epilogue needs whole rows in one wave");
    const float inv_n = 1.0f / (float)BN;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int rbase = m0 + (wm * MT + i) * 32 + 4 * half;
      float sum[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) sum[r] = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + j * 32 + l31;
        const float bz = p.bias[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          float v = acc[i][j][r] + bz;
          if (row < p.M) v += p.resid[(size_t)row * p.N + col];
          acc[i][j][r] = v;
          sum[r] += v;
        }
      }
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum[r] += __shfl_xor(sum[r], off);
      float mean[16], var[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        mean[r] = sum[r] * inv_n;
        var[r] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float dlt = acc[i][j][r] - mean[r];
          var[r] += dlt * dlt;
        }
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1)
#pragma unroll
        for (int r = 0; r < 16; ++r) var[r] += __shfl_xor(var[r], off);
#pragma unroll
      for (int r = 0; r < 16; ++r) var[r] = 1.0f / sqrtf(var[r] * inv_n + p.eps);  // rstd
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + j * 32 + l31;
        const float gm = p.gamma[col], bt = p.beta[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          if (row < p.M) p.C[(size_t)row * p.N + col] = (acc[i][j][r] - mean[r]) * var[r] * gm + bt;
        }
      }
    }
  } else {
    const bool full = m0 + (wm + 1) * MT * 32 <= p.M && n0 + (wn + 1) * NT * 32 <= p.N;
    // all loads unconditional (clamped) and every value finished before the predicated stores: a
    // load result consumed inside a per-row branch makes hipcc emit `s_waitcnt vmcnt(0)` in front of
    // every store, serialising them
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + (wn * NT + j) * 32 + l31;
        const int colc = col < p.N ? col : p.N - 1;
        const float bz = p.bias[colc];
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const int rowc = row < p.M ? row : p.M - 1;
          v[r] = acc[i][j][r] + bz;
          if constexpr (EPI == EPI_BIAS_GELU) v[r] = gelu_erf(v[r]);
          if constexpr (EPI == EPI_BIAS_RESID) v[r] += p.resid[(size_t)rowc * p.N + colc];
        }
        if (full) {  // wave-uniform: interior block, straight-line stores
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            p.C[(size_t)row * p.N + col] = v[r];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(v[r]));  // values are final before any branch
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row < p.M && col < p.N) p.C[(size_t)row * p.N + col] = v[r];
          }
        }
      }
  }
}

template <int EPI>
static void launch_generic(const GemmArgs& p, hipStream_t s) {
  constexpr int BM = 128, BN = 128;
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  hipLaunchKernelGGL((gemm_f32_kernel<2, 2, 2, 2, 32, EPI>), dim3(tiles), dim3(256), 0, s, p);
}

void launch_gemm_f32(int epilogue, const float* A, const float* W, const float* bias, const float* resid, float* C,
                     int M, int N, int K, hipStream_t s) {
  GemmArgs p{A, W, bias, resid, nullptr, nullptr, C, M, N, K, 0.f};
  switch (epilogue) {
    case EPI_BIAS: launch_generic<EPI_BIAS>(p, s); break;
    case EPI_BIAS_GELU: launch_generic<EPI_BIAS_GELU>(p, s); break;
    default: launch_generic<EPI_BIAS_RESID>(p, s); break;
  }
}

bool launch_gemm_f32_ln(const float* A, const float* W, const float* bias, const float* resid, const float* gamma,
                        const float* beta, float eps, float* C, int M, int N, int K, hipStream_t s) {
  GemmArgs p{A, W, bias, resid, gamma, beta, C, M, N, K, eps};
  const int tiles = (M + 127) / 128;
  if (K % 16 != 0) return false;
  if (N == 384) {
    hipLaunchKernelGGL((gemm_f32_kernel<4, 1, 1, 12, 16, EPI_LN_INTERNAL>), dim3(tiles), dim3(256), 0, s, p);
    return true;
  }
  if (N == 192) {
    hipLaunchKernelGGL((gemm_f32_kernel<4, 1, 1, 6, 16, EPI_LN_INTERNAL>), dim3(tiles), dim3(256), 0, s, p);
    return true;
  }
  return false;
}

}  // namespace fdmi
