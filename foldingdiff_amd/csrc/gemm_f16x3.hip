// Token GEMMs with fp32-class accuracy on the fp16 matrix cores ("fp16x3 split").
//
//   C[M,N] = A[M,K] * W[N,K]^T + bias[N]   (+ GELU | + residual)
//
// Same role and epilogues as gemm_f32.hip (HF BertSelfAttention q/k/v, BertSelfOutput.dense,
// BertIntermediate.dense, BertOutput.dense, AnglesPredictor.dense1 -- transformers 4.11.3 via
// foldingdiff/modelling.py:473-480, :203-205), but each fp32 operand x is carried as two fp16
// numbers
//        x * s  =  hi + lo,      hi = fp16(x*s),  lo = fp16(x*s - hi)
// (s a power of two chosen so that lo stays a normal fp16 for operands of typical size: 16 for
// activations, per-tensor for weights), i.e. 22 significant bits, and a product a*w is formed as
//        a_hi*w_hi + a_hi*w_lo + a_lo*w_hi              (the lo*lo term, 2^-22 relative, is dropped)
// by three v_mfma_f32_32x32x16_f16 into ONE fp32 accumulator.  fp16 products are exact in
// fp32, so the only error beyond the fp32 accumulation itself is the operand truncation at
// 2^-22..2^-23 -- numerically indistinguishable from a plain fp32 GEMM (3.56e-7 vs 3.55e-7
// rel-rms against fp64 at K=768; tests/test_gpu_parity.py measures it on the device).  MFMA cost: 3 x 32 cycles per 16 k  vs  8 x 64 cycles for v_mfma_f32_32x32x2_f32,
// i.e. 5.3x the fp32-MFMA rate (838 TFLOP/s fp32-equivalent peak).
//
// Activations stay fp32 in HBM; they are split while being staged into LDS (VALU work that
// overlaps the MFMAs).  Weights are split once at fd_finalize into the LDS row image
// [n][k/16][hi x16 | lo x16] (64 B).
//
// Tiling: 256 x 128 block, 4 waves as 2 x 2, each 128 x 64 (4 x 2 MFMA tiles, 128 accumulator
// registers), BK = 16, LDS double buffered (one barrier per k-tile).  LDS rows
// are 64 B of payload padded to 80 B => the 16-byte operand fetches are conflict free.
// 32 FLOP per staged byte -> 21 B/clk/CU from L2 at the fp16x3 peak (tile sized for that).
#include "fdmi_kernels.h"

namespace fdmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // 16-byte LDS / global transfer unit

struct GemmSplitArgs {
  const float* A;
  const u32x4* Wp;  // packed split weight, [Npad][K/16][4] x 16 B
  const float* bias;
  const float* resid;
  float* C;
  int M, N, K;
  float a_scale;    // power of two applied to A before splitting
  float out_scale;  // 1 / (a_scale * w_scale)
};

__device__ __forceinline__ int xcd_remap16(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

__device__ __forceinline__ float gelu_erf16(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// split 8 fp32 values (two float4) into hi / lo fp16 octets; pure register code (vector
// element inserts + bitcasts: nothing for the compiler to demote to scratch or LDS)
__device__ __forceinline__ void split8(const float4& p, const float4& q, float s, u32x4& hi, u32x4& lo) {
  const float x[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
  f16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float xs = x[i] * s;
    const _Float16 h = (_Float16)xs;            // round to nearest even
    a[i] = h;
    b[i] = (_Float16)(xs - (float)h);           // xs - h is exact in fp32
  }
  hi = __builtin_bit_cast(u32x4, a);
  lo = __builtin_bit_cast(u32x4, b);
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f16x3_kernel(GemmSplitArgs p) {
  constexpr int BM = 256, BN = 128, BK = 16, RQ = 5;  // RQ: uint4 (16 B) per padded LDS row
  constexpr int STAGE = (BM + BN) * RQ;
  __shared__ u32x4 smem[2 * STAGE];  // 2 x 30,720 B

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int bid = xcd_remap16(blockIdx.x, gridDim.x);
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  const int K = p.K, nk = K / BK;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging roles: thread t owns A row t (16 floats = 64 B per k-tile); W image chunks 2 per thread
  float4 ra[4];
  u32x4 rw[2];
  const int arow = m0 + tid;
  const float* aptr = p.A + (size_t)(arow < p.M ? arow : 0) * K;
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      ra[i] = arow < p.M ? *reinterpret_cast<const float4*>(aptr + kt * BK + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + 256 * i, row = c >> 2, part = c & 3;
      rw[i] = p.Wp[((size_t)(n0 + row) * nk + kt) * 4 + part];
    }
  };
  auto lstore = [&](int buf) {
    u32x4* S = smem + buf * STAGE;
    u32x4 h0, l0, h1, l1;
    split8(ra[0], ra[1], p.a_scale, h0, l0);
    split8(ra[2], ra[3], p.a_scale, h1, l1);
    u32x4* row = S + tid * RQ;  // [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15 | pad]
    row[0] = h0; row[1] = h1; row[2] = l0; row[3] = l1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + 256 * i, r = c >> 2, part = c & 3;
      S[(BM + r) * RQ + part] = rw[i];
    }
  };

  gload(0);
  lstore(0);
  if (nk > 1) gload(1);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const u32x4* S = smem + (kt & 1) * STAGE;
    f16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32x4* row = S + (wm * 128 + i * 32 + l31) * RQ;
      ah[i] = __builtin_bit_cast(f16x8, row[half]);
      al[i] = __builtin_bit_cast(f16x8, row[2 + half]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const u32x4* row = S + (BM + wn * 64 + j * 32 + l31) * RQ;
      bh[j] = __builtin_bit_cast(f16x8, row[half]);
      bl[j] = __builtin_bit_cast(f16x8, row[2 + half]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
    if (kt + 1 < nk) {
      lstore((kt + 1) & 1);  // buffer last read in iteration kt-1; every wave passed that barrier
      if (kt + 2 < nk) gload(kt + 2);
    }
    __syncthreads();
  }

  // epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l31;
      if (col < p.N) {
        const float bz = p.bias[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (row < p.M) {
            float v = acc[i][j][r] * p.out_scale + bz;
            if constexpr (EPI == EPI_BIAS_GELU) v = gelu_erf16(v);
            if constexpr (EPI == EPI_BIAS_RESID) v += p.resid[(size_t)row * p.N + col];
            p.C[(size_t)row * p.N + col] = v;
          }
        }
      }
    }
}

void launch_gemm_f16x3(int epilogue, const float* A, const void* Wp, float w_scale, const float* bias,
                       const float* resid, float* C, int M, int N, int K, hipStream_t s) {
  const float a_scale = 16.0f;  // |a| < 4094 stays finite in fp16; lo of |a| > 0.008 is a normal fp16
  GemmSplitArgs p{A, static_cast<const u32x4*>(Wp), bias, resid, C, M, N, K, a_scale, 1.0f / (a_scale * w_scale)};
  const int tiles = ((M + 255) / 256) * ((N + 127) / 128);
  switch (epilogue) {
    case EPI_BIAS: hipLaunchKernelGGL((gemm_f16x3_kernel<EPI_BIAS>), dim3(tiles), dim3(256), 0, s, p); break;
    case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm_f16x3_kernel<EPI_BIAS_GELU>), dim3(tiles), dim3(256), 0, s, p); break;
    default: hipLaunchKernelGGL((gemm_f16x3_kernel<EPI_BIAS_RESID>), dim3(tiles), dim3(256), 0, s, p); break;
  }
}

}  // namespace fdmi
