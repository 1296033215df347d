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
// [n][k/32][hi x32 | lo x32] (128 B).
//
// Tiling: (64*WM) x 128 block, WM x 2 waves, each 64 x 64 (2 x 2 MFMA tiles, 64 accumulator
// registers), BK = 32, LDS double buffered (one barrier per k-tile), and TWO k-tiles of global
// loads in flight per thread (register prefetch sets): the kernel is L2-latency bound with one.
// LDS rows are 128 B of payload [hi k0-31 | lo k0-31] padded to 144 B => the 16-byte operand
// fetches are conflict free.  32 FLOP per staged byte -> 21 B/clk/CU from L2 at the fp16x3 peak.
#include <cstdlib>

#include "fdmi_kernels.h"

namespace fdmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // 16-byte LDS / global transfer unit

struct GemmSplitArgs {
  const float* A;
  const u32x4* Wp;  // packed split weight, [Npad128][K/32][8] x 16 B  (hi k0-31 | lo k0-31)
  const float* bias;
  const float* resid;
  float* C;
  int M, N, K;
  float a_scale;    // power of two applied to A before splitting
  float out_scale;  // 1 / (a_scale * w_scale)
};

struct StreamNo { static constexpr bool value = false; };
struct StreamYes { static constexpr bool value = true; };

__device__ __forceinline__ int xcd_remap16(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// erf(x) as the (odd degree-13) / (even degree-8) rational minimax on [-4, 4] (the single-precision
// form used by Eigen / XLA; |x| > 4 is +-1 in fp32): max abs error 4.5e-7 vs float64 erf (checked in
// tests/test_host.py on 3M points) -- the same class as libm's erff -- at 18 branch-free VALU ops per
// GELU instead of erff's two divergent branches (the GELU epilogue runs 50M times per FFN-up launch).
__device__ __forceinline__ float erf_rational(float x) {
  x = __builtin_fminf(__builtin_fmaxf(x, -4.0f), 4.0f);
  const float x2 = x * x;
  float p = -2.72614225801306e-10f;
  p = __builtin_fmaf(p, x2, 2.77068142495902e-08f);
  p = __builtin_fmaf(p, x2, -2.10102402082508e-06f);
  p = __builtin_fmaf(p, x2, -5.69250639462346e-05f);
  p = __builtin_fmaf(p, x2, -7.34990630326855e-04f);
  p = __builtin_fmaf(p, x2, -2.95459980854025e-03f);
  p = __builtin_fmaf(p, x2, -1.60960333262415e-02f);
  float q = -1.45660718464996e-05f;
  q = __builtin_fmaf(q, x2, -2.13374055278905e-04f);
  q = __builtin_fmaf(q, x2, -1.68282697438203e-03f);
  q = __builtin_fmaf(q, x2, -7.37332916720468e-03f);
  q = __builtin_fmaf(q, x2, -1.42647390514189e-02f);
  return (p * x) * __builtin_amdgcn_rcpf(q);
}

// exact-erf GELU (HF "gelu", modelling.py:195-196): 0.5 x (1 + erf(x / sqrt(2)))
__device__ __forceinline__ float gelu_erf16(float x) {
  const float h = 0.5f * x;
  return __builtin_fmaf(h, erf_rational(x * 0.70710678118654752440f), h);
}

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

// Epilogue of one wave's 64 x 64 accumulator block (2 x 2 MFMA tiles) at (row0, col0).  All loads
// (bias, residual) are unconditional with clamped indices and every value is finished BEFORE the
// predicated stores: a load result consumed inside a per-row branch makes hipcc put
// `s_waitcnt vmcnt(0)` in front of every store, which serialises the stores and drains the
// prefetched loads of the next tile.
template <int EPI>
__device__ __forceinline__ void store_tile(const GemmSplitArgs& p, const f32x16 (&acc)[2][2], int row0, int col0,
                                           int half, int l31) {
  const bool full = row0 + 64 <= p.M && col0 + 64 <= p.N;
  float bz[2];
  int colc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = col0 + j * 32 + l31;
    colc[j] = col < p.N ? col : p.N - 1;
    bz[j] = p.bias[colc[j]];
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = col0 + j * 32 + l31;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int rowc = row < p.M ? row : p.M - 1;
        v[r] = acc[i][j][r] * p.out_scale + bz[j];
        if constexpr (EPI == EPI_BIAS_GELU) v[r] = gelu_erf16(v[r]);
        if constexpr (EPI == EPI_BIAS_RESID) v[r] += p.resid[(size_t)rowc * p.N + colc[j]];
      }
      if (full) {  // wave-uniform: interior block, straight-line stores
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          p.C[(size_t)row * p.N + col] = v[r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(v[r]));  // values are final before any branch
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (row < p.M && col < p.N) p.C[(size_t)row * p.N + col] = v[r];
        }
      }
    }
}

// WM = waves along M: 4 -> 256 x 128 block, 8 waves, one block per CU;
//                     2 -> 128 x 128 block, 4 waves, two independent blocks per CU (their barrier /
//                          staging phases overlap each other's MFMA phases).
// AL = A-staging layout: 0 -> thread = (row, 64-byte half of the 128-byte k-tile row), four dwordx4 loads
//                        1 -> thread = two (row, 32-byte octet) pairs: 4 consecutive lanes cover one full
//                             128-byte line with two dwordx4 loads (half the cache-line requests per instruction)
// DBG (ablation builds, never used by the product path; results are wrong by design):
//   1 = no MFMAs (operands still fetched from LDS), 2 = no global loads / LDS staging inside the
//   k-loop, 3 = staging without the fp32 -> hi/lo conversion.
template <int EPI, int PF, int WM, int AL, int DBG = 0>
__global__ __launch_bounds__(128 * WM) void gemm_f16x3_kernel(GemmSplitArgs p) {
  constexpr int BM = 64 * WM, BN = 128, BK = 32, RQ = 9;  // RQ: 16-byte units per padded LDS row (144 B)
  constexpr int NTHR = 128 * WM;
  constexpr int WU = BN * 8 / NTHR;                       // W image units (16 B) per thread per k-tile
  constexpr int STAGE = (BM + BN) * RQ;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* smem = reinterpret_cast<u32x4*>(smem_raw);  // 2 stages x 55,296 B

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int bid = xcd_remap16(blockIdx.x, gridDim.x);
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  const int K = p.K, nk = K / BK;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging roles: thread t -> A row t/2, 16 floats (k = 16*(t&1) ..); W row t/(8/WU), WU units of its image
  // Loads are UNCONDITIONAL (row / tile indices are clamped instead of predicated): a load inside
  // a branch makes hipcc lose count of the outstanding loads and fall back to s_waitcnt vmcnt(2),
  // which drains the younger prefetch set as well.  Rows >= M therefore accumulate copies of row
  // M-1; they are never stored.
  const int arow = AL == 0 ? tid >> 1 : tid >> 2, au = AL == 0 ? tid & 1 : tid & 3;
  const int arow2 = arow + NTHR / 4;  // AL == 1: second (row, octet) pair of this thread
  const float* aptr = p.A + (size_t)(m0 + arow < p.M ? m0 + arow : p.M - 1) * K + (AL == 0 ? 16 : 8) * au;
  const float* aptr2 = p.A + (size_t)(m0 + arow2 < p.M ? m0 + arow2 : p.M - 1) * K + 8 * au;
  const int wrow = tid / (8 / WU), wpart = WU * (tid % (8 / WU));
  const u32x4* wptr = p.Wp + ((size_t)(n0 + wrow) * nk) * 8 + wpart;

  float4 ra0[4], ra1[4];
  u32x4 rw0[WU], rw1[WU];
  auto gload = [&](float4 (&ra)[4], u32x4 (&rw)[WU], int kt) {
    kt = kt < nk ? kt : nk - 1;  // past the end: re-load the last tile (never consumed)
    if constexpr (AL == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const float4*>(aptr + kt * BK + 4 * i);
    } else {
      ra[0] = *reinterpret_cast<const float4*>(aptr + kt * BK);
      ra[1] = *reinterpret_cast<const float4*>(aptr + kt * BK + 4);
      ra[2] = *reinterpret_cast<const float4*>(aptr2 + kt * BK);
      ra[3] = *reinterpret_cast<const float4*>(aptr2 + kt * BK + 4);
    }
#pragma unroll
    for (int i = 0; i < WU; ++i) rw[i] = wptr[(size_t)kt * 8 + i];
  };
  auto lstore = [&](const float4 (&ra)[4], const u32x4 (&rw)[WU], int buf) {
    u32x4* S = smem + buf * STAGE;
    u32x4 h0, l0, h1, l1;
    if constexpr (DBG == 3) {
      h0 = __builtin_bit_cast(u32x4, ra[0]); l0 = __builtin_bit_cast(u32x4, ra[1]);
      h1 = __builtin_bit_cast(u32x4, ra[2]); l1 = __builtin_bit_cast(u32x4, ra[3]);
    } else {
      split8(ra[0], ra[1], p.a_scale, h0, l0);
      split8(ra[2], ra[3], p.a_scale, h1, l1);
    }
    u32x4* row = S + arow * RQ;  // [hi k0-31 (4 units) | lo k0-31 (4 units) | pad]
    if constexpr (AL == 0) {
      row[2 * au] = h0; row[2 * au + 1] = h1; row[4 + 2 * au] = l0; row[4 + 2 * au + 1] = l1;
    } else {
      u32x4* rowb = S + arow2 * RQ;
      row[au] = h0; row[4 + au] = l0; rowb[au] = h1; rowb[4 + au] = l1;
    }
    u32x4* wr = S + (BM + wrow) * RQ + wpart;
#pragma unroll
    for (int i = 0; i < WU; ++i) wr[i] = rw[i];
  };
  auto compute = [&](int buf) {
    const u32x4* S = smem + buf * STAGE;
#pragma unroll
    for (int c = 0; c < 2; ++c) {  // two k16 steps
      f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u32x4* row = S + (wm * 64 + i * 32 + l31) * RQ;
        ah[i] = __builtin_bit_cast(f16x8, row[2 * c + half]);
        al[i] = __builtin_bit_cast(f16x8, row[4 + 2 * c + half]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const u32x4* row = S + (BM + wn * 64 + j * 32 + l31) * RQ;
        bh[j] = __builtin_bit_cast(f16x8, row[2 * c + half]);
        bl[j] = __builtin_bit_cast(f16x8, row[4 + 2 * c + half]);
      }
      if constexpr (DBG == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(ah[i]), "v"(al[i]), "v"(bh[i]), "v"(bl[i]));
        continue;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
    }
  };

  // register set 1 carries the even k-tiles, set 0 the odd ones
  gload(ra1, rw1, 0);
  lstore(ra1, rw1, 0);
  gload(ra0, rw0, 1);
  if constexpr (PF == 2) gload(ra1, rw1, 2);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    // even tile kt is in buffer 0; stage odd tile kt+1 into buffer 1 (last read in iteration
    // kt-1: every wave has passed that barrier).  Staging past the last tile is harmless.
    compute(0);
    if constexpr (DBG != 2) {
      lstore(ra0, rw0, 1);
      if constexpr (PF == 2) gload(ra0, rw0, kt + 3);
      else gload(ra1, rw1, kt + 2);
    }
    __syncthreads();
    if (kt + 1 < nk) compute(1);
    if constexpr (DBG != 2) {
      lstore(ra1, rw1, 0);
      if constexpr (PF == 2) gload(ra1, rw1, kt + 4);
      else gload(ra0, rw0, kt + 3);
    }
    __syncthreads();
  }

  // epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  store_tile<EPI>(p, acc, m0 + wm * 64, n0 + wn * 64, half, l31);
}

// ---------------------------------------------------------------------------------------------
// Persistent variant: one workgroup per CU walks a list of output tiles and treats the
// (tile, k-tile) pairs as ONE continuous stream through the same two-deep register prefetch and
// LDS double buffer.  The first k-tiles of the next output tile are therefore already in flight /
// in LDS while the epilogue of the current tile issues its (asynchronous) stores: neither the
// per-tile load latency nor the store drain is exposed any more (with one 8-wave workgroup per
// CU nothing else could hide them -- measured: staging alone 184 us, MFMA loop + epilogue alone
// 236 us, both together 304 us on the QKV shape in the non-persistent kernel).
// Requires an even number of k-tiles (buffer parity is then the same for every tile).
// Tiles are dealt XCD-aware: XCD x (= workgroups with id % 8 == x) owns a contiguous range of the
// n-fastest tile order, and its workgroups take neighbouring tiles at the same time (shared A panel
// in that XCD's L2).
// VAR (compute-phase schedule): 0 = fetch the fragments of one k16 step, 12 MFMAs, next step;
//                               1 = fetch the fragments of both k16 steps up front, then 24 MFMAs;
//                               2 = as 1 with s_setprio 1 around the MFMA cluster.
template <int EPI, int VAR = 0, int WM = 4>
__global__ __launch_bounds__(128 * WM) void gemm_f16x3_persist_kernel(GemmSplitArgs p) {
  constexpr int BM = 64 * WM, BN = 128, BK = 32, RQ = 9, NTHR = 128 * WM;
  constexpr int WU = BN * 8 / NTHR;  // W image units (16 B) per thread per k-tile
  constexpr int STAGE = (BM + BN) * RQ;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* smem = reinterpret_cast<u32x4*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int ntiles = tiles_m * tiles_n;
  const int K = p.K, nk = K / BK;
  // this workgroup's tiles: first, first + stride, ... (cnt of them)
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int lo = (int)((long long)ntiles * xcd / 8), hi = (int)((long long)ntiles * (xcd + 1) / 8);
  const int first = lo + j, stride = per;
  const int cnt = first < hi ? (hi - first + stride - 1) / stride : 0;
  if (cnt == 0) return;
  const int G = cnt * nk;  // length of the (tile, k-tile) stream

  const int arow = tid >> 2, au = tid & 3;        // A: two (row, 8-float octet) pairs per thread
  const int arow2 = arow + NTHR / 4;
  const int wrow = tid / (8 / WU), wpart = WU * (tid % (8 / WU));  // W image: WU x 16 B per thread

  f32x16 acc[2][2];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
  };
  zero_acc();

  float4 ra0[4], ra1[4];
  u32x4 rw0[WU], rw1[WU];
  // unconditional loads, indices clamped (see the non-persistent kernel)
  auto gload = [&](float4 (&ra)[4], u32x4 (&rw)[WU], int g) __attribute__((always_inline)) {
    g = g < G ? g : G - 1;
    const int ti = g / nk, kt = g - ti * nk;
    const int tile = first + ti * stride;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int r1 = m0 + arow < p.M ? m0 + arow : p.M - 1, r2 = m0 + arow2 < p.M ? m0 + arow2 : p.M - 1;
    const float* a1 = p.A + (size_t)r1 * K + kt * BK + 8 * au;
    const float* a2 = p.A + (size_t)r2 * K + kt * BK + 8 * au;
    ra[0] = *reinterpret_cast<const float4*>(a1);
    ra[1] = *reinterpret_cast<const float4*>(a1 + 4);
    ra[2] = *reinterpret_cast<const float4*>(a2);
    ra[3] = *reinterpret_cast<const float4*>(a2 + 4);
    const u32x4* w = p.Wp + ((size_t)(n0 + wrow) * nk + kt) * 8 + wpart;
#pragma unroll
    for (int i = 0; i < WU; ++i) rw[i] = w[i];
  };
  auto lstore = [&](const float4 (&ra)[4], const u32x4 (&rw)[WU], int buf) __attribute__((always_inline)) {
    u32x4* S = smem + buf * STAGE;
    u32x4 h0, l0, h1, l1;
    split8(ra[0], ra[1], p.a_scale, h0, l0);
    split8(ra[2], ra[3], p.a_scale, h1, l1);
    u32x4* row = S + arow * RQ;
    u32x4* rowb = S + arow2 * RQ;
    row[au] = h0; row[4 + au] = l0; rowb[au] = h1; rowb[4 + au] = l1;
    u32x4* wr = S + (BM + wrow) * RQ + wpart;
#pragma unroll
    for (int i = 0; i < WU; ++i) wr[i] = rw[i];
  };
  auto compute = [&](int buf) __attribute__((always_inline)) {
    const u32x4* S = smem + buf * STAGE;
    if constexpr (VAR == 0 || VAR == 3) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const u32x4* row = S + (wm * 64 + i * 32 + l31) * RQ;
          ah[i] = __builtin_bit_cast(f16x8, row[2 * c + half]);
          al[i] = __builtin_bit_cast(f16x8, row[4 + 2 * c + half]);
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const u32x4* row = S + (BM + wn * 64 + jj * 32 + l31) * RQ;
          bh[jj] = __builtin_bit_cast(f16x8, row[2 * c + half]);
          bl[jj] = __builtin_bit_cast(f16x8, row[4 + 2 * c + half]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[jj], acc[i][jj], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[jj], acc[i][jj], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[jj], acc[i][jj], 0, 0, 0);
      }
    } else {
      f16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];  // [k16 step][tile]
#pragma unroll
      for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const u32x4* row = S + (wm * 64 + i * 32 + l31) * RQ;
          ah[c][i] = __builtin_bit_cast(f16x8, row[2 * c + half]);
          al[c][i] = __builtin_bit_cast(f16x8, row[4 + 2 * c + half]);
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const u32x4* row = S + (BM + wn * 64 + jj * 32 + l31) * RQ;
          bh[c][jj] = __builtin_bit_cast(f16x8, row[2 * c + half]);
          bl[c][jj] = __builtin_bit_cast(f16x8, row[4 + 2 * c + half]);
        }
      }
      if constexpr (VAR == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[c][i], bh[c][jj], acc[i][jj], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[c][i], bl[c][jj], acc[i][jj], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[c][i], bh[c][jj], acc[i][jj], 0, 0, 0);
      }
      if constexpr (VAR == 2) __builtin_amdgcn_s_setprio(0);
    }
  };
  auto epilogue = [&](int ti) __attribute__((always_inline)) {
    const int tile = first + ti * stride;
    store_tile<EPI>(p, acc, (tile / tiles_n) * BM + wm * 64, (tile % tiles_n) * BN + wn * 64, half, l31);
  };

  if constexpr (VAR == 3) {
    // Ping-pong schedule (8 waves): waves 0-3 ("A") and 4-7 ("B") share the four SIMDs pairwise and run
    // one barrier phase apart, so on every SIMD one wave is in its MFMA phase while the other splits and
    // stages operands (in the lock-step schedule both waves of a SIMD are in the same phase and the MFMA
    // pipe idles during every staging phase).  Phase table (stream position p; buffer p & 1):
    //     A: store(p) at phase 2p,     compute(p) at phase 2p + 3
    //     B: store(p) at phase 2p + 1, compute(p) at phase 2p + 2
    // every phase ends in a workgroup barrier; position p is complete after phase 2p + 1, read in phases
    // 2p + 2 / 2p + 3, and its buffer is rewritten (position p + 2) from phase 2p + 4 on.
    static_assert(VAR != 3 || WM == 4, "ping-pong needs the 8-wave workgroup");
    const int role = __builtin_amdgcn_readfirstlane(wid >> 2);
    const int npairs = nk / 2;  // >= 2 (host guarantees)
    gload(ra1, rw1, 0);
    gload(ra0, rw0, 1);
    if (role == 0) {
      lstore(ra1, rw1, 0);
      gload(ra1, rw1, 2);
      __syncthreads();  // phase 0
      __syncthreads();  // phase 1 (B stores position 0)
      auto step = [&](int g, int ti, auto last) __attribute__((always_inline)) {
        lstore(ra0, rw0, 1);  // position g + 1
        gload(ra0, rw0, g + 3);
        __syncthreads();
        compute(0);           // position g
        __syncthreads();
        lstore(ra1, rw1, 0);  // position g + 2 (the last pair of a tile: already the next tile)
        gload(ra1, rw1, g + 4);
        __syncthreads();
        compute(1);           // position g + 1
        if constexpr (decltype(last)::value) {
          epilogue(ti);
          zero_acc();
        }
        __syncthreads();
      };
      for (int ti = 0; ti < cnt; ++ti) {
        const int g0 = ti * nk;
        step(g0, ti, StreamNo{});
        for (int pi = 1; pi < npairs - 1; ++pi) step(g0 + 2 * pi, ti, StreamNo{});
        step(g0 + nk - 2, ti, StreamYes{});
      }
    } else {
      __syncthreads();  // phase 0 (A stores position 0)
      auto step = [&](int g, int ti, auto last) __attribute__((always_inline)) {
        lstore(ra1, rw1, 0);  // position g
        gload(ra1, rw1, g + 2);
        __syncthreads();
        compute(0);           // position g
        __syncthreads();
        lstore(ra0, rw0, 1);  // position g + 1
        gload(ra0, rw0, g + 3);
        __syncthreads();
        compute(1);           // position g + 1
        if constexpr (decltype(last)::value) {
          epilogue(ti);
          zero_acc();
        }
        __syncthreads();
      };
      for (int ti = 0; ti < cnt; ++ti) {
        const int g0 = ti * nk;
        step(g0, ti, StreamNo{});
        for (int pi = 1; pi < npairs - 1; ++pi) step(g0 + 2 * pi, ti, StreamNo{});
        step(g0 + nk - 2, ti, StreamYes{});
      }
      __syncthreads();  // pairs with A's last barrier
    }
    return;
  }

  // stream positions: even g -> LDS buffer 0 / register set 1, odd g -> buffer 1 / set 0
  auto pair = [&](int g) __attribute__((always_inline)) {
    compute(0);
    lstore(ra0, rw0, 1);
    gload(ra0, rw0, g + 3);
    __syncthreads();
    compute(1);
    lstore(ra1, rw1, 0);  // position g+2: for the last k-pair of a tile this is already the NEXT tile
    gload(ra1, rw1, g + 4);
    __syncthreads();
  };
  gload(ra1, rw1, 0);
  lstore(ra1, rw1, 0);
  gload(ra0, rw0, 1);
  gload(ra1, rw1, 2);
  __syncthreads();
  // The first and the last k-pair of every tile are peeled so that the steady-state loop has the
  // same number of younger memory operations on every path into it: hipcc then keeps exact counted
  // s_waitcnt vmcnt(N) there, and the epilogue's stores are never drained by a conservative wait.
  const int npairs = nk / 2;  // >= 2 (host guarantees)
  for (int ti = 0; ti < cnt; ++ti) {
    const int g0 = ti * nk;
    pair(g0);
    for (int pi = 1; pi < npairs - 1; ++pi) pair(g0 + 2 * pi);
    pair(g0 + nk - 2);
    epilogue(ti);  // stores drain under the next tile's MFMAs; its first k-tiles are already staged
    zero_acc();
  }
}

template <int EPI, int VAR, int WM>
static void launch_persist(const GemmSplitArgs& p, hipStream_t s) {
  constexpr int BM = 64 * WM;
  constexpr int smem = 2 * (BM + 128) * 9 * 16;
  static bool attr_set = false;
  static int n_cu = 256;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x3_persist_kernel<EPI, VAR, WM>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      n_cu = prop.multiProcessorCount;
    attr_set = true;
  }
  const int ntiles = ((p.M + BM - 1) / BM) * ((p.N + 127) / 128);
  int grid = n_cu * (WM == 2 ? 2 : 1) / 8 * 8;  // one 8-wave or two 4-wave workgroups per CU, a multiple of the 8 XCDs
  if (grid > ntiles) grid = (ntiles + 7) / 8 * 8;
  hipLaunchKernelGGL((gemm_f16x3_persist_kernel<EPI, VAR, WM>), dim3(grid), dim3(128 * WM), smem, s, p);
}

// ---------------------------------------------------------------------------------------------
// GEMM + bias + residual + LayerNorm over full rows (BertSelfOutput / BertOutput, transformers
// 4.11.3: LN(dense(x) + input)), N == 384 == the whole row in one workgroup.
// Same persistent (tile, k-tile) stream as above with a 128 x 384 block: 8 waves as 2 (M) x 4 (N),
// each 64 x 96 (2 x 3 MFMA tiles, 96 accumulator registers, 18 MFMAs per 10 fragment fetches).
// The epilogue finishes v = acc + bias + resid in registers, reduces each row's 96 columns inside
// the wave (DPP butterflies over the 32 lanes that share a row), combines the four N-waves through a
// small LDS scratch, and repeats that for the centred squares (two-pass variance, as
// rowwise.hip:row_layernorm): the pre-LN tensor never goes to HBM and the standalone LayerNorm
// launch (201 MB of traffic per call at M = 65536) disappears.
static int env_int(const char* name, int dflt);

struct GemmLnArgs {
  const float* A;
  const u32x4* Wp;
  const float* bias;
  const float* resid;
  const float* gamma;
  const float* beta;
  float* C;
  int M, K;
  float a_scale, out_scale, eps;
  int N;  // row length of C (= 384 for the LayerNorm epilogue; a multiple of 384 for the plain epilogues)
};

// Sum over the 32 lanes of this lane's half-wave, DPP only (no LDS round trip).  The result is
// valid in the UPPER 16 lanes of each half (lanes 16-31 and 48-63): row_bcast15 adds the lower
// row's total into the upper row only.
__device__ __forceinline__ float half_wave_sum_hi(float v) {
#define FD_DPP_ADD(ctrl, rmask) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
  FD_DPP_ADD(0xB1, 0xf);   // quad_perm [1,0,3,2]
  FD_DPP_ADD(0x4E, 0xf);   // quad_perm [2,3,0,1]
  FD_DPP_ADD(0x141, 0xf);  // row_half_mirror: lanes of a quad hold equal sums, so i <-> 7-i adds the other quad
  FD_DPP_ADD(0x140, 0xf);  // row_mirror: i <-> 15-i adds the other 8 lanes
  FD_DPP_ADD(0x142, 0xa);  // row_bcast15 into rows 1 and 3: + the total of the 16 lanes below
#undef FD_DPP_ADD
  return v;
}

// DBG (ablation builds via FDMI_LN_DBG, results wrong by design): 1 = no residual loads, 2 = no output
// stores, 3 = no LayerNorm reductions.
// PLAIN >= 0: the same 128 x 384 tiling with a plain bias (EPI_BIAS) / bias + GELU (EPI_BIAS_GELU) epilogue for
// N a multiple of 384 (QKV, FFN-up, head dense1; launch_gemm_f16x3_wide) -- each A panel is split 3x less
// often than with 128-column tiles and a wave issues 18 MFMAs per 10 fragment fetches instead of 12 per 8.
template <int DBG, int PLAIN = -1>
__global__ __launch_bounds__(512) void gemm_f16x3_ln_kernel(GemmLnArgs p) {
  constexpr int BM = 128, BN = 384, BK = 32, RQ = 9, NTHR = 512, NT = 3;
  constexpr int WU = BN * 8 / NTHR;  // 6 W image units (16 B) per thread per k-tile
  constexpr int STAGE = (BM + BN) * RQ;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* smem = reinterpret_cast<u32x4*>(smem_raw);
  float* red = reinterpret_cast<float*>(smem + 2 * STAGE);  // 2 passes x (part[128][4] + tot[128])

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3;
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles_n = PLAIN >= 0 ? p.N / BN : 1;
  const int ntiles = (p.M / BM) * tiles_n;
  const int K = p.K, nk = K / BK;
  // tiles are dealt XCD-aware, n fastest (see the persistent kernel above): the workgroups of an XCD take
  // neighbouring tiles at the same time, so the column tiles of one A panel share that XCD's L2
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int tlo = (int)((long long)ntiles * xcd / 8), thi = (int)((long long)ntiles * (xcd + 1) / 8);
  const int first = tlo + jx, stride = per;
  const int cnt = first < thi ? (thi - first + stride - 1) / stride : 0;
  if (cnt == 0) return;
  const int G = cnt * nk;

  const int arow = tid >> 2, au = tid & 3;   // A: one (row, 8-float octet) per thread
  const int wrow = tid >> 3, wu = tid & 7;   // W image: unit wu of rows wrow + 64 i

  f32x16 acc[2][NT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < NT; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
  };
  zero_acc();

  float4 ra0[2], ra1[2];
  u32x4 rw0[WU], rw1[WU];
  auto gload = [&](float4 (&ra)[2], u32x4 (&rw)[WU], int g) {
    g = g < G ? g : G - 1;
    const int ti = g / nk, kt = g - ti * nk;
    const int tile = first + ti * stride;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int r1 = m0 + arow < p.M ? m0 + arow : p.M - 1;
    const float* a1 = p.A + (size_t)r1 * K + kt * BK + 8 * au;
    ra[0] = *reinterpret_cast<const float4*>(a1);
    ra[1] = *reinterpret_cast<const float4*>(a1 + 4);
    const u32x4* w = p.Wp + ((size_t)(n0 + wrow) * nk + kt) * 8 + wu;
#pragma unroll
    for (int i = 0; i < WU; ++i) rw[i] = w[(size_t)i * 64 * nk * 8];
  };
  auto lstore = [&](const float4 (&ra)[2], const u32x4 (&rw)[WU], int buf) {
    u32x4* S = smem + buf * STAGE;
    u32x4 h0, l0;
    split8(ra[0], ra[1], p.a_scale, h0, l0);
    u32x4* row = S + arow * RQ;
    row[au] = h0;
    row[4 + au] = l0;
#pragma unroll
    for (int i = 0; i < WU; ++i) S[(BM + wrow + 64 * i) * RQ + wu] = rw[i];
  };
  auto compute = [&](int buf) {
    const u32x4* S = smem + buf * STAGE;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      f16x8 ah[2], al[2], bh[NT], bl[NT];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u32x4* row = S + (wm * 64 + i * 32 + l31) * RQ;
        ah[i] = __builtin_bit_cast(f16x8, row[2 * c + half]);
        al[i] = __builtin_bit_cast(f16x8, row[4 + 2 * c + half]);
      }
#pragma unroll
      for (int jj = 0; jj < NT; ++jj) {
        const u32x4* row = S + (BM + wn * 96 + jj * 32 + l31) * RQ;
        bh[jj] = __builtin_bit_cast(f16x8, row[2 * c + half]);
        bl[jj] = __builtin_bit_cast(f16x8, row[4 + 2 * c + half]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < NT; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[jj], acc[i][jj], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < NT; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[jj], acc[i][jj], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < NT; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[jj], acc[i][jj], 0, 0, 0);
    }
  };

  // Row statistic of the workgroup.  Stage 1: every wave reduces its 96 columns of each of its 64 rows
  // (DPP) and the upper-row lanes write the per-wave partials to part[row][N-wave]; stage 2: thread `row`
  // adds the four partials in a fixed order (deterministic) into tot[row]; stage 3: each lane fetches the
  // totals of its 32 rows as 8 float4 (rows r&3 are consecutive).  LDS addresses are one per-lane base
  // (re-derived per call, so that nothing is hoisted out of the tile loop and spilled) + constants.
  auto block_row_sum = [&](float (&s)[2][16], float* part) {
    float* tot = part + BM * 4;
    float m0 = 0.f, m1 = 0.f;
    const int l15 = lane & 15;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[0][r] = half_wave_sum_hi(s[0][r]);
      s[1][r] = half_wave_sum_hi(s[1][r]);
      m0 = (l15 == r) ? s[0][r] : m0;
      m1 = (l15 == r) ? s[1][r] : m1;
    }
    int rbase = wm * 64 + 4 * half;  // first tile row of this lane's row set
    asm volatile("" : "+v"(rbase));
    if (lane & 16) {  // lanes holding complete half-wave sums; lane l15 publishes rows r = l15 of both i
      const int rr = (l15 & 3) + 8 * (l15 >> 2);
      part[(rbase + rr) * 4 + wn] = m0;
      part[(rbase + 32 + rr) * 4 + wn] = m1;
    }
    __syncthreads();
    if (tid < BM) {
      const float4 q = *reinterpret_cast<const float4*>(part + tid * 4);
      tot[tid] = (q.x + q.y) + (q.z + q.w);
    }
    __syncthreads();
    const float* rd = tot + rbase;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 q = *reinterpret_cast<const float4*>(rd + i * 32 + 8 * g4);
        s[i][4 * g4 + 0] = q.x;
        s[i][4 * g4 + 1] = q.y;
        s[i][4 * g4 + 2] = q.z;
        s[i][4 * g4 + 3] = q.w;
      }
  };

  // Register budget (256 per lane at 2 waves/SIMD): 96 accumulators + ONE prefetch set (32) stay live
  // across the epilogue; the second set is re-issued after it (see the stream loop below).
  // Addresses are (wave-uniform row base in SGPRs) + (one per-lane offset): a per-row VGPR address
  // pair would cost 64 registers and spill.
  auto epilogue = [&](int ti, int g_next) {   // M % 128 == 0 (host guarantees): every tile is full
    if constexpr (PLAIN >= 0) {
      const int tile = first + ti * stride;
      const int prow0 = __builtin_amdgcn_readfirstlane((tile / tiles_n) * BM + wm * 64);
      const int n0 = (tile % tiles_n) * BN;
      int poff = 4 * half * p.N + n0 + wn * 96 + l31;
      asm volatile("" : "+v"(poff));
      float* pc = p.C + (size_t)prow0 * p.N;
      float pb[NT];
#pragma unroll
      for (int jj = 0; jj < NT; ++jj) pb[jj] = p.bias[n0 + wn * 96 + jj * 32 + l31];
      gload(ra1, rw1, g_next);  // ahead of the stores (see the LayerNorm epilogue)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int jj = 0; jj < NT; ++jj)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float o = acc[i][jj][r] * p.out_scale + pb[jj];
            if constexpr (PLAIN == EPI_BIAS_GELU) o = gelu_erf16(o);
            (pc + (size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * p.N)[poff + jj * 32] = o;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      return;
    }
    const int row0 = __builtin_amdgcn_readfirstlane((first + ti * stride) * BM + wm * 64);
    int loff = 4 * half * BN + wn * 96 + l31;   // lane part of every element offset
    asm volatile("" : "+v"(loff));               // re-derived per tile: nothing address-like is hoisted out of the tile loop
    const float* rbase = p.resid + (size_t)row0 * BN;
    float* cbase = p.C + (size_t)row0 * BN;
    float s[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[i][r] = 0.f;
    // residual: 6 batches (jj, i) of 16 loads, two batches in flight
    float rv[2][16];
    float bz[NT];
#pragma unroll
    for (int jj = 0; jj < NT; ++jj) bz[jj] = p.bias[wn * 96 + jj * 32 + l31];
#pragma unroll
    for (int r = 0; r < 16; ++r) rv[0][r] = DBG == 1 ? 0.f : (rbase + ((r & 3) + 8 * (r >> 2)) * BN)[loff];
#pragma unroll
    for (int b = 0; b < 2 * NT; ++b) {
      const int jj = b >> 1, i = b & 1;
      if (b + 1 < 2 * NT) {
        const int jn = (b + 1) >> 1, in = (b + 1) & 1;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          rv[(b + 1) & 1][r] = DBG == 1 ? 0.f : (rbase + (in * 32 + (r & 3) + 8 * (r >> 2)) * BN)[loff + jn * 32];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[i][jj][r] * p.out_scale + bz[jj] + rv[b & 1][r];
        acc[i][jj][r] = v;
        s[i][r] += v;
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the batches apart (register pressure)
    }
    float gm[NT], bt[NT];  // fetched here (the residual registers are free again), long before the stores
#pragma unroll
    for (int jj = 0; jj < NT; ++jj) {
      gm[jj] = p.gamma[wn * 96 + jj * 32 + l31];
      bt[jj] = p.beta[wn * 96 + jj * 32 + l31];
    }
    if constexpr (DBG != 3) block_row_sum(s, red);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float mean = s[i][r] * (1.0f / BN);
        float t = 0.f;
#pragma unroll
        for (int jj = 0; jj < NT; ++jj) {
          const float dl = acc[i][jj][r] - mean;
          acc[i][jj][r] = dl;
          t += dl * dl;
        }
        s[i][r] = t;
      }
    if constexpr (DBG != 3) block_row_sum(s, red + BM * 5);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[i][r] = 1.0f / sqrtf(s[i][r] * (1.0f / BN) + p.eps);
    // The second prefetch set of the next tile goes out BEFORE the 96 stores: vmcnt retires in order, so a
    // load issued behind them could only be consumed after every store of this tile has drained.
    gload(ra1, rw1, g_next);
    __builtin_amdgcn_sched_barrier(0);
    const bool do_store = DBG != 2 || p.M < 0;
#pragma unroll
    for (int jj = 0; jj < NT; ++jj) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float o = acc[i][jj][r] * s[i][r] * gm[jj] + bt[jj];
          if (do_store) (cbase + (i * 32 + (r & 3) + 8 * (r >> 2)) * BN)[loff + jj * 32] = o;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // stream positions: even g -> LDS buffer 0 / register set 1, odd g -> buffer 1 / set 0
  auto pair = [&](int g, auto last) {
    compute(0);
    lstore(ra0, rw0, 1);
    gload(ra0, rw0, g + 3);
    __syncthreads();
    compute(1);
    lstore(ra1, rw1, 0);  // position g+2: for the last k-pair of a tile this is already the NEXT tile
    if constexpr (!decltype(last)::value) gload(ra1, rw1, g + 4);
    __syncthreads();
  };
  gload(ra1, rw1, 0);
  lstore(ra1, rw1, 0);
  gload(ra0, rw0, 1);
  gload(ra1, rw1, 2);
  __syncthreads();
  const int npairs = nk / 2;  // >= 2 (host guarantees)
  for (int ti = 0; ti < cnt; ++ti) {
    const int g0 = ti * nk;
    pair(g0, StreamNo{});
    for (int pi = 1; pi < npairs - 1; ++pi) pair(g0 + 2 * pi, StreamNo{});
    pair(g0 + nk - 2, StreamYes{});   // the load of stream position g0+nk+2 is issued inside the epilogue
    epilogue(ti, g0 + nk + 2);
    zero_acc();
  }
}

bool launch_gemm_f16x3_ln(const float* A, const void* Wp, float w_scale, const float* bias, const float* resid,
                          const float* gamma, const float* beta, float eps, float* C, int M, int N, int K,
                          hipStream_t s) {
  if (N != 384 || K % 64 != 0 || K < 128 || M % 128 != 0) return false;
  constexpr int smem = 2 * (128 + 384) * 9 * 16 + 2 * 128 * 5 * 4;  // 152,576 B
  static bool attr_set = false;
  static int n_cu = 256;
  static const int dbg = env_int("FDMI_LN_DBG", 0);  // ablation only
  if (!attr_set) {
    for (const void* f : {reinterpret_cast<const void*>(&gemm_f16x3_ln_kernel<0>), reinterpret_cast<const void*>(&gemm_f16x3_ln_kernel<1>),
                          reinterpret_cast<const void*>(&gemm_f16x3_ln_kernel<2>), reinterpret_cast<const void*>(&gemm_f16x3_ln_kernel<3>)})
      (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      n_cu = prop.multiProcessorCount;
    attr_set = true;
  }
  const float a_scale = 16.0f;
  GemmLnArgs p{A, static_cast<const u32x4*>(Wp), bias, resid, gamma, beta, C, M, K, a_scale, 1.0f / (a_scale * w_scale), eps, N};
  const int ntiles = (M + 127) / 128;
  int grid = n_cu / 8 * 8;
  if (grid > ntiles) grid = (ntiles + 7) / 8 * 8;
  switch (dbg) {
    case 1: hipLaunchKernelGGL(gemm_f16x3_ln_kernel<1>, dim3(grid), dim3(512), smem, s, p); break;
    case 2: hipLaunchKernelGGL(gemm_f16x3_ln_kernel<2>, dim3(grid), dim3(512), smem, s, p); break;
    case 3: hipLaunchKernelGGL(gemm_f16x3_ln_kernel<3>, dim3(grid), dim3(512), smem, s, p); break;
    default: hipLaunchKernelGGL(gemm_f16x3_ln_kernel<0>, dim3(grid), dim3(512), smem, s, p); break;
  }
  return true;
}

// "Wide tile" variant of the plain GEMMs (see gemm_f16x3_ln_kernel; default, FDMI_GEMM_WIDE=0 turns it off:
// measured QKV 199 -> 191 us, FFN-up 153 -> 140, head 83 -> 75): false when disabled or the shape does not
// fit (N % 384, M % 128, K % 64).
bool launch_gemm_f16x3_wide(int epilogue, const float* A, const void* Wp, float w_scale, const float* bias, float* C,
                            int M, int N, int K, hipStream_t s) {
  static const int enabled = env_int("FDMI_GEMM_WIDE", 1);
  if (!enabled || (epilogue != EPI_BIAS && epilogue != EPI_BIAS_GELU) || N % 384 != 0 || K % 64 != 0 || K < 128 ||
      M % 128 != 0)
    return false;
  constexpr int smem = 2 * (128 + 384) * 9 * 16 + 2 * 128 * 5 * 4;
  static bool attr_set = false;
  static int n_cu = 256;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x3_ln_kernel<0, EPI_BIAS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x3_ln_kernel<0, EPI_BIAS_GELU>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      n_cu = prop.multiProcessorCount;
    attr_set = true;
  }
  const float a_scale = 16.0f;
  GemmLnArgs p{A, static_cast<const u32x4*>(Wp), bias, nullptr, nullptr, nullptr, C, M, K, a_scale, 1.0f / (a_scale * w_scale), 0.f, N};
  const int ntiles = (M / 128) * (N / 384);
  int grid = n_cu / 8 * 8;
  if (grid > ntiles) grid = (ntiles + 7) / 8 * 8;
  if (epilogue == EPI_BIAS) hipLaunchKernelGGL((gemm_f16x3_ln_kernel<0, EPI_BIAS>), dim3(grid), dim3(512), smem, s, p);
  else hipLaunchKernelGGL((gemm_f16x3_ln_kernel<0, EPI_BIAS_GELU>), dim3(grid), dim3(512), smem, s, p);
  return true;
}

template <int EPI, int PF, int WM, int AL>
static void launch_one(const GemmSplitArgs& p, hipStream_t s) {
  constexpr int BM = 64 * WM;
  constexpr int smem = 2 * (BM + 128) * 9 * 16;  // WM=4: 110,592 B; WM=2: 73,728 B
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x3_kernel<EPI, PF, WM, AL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + 127) / 128);
  hipLaunchKernelGGL((gemm_f16x3_kernel<EPI, PF, WM, AL>), dim3(tiles), dim3(128 * WM), smem, s, p);
}

template <int WM, int AL>
static void launch_pf(int epilogue, const GemmSplitArgs& p, hipStream_t s) {
  switch (epilogue) {
    case EPI_BIAS: launch_one<EPI_BIAS, 2, WM, AL>(p, s); break;
    case EPI_BIAS_GELU: launch_one<EPI_BIAS_GELU, 2, WM, AL>(p, s); break;
    default: launch_one<EPI_BIAS_RESID, 2, WM, AL>(p, s); break;
  }
}

// experiment knobs (environment, read once): FDMI_GEMM_BM = 128|256 rows per workgroup,
// FDMI_GEMM_AL = 0|1 A-staging layout
static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

template <int DBG>
static void launch_dbg(const GemmSplitArgs& p, hipStream_t s) {
  constexpr int smem = 2 * (256 + 128) * 9 * 16;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x3_kernel<EPI_BIAS, 2, 4, 1, DBG>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  const int tiles = ((p.M + 255) / 256) * ((p.N + 127) / 128);
  hipLaunchKernelGGL((gemm_f16x3_kernel<EPI_BIAS, 2, 4, 1, DBG>), dim3(tiles), dim3(512), smem, s, p);
}

void launch_gemm_f16x3(int epilogue, const float* A, const void* Wp, float w_scale, const float* bias,
                       const float* resid, float* C, int M, int N, int K, hipStream_t s) {
  static const int al = env_int("FDMI_GEMM_AL", 1) == 0 ? 0 : 1;
  static const int dbg = env_int("FDMI_GEMM_DBG", 0);  // ablation only (see DBG above)
  static const int bm = env_int("FDMI_GEMM_BM", 256) == 128 ? 128 : 256;
  const float a_scale = 16.0f;  // |a| < 4094 stays finite in fp16; lo of |a| > 0.008 is a normal fp16
  GemmSplitArgs p{A, static_cast<const u32x4*>(Wp), bias, resid, C, M, N, K, a_scale, 1.0f / (a_scale * w_scale)};
  if (dbg == 0 && !resid && launch_gemm_f16x3_wide(epilogue, A, Wp, w_scale, bias, C, M, N, K, s)) return;
  static const int persist = env_int("FDMI_GEMM_PERSIST", 1);
  if (persist && dbg == 0 && (K / 32) % 2 == 0 && K >= 128) {
    static const int pbm = env_int("FDMI_GEMM_PBM", 256);  // rows per persistent workgroup: 256 (8 waves) | 128 (4 waves, 2 per CU)
#define FD_PERSIST(W)                                                     \
  switch (epilogue) {                                                     \
    case EPI_BIAS: launch_persist<EPI_BIAS, 0, W>(p, s); break;           \
    case EPI_BIAS_GELU: launch_persist<EPI_BIAS_GELU, 0, W>(p, s); break; \
    default: launch_persist<EPI_BIAS_RESID, 0, W>(p, s); break;           \
  }
    static const int var = env_int("FDMI_GEMM_VAR", 0);  // 3: ping-pong wave schedule (256-row workgroups)
    if (var == 3 && pbm != 128) {
      switch (epilogue) {
        case EPI_BIAS: launch_persist<EPI_BIAS, 3, 4>(p, s); break;
        case EPI_BIAS_GELU: launch_persist<EPI_BIAS_GELU, 3, 4>(p, s); break;
        default: launch_persist<EPI_BIAS_RESID, 3, 4>(p, s); break;
      }
      return;
    }
    if (pbm == 128) { FD_PERSIST(2) } else { FD_PERSIST(4) }
#undef FD_PERSIST
    return;
  }
  if (dbg >= 1 && dbg <= 3) {
    if (dbg == 1) launch_dbg<1>(p, s);
    else if (dbg == 2) launch_dbg<2>(p, s);
    else launch_dbg<3>(p, s);
    return;
  }
  if (bm == 256) {
    if (al == 0) launch_pf<4, 0>(epilogue, p, s);
    else launch_pf<4, 1>(epilogue, p, s);
  } else {
    if (al == 0) launch_pf<2, 0>(epilogue, p, s);
    else launch_pf<2, 1>(epilogue, p, s);
  }
}

}  // namespace fdmi
