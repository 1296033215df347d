// LayerNorm-fused token GEMM for launches of FEW ROWS (round 4): out = LN(A W^T + bias + resid) * gamma + beta with N = 384 columns
// and K = 384 or 768 -- BertSelfOutput (attention output projection) and BertOutput (FFN down projection) of HF 4.11.3, called at
// foldingdiff/modelling.py:473-480.  Same split arithmetic, same image layouts and -- operation for operation -- the same epilogue
// as gi::gemm_img_kernel<EPI_IMG_LN> (gemm_img.hip), so the two kernels agree BIT FOR BIT (tests/test_gpu_parity.py:
// test_few_rows_gemm_path_leaves_every_bit_of_the_model_output_unchanged) and a sequence's result does not depend on the size of the
// batch it is sampled in.
//
// Why: the tile kernel works on 128-row tiles with two loader waves feeding a ring; a launch costs at least one tile's k-loop per
// CU (17-19 us at K = 384, more at 768) even when only a handful of CUs have a tile at all.  At batch 1-16 the 24 LayerNorm GEMMs of
// a reverse step were 0.45 ms of its 1.07.  Here ONE WORKGROUP OWNS A 32-ROW GROUP (rows / 32 workgroups: 32 at batch 8):
//   * the group's A rows (48 or 96 KiB, contiguous in HBM, already in fragment order) land in LDS with one burst of LDS-DMA;
//   * four waves = the four 96-column quarters of the row (the tile kernel's `wn`), three 32 x 32 accumulators each, a lane owns a
//     token row (D^T = W A^T);
//   * the weights are NOT staged through LDS: no two waves of the workgroup want the same ones, so each wave fetches its own
//     fragments from L2 straight into registers, two k-tiles ahead (12 x 16 bytes per lane and k-tile);
//   * epilogue as in gemm_img.hip: + bias + residual (sum in the same order), row sums through LDS in the same fixed order,
//     centred second moment, normalise, split, store.
// Every workgroup streams the whole weight matrix from L2 (0.6 / 1.2 MB), which is why this is the path of few rows only.
#include <cstdlib>

#include "fdmi_kernels.h"
#include "img_common.h"

namespace fdmi {
namespace lr {

constexpr int BN = 384;  // columns = one LayerNorm row

template <int V> struct IC { static constexpr int value = V; };

template <int NKT>
__global__ __launch_bounds__(256) void gemm_ln_rows_kernel(GemmImgArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int A_BYTES = NKT * 4096;
  constexpr int OFF_PAR = A_BYTES;                  // bias | gamma | beta (at the output image's scale)
  constexpr int OFF_RED = OFF_PAR + 3 * BN * 4;     // 2 x part[32][4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int g = blockIdx.x;                         // the 32-row group
  constexpr int nb = BN >> 5;
  if (g * 32 >= p.dims[1]) return;                  // (a group beyond the launch's rows: the grid covers the workspace's capacity)

  {  // the group's A rows: NKT * 4 pieces of 1 KiB, NKT per wave
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(p.A) + (size_t)g * A_BYTES, 0, A_BYTES, 0x00020000);
#pragma unroll
    for (int k = 0; k < NKT; ++k) {
      const int piece = wn + 4 * k;
      dma16(rs, (lds_ptr_t)(smem) + piece * 1024, lane * 16, piece * 1024);
    }
  }
  {
    float* par = reinterpret_cast<float*>(smem + OFF_PAR);
    for (int i = tid; i < BN; i += 256) {
      par[i] = p.bias[i];
      par[BN + i] = p.gamma[i];
      par[2 * BN + i] = p.beta[i] * p.out_scale;    // beta at the OUTPUT image's scale (a power of two: exact)
    }
  }

  // ---- weights: row R = wn * 96 + 32 jn + l31 of W, k-tile kt, k16 step c: unit 2c + half (hi), 4 + 2c + half (lo) of the weight
  // image ([384-row tile][k-tile][48 KiB stage], pieces of 8 rows, unit u of row r at ((u ^ (piece & 1)) * 8 + r % 8) * 16: api.hip)
  struct WT {
    f16x8 h[3][2], l[3][2];
  };
  const unsigned char* wrow[3];
  int sw[3];
#pragma unroll
  for (int jn = 0; jn < 3; ++jn) {
    const int R = wn * 96 + 32 * jn + l31, piece = R >> 3;
    wrow[jn] = p.W + piece * 1024 + (R & 7) * 16;
    sw[jn] = piece & 1;
  }
  auto load_w = [&](WT& w, int kt) {
#pragma unroll
    for (int jn = 0; jn < 3; ++jn)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        w.h[jn][c] = *reinterpret_cast<const f16x8*>(wrow[jn] + (size_t)kt * 49152 + (((2 * c + half) ^ sw[jn]) << 7));
        w.l[jn][c] = *reinterpret_cast<const f16x8*>(wrow[jn] + (size_t)kt * 49152 + (((4 + 2 * c + half) ^ sw[jn]) << 7));
      }
  };
  WT w0, w1, w2;
  load_w(w0, 0);
  load_w(w1, 1);

  f32x16 acc[3];
#pragma unroll
  for (int jn = 0; jn < 3; ++jn)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[jn][r] = 0.f;

  FD_WAIT_VM(0);      // the A pieces (and the first two weight k-tiles, requested at the same moment) have landed ...
  barrier_keep_vm();  // ... those of every wave: the group and the parameter image are in LDS

  const unsigned char* ab = smem + l31 * 16 + half * 512;
  auto mm = [&](const WT& w, int kt) {  // per accumulator: wh ah | wh al | wl ah per k16 step, k ascending -- the tile kernel's order
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const f16x8 ah = *reinterpret_cast<const f16x8*>(ab + kt * 4096 + c * 1024);
      const f16x8 al = *reinterpret_cast<const f16x8*>(ab + kt * 4096 + c * 1024 + 2048);
#pragma unroll
      for (int jn = 0; jn < 3; ++jn) acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.h[jn][c], ah, acc[jn], 0, 0, 0);
#pragma unroll
      for (int jn = 0; jn < 3; ++jn) acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.h[jn][c], al, acc[jn], 0, 0, 0);
#pragma unroll
      for (int jn = 0; jn < 3; ++jn) acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.l[jn][c], ah, acc[jn], 0, 0, 0);
    }
  };
  // residual half-blocks (one hi and one lo 16-byte unit per lane), requested while the last k-tiles are multiplied
  const unsigned char* rbase = p.resid + (size_t)g * nb * 4096;
  const unsigned roff = (unsigned)(l31 * 16 + half * 1024);
  u32x4 rres[6][2];
  auto request_all = [&]() {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int jn = i >> 1, hb = i & 1;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<unsigned char*>(rbase + (size_t)(wn * 3 + jn) * 4096), 0, 4096, 0x00020000);
      rres[i][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(roff + hb * 512), 0, 0));
      rres[i][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(roff + 2048 + hb * 512), 0, 0));
    }
  };
  static_assert(NKT % 3 == 0, "the weight buffers rotate in threes");
#pragma unroll 1
  for (int kt = 0; kt < NKT - 3; kt += 3) {
    load_w(w2, kt + 2);
    mm(w0, kt);
    load_w(w0, kt + 3);
    mm(w1, kt + 1);
    load_w(w1, kt + 4);
    mm(w2, kt + 2);
  }
  load_w(w2, NKT - 1);
  mm(w0, NKT - 3);
  request_all();
  mm(w1, NKT - 2);
  mm(w2, NKT - 1);

  // ---- epilogue: gi::gemm_img_kernel<EPI_IMG_LN> for one 32-row group (im = 0), statement for statement
  const float* par = reinterpret_cast<const float*>(smem + OFF_PAR);
  float* red = reinterpret_cast<float*>(smem + OFF_RED);
  const float os = p.acc_scale;
  const float inv_n = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 1.0f / (float)BN)));
  float s = 0.f;
  auto pass1_half = [&](auto JN, auto HB, const u32x4& rh, const u32x4& rl) {
    constexpr int jn = decltype(JN)::value, hb = decltype(HB)::value;
    const int cb = wn * 3 + jn;
    unsigned H[4] = {rh[0], rh[1], rh[2], rh[3]}, Lo[4] = {rl[0], rl[1], rl[2], rl[3]};
    swap32(H[0], H[2]);
    swap32(H[1], H[3]);
    swap32(Lo[0], Lo[2]);
    swap32(Lo[1], Lo[3]);
    const float ri = p.resid_inv;
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
      const int q = hb + 2 * qi;
      const float4 b4 = *reinterpret_cast<const float4*>(par + cb * 32 + 8 * q + 4 * half);
      const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int dd = 0; dd < 2; ++dd) {
        float v0 = __builtin_fmaf(acc[jn][4 * q + 2 * dd], os, bb[2 * dd]);
        float v1 = __builtin_fmaf(acc[jn][4 * q + 2 * dd + 1], os, bb[2 * dd + 1]);
        v0 = fma_mix_lo(H[2 * qi + dd], ri, v0);
        v1 = fma_mix_hi(H[2 * qi + dd], ri, v1);
        v0 = fma_mix_lo(Lo[2 * qi + dd], ri, v0);
        v1 = fma_mix_hi(Lo[2 * qi + dd], ri, v1);
        acc[jn][4 * q + 2 * dd] = v0;
        acc[jn][4 * q + 2 * dd + 1] = v1;
        s += v0;
        s += v1;
      }
    }
  };
  pass1_half(IC<0>{}, IC<0>{}, rres[0][0], rres[0][1]);
  pass1_half(IC<0>{}, IC<1>{}, rres[1][0], rres[1][1]);
  pass1_half(IC<1>{}, IC<0>{}, rres[2][0], rres[2][1]);
  pass1_half(IC<1>{}, IC<1>{}, rres[3][0], rres[3][1]);
  pass1_half(IC<2>{}, IC<0>{}, rres[4][0], rres[4][1]);
  pass1_half(IC<2>{}, IC<1>{}, rres[5][0], rres[5][1]);
  // row sums: in-lane (48 columns) + the other half-wave + the four column quarters through LDS, fixed order
  auto block_sum = [&](float& t, float* part) {
    {
      unsigned lo_side = __builtin_bit_cast(unsigned, t), hi_side = lo_side;
      swap32(lo_side, hi_side);
      t = __builtin_bit_cast(float, lo_side) + __builtin_bit_cast(float, hi_side);
    }
    if (half == 0) part[l31 * 4 + wn] = t;
    barrier_keep_vm();
    const float4 q4 = *reinterpret_cast<const float4*>(part + l31 * 4);
    t = (q4.x + q4.y) + (q4.z + q4.w);
  };
  block_sum(s, red);
  float t2 = 0.f;
  {
    float mean = s * inv_n;
    asm volatile("" : "+v"(mean));  // (as in gemm_img.hip: a value, not a product)
#pragma unroll
    for (int jn = 0; jn < 3; ++jn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float dl = acc[jn][r] - mean;
        acc[jn][r] = dl;
        t2 = __builtin_fmaf(dl, dl, t2);
      }
  }
  block_sum(t2, red + 32 * 4);
  const float rstd = (1.0f / sqrtf(__builtin_fmaf(t2, inv_n, p.eps))) * p.out_scale;  // at the output image's scale (beta in LDS too)
#pragma unroll
  for (int jn = 0; jn < 3; ++jn) {
    const int cb = wn * 3 + jn;
    float o[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 g4 = *reinterpret_cast<const float4*>(par + BN + cb * 32 + 8 * q + 4 * half);
      const float4 e4 = *reinterpret_cast<const float4*>(par + 2 * BN + cb * 32 + 8 * q + 4 * half);
      o[4 * q + 0] = __builtin_fmaf(acc[jn][4 * q + 0] * rstd, g4.x, e4.x);
      o[4 * q + 1] = __builtin_fmaf(acc[jn][4 * q + 1] * rstd, g4.y, e4.y);
      o[4 * q + 2] = __builtin_fmaf(acc[jn][4 * q + 2] * rstd, g4.z, e4.z);
      o[4 * q + 3] = __builtin_fmaf(acc[jn][4 * q + 3] * rstd, g4.w, e4.w);
    }
    store_group_block(p.out + ((size_t)g * nb + cb) * 4096, o, 1.0f, l31, half);
  }
}

template <int NKT>
static void launch(const GemmImgArgs& p, int max_rows, hipStream_t s) {
  constexpr int SMEM = NKT * 4096 + 3 * BN * 4 + 2 * 32 * 4 * 4;
  static bool attr_set[64] = {false};  // (the attribute is per device)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ln_rows_kernel<NKT>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr_set[dev] = true;
  }
  // one workgroup per 32-row group of the workspace's capacity (groups beyond the actual row count compute padding rows, as the
  // tile kernel's last tile does)
  hipLaunchKernelGGL((gemm_ln_rows_kernel<NKT>), dim3((max_rows + 127) / 128 * 4), dim3(256), SMEM, s, p);
}

}  // namespace lr

bool gemm_ln_rows_supported(const GemmImgArgs& p) {
  return p.N == lr::BN && (p.K == 384 || p.K == 768) && p.out_f32 == nullptr && p.resid != nullptr;
}

void launch_gemm_ln_rows(const GemmImgArgs& p, int max_rows, hipStream_t s) {
  if (p.K == 384) lr::launch<12>(p, max_rows, s);
  else lr::launch<24>(p, max_rows, s);
}

}  // namespace fdmi
