// Token GEMMs on row images, 256 x 384 tiles: the same products, images and epilogues as gemm_img.hip with HALF the weight
// ingest per matrix instruction.
//
// Why a second tile shape: the k-loop of gemm_img.hip (128 x 384 tile, 8 compute + 2 loader waves) runs at ~2900 cycles per
// k-tile for 2304 matrix cycles because a CU ingests only ~22 B/clk out of L2 with every CU streaming (64 KiB per k-tile:
// profiles/r03_pingpong_stamps.log).  A 256-row tile reads the same 48 KiB of weights per k-tile for twice the matrix work:
// 80 KiB per 4608 matrix cycles = 17 B/clk, under what the CU can take, so the loop is bound by the matrix pipe.
//
// What that costs: 256 x 384 fp32 accumulators are 384 registers per lane of FOUR waves -- one wave per SIMD with the full
// 512-register file (accumulators in AGPRs + VGPRs).  So there are no loader waves (every wave issues its share of the
// LDS-DMA pieces between its MFMAs) and no second wave per SIMD to hide anything: fragment reads run one MFMA pass ahead
// of their use, the DMA three stages ahead.
//
//  * a stage is ONE k16 step: W 384 columns x (hi 32 B | lo 32 B) = 24 KiB + A 256 rows x 64 B = 16 KiB, ring of three stages.
//    In LDS a stage is [32-row group][plane hi | lo][half-wave][row % 32][16 B]: the fragment of (group, plane) is the 1 KiB
//    at lane * 16 -- one ds_read_b128 per lane, conflict free, the stage and group offsets are immediates.
//  * both operands are read from the SAME HBM images as gemm_img.hip: the grouped activation image gives a (group, plane)
//    piece as 1 KiB of contiguous bytes; the weight image (the 48 KiB k-tile stage of gemm_img.hip) is gathered by 128-byte
//    lines (per-lane source offsets).
//  * per stage a wave runs 72 MFMAs as three passes of 24 (w_hi a_hi | w_hi a_lo | w_lo a_hi) over its 192 x 128 part of the
//    tile (6 x 4 MFMA tiles, swapped form: lane = token row), ONE workgroup barrier per stage before the third pass; behind
//    it the wave requests the stage three ahead and reads the first fragments of the next one.
//  * epilogues: as in gemm_img.hip, per 32 x 32 block.
//
// Used when the row count fills whole 256-row tiles evenly over the CUs (launch_gemm_img decides); everything else runs the
// 128-row kernel.
#include <cstdlib>
#include <type_traits>

#include "fdmi_kernels.h"
#include "img_common.h"

namespace fdmi {
namespace gb {

template <int V> using IC = std::integral_constant<int, V>;

constexpr int BM = 256, BN = 384, NTHR = 256;
constexpr int W_ST = BN * 64, A_ST = BM * 64, STAGE = W_ST + A_ST;  // 24,576 + 16,384 = 40,960 B per k16 stage
constexpr int NS = 3;
constexpr int OFF_PAR = NS * STAGE;                 // 122,880: bias | gamma | beta
constexpr int OFF_RED = OFF_PAR + 3 * BN * 4;       // 127,488: 2 x part[256][2]
constexpr int OFF_RI = OFF_RED + 2 * BM * 2 * 4;    // 131,584: (sequence, position) of each wave's 128 token rows
constexpr int SMEM = OFF_RI + 4 * 1024;             // 135,680 B
constexpr int W_KTILE = BN * 128;                   // bytes of one k-tile (32 k) of a 384-column weight tile in HBM
#ifndef FDMI_BIG_STAGGER
#define FDMI_BIG_STAGGER 0  // wave w issues its copy piece behind the w-th MFMA of a row: the four waves' pieces do not meet in the address unit
#endif
#ifndef FDMI_BIG_DBG
#define FDMI_BIG_DBG 0  // ablation builds (wrong results): 1 no DMA in the loop, 2 no MFMAs, 4 no epilogue, 8 linear W pieces, 16 no fragment reads
#endif
constexpr int NDMA = 10;                            // LDS-DMA instructions per wave and stage (6 W + 4 A)

// PROF (FDMI_STAMPS=1): workgroup 0 records s_memtime stamps per stage: stamps[EPI][wave][slot][6] = {stage top, after pass 2, after
// the barrier, after the DMA issue + first reads of the next stage, after pass 3, after the epilogue (a tile's last stage)};
// slot 63 = {s_memtime, s_memrealtime (100 MHz) at kernel start, the same at the end}
template <int EPI, bool PROF>
__global__ __launch_bounds__(NTHR) void gemm_big_kernel(GemmImgArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wid & 1, wm = wid >> 1;  // wave tile: rows wm*128 .. +127, columns wn*192 .. +191
  const int nk = p.K >> 5, rb = nk * 128, nks = nk * 2;  // k-tiles; bytes per image row; k16 stages per tile
  const int Mp = p.dims[1];
  const int tiles_m = (Mp + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN, ntiles = tiles_m * tiles_n;
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int tlo = (int)((long long)ntiles * xcd / 8), thi = (int)((long long)ntiles * (xcd + 1) / 8);
  const int first = tlo + jx, stride = per;
  const int cnt = first < thi ? (thi - first + stride - 1) / stride : 0;
  if (cnt == 0) return;
  const int G = cnt * nks;  // stream positions (stages)

  {  // parameters -> LDS (published by the first barrier); same images as gemm_img.hip
    float* par = reinterpret_cast<float*>(smem + OFF_PAR);
    if constexpr (EPI == EPI_IMG_LN) {
      for (int i = tid; i < BN; i += NTHR) {
        const bool ok = i < p.N;
        par[i] = ok ? p.bias[i] : 0.f;
        par[BN + i] = ok ? p.gamma[i] : 0.f;
        par[2 * BN + i] = ok ? p.beta[i] * p.out_scale : 0.f;
      }
    } else if constexpr (EPI == EPI_IMG_QKV) {
      const int nq = p.H * 32;
      for (int i = tid; i < 3 * BN; i += NTHR) {
        const float sc = i < nq ? p.q_scale : (i < 2 * nq ? p.k_scale : p.v_scale);
        par[i] = i < p.N ? p.bias[i] * sc : 0.f;
      }
    } else {
      for (int i = tid; i < 3 * BN; i += NTHR) par[i] = i < p.N ? p.bias[i] : 0.f;
    }
  }

  auto tile_mn = [&](int ti, int& m0, int& n0) {
    const int tile = first + ti * stride;
    m0 = (tile / tiles_n) * BM;
    n0 = (tile - (tile / tiles_n) * tiles_n) * BN;
  };

  // ---------------------------------------------------------------- the copy stream (every wave issues 10 of a stage's 40 pieces)
  // W piece j (0..23) = column group j / 2, plane j % 2 of the stage: lane (half, l31) fetches the 16-byte unit
  // 2c + half + 4 plane of weight row 32 g + l31 out of gemm_img's k-tile stage ([8-row piece][unit ^ (piece & 1)][row % 8]).
  // A piece j (0..15) = row group j / 2, plane j % 2: units 2c, 2c + 1 (+ 4 plane) of the group are 1 KiB of contiguous bytes.
  // Issue order: the ten pieces of a stage are spread over 48 MFMAs (one piece behind every fourth MFMA of pass 3 and of the
  // next stage's pass 1): issued in one burst behind the barrier they cost ~940 cycles per stage with the matrix pipe idle
  // (40 KiB through the CU's address unit; profiles/r03_bigtile_stamps.log).
  int i_ti = 0, i_ks = 0, i_slot = 0, i_m0, i_n0;
  tile_mn(0, i_m0, i_n0);
  auto issue_piece = [&](auto I) __attribute__((always_inline)) {
    constexpr int i = decltype(I)::value;
    if (FDMI_BIG_DBG & 1) return;
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const int kt = i_ks >> 1, c = i_ks & 1;
    lds_ptr_t dst = (lds_ptr_t)(smem) + i_slot * STAGE;
    if constexpr (i < 6) {
      const int l31 = ln & 31, hf = ln >> 5;
      const int wlane = (l31 >> 3) * 1024 + ((hf ^ ((l31 >> 3) & 1)) * 128) + (l31 & 7) * 16;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<unsigned char*>(p.W) + (size_t)i_n0 * rb, 0, BN * rb, 0x00020000);
      const int j = wid * 6 + i;
      dma16(rs, dst + j * 1024, (FDMI_BIG_DBG & 8) ? ln * 16 : wlane, kt * W_KTILE + c * 256 + (j >> 1) * 4096 + (j & 1) * 512);
    } else {
      const int groups = (Mp - i_m0) >> 5;  // row groups of this tile that exist (>= 4: Mp is a multiple of 128)
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<unsigned char*>(p.A) + (size_t)(i_m0 >> 5) * nk * 4096, 0, (groups < 8 ? groups : 8) * nk * 4096, 0x00020000);
      const int j = wid * 4 + (i - 6);
      int g = j >> 1;
      g = g < groups ? g : 0;  // the missing half of a last half tile: any rows, never stored
      dma16(rs, dst + W_ST + j * 1024, ln * 16, kt * 4096 + c * 1024 + g * nk * 4096 + (j & 1) * 2048);
    }
    if constexpr (i == NDMA - 1) {
      i_slot = i_slot == NS - 1 ? 0 : i_slot + 1;
      if (i_ti * nks + i_ks + 1 < G) {  // past the end: re-issue the last stage (lands in a free slot, never read)
        if (++i_ks == nks) {
          i_ks = 0;
          ++i_ti;
          tile_mn(i_ti, i_m0, i_n0);
        }
      }
    }
  };
  auto issue_first_half = [&]() __attribute__((always_inline)) {
    issue_piece(IC<0>{}); issue_piece(IC<1>{}); issue_piece(IC<2>{}); issue_piece(IC<3>{}); issue_piece(IC<4>{});
  };
  auto issue_second_half = [&]() __attribute__((always_inline)) {
    issue_piece(IC<5>{}); issue_piece(IC<6>{}); issue_piece(IC<7>{}); issue_piece(IC<8>{}); issue_piece(IC<9>{});
  };

  // ---------------------------------------------------------------- fragments and accumulators
  // 24 accumulator tiles [jn][im] = 384 registers: tiles jn 0..3 live in AGPRs (256), jn 4, 5 in VGPRs (128).  hipcc selects ONE
  // MFMA form per function (AGPR destination) and, with more accumulators than AGPRs, copies every tile through a[0:15]
  // around every MFMA; the matrix instructions are therefore inline assembly with the register class spelled out.  Nothing
  // the compiler knows about MFMA hazards applies to them: inside the k-loop two MFMAs on one accumulator are 24 MFMAs
  // apart, and the epilogue starts behind FD_MFMA_DRAIN().
  f32x16 accA[16], accV[8];
#define FD_ACC(jn, im) ((jn) < 4 ? accA[(jn) * 4 + (im)] : accV[((jn) - 4) * 4 + (im)])
#define FD_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")
  auto zero_acc = [&]() {
#pragma unroll
    for (int b = 0; b < 16; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) accA[b][r] = 0.f;
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) accV[b][r] = 0.f;
  };
  // 24 MFMAs, row jn of the wave tile after row jn; BASE >= 0: copy piece BASE + jn is issued behind row jn (jn < 5)
  auto mm24 = [&](const f16x8 (&wf)[6], const f16x8 (&af)[4], auto BASE) __attribute__((always_inline)) {
    constexpr int base = decltype(BASE)::value;
    if (FDMI_BIG_DBG & 2) {
#pragma unroll
      for (int jn = 0; jn < 6; ++jn) asm volatile("" ::"v"(wf[jn]));
#pragma unroll
      for (int im = 0; im < 4; ++im) asm volatile("" ::"v"(af[im]));
    }
#define FD_ROW(jn, CLS, ARR, IDX)                                                                                              \
  do {                                                                                                                         \
    _Pragma("unroll") for (int im = 0; im < 4; ++im) {                                                                         \
      if (!(FDMI_BIG_DBG & 2))                                                                                                 \
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : CLS(ARR[(IDX) * 4 + im]) : "v"(wf[jn]), "v"(af[im]));          \
      if constexpr (base >= 0 && (jn) < 5) {                                                                                   \
        if (FDMI_BIG_STAGGER ? wid == im : im == 3) issue_piece(IC<(base >= 0 ? base : 0) + ((jn) < 5 ? (jn) : 0)>{});         \
      }                                                                                                                        \
    }                                                                                                                          \
  } while (0)
    FD_ROW(0, "+a", accA, 0); FD_ROW(1, "+a", accA, 1); FD_ROW(2, "+a", accA, 2); FD_ROW(3, "+a", accA, 3);
    FD_ROW(4, "+v", accV, 0); FD_ROW(5, "+v", accV, 1);
#undef FD_ROW
  };
  // fragment (group, plane) = the 1 KiB at lane * 16
  auto ldw = [&](f16x8 (&d)[6], int slot, int plane) __attribute__((always_inline)) {
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const unsigned char* b = smem + slot * STAGE + wn * 6 * 2048 + plane * 1024 + ln * 16;
#pragma unroll
    for (int jn = 0; jn < 6; ++jn) d[jn] = *reinterpret_cast<const f16x8*>(b + jn * 2048);
  };
  auto lda = [&](f16x8 (&d)[4], int slot, int plane) __attribute__((always_inline)) {
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const unsigned char* b = smem + slot * STAGE + W_ST + wm * 4 * 2048 + plane * 1024 + ln * 16;
#pragma unroll
    for (int im = 0; im < 4; ++im) d[im] = *reinterpret_cast<const f16x8*>(b + im * 2048);
  };

  // ---------------------------------------------------------------- epilogues (block by block as in gemm_img.hip)
  auto epilogue = [&](int ti) {
    int m0, n0, ln;
    tile_mn(ti, m0, n0);
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const int l31 = ln & 31, half = ln >> 5;
    const float os = p.acc_scale;
    const float* par0 = reinterpret_cast<const float*>(smem + OFF_PAR);
    const bool live = m0 + wm * 128 < Mp;  // the second half of a last half tile has no rows
    auto bias4 = [&](int cbg, int q, float sc) -> float4 {
      if (cbg * 32 < 3 * BN) return *reinterpret_cast<const float4*>(par0 + cbg * 32 + 8 * q + 4 * half);
      float4 b = *reinterpret_cast<const float4*>(p.bias + cbg * 32 + 8 * q + 4 * half);
      b.x *= sc; b.y *= sc; b.z *= sc; b.w *= sc;
      return b;
    };
    if constexpr (EPI == EPI_IMG_GELU || EPI == EPI_IMG_BIAS) {
      const int nb = p.N >> 5;
      if (!live) return;
#pragma unroll
      for (int jn = 0; jn < 6; ++jn) {
        const int cb = (n0 >> 5) + wn * 6 + jn;  // wave-uniform
        if (cb >= nb) continue;
        float4 b4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b4[q] = bias4(cb, q, 1.0f);
#pragma unroll
        for (int im = 0; im < 4; ++im) {
          float o[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            o[4 * q + 0] = __builtin_fmaf(FD_ACC(jn, im)[4 * q + 0], os, b4[q].x);
            o[4 * q + 1] = __builtin_fmaf(FD_ACC(jn, im)[4 * q + 1], os, b4[q].y);
            o[4 * q + 2] = __builtin_fmaf(FD_ACC(jn, im)[4 * q + 2], os, b4[q].z);
            o[4 * q + 3] = __builtin_fmaf(FD_ACC(jn, im)[4 * q + 3], os, b4[q].w);
          }
          if constexpr (EPI == EPI_IMG_GELU) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              const gf2 g = gelu_erf2(gf2{o[r], o[r + 1]});
              o[r] = g[0];
              o[r + 1] = g[1];
            }
          }
          store_group_block(p.out + ((size_t)((m0 + wm * 128 + im * 32) >> 5) * nb + cb) * 4096, o, p.out_scale, l31, half);
        }
      }
    }
  };

  // ---------------------------------------------------------------- the stream
  //   stage s:  pass 1 (w_hi a_hi)  [reads: a_lo, w_lo]   pass 2 (w_hi a_lo)   vmcnt: own pieces of s + 1 landed   BARRIER s + 1
  //             pass 3 (w_lo a_hi)  [issue stage s + 3 into the slot of s; reads of s + 1: w_hi, a_hi]
  // The a_hi / a_lo buffers swap roles from one stage to the next (the buffer free during pass 3 is the a_lo one), so the
  // loop body is a k-tile = two stages.
#define FD_SB() __builtin_amdgcn_sched_barrier(0)
  const bool rec = PROF && blockIdx.x == 0 && p.stamps != nullptr;
  unsigned long long* st = PROF ? p.stamps + ((size_t)(EPI == EPI_IMG_QKV ? (int)EPI_IMG_QK : EPI) * 8 + wid) * 64 * 6 : nullptr;
  int slot = 0;
#define FD_STAMP(i) do { if (PROF) { if (rec && slot < 63) { int ln_; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln_)); if (ln_ == 0) st[slot * 6 + (i)] = __builtin_amdgcn_s_memtime(); } } } while (0)
  if (PROF) {
    if (rec) {
      int ln_;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln_));
      if (ln_ == 0) {
        st[63 * 6 + 0] = __builtin_amdgcn_s_memtime();
        st[63 * 6 + 1] = __builtin_amdgcn_s_memrealtime();
      }
    }
  }
  f16x8 Xa[6], Xb[6], Ya[4], Yb[4];
  int cs = 0;  // slot of the stage being computed
  auto next_slot = [&](int s) { return s == NS - 1 ? 0 : s + 1; };
  // one stage with a_hi in H and a_lo going to L; `more`: the next stage belongs to the same tile (fetch its first fragments)
  auto stage = [&](f16x8 (&H)[4], f16x8 (&L)[4], bool wait, auto MORE) __attribute__((always_inline)) {
    FD_SB();
    FD_STAMP(0);
    lda(L, cs, 1);
    ldw(Xb, cs, 1);
    FD_SB();
    mm24(Xa, H, IC<5>{});   // + the second half of the copy stage begun in the previous pass 3
    FD_SB();
    mm24(Xa, L, IC<-1>{});
    FD_SB();
    FD_STAMP(1);
    if (wait && !(FDMI_BIG_DBG & 1)) FD_WAIT_VM(NDMA);
    barrier_keep_vm();
    FD_SB();
    FD_STAMP(2);
    cs = next_slot(cs);
    if constexpr (decltype(MORE)::value) {
      ldw(Xa, cs, 0);
      lda(L, cs, 0);
    }
    FD_SB();
    FD_STAMP(3);
    mm24(Xb, H, IC<0>{});   // + the first half of the copy stage three ahead (into the slot this barrier freed)
    FD_SB();
    FD_STAMP(4);
    if constexpr (decltype(MORE)::value) ++slot;
  };

  issue_first_half(); issue_second_half();
  issue_first_half(); issue_second_half();
  issue_first_half();
  FD_WAIT_VM(NDMA + NDMA / 2);
  barrier_keep_vm();  // stage 0 landed (also publishes the parameter image)
  zero_acc();
  ldw(Xa, cs, 0);
  lda(Ya, cs, 0);
  for (int ti = 0; ti < cnt; ++ti) {
    // the first stage of a later tile: its pieces were waited for before the previous epilogue (whose stores are now
    // between them and the next stage's pieces in the queue)
    stage(Ya, Yb, ti == 0, IC<1>{});
    stage(Yb, Ya, true, IC<1>{});
    for (int kt = 1; kt + 1 < nk; ++kt) {
      stage(Ya, Yb, true, IC<1>{});
      stage(Yb, Ya, true, IC<1>{});
    }
    stage(Ya, Yb, true, IC<1>{});
    stage(Yb, Ya, true, IC<0>{});
    FD_MFMA_DRAIN();
    FD_WAIT_VM(NDMA / 2);  // this wave's pieces of the next tile's SECOND stage landed: its barrier follows the epilogue's stores in the queue
    if (!(FDMI_BIG_DBG & 4)) epilogue(ti);
    FD_STAMP(5);
    ++slot;
    zero_acc();
    if (ti + 1 < cnt) {  // (cs already points at the next tile's first stage, published by the last barrier)
      ldw(Xa, cs, 0);
      lda(Ya, cs, 0);
    }
  }
  FD_WAIT_VM(0);  // nothing may land in LDS after the workgroup has exited
  if (PROF) {
    if (rec) {
      int ln_;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln_));
      if (ln_ == 0) {
        st[63 * 6 + 2] = __builtin_amdgcn_s_memtime();
        st[63 * 6 + 3] = __builtin_amdgcn_s_memrealtime();
      }
    }
  }
#undef FD_STAMP
#undef FD_SB
#undef FD_ACC
#undef FD_MFMA_DRAIN
}

static int n_cu_of_current_device() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    hipDeviceProp_t prop;
    cached[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return cached[dev];
}

template <int EPI>
static void launch(const GemmImgArgs& p, int max_rows, hipStream_t s) {
  static bool attr_set[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_big_kernel<EPI, false>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_big_kernel<EPI, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr_set[dev] = true;
  }
  const int ntiles_max = ((max_rows + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  int grid = n_cu_of_current_device() / 8 * 8;
  if (grid > ntiles_max) grid = (ntiles_max + 7) / 8 * 8;
  if (grid < 8) grid = 8;
  if (p.stamps) hipLaunchKernelGGL((gemm_big_kernel<EPI, true>), dim3(grid), dim3(NTHR), SMEM, s, p);
  else hipLaunchKernelGGL((gemm_big_kernel<EPI, false>), dim3(grid), dim3(NTHR), SMEM, s, p);
}

}  // namespace gb

bool gemm_big_supported(int epilogue) { return epilogue == EPI_IMG_GELU || epilogue == EPI_IMG_BIAS; }

void launch_gemm_big(int epilogue, const GemmImgArgs& p, int max_rows, hipStream_t s) {
  switch (epilogue) {
    case EPI_IMG_GELU: gb::launch<EPI_IMG_GELU>(p, max_rows, s); break;
    default: gb::launch<EPI_IMG_BIAS>(p, max_rows, s); break;
  }
}

}  // namespace fdmi
