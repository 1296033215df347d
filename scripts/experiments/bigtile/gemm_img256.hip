// Token GEMMs on row images, 256 x 384 tiles: the products, images and epilogues of gemm_img.hip with 30 % fewer bytes
// between L2 and the CU per matrix instruction.
//
// Why: every launch of gemm_img.hip (128 x 384 tiles) takes the time its workgroup needs to move its bytes at ~20 B/clk --
// per tile 12 k-tiles x 64 KiB of operands plus the 192 KiB it stores (FFN-up: 964 KiB = 48.2 k cycles, measured 48.3 k;
// the same sum fits q|k|v, attn-out and FFN-down within 8 %).  Three quarters of those bytes are the
// weight tile, which every 128-row tile reads again.  A 256-row tile reads it once for twice the rows.
//
// What that costs: 256 x 384 fp32 accumulators are 192 registers per lane of EIGHT waves, so the workgroup is eight waves
// with 256 registers each (two per SIMD) and nothing else: no loader waves -- every wave issues five of a stage's forty
// LDS-DMA pieces between its MFMAs, and the other wave of its SIMD covers the issue stall.  (With ONE 512-register wave per
// SIMD the same tile is slower than the 128-row kernel: scripts/experiments/bigtile/README.md.)
//
//  * a stage is ONE k16 step: W 384 columns x (hi 32 B | lo 32 B) = 24 KiB + A 256 rows x 64 B = 16 KiB, ring of three stages.
//    In LDS a stage is [32-row group][plane hi | lo][half-wave][row % 32][16 B]: the fragment of (group, plane) is the 1 KiB
//    at lane * 16 -- one ds_read_b128 per lane, conflict free.
//  * both operands are read from the SAME HBM images as gemm_img.hip: the grouped activation image gives a (group, plane)
//    piece as 1 KiB of contiguous bytes; the weight image (gemm_img's 48 KiB k-tile stage) is gathered by 128-byte lines.
//  * 8 waves as 2 (M) x 4 (N), wave tile 128 x 96 = 4 x 3 MFMA tiles (swapped form: lane = token row).  Per stage a wave
//    runs 36 MFMAs as three passes of 12 (w_lo a_hi | w_hi a_hi | w_hi a_lo), ONE workgroup barrier per stage before the
//    third pass; behind it the wave requests its pieces of the stage three ahead and the first fragments of the next one.
//  * the accumulators of eight MFMA tiles live in AGPRs, four in VGPRs; the MFMAs are inline assembly with the register class
//    spelled out (hipcc selects one MFMA form per function and, with accumulators in both files, copies every tile through
//    a[0:15] around every MFMA).  Nothing the compiler knows about MFMA hazards applies to them: inside the k-loop two MFMAs
//    on one accumulator are 12 MFMAs apart, and the epilogue starts behind FD_MFMA_DRAIN().
//  * epilogues: as in gemm_img.hip, per 32 x 32 block.
#include <cstdlib>
#include <type_traits>

#include "fdmi_kernels.h"
#include "img_common.h"

namespace fdmi {
namespace g2 {

template <int V> using IC = std::integral_constant<int, V>;

#ifndef FDMI_256_DBG
#define FDMI_256_DBG 0  // ablation builds (wrong results): 1 no DMA in the loop, 2 no MFMAs, 4 no epilogue
#endif
constexpr int BM = 256, BN = 384, NTHR = 512;
constexpr int W_ST = BN * 64, A_ST = BM * 64, STAGE = W_ST + A_ST;  // 24,576 + 16,384 = 40,960 B per k16 stage
constexpr int NS = 3;
constexpr int OFF_PAR = NS * STAGE;                 // 122,880: bias | gamma | beta
constexpr int OFF_RED = OFF_PAR + 3 * BN * 4;       // 127,488: 2 x part[256][4]
constexpr int OFF_RI = OFF_RED + 2 * BM * 4 * 4;    // 135,680: (sequence, position) of the tile's 256 token rows
constexpr int SMEM = OFF_RI + BM * 8;               // 137,728 B
constexpr int W_KTILE = BN * 128;                   // bytes of one k-tile (32 k) of a 384-column weight tile in HBM
constexpr int NDMA = 5;                             // LDS-DMA instructions per wave and stage (3 W + 2 A)
constexpr int NACC_A = 8;                           // MFMA tiles [jn][im] (index jn * 4 + im) held in AGPRs; the other four in VGPRs

// PROF (FDMI_STAMPS=1): workgroup 0 records s_memtime stamps per stage: stamps[EPI][wave][slot][6] = {stage top, after pass 2, after
// the barrier, after the first reads of the next stage, after pass 3, after the epilogue (a tile's last stage)};
// slot 63 = {s_memtime, s_memrealtime (100 MHz) at kernel start, the same at the end}
template <int EPI, bool PROF>
__global__ __launch_bounds__(NTHR) void gemm_img256_kernel(GemmImgArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wid & 3, wm = wid >> 2;  // wave tile: rows wm*128 .. +127, columns wn*96 .. +95
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

  // ---------------------------------------------------------------- the copy stream (every wave issues 5 of a stage's 40 pieces)
  // W piece j (0..23) = column group j / 2, plane j % 2 of the stage: lane (half, l31) fetches the 16-byte unit
  // 2c + half + 4 plane of weight row 32 g + l31 out of gemm_img's k-tile stage ([8-row piece][unit ^ (piece & 1)][row % 8]).
  // A piece j (0..15) = row group j / 2, plane j % 2: units 2c, 2c + 1 (+ 4 plane) of the group are 1 KiB of contiguous bytes.
  int i_ti = 0, i_ks = 0, i_slot = 0, i_m0, i_n0;
  tile_mn(0, i_m0, i_n0);
  auto issue_piece = [&](auto I) __attribute__((always_inline)) {
    constexpr int i = decltype(I)::value;
    if (FDMI_256_DBG & 1) return;
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const int kt = i_ks >> 1, c = i_ks & 1;
    lds_ptr_t dst = (lds_ptr_t)(smem) + i_slot * STAGE;
    if constexpr (i < 3) {
      const int l31 = ln & 31, hf = ln >> 5;
      const int wlane = (l31 >> 3) * 1024 + ((hf ^ ((l31 >> 3) & 1)) * 128) + (l31 & 7) * 16;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<unsigned char*>(p.W) + (size_t)i_n0 * rb, 0, BN * rb, 0x00020000);
      const int j = wid * 3 + i;
      dma16(rs, dst + j * 1024, wlane, kt * W_KTILE + c * 256 + (j >> 1) * 4096 + (j & 1) * 512);
    } else {
      const int groups = (Mp - i_m0) >> 5;  // row groups of this tile that exist (>= 4: Mp is a multiple of 128)
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<unsigned char*>(p.A) + (size_t)(i_m0 >> 5) * nk * 4096, 0, (groups < 8 ? groups : 8) * nk * 4096, 0x00020000);
      const int j = wid * 2 + (i - 3);
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
  auto issue_stage = [&]() __attribute__((always_inline)) {
    issue_piece(IC<0>{}); issue_piece(IC<1>{}); issue_piece(IC<2>{}); issue_piece(IC<3>{}); issue_piece(IC<4>{});
  };

  // ---------------------------------------------------------------- fragments and accumulators
  f32x16 accA[NACC_A], accV[12 - NACC_A];
#define FD_ACC(jn, im) ((jn) * 4 + (im) < NACC_A ? accA[(jn) * 4 + (im)] : accV[(jn) * 4 + (im) - NACC_A])
#define FD_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")
  auto zero_acc = [&]() {
#pragma unroll
    for (int b = 0; b < NACC_A; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) accA[b][r] = 0.f;
#pragma unroll
    for (int b = 0; b < 12 - NACC_A; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) accV[b][r] = 0.f;
  };
  // 12 MFMAs, row jn of the wave tile after row jn; PIECES: the wave's five copy pieces are issued behind MFMAs 2, 4, 6, 8, 10
  auto mm12 = [&](auto SW, const f16x8 (&wf)[3], const f16x8 (&af)[4], auto PIECES) __attribute__((always_inline)) {
    constexpr bool sw = decltype(SW)::value != 0, pieces = decltype(PIECES)::value != 0;
    if (FDMI_256_DBG & 2) {
#pragma unroll
      for (int jn = 0; jn < 3; ++jn) asm volatile("" ::"v"(wf[jn]));
#pragma unroll
      for (int im = 0; im < 4; ++im) asm volatile("" ::"v"(af[im]));
    }
#define FD_ONE(jn, im)                                                                                                          \
  do {                                                                                                                          \
    if (!(FDMI_256_DBG & 2)) {                                                                                                  \
      if constexpr ((jn) * 4 + (im) < NACC_A) {                                                                                 \
        if constexpr (sw) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(accA[(jn) * 4 + (im)]) : "v"(wf[jn]), "v"(af[im])); \
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(accA[(jn) * 4 + (im)]) : "v"(af[im]), "v"(wf[jn])); \
      } else {                                                                                                                  \
        if constexpr (sw) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(accV[(jn) * 4 + (im) - NACC_A]) : "v"(wf[jn]), "v"(af[im])); \
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(accV[(jn) * 4 + (im) - NACC_A]) : "v"(af[im]), "v"(wf[jn])); \
      }                                                                                                                         \
    }                                                                                                                           \
    if constexpr (pieces && ((jn) * 4 + (im)) % 2 == 1 && ((jn) * 4 + (im)) / 2 < NDMA) issue_piece(IC<(((jn) * 4 + (im)) / 2) % NDMA>{}); \
  } while (0)
    FD_ONE(0, 0); FD_ONE(0, 1); FD_ONE(0, 2); FD_ONE(0, 3);
    FD_ONE(1, 0); FD_ONE(1, 1); FD_ONE(1, 2); FD_ONE(1, 3);
    FD_ONE(2, 0); FD_ONE(2, 1); FD_ONE(2, 2); FD_ONE(2, 3);
#undef FD_ONE
  };
  // fragment (group, plane) = the 1 KiB at lane * 16
  auto ldw = [&](f16x8 (&d)[3], int slot, int plane) __attribute__((always_inline)) {
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const unsigned char* b = smem + slot * STAGE + wn * 3 * 2048 + plane * 1024 + ln * 16;
#pragma unroll
    for (int jn = 0; jn < 3; ++jn) d[jn] = *reinterpret_cast<const f16x8*>(b + jn * 2048);
  };
  auto lda = [&](f16x8 (&d)[4], int slot, int plane) __attribute__((always_inline)) {
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const unsigned char* b = smem + slot * STAGE + W_ST + wm * 4 * 2048 + plane * 1024 + ln * 16;
#pragma unroll
    for (int im = 0; im < 4; ++im) d[im] = *reinterpret_cast<const f16x8*>(b + im * 2048);
  };

  // ---------------------------------------------------------------- epilogues (block by block as in gemm_img.hip)
  auto epilogue = [&](auto SW, int ti) {
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
      for (int jn = 0; jn < 3; ++jn) {
        const int cb = (n0 >> 5) + wn * 3 + jn;  // wave-uniform
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
#define FD_SB() __builtin_amdgcn_sched_barrier(0)
  const bool rec = PROF && blockIdx.x == 0 && p.stamps != nullptr;
  unsigned long long* st = PROF ? p.stamps + ((size_t)(EPI == EPI_IMG_QKV ? (int)EPI_IMG_QK : EPI) * 8 + wid) * 64 * 6 : nullptr;
  int slot = 0;
#define FD_STAMP(i) do { if (PROF) { if (rec && slot < 63) { int ln_; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln_)); if (ln_ == 0) st[slot * 6 + (i)] = __builtin_amdgcn_s_memtime(); } } } while (0)
#define FD_STAMP2(i) do { if (PROF) { if (rec) { int ln_; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln_)); if (ln_ == 0) { st[63 * 6 + (i)] = __builtin_amdgcn_s_memtime(); st[63 * 6 + (i) + 1] = __builtin_amdgcn_s_memrealtime(); } } } } while (0)
  FD_STAMP2(0);
  // Fragment registers: X (3 weight fragments), Y (a_hi), Z (a_lo) = 44 VGPRs.  The 128 VGPRs of a 256-register wave also hold
  // four accumulator tiles (hipcc gives a kernel that uses AGPRs half of its budget in each file), so the weight fragments are
  // single-buffered: w_hi is requested when pass 1 has been issued, the next stage's w_lo when pass 3 has -- the other wave
  // of the SIMD runs its MFMAs meanwhile.
  //   stage s:  [a_lo -> Z]  pass 1 (w_lo a_hi)  [w_hi -> X]  pass 2 (w_hi a_hi)   vmcnt: own pieces of s + 1 landed   BARRIER s + 1
  //             [a_hi of s + 1 -> Y]  pass 3 (w_hi a_lo) + the wave's pieces of stage s + 3 (into the slot of s)  [w_lo of s + 1 -> X]
  f16x8 X[3], Y[4], Z[4];
  int cs = 0;  // slot of the stage being computed
  auto next_slot = [&](int s) { return s == NS - 1 ? 0 : s + 1; };
  // MORE: the next stage belongs to the same tile (fetch its first fragments)
  auto stage = [&](auto SW, bool wait, auto MORE) __attribute__((always_inline)) {
    FD_SB();
    FD_STAMP(0);
    lda(Z, cs, 1);
    FD_SB();
    mm12(SW, X, Y, IC<0>{});
    FD_SB();
    ldw(X, cs, 0);
    FD_SB();
    mm12(SW, X, Y, IC<0>{});
    FD_SB();
    FD_STAMP(1);
    if (wait && !(FDMI_256_DBG & 1)) FD_WAIT_VM(NDMA);
    barrier_keep_vm();
    FD_SB();
    FD_STAMP(2);
    cs = next_slot(cs);
    if constexpr (decltype(MORE)::value) lda(Y, cs, 0);
    FD_SB();
    FD_STAMP(3);
    mm12(SW, X, Z, IC<1>{});
    FD_SB();
    if constexpr (decltype(MORE)::value) ldw(X, cs, 1);
    FD_SB();
    FD_STAMP(4);
    if constexpr (decltype(MORE)::value) ++slot;
  };
  auto run_tile = [&](auto SW, int ti) {
    // the first stage of a later tile: its pieces were waited for before the previous epilogue (whose stores are now
    // between them and the next stage's pieces in the queue)
    stage(SW, ti == 0, IC<1>{});
    for (int ks = 1; ks + 1 < nks; ++ks) stage(SW, true, IC<1>{});
    stage(SW, true, IC<0>{});
    FD_MFMA_DRAIN();
    FD_WAIT_VM(NDMA);  // this wave's pieces of the next tile's SECOND stage landed: its barrier follows the epilogue's stores in the queue
    if (!(FDMI_256_DBG & 4)) epilogue(SW, ti);
    FD_STAMP(5);
    ++slot;
    zero_acc();
    if (ti + 1 < cnt) {  // (cs already points at the next tile's first stage, published by the last barrier)
      ldw(X, cs, 1);
      lda(Y, cs, 0);
    }
  };

  issue_stage();
  issue_stage();
  issue_stage();
  FD_WAIT_VM(2 * NDMA);
  barrier_keep_vm();  // stage 0 landed (also publishes the parameter image)
  zero_acc();
  ldw(X, cs, 1);
  lda(Y, cs, 0);
  for (int ti = 0; ti < cnt; ++ti) {
    if constexpr (EPI == EPI_IMG_QKV) {  // the v tiles run the normal MFMA form (lane = feature), the q | k tiles the swapped one
      int m0, n0;
      tile_mn(ti, m0, n0);
      if (n0 >= 2 * p.H * 32) run_tile(IC<0>{}, ti);
      else run_tile(IC<1>{}, ti);
    } else {
      run_tile(IC<1>{}, ti);
    }
  }
  FD_WAIT_VM(0);  // nothing may land in LDS after the workgroup has exited
  FD_STAMP2(2);
#undef FD_STAMP
#undef FD_STAMP2
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
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_img256_kernel<EPI, false>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_img256_kernel<EPI, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr_set[dev] = true;
  }
  const int ntiles_max = ((max_rows + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  int grid = n_cu_of_current_device() / 8 * 8;
  if (grid > ntiles_max) grid = (ntiles_max + 7) / 8 * 8;
  if (grid < 8) grid = 8;
  if (p.stamps) hipLaunchKernelGGL((gemm_img256_kernel<EPI, true>), dim3(grid), dim3(NTHR), SMEM, s, p);
  else hipLaunchKernelGGL((gemm_img256_kernel<EPI, false>), dim3(grid), dim3(NTHR), SMEM, s, p);
}

}  // namespace g2

bool gemm_img256_supported(int epilogue) { return epilogue == EPI_IMG_GELU || epilogue == EPI_IMG_BIAS; }

void launch_gemm_img256(int epilogue, const GemmImgArgs& p, int max_rows, hipStream_t s) {
  switch (epilogue) {
    case EPI_IMG_GELU: g2::launch<EPI_IMG_GELU>(p, max_rows, s); break;
    default: g2::launch<EPI_IMG_BIAS>(p, max_rows, s); break;
  }
}

}  // namespace fdmi
