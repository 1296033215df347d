// Token GEMMs of the encoder on row images, "ping-pong" form: the epilogue of one half of the workgroup runs UNDER the
// matrix instructions of the other half.
//
//   C[M,N] = A[M,K] * W[N,K]^T + bias[N]   then one of the epilogues of gemm_img.hip (same math, same images, same
//   fp16 hi/lo split arithmetic: three v_mfma_f32_32x32x16_f16 per product into one fp32 accumulator).
//   Reference: HF BertSelfAttention / BertSelfOutput / BertIntermediate / BertOutput (transformers 4.11.3) as called from
//   foldingdiff/modelling.py:473-480, AnglesPredictor.dense1 (modelling.py:195-196, :203-205).
//
// Why (profiles/r03_coissue2_probe.log): on one SIMD the plain (non-packed) VALU instructions of one wave issue beside the
// MFMAs of the other wave at ~89 % of their solo rate while the MFMA stream keeps its full rate; packed fp32 instructions
// (v_pk_fma_f32 ...) do not -- they serialize with the matrix pipe.  gemm_img.hip keeps its eight compute waves in
// lockstep, so both waves of a SIMD are in the k-loop together and in the epilogue together and the matrix pipe idles for
// the whole epilogue (~40 % of a launch).  Here the two waves of a SIMD belong to different GROUPS that are half a tile
// period out of phase.  This file is compiled with -fno-slp-vectorize (build.py): no packed fp32 in the epilogues.
//
// Structure (one persistent workgroup per CU, 8 compute waves + NL loader waves, LDS rings as in gemm_img.hip):
//  * group g = waves 4g .. 4g+3 (one per SIMD) owns 64-row x 384-column tiles, wave tile 64 x 96 (96 accumulators).
//  * the workgroup walks a stream of POSITIONS.  Position p carries one W k-tile stage (48 KiB, k-tile p mod nk of the
//    workgroup's column tile -- the column tile is fixed per workgroup, so the weight k-tiles simply cycle) and, for each
//    group that computes at p, the 64 rows x 32 k of ITS current tile (8 KiB).  A tile of a group is any nk consecutive
//    positions (the k order is rotated, the sum is the same); after them the group spends S positions in its epilogue,
//    one chunk per position, while the other group keeps computing: its period is T = nk + S positions, group 1 runs
//    T / 2 positions behind group 0.  One workgroup barrier per position, exactly as in gemm_img.hip; the LayerNorm
//    epilogue's row statistics cross the four waves of a group through LDS and are published by the position barriers.
//  * loader waves: issue W(p+1) and the computing groups' A(p+2) after barrier p, counted vmcnt waits.
//  * tiles are dealt XCD-aware: XCD x owns a contiguous range of 128-row pairs; its workgroups are split into one class
//    per column tile; workgroup ic of a class takes pairs ic, ic + nc, ...: group g takes half g of each pair.
#include <cstdlib>
#include <type_traits>

#include "fdmi_kernels.h"
#include "img_common.h"

namespace fdmi {
namespace gp {

template <int V> using IC = std::integral_constant<int, V>;

#ifndef FDMI_PP_NL
#define FDMI_PP_NL 2
#endif
#ifndef FDMI_PP_PRIO
#define FDMI_PP_PRIO 0   // s_setprio of a wave inside its k-loop (the epilogue runs at 0)
#endif
#ifndef FDMI_PP_DBG
#define FDMI_PP_DBG 0    // ablation builds (wrong results): 1 no W copies after the first two positions, 2 no epilogue chunks,
#endif                   // 3 no fragment reads in the k-loop, 4 no A copies after the first three positions
constexpr int NL = FDMI_PP_NL;                                   // loader waves
constexpr int BMH = 64, BN = 384, NTHR = 64 * (8 + NL);          // a group's tile: 64 x 384
constexpr int W_STAGE = BN * 128, A_HALF = BMH * 128, A_STAGE = 2 * A_HALF;
constexpr int NWS = 2, NAS = 3;
constexpr int OFF_A = NWS * W_STAGE;                             //  98,304
constexpr int OFF_PAR = OFF_A + NAS * A_STAGE;                   // 147,456: bias | gamma | beta
constexpr int OFF_RED = OFF_PAR + 3 * BN * 4;                    // 152,064: per group 2 x part[64][4]
constexpr int SMEM = OFF_RED + 2 * 2 * BMH * 4 * 4;              // 156,160 B
constexpr int APL = 8 / NL;                                      // A pieces per loader wave and computing group

// epilogue positions per tile: the tile's 12 half-blocks (jn 0-2 x im 0-1 x the quad pairs {0,2} / {1,3} of a 32 x 32 MFMA
// tile; a half-block is one 16-byte hi unit + one 16-byte lo unit per lane) are spread over S chunks
template <int EPI> struct Epi { static constexpr int S = 12; };          // GELU / BIAS: one half-block per position
template <> struct Epi<EPI_IMG_QK> { static constexpr int S = 3; };
template <> struct Epi<EPI_IMG_VT> { static constexpr int S = 3; };
template <> struct Epi<EPI_IMG_QKV> { static constexpr int S = 3; };
template <> struct Epi<EPI_IMG_LN> { static constexpr int S = 8; };      // 4 x (acc + bias + residual, row sum) | deviations | 3 x store

// PROF (FDMI_STAMPS=1): workgroup 0 records s_memtime stamps per position:  stamps[EPI slot][wave 0-9][position < 100][3] =
//   compute waves {start of the position's work, before the barrier of the next position, behind it};
//   loader waves  {before the vmcnt wait, behind it, behind the barrier}            (scripts/stamps_pp.py)
template <int EPI, bool PROF>
__global__ __launch_bounds__(NTHR) void gemm_pp_kernel(GemmImgArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int S = Epi<EPI>::S;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wid & 3, grp = (wid >> 2) & 1;
  const bool rec = PROF && blockIdx.x == 0 && p.stamps != nullptr;
  unsigned long long* st = PROF ? p.stamps + ((size_t)(EPI == EPI_IMG_QKV ? (int)EPI_IMG_QK : EPI) * 3072 + wid * 300) : nullptr;
#define FD_STAMP(pos, i) do { if (PROF) { if (rec && (pos) < 100 && lane == 0) st[(pos) * 3 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
  const int nk = p.K >> 5, rb = nk * 128;                 // k-tiles; bytes per image row (A and W share K)
  const int Mp = p.dims[1];
  const int tiles_n = (p.N + BN - 1) / BN, npairs = Mp / 128;
  const int T = nk + S, off1 = T >> 1;
  // work of this workgroup
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int plo = (int)((long long)npairs * xcd / 8), phi = (int)((long long)npairs * (xcd + 1) / 8);
  const int cls = jx % tiles_n, ic = jx / tiles_n;
  const int nc = (per - cls + tiles_n - 1) / tiles_n;   // workgroups of this class on this XCD
  const int cnt = plo + ic < phi ? (phi - plo - ic + nc - 1) / nc : 0;   // tiles per group
  if (cnt == 0) return;
  const int n0 = cls * BN;
  const int P = off1 + cnt * T;                           // positions of the stream (group 1 ends it)
  auto tile_m0 = [&](int g, int ti) { return ((plo + ic + ti * nc) * 2 + g) * BMH; };

  {  // bias (all N <= 3 BN columns) or bias | gamma | beta (EPI_LN, N <= BN) -> LDS, published by the first barrier
    float* par = reinterpret_cast<float*>(smem + OFF_PAR);
    if constexpr (EPI == EPI_IMG_LN) {
      for (int i = tid; i < BN; i += NTHR) {
        const bool ok = i < p.N;
        par[i] = ok ? p.bias[i] : 0.f;
        par[BN + i] = ok ? p.gamma[i] : 0.f;
        par[2 * BN + i] = ok ? p.beta[i] : 0.f;
      }
    } else {
      for (int i = tid; i < 3 * BN; i += NTHR) par[i] = i < p.N ? p.bias[i] : 0.f;
    }
  }

  // a group's clock: phase ph inside its period (negative: not started), tile index ti
  struct Clock {
    int ph, ti;
  };
  auto advance = [&](Clock& c) {
    if (++c.ph == T) {
      c.ph = 0;
      ++c.ti;
    }
  };
  auto computes = [&](const Clock& c) { return c.ph >= 0 && c.ph < nk && c.ti < cnt; };

  // ================================================================ the loader waves
  //   prologue A(0) W(0) A(1);  position p:  [vmcnt: W(p), A(p) landed] [barrier p] W(p+1) A(p+2)
  // The barrier publishes position p to the compute waves and tells the loaders that compute(p-1) is over, which frees W slot
  // (p+1) & 1 and A slot (p+2) % 3.
  if (wid >= 8) {
    const int li = wid - 8;  // pieces j = li, li + NL, ...: all of one parity
    const int vw = lane * 16;  // W: the HBM image of a (column tile, k-tile) IS the LDS stage (api.hip: pack_weight_tiles)
    // A: grouped image [row / 32][k-tile][unit][row % 32][16 B]; a piece = 8 rows x 8 units, unit-major in LDS (gemm_img.hip)
    const int va = (((lane >> 3) ^ (li & 1)) * 512) + (lane & 7) * 16;
    Clock cw[2] = {{0, 0}, {-off1, 0}}, ca[2] = {{0, 0}, {-off1, 0}};
    int w_kt = 0, a_kt = 0, w_slot = 0, a_slot = 0;
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(p.W) + (size_t)n0 * rb, 0, BN * rb, 0x00020000);
    auto issue_w = [&]() {  // the position of clocks cw
      if ((computes(cw[0]) || computes(cw[1])) && !(FDMI_PP_DBG == 1 && cw[0].ph + cw[0].ti * T > 1)) {
        lds_ptr_t dst = (lds_ptr_t)(smem) + w_slot * W_STAGE;
        const int so = w_kt * W_STAGE;
#pragma unroll
        for (int i = 0; i < 48 / NL; ++i) {
          const int j = li + i * NL;
          dma16(rsw, dst + j * 1024, vw, so + j * 1024);
        }
      }
      w_slot ^= 1;
      if (++w_kt == nk) w_kt = 0;
      advance(cw[0]);
      advance(cw[1]);
    };
    auto issue_a = [&]() -> int {  // the position of clocks ca; returns the number of pieces this wave issued
      int n = 0;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        if (computes(ca[g]) && !(FDMI_PP_DBG == 4 && ca[0].ph + ca[0].ti * T > 2)) {
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
              const_cast<unsigned char*>(p.A) + (size_t)tile_m0(g, ca[g].ti) * rb, 0, BMH * rb, 0x00020000);  // two 32-row groups
          lds_ptr_t dst = (lds_ptr_t)(smem) + OFF_A + a_slot * A_STAGE + g * A_HALF;
          const int so = a_kt * 4096;
#pragma unroll
          for (int i = 0; i < APL; ++i) {
            const int j = li + i * NL;
            dma16(rs, dst + j * 1024, va, so + (j >> 2) * 32 * rb + (j & 3) * 128);
          }
          n += APL;
        }
        advance(ca[g]);
      }
      a_slot = a_slot == NAS - 1 ? 0 : a_slot + 1;
      if (++a_kt == nk) a_kt = 0;
      return n;
    };
    (void)issue_a();
    issue_w();
    int na1 = issue_a();  // pieces of A(p+1) in flight behind W(p)
    for (int pp = 0; pp < P; ++pp) {
      FD_STAMP(pp, 0);
      if (na1 == 0) FD_WAIT_VM(0);
      else if (na1 == APL) FD_WAIT_VM(APL);
      else FD_WAIT_VM(2 * APL);
      FD_STAMP(pp, 1);
      barrier_keep_vm();
      FD_STAMP(pp, 2);
      issue_w();
      na1 = issue_a();
    }
    FD_WAIT_VM(0);  // nothing may land in LDS after the workgroup has exited
    return;
  }

  // ================================================================ the compute waves
  const int wbase = wn * 96 * 128, abase = OFF_A + grp * A_HALF;
  const float* par0 = reinterpret_cast<const float*>(smem + OFF_PAR);

  f32x16 acc[3][2];  // [jn][im]
  auto zero_acc = [&]() {
#pragma unroll
    for (int jn = 0; jn < 3; ++jn)
#pragma unroll
      for (int im = 0; im < 2; ++im)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jn][im][r] = 0.f;
  };
  zero_acc();

  auto mm6 = [&](auto SW, const f16x8 (&wf)[3], const f16x8 (&af)[2]) {  // SW: swapped form (D^T = W A^T)
#pragma unroll
    for (int jn = 0; jn < 3; ++jn)
#pragma unroll
      for (int im = 0; im < 2; ++im)
        acc[jn][im] = decltype(SW)::value ? __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[jn], af[im], acc[jn][im], 0, 0, 0)
                                          : __builtin_amdgcn_mfma_f32_32x32x16_f16(af[im], wf[jn], acc[jn][im], 0, 0, 0);
  };
  auto ldw = [&](f16x8 (&d)[3], const unsigned char* wb, int off) {
    if (FDMI_PP_DBG == 3) return;
#pragma unroll
    for (int jn = 0; jn < 3; ++jn) d[jn] = *reinterpret_cast<const f16x8*>(wb + off + jn * 4096);
  };
  auto lda = [&](f16x8 (&d)[2], const unsigned char* ab, int off) {
    if (FDMI_PP_DBG == 3) return;
#pragma unroll
    for (int im = 0; im < 2; ++im) d[im] = *reinterpret_cast<const f16x8*>(ab + off + im * 4096);
  };
  auto lane_id = [&]() {  // opaque copy of the lane id: values derived from it inside a block do not live across the loop
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    return ln;
  };

  // ---------------------------------------------------------------- epilogue pieces
  // SWAP form: lane (l31, half) owns token row  m0 + 32 im + l31  and, per MFMA tile jn, the columns
  // n0 + wn*96 + 32 jn + 8q + 4 half + e  (register r = 4q + e).  Half-block hb = quads {hb, hb + 2}: after the half-wave
  // exchange the lane pair holds hi unit 2 half' + hb... (img_common.h): one hi and one lo 16-byte store per lane.
  // values of a half-block: v[4 qi + e] = acc[jn][im][4 (hb + 2 qi) + e]
  auto pack_half = [&](const float (&v)[8], float s, u32x4& hi, u32x4& lo) {
    unsigned H[4], Lo[4];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
      for (int dd = 0; dd < 2; ++dd) split_pair(v[4 * qi + 2 * dd] * s, v[4 * qi + 2 * dd + 1] * s, H[2 * qi + dd], Lo[2 * qi + dd]);
    swap32(H[0], H[2]);
    swap32(H[1], H[3]);
    swap32(Lo[0], Lo[2]);
    swap32(Lo[1], Lo[3]);
    hi = u32x4{H[0], H[1], H[2], H[3]};
    lo = u32x4{Lo[0], Lo[1], Lo[2], Lo[3]};
  };
  auto bias_half = [&](auto JN, auto IM, auto HB, int cb, float (&o)[8], int half) {
    constexpr int jn = decltype(JN)::value, im = decltype(IM)::value, hb = decltype(HB)::value;
    const float os = p.acc_scale;
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
      const int q = hb + 2 * qi;
      const float4 b4 = *reinterpret_cast<const float4*>(par0 + cb * 32 + 8 * q + 4 * half);
      o[4 * qi + 0] = __builtin_fmaf(acc[jn][im][4 * q + 0], os, b4.x);
      o[4 * qi + 1] = __builtin_fmaf(acc[jn][im][4 * q + 1], os, b4.y);
      o[4 * qi + 2] = __builtin_fmaf(acc[jn][im][4 * q + 2], os, b4.z);
      o[4 * qi + 3] = __builtin_fmaf(acc[jn][im][4 * q + 3], os, b4.w);
    }
  };
  // half-block index hbi 0..11 -> (im, jn, hb)
#define FD_HBI(hbi) IC<((hbi) % 6) / 2>{}, IC<(hbi) / 6>{}, IC<(hbi) % 2>{}

  // ---- GELU / BIAS
  auto epi_gelu_half = [&](auto JN, auto IM, auto HB, int m0) {
    constexpr int jn = decltype(JN)::value, im = decltype(IM)::value, hb = decltype(HB)::value;
    const int nb = p.N >> 5;
    const int cb = (n0 >> 5) + wn * 3 + jn;  // wave-uniform
    if (cb >= nb) return;
    const int ln = lane_id();
    const int l31 = ln & 31, half = ln >> 5;
    float o[8];
    bias_half(JN, IM, HB, cb, o, half);
    if constexpr (EPI == EPI_IMG_GELU) {
#pragma unroll
      for (int r = 0; r < 8; ++r) o[r] = gelu_erf(o[r]);
    }
    u32x4 hi, lo;
    pack_half(o, p.out_scale, hi, lo);
    unsigned char* blk0 = p.out + ((size_t)((m0 + im * 32) >> 5) * nb + cb) * 4096;
    const unsigned off = (unsigned)(l31 * 16 + half * 1024 + hb * 512);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(blk0, 0, 4096, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(hi, rs, (int)off, 0, FD_STORE_AUX);
    __builtin_amdgcn_raw_buffer_store_b128(lo, rs, (int)off + 2048, 0, FD_STORE_AUX);
  };

  // ---- q | k: grouped images per (sequence, head): [position / 32][unit][position % 32][16 B]
  auto epi_qk_half = [&](auto JN, auto IM, auto HB, int2 ri) {
    constexpr int hb = decltype(HB)::value, jn = decltype(JN)::value;
    const int H = p.H;
    const int cb = (n0 >> 5) + wn * 3 + jn;  // wave-uniform: block of the [q | k] column space
    if (cb >= 2 * H) return;
    const int isk = cb >= H ? 1 : 0, h = cb - isk * H;
    const int half = lane_id() >> 5;
    float o[8];
    bias_half(JN, IM, HB, cb, o, half);
    u32x4 hi, lo;
    pack_half(o, isk ? p.k_scale : p.q_scale, hi, lo);
    if (ri.x >= 0) {  // rows that are no token (alignment / tail rows) are not stored
      const long long row = ((long long)ri.x * H + h) * p.LTOT + ri.y;
      unsigned char* u0 = (isk ? p.kbuf : p.qbuf) + img_unit_offset(row, 1, 0, 2 * half);
      *reinterpret_cast<u32x4*>(u0 + hb * 512) = hi;
      *reinterpret_cast<u32x4*>(u0 + (4 + hb) * 512) = lo;
    }
  };

  // ---- v^T (normal MFMA form: lane = feature d = l31 of head cb, register r = 4q + e <-> token row 8q + 4 half + e).  Half-block
  // hb = token octet (2 half + hb) of the 32-row MFMA tile after the exchange; layout and swizzle as in gemm_img.hip
  auto epi_vt_half = [&](auto JN, auto IM, auto HB, int2 ri) {
    constexpr int jn = decltype(JN)::value, im = decltype(IM)::value, hb = decltype(HB)::value;
    const int H = p.H, nkb = p.LTOT >> 5;
    constexpr bool merged = EPI == EPI_IMG_QKV;  // v columns follow the 2 H blocks of q | k
    const int cb = (n0 >> 5) + wn * 3 + jn - (merged ? 2 * H : 0);
    if (cb >= H) return;
    const int l31 = lane_id() & 31;
    const int sz = (l31 >> 1) & 15;
    const float bz = par0[(cb + (merged ? 2 * H : 0)) * 32 + l31];
    const float os = p.acc_scale;
    float o[8];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
      for (int e = 0; e < 4; ++e) o[4 * qi + e] = __builtin_fmaf(acc[jn][im][4 * (hb + 2 * qi) + e], os, bz);
    u32x4 vh, vl;
    pack_half(o, p.v_scale, vh, vl);
    const bool ok = ri.x >= 0;
    const int lpos = ok ? ri.y : 0;
    const int kb = lpos >> 5, oc = (lpos & 31) >> 3;
    if (sz & 1) {
      vh = u32x4{vh[2], vh[3], vh[0], vh[1]};
      vl = u32x4{vl[2], vl[3], vl[0], vl[1]};
    }
    unsigned char* row = ok ? p.vbuf + ((((size_t)ri.x * H + cb) * nkb + kb) * 32 + l31) * 128 : p.trash;
    *reinterpret_cast<u32x4*>(row + ((oc ^ (sz >> 1)) << 4)) = vh;
    *reinterpret_cast<u32x4*>(row + (((4 + oc) ^ (sz >> 1)) << 4)) = vl;
  };

  // ---- LayerNorm(dense + bias + residual): state that lives across the chunks of one epilogue
  float ln_s[2], ln_m[2];            // row sums -> mean; squared deviations -> rstd
  u32x4 rres[3][2];                  // residual half-blocks of the NEXT pass-1 chunk: [i][hi | lo]
  auto ln_load_resid = [&](int c, int m0) {  // the three half-blocks of pass-1 chunk c
    const int ln = lane_id();
    const int l31 = ln & 31, half = ln >> 5;
    const int nb = p.N >> 5;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int hbi = 3 * c + i, im = hbi / 6, jn = (hbi % 6) / 2, hb = hbi % 2;
      int cb = wn * 3 + jn;
      cb = cb < nb ? cb : 0;
      const unsigned char* blk0 = p.resid + ((size_t)((m0 + im * 32) >> 5) * nb + cb) * 4096;
      const unsigned off = (unsigned)(l31 * 16 + half * 1024 + hb * 512);
      rres[i][0] = *reinterpret_cast<const u32x4*>(blk0 + (size_t)off);
      rres[i][1] = *reinterpret_cast<const u32x4*>(blk0 + (size_t)(off + 2048));
    }
  };
  // pass 1, one half-block: acc <- acc / (a_scale w_scale) + bias + residual, row sum
  auto ln_pass1_half = [&](auto JN, auto IM, auto HB, const u32x4& rh, const u32x4& rl) {
    constexpr int jn = decltype(JN)::value, im = decltype(IM)::value, hb = decltype(HB)::value;
    const int nb = p.N >> 5;
    const int cb = wn * 3 + jn;
    if (cb >= nb) {  // columns beyond N (d_model < 384): contribute nothing
#pragma unroll
      for (int qi = 0; qi < 2; ++qi)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[jn][im][4 * (hb + 2 * qi) + e] = 0.f;
      return;
    }
    const int half = lane_id() >> 5;
    unsigned H[4] = {rh[0], rh[1], rh[2], rh[3]}, Lo[4] = {rl[0], rl[1], rl[2], rl[3]};
    swap32(H[0], H[2]);
    swap32(H[1], H[3]);
    swap32(Lo[0], Lo[2]);
    swap32(Lo[1], Lo[3]);
    float o[8];
    bias_half(JN, IM, HB, cb, o, half);
    const float ri = p.resid_inv;
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
      for (int dd = 0; dd < 2; ++dd) {
        const float r0 = (h2f_lo(H[2 * qi + dd]) + h2f_lo(Lo[2 * qi + dd])) * ri;
        const float r1 = (h2f_hi(H[2 * qi + dd]) + h2f_hi(Lo[2 * qi + dd])) * ri;
        const float v0 = o[4 * qi + 2 * dd] + r0, v1 = o[4 * qi + 2 * dd + 1] + r1;
        acc[jn][im][4 * (hb + 2 * qi) + 2 * dd] = v0;
        acc[jn][im][4 * (hb + 2 * qi) + 2 * dd + 1] = v1;
        ln_s[im] += v0;
        ln_s[im] += v1;
      }
  };
  // row statistic of the group's 64 rows: in-lane (48 columns) + the other half-wave, then the four N-waves through LDS
  // (fixed order); the position barrier between `put` and `get` publishes it
  auto ln_put = [&](float (&t)[2], int which) {
    float* part = reinterpret_cast<float*>(smem + OFF_RED) + (grp * 2 + which) * (BMH * 4);
    const int ln = lane_id();
    const int l31 = ln & 31, half = ln >> 5;
#pragma unroll
    for (int im = 0; im < 2; ++im) {
      unsigned a = __builtin_bit_cast(unsigned, t[im]), b = a;
      swap32(a, b);
      const float tot = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
      if (half == 0) part[(im * 32 + l31) * 4 + wn] = tot;
    }
  };
  auto ln_get = [&](float (&t)[2], int which) {
    const float* part = reinterpret_cast<const float*>(smem + OFF_RED) + (grp * 2 + which) * (BMH * 4);
    const int l31 = lane_id() & 31;
#pragma unroll
    for (int im = 0; im < 2; ++im) {
      const float4 q4 = *reinterpret_cast<const float4*>(part + (im * 32 + l31) * 4);
      t[im] = (q4.x + q4.y) + (q4.z + q4.w);
    }
  };
  auto ln_store_half = [&](auto JN, auto IM, auto HB, int m0) {
    constexpr int jn = decltype(JN)::value, im = decltype(IM)::value, hb = decltype(HB)::value;
    const int nb = p.N >> 5;
    const int cb = wn * 3 + jn;
    if (cb >= nb) return;
    const int ln = lane_id();
    const int l31 = ln & 31, half = ln >> 5;
    const float rstd = ln_m[im];
    float o[8];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
      const int q = hb + 2 * qi;
      const float4 g4 = *reinterpret_cast<const float4*>(par0 + BN + cb * 32 + 8 * q + 4 * half);
      const float4 e4 = *reinterpret_cast<const float4*>(par0 + 2 * BN + cb * 32 + 8 * q + 4 * half);
      o[4 * qi + 0] = acc[jn][im][4 * q + 0] * rstd * g4.x + e4.x;
      o[4 * qi + 1] = acc[jn][im][4 * q + 1] * rstd * g4.y + e4.y;
      o[4 * qi + 2] = acc[jn][im][4 * q + 2] * rstd * g4.z + e4.z;
      o[4 * qi + 3] = acc[jn][im][4 * q + 3] * rstd * g4.w + e4.w;
    }
    u32x4 hi, lo;
    pack_half(o, p.out_scale, hi, lo);
    unsigned char* blk0 = p.out + ((size_t)((m0 + im * 32) >> 5) * nb + cb) * 4096;
    const unsigned off = (unsigned)(l31 * 16 + half * 1024 + hb * 512);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(blk0, 0, 4096, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(hi, rs, (int)off, 0, FD_STORE_AUX);
    __builtin_amdgcn_raw_buffer_store_b128(lo, rs, (int)off + 2048, 0, FD_STORE_AUX);
  };

  // chunk c (compile-time) of the epilogue of tile ti
  auto chunk = [&](auto SW, auto C, int m0) {
    constexpr int c = decltype(C)::value;
    if (FDMI_PP_DBG == 2) return;
    constexpr bool kQK = EPI == EPI_IMG_QK || (EPI == EPI_IMG_QKV && decltype(SW)::value);
    constexpr bool kVT = EPI == EPI_IMG_VT || (EPI == EPI_IMG_QKV && !decltype(SW)::value);
    if constexpr (EPI == EPI_IMG_GELU || EPI == EPI_IMG_BIAS) {
      epi_gelu_half(FD_HBI(c), m0);
    } else if constexpr (kQK) {
      // (sequence, position) of the lane's two token rows: re-read per chunk (L2 hits; nothing lives across positions)
      const int l31 = lane_id() & 31;
      int2 ri[2];
#pragma unroll
      for (int im = 0; im < 2; ++im) ri[im] = p.rowinfo[m0 + im * 32 + l31];
      epi_qk_half(FD_HBI(4 * c + 0), ri[(4 * c + 0) / 6]);
      epi_qk_half(FD_HBI(4 * c + 1), ri[(4 * c + 1) / 6]);
      epi_qk_half(FD_HBI(4 * c + 2), ri[(4 * c + 2) / 6]);
      epi_qk_half(FD_HBI(4 * c + 3), ri[(4 * c + 3) / 6]);
    } else if constexpr (kVT) {
      const int half = lane_id() >> 5;
      auto one = [&](auto HBI) {
        constexpr int hbi = decltype(HBI)::value, im = hbi / 6, hb = hbi % 2;
        const int2 ri = p.rowinfo[m0 + im * 32 + 16 * half + 8 * hb];
        epi_vt_half(FD_HBI(hbi), ri);
      };
      one(IC<4 * c + 0>{});
      one(IC<4 * c + 1>{});
      one(IC<4 * c + 2>{});
      one(IC<4 * c + 3>{});
    } else if constexpr (EPI == EPI_IMG_LN) {
      if constexpr (c < 4) {  // pass 1: three half-blocks; the next chunk's residual is requested as soon as this one's is consumed
        if constexpr (c == 0) ln_s[0] = ln_s[1] = 0.f;
        ln_pass1_half(FD_HBI(3 * c + 0), rres[0][0], rres[0][1]);
        ln_pass1_half(FD_HBI(3 * c + 1), rres[1][0], rres[1][1]);
        ln_pass1_half(FD_HBI(3 * c + 2), rres[2][0], rres[2][1]);
        if constexpr (c < 3) ln_load_resid(c + 1, m0);
        if constexpr (c == 3) ln_put(ln_s, 0);
      } else if constexpr (c == 4) {  // mean, deviations, their squares
        ln_get(ln_s, 0);
        const float inv_n = 1.0f / (float)p.N;
        const int nb = p.N >> 5;
        float t2[2] = {0.f, 0.f};
#pragma unroll
        for (int im = 0; im < 2; ++im) {
          const float mean = ln_s[im] * inv_n;
#pragma unroll
          for (int jn = 0; jn < 3; ++jn) {
            if (wn * 3 + jn >= nb) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float dl = acc[jn][im][r] - mean;
              acc[jn][im][r] = dl;
              t2[im] += dl * dl;
            }
          }
        }
        ln_put(t2, 1);
      } else {  // normalise + store: four half-blocks per chunk
        if constexpr (c == 5) {
          ln_get(ln_m, 1);
          const float inv_n = 1.0f / (float)p.N;
#pragma unroll
          for (int im = 0; im < 2; ++im) ln_m[im] = 1.0f / sqrtf(ln_m[im] * inv_n + p.eps);
        }
        ln_store_half(FD_HBI(4 * (c - 5) + 0), m0);
        ln_store_half(FD_HBI(4 * (c - 5) + 1), m0);
        ln_store_half(FD_HBI(4 * (c - 5) + 2), m0);
        ln_store_half(FD_HBI(4 * (c - 5) + 3), m0);
      }
    }
  };
  // ---------------------------------------------------------------- the stream of the compute waves
#define FD_SB() __builtin_amdgcn_sched_barrier(0)
  // Fragment registers: FOUR buffers (40 VGPRs) serve the eight fragment sets of a position.  Every set is requested at the
  // START of the MFMA group before the one that consumes it, into the buffer whose last reader was the group before that:
  //     group            1: wh0 ah0   2: wh0 al0   3: wl0 ah0   4: wh1 ah1   5: wh1 al1   6: wl1 ah1
  //     requested during    al0 -> Yb    wl0 -> Xb    wh1 -> Xa    al1 -> Ya    wl1 -> Xb    wh0' -> Xa
  //                                                   ah1 -> Yb                              ah0' -> Ya
  // (wh0 / ah0 of the position arrive in Xa / Ya).  Left to itself hipcc reuses the registers of the set still being read and
  // sinks the requests to one MFMA (32 cycles) before their consumer: a wave alone on its SIMD then waits for the LDS in
  // every group (cycle stamps: 1750 cycles for 30 MFMAs).
  f16x8 Xa[3], Xb[3], Ya[2], Yb[2];
  int cw = 0, ca = 0;  // slots of the current position
  // first fragments (k16 step 0, hi planes) of the position in slots (cw, ca)
  auto first_fragments = [&]() {
    const int ln = lane_id();
    const int r00 = ((ln & 31) >> 3) * 1024 + ((((ln >> 5) ^ ((ln >> 3) & 1)) * 8 + (ln & 7)) << 4);
    lda(Ya, smem + abase + ca * A_STAGE, r00);
    ldw(Xa, smem + wbase + cw * W_STAGE, r00);
  };
  auto groups_1_to_5 = [&](auto SW) {
    const unsigned char* wb = smem + wbase + cw * W_STAGE;
    const unsigned char* ab = smem + abase + ca * A_STAGE;
    // the four fragment offsets, re-derived from the lane id per position (no loop-invariant VGPRs to spill)
    int rd[2][2];
    {
      const int ln = lane_id();
      const int l31_ = ln & 31, half_ = ln >> 5;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          rd[c][pl] = (l31_ >> 3) * 1024 + ((((2 * c + half_ + 4 * pl) ^ ((l31_ >> 3) & 1)) * 8 + (l31_ & 7)) << 4);
    }
    FD_SB();
    lda(Yb, ab, rd[0][1]);             // al0
    FD_SB();
    mm6(SW, Xa, Ya);                   // 1: wh0 ah0
    FD_SB();
    ldw(Xb, wb, rd[0][1]);             // wl0
    FD_SB();
    mm6(SW, Xa, Yb);                   // 2: wh0 al0
    FD_SB();
    ldw(Xa, wb, rd[1][0]);             // wh1
    lda(Yb, ab, rd[1][0]);             // ah1
    FD_SB();
    mm6(SW, Xb, Ya);                   // 3: wl0 ah0
    FD_SB();
    lda(Ya, ab, rd[1][1]);             // al1
    FD_SB();
    mm6(SW, Xa, Yb);                   // 4: wh1 ah1
    FD_SB();
    ldw(Xb, wb, rd[1][1]);             // wl1
    FD_SB();
    mm6(SW, Xa, Ya);                   // 5: wh1 al1
    FD_SB();
  };
  auto next_slots = [&]() {
    cw ^= 1;
    ca = ca == NAS - 1 ? 0 : ca + 1;
  };

  // A group's program is periodic: nk compute positions, then the S chunks of the epilogue as STRAIGHT-LINE code with the
  // position barriers between them (chunks selected by a switch inside a position loop make every accumulator update a
  // phi of 16-register tuples: hipcc then spills whole accumulator tiles).  Group 1 idles off1 positions before its first
  // tile, group 0 after its last one: both execute exactly P barriers.
  auto run = [&](auto SW) {
    int pp = 0;  // position index
    auto end_position = [&]() {  // every wave passes the barrier of position pp + 1 (there is none after the last position)
      next_slots();
      FD_STAMP(pp, 1);
      if (++pp < P) barrier_keep_vm();
      FD_STAMP(pp - 1, 2);
      FD_STAMP(pp, 0);
    };
    barrier_keep_vm();  // position 0 landed (also publishes the parameter image)
    if (grp != 0)
      for (int i = 0; i < off1; ++i) end_position();
    for (int ti = 0; ti < cnt; ++ti) {
      const int m0 = tile_m0(grp, ti);
      zero_acc();
      if (FDMI_PP_PRIO) __builtin_amdgcn_s_setprio(FDMI_PP_PRIO);
      first_fragments();
      for (int kt = 0; kt < nk; ++kt) {
        groups_1_to_5(SW);
        next_slots();
        FD_STAMP(pp, 1);
        if (++pp < P) {
          FD_WAIT_LGKM0();
          barrier_keep_vm();  // every fragment of this position is in registers: its slots are free; the next position landed
        }
        FD_STAMP(pp - 1, 2);
        FD_STAMP(pp, 0);
        const bool more = kt + 1 < nk;
        if (more) first_fragments();   // into Xa / Ya: their last readers were groups 4 and 5
        FD_SB();
        mm6(SW, Xb, Yb);               // 6: wl1 ah1
        FD_SB();
      }
      if (FDMI_PP_PRIO) __builtin_amdgcn_s_setprio(0);
      if constexpr (EPI == EPI_IMG_LN) ln_load_resid(0, m0);  // the tile is complete: request the first residual half-blocks
      chunk(SW, IC<0>{}, m0); end_position();
      chunk(SW, IC<1>{}, m0); end_position();
      chunk(SW, IC<2>{}, m0); end_position();
      if constexpr (S > 3) {
        chunk(SW, IC<3>{}, m0); end_position();
        chunk(SW, IC<4>{}, m0); end_position();
        chunk(SW, IC<5>{}, m0); end_position();
        chunk(SW, IC<6>{}, m0); end_position();
        chunk(SW, IC<7>{}, m0); end_position();
      }
      if constexpr (S > 8) {
        chunk(SW, IC<8>{}, m0); end_position();
        chunk(SW, IC<9>{}, m0); end_position();
        chunk(SW, IC<10>{}, m0); end_position();
        chunk(SW, IC<11>{}, m0); end_position();
      }
      static_assert(S == 3 || S == 8 || S == 12, "chunk list above");
    }
    if (grp == 0)
      for (int i = 0; i < off1; ++i) end_position();
  };
  if constexpr (EPI == EPI_IMG_QKV) {  // the v tiles run the normal MFMA form (lane = feature), the q | k tiles the swapped one
    if (n0 >= 2 * p.H * 32) run(IC<0>{});
    else run(IC<1>{});
  } else {
    run(IC<EPI == EPI_IMG_VT ? 0 : 1>{});
  }
#undef FD_SB
#undef FD_STAMP
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
static void launch(const GemmImgArgs& p, hipStream_t s) {
  static bool attr_set[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp_kernel<EPI, false>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp_kernel<EPI, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr_set[dev] = true;
  }
  int grid = n_cu_of_current_device() / 8 * 8;  // every XCD needs at least one workgroup per column tile (<= 3)
  if (grid < 32) grid = 32;
  if (p.stamps) hipLaunchKernelGGL((gemm_pp_kernel<EPI, true>), dim3(grid), dim3(NTHR), SMEM, s, p);
  else hipLaunchKernelGGL((gemm_pp_kernel<EPI, false>), dim3(grid), dim3(NTHR), SMEM, s, p);
}

}  // namespace gp

void launch_gemm_pp(int epilogue, const GemmImgArgs& p, hipStream_t s) {
  switch (epilogue) {
    case EPI_IMG_GELU: gp::launch<EPI_IMG_GELU>(p, s); break;
    case EPI_IMG_LN: gp::launch<EPI_IMG_LN>(p, s); break;
    case EPI_IMG_QK: gp::launch<EPI_IMG_QK>(p, s); break;
    case EPI_IMG_BIAS: gp::launch<EPI_IMG_BIAS>(p, s); break;
    case EPI_IMG_QKV: gp::launch<EPI_IMG_QKV>(p, s); break;
    default: gp::launch<EPI_IMG_VT>(p, s); break;
  }
}

}  // namespace fdmi
