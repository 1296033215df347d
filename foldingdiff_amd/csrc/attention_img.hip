// Multi-head self-attention on row images (HF BertSelfAttention 4.11.3 semantics incl. relative_key and the
// additive -10000 key mask; math and citations in attention_f32.hip), three contractions as fp16 hi/lo split
// triples on v_mfma_f32_32x32x16_f16 (fp32-class accuracy, see gemm_img.hip).
//
// Inputs are what the QK / V^T GEMM epilogues wrote, already split and already in the LDS layout:
//     q, k  [b][h] grouped images (img_common.h): [position / 32][unit 0-7][position % 32][16 B], units 0-3 hi d0-31, 4-7 lo.
//           k goes to LDS as unit-major pieces of 8 rows ([piece j][position p][row % 8][16 B], p holding unit p ^ (j & 1):
//           conflict-free 16-byte operand fetches), copied with per-lane source offsets: eight neighbouring lanes read the
//           eight rows of one unit = one 128-byte line
//     vt  [b][h][32-key block][32 d][128 B]    V transposed: hi keys | lo keys as sixteen 8-byte units, unit u stored
//                                               at u ^ vt_swz(d)  (img_common.h: conflict-free 8-byte operand fetches)
// so filling LDS is a linear LDS-DMA copy: no VGPR round trip, no split arithmetic, no ds_write.
//
// One 8-wave workgroup per CU, persistent; its waves form two 4-wave groups, each walking its own stream of
// (sequence, head, query group) x key-tile positions with split-phase prefetch (per group one K buffer, one V buffer):
//     [A] K(p) landed       | S^T = K Q^T and the relative-key band  | [B] K region free -> DMA K(p+1)
//         softmax                                                    | [C] V(p) landed
//         O^T += V^T P^T                                             | [D] V region free -> DMA V(p+1)
// Q goes from global memory straight to registers, one item ahead.
// so every copy has a whole compute phase to land.  Waits are counted (s_waitcnt vmcnt(N)), never 0 in the loop.
// A wave owns one 32-query row block; S^T (keys x queries) puts a query's scores in one lane pair, so softmax is
// in-register and P is already the B operand of the PV MFMA; the relative_key term is dense 32x32 tiles
// R^T = E Q^T over the band (rows = band, columns = queries), skewed through a per-wave LDS scratch of TWO tiles in which
// the band rows of a tile pair are consecutive 128-byte rows: a score's band value sits at an address LINEAR in its key,
// so the gather is one ds_read_b32 with an immediate offset and one fma per score (see the comments inside).
// ctx leaves as a grouped row image (img_common.h) with one 128-byte block per (token row, head) for the attention-output GEMM.
#include <cstdlib>

#include "fdmi_kernels.h"
#include "img_common.h"

#include <type_traits>

#ifndef FDMI_ATTN_STAG
#define FDMI_ATTN_STAG 1  // the two wave groups of a workgroup run half a position apart (0: in lockstep, as in round 2)
#endif

namespace fdmi {
namespace ai {

template <int V> using IC = std::integral_constant<int, V>;

// 32-bit LDS addresses kept in registers across the item loop (a generic pointer would pin a register PAIR each)
typedef const __attribute__((address_space(3))) float* lds_cf32_t;
typedef const __attribute__((address_space(3))) u32x4* lds_cu128_t;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long long)(lds_ptr_t)(const_cast<void*>(p)); }
__device__ __forceinline__ float lds_f32(unsigned a) { return *(lds_cf32_t)(unsigned long long)a; }
__device__ __forceinline__ u32x4 lds_u128(unsigned a) { return *(lds_cu128_t)(unsigned long long)a; }


constexpr float PS = 1024.0f;  // probabilities are <= 1
constexpr float kLog2e = 1.44269504088896341f;
constexpr float kInvSqrtD = 0.17677669529663687f;  // 1 / sqrt(32)

__device__ __forceinline__ float exp2_neg(float x) { return __builtin_amdgcn_exp2f(x); }
// reductions over a lane pair (lane, lane ^ 32): one v_permlane32_swap leaves [x lo | x lo] and [x hi | x hi], and the symmetric
// operation gives BOTH halves the same bits (__shfl_xor is a ds_bpermute: six address instructions and an LDS round trip)
__device__ __forceinline__ void pair_halves(float x, float& lo, float& hi) {
  unsigned a = __builtin_bit_cast(unsigned, x), b = a;
  swap32(a, b);  // lanes 32-63 of a <-> lanes 0-31 of b
  lo = __builtin_bit_cast(float, a);
  hi = __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float pair_max(float x) {
  float lo, hi;
  pair_halves(x, lo, hi);
  return fmaxf(lo, hi);
}
__device__ __forceinline__ float pair_sum(float x) {
  float lo, hi;
  pair_halves(x, lo, hi);
  return lo + hi;
}

constexpr int ceil_div(int a, int b) { return (a + b - 1) / b; }

template <int T, bool ELDS>
struct Geo {
  static constexpr int LP = 32 * T;
  static constexpr int K_BYTES = LP * 128, V_BYTES = LP * 128;
  static constexpr int KC = ceil_div(K_BYTES, 1024), VC = ceil_div(V_BYTES, 1024);
  static constexpr int KW = ceil_div(KC, 4), VW = ceil_div(VC, 4);  // DMA pieces per wave (4 waves per group)
  static constexpr int E_BYTES = ELDS ? 32 * 1024 : 0;              // distance table image, 255 rows x 128 B (maxpos <= 128)
  static constexpr int OFF_K = 0, OFF_V = OFF_K + KW * 4 * 1024, OFF_R = OFF_V + VW * 4 * 1024;
  static constexpr int G_REL = OFF_R + 4 * 2 * 32 * 32 * 4, G_ABS = OFF_R;  // bytes per group (skew scratch: two 4 KiB tile slots per wave)
};

// compile-time loop
template <int LO, int HI, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (LO < HI) {
    f(IC<LO>{});
    static_for<LO + 1, HI>(f);
  }
}

// The relative_key band of a position as a sequence of operations on band tiles q = 0 .. T (32 band rows each):
//     M(q)  R^T tile q = E_tile Q^T, six MFMAs into accumulator q & 1
//     W(q)  accumulator q & 1 -> scratch slot q & 1 (16 ds_write_addtid_b32)
//     G(q)  S^T tile T-1-q += the band values of the tile PAIR (q, q+1): 16 gathers + 16 fmas
// software pipelined so that the MFMAs of tile q+2 are issued BEFORE the gathers of pair q (they run under the gather's LDS round
// trip and fmas; round 3 ran MFMA -> write -> read -> fma of one tile strictly in sequence):
//     M0 M1 W0 W1 M2 | G0 W2 M3 | G1 W3 M4 | G2 W4 | G3          (T = 4; operations on tiles beyond T do not exist)
// W(q+2) follows G(q) in program order: it overwrites the slot pair q read (LDS operations of one wave execute in order).
struct BandOp {
  int kind, q;  // 0 M, 1 W, 2 G, -1 nothing
};
template <int T>
constexpr BandOp band_op(int i) {
  BandOp o{-1, 0};
  if (i < 5) {
    const int kinds[5] = {0, 0, 1, 1, 0}, qs[5] = {0, 1, 0, 1, 2};
    o = BandOp{kinds[i], qs[i]};
  } else {
    const int k = i - 5, q = k / 3, w = k % 3;
    o = w == 0 ? BandOp{2, q} : (w == 1 ? BandOp{1, q + 2} : BandOp{0, q + 3});
  }
  if (o.kind == 2 ? o.q > T - 1 : o.q > T) o.kind = -1;
  return o;
}

// Two 4-wave groups per workgroup (8 waves, one workgroup per CU), each group walking its own stream of
// (item, key tile) positions in lockstep with the other (shared barriers), both sharing ONE copy of the distance
// table in LDS (ELDS: single key tile and maxpos <= 128, i.e. every released configuration).  With the table in LDS the
// S / band phase issues no vector-memory instruction at all, so the K / V copies in flight are never waited for early
// (loads retire in order: a table fetch from L2 behind a V copy used to stall the band phase until V had landed).
// Q comes straight from global memory into registers, one item ahead.
// SAFE: every wait is vmcnt(0) (debug aid for the counted-wait bookkeeping)
// RKQ: position_embedding_type = "relative_key_query" (HF BertSelfAttention 4.11.3; offered by the reference's training CLI,
// bin/train.py:305-307): the score also gets  k_r . E[l - r + maxpos - 1]  -- the same band of the distance table paired with
// the KEYS.  Per S^T tile t that is two more dense 32 x 32 tiles K_t E^T (rows = keys, i.e. the S^T tile's own rows) against the
// band tiles T-1-t and T-t, skewed through a scratch slot while it is free: lane (query l31) register r (key kl) reads
// scratch[r][half][(l31 - kl + 31) & 31] -- the scratch ROW is the register's own, only the column is skewed.
template <int T, bool REL, bool ELDS, int NG, bool SAFE, bool PROF = false, bool RKQ = false>
__global__ __launch_bounds__(256 * NG) void attn_img_kernel(AttnImgArgs p) {
  static_assert(REL || !RKQ, "relative_key_query is a relative position type");
  // STAG: group 1 runs HALF A POSITION behind group 0.  A position is two halves of three barrier-separated segments each:
  //     H1  [A] ctx store(p-1), copy V(p), S^T tiles + band operations [0, BP1)  |  copy K(p+1), band operations [BP1, BP2)  |  band operations [BP2, end)
  //     H2  [B] fetch Q(p+1), softmax                                             |  [C] P V                                 |  [D] item bookkeeping
  // (round 4: cycle stamps showed H2 on the critical path in every segment -- 7.4 k cycles of which ~2.5 k were the ISSUE of its
  // vector-memory instructions, ~100-200 cycles apiece -- while H1 needed 3.5 k of its 7.1 k: the copies and the ctx store now sit in H1)
  // so on every SIMD one wave is in H1 while the other is in H2: the plain fp32 VALU instructions of the one issue beside the
  // MFMAs of the other (profiles/r03_coissue2_probe.log; this file is compiled with -fno-slp-vectorize, packed fp32 would
  // serialize with the matrix pipe).  In lockstep both waves of a SIMD ran the same phase and the matrix pipe idled through
  // every softmax (29 % MFMA utilisation, cycle stamps).  Group 1 starts with an empty half, group 0 ends with one.
  constexpr bool STAG = FDMI_ATTN_STAG != 0 && NG == 2;
#ifndef FDMI_ATTN_KEARLY
#define FDMI_ATTN_KEARLY 0  // (measured: no gain, profiles/r04_attention_ab4.log)
#endif
#ifndef FDMI_ATTN_ILP
#define FDMI_ATTN_ILP 0  // (measured: no gain, profiles/r04_attention_ab4.log)  S^T tiles in pairs and band tiles 0 / 1 together: two independent accumulators alternate on the matrix pipe
#endif
#ifndef FDMI_ATTN_SNEXT
#define FDMI_ATTN_SNEXT 0  // (measured: 11 % SLOWER, profiles/r04_attention_ab5_snext.log)  1: the S^T tiles of position p + 1 are multiplied at
                           // the END of position p (behind barrier [D], in registers that are free since P V) instead of in front of
                           // the band operations of p + 1.  The stamps of one group had suggested it: 7.0 k cycles in H1 against 4.3 k of
                           // work in H2 -- but the third of H2 behind [D] (V copy issue, ctx pack + store) is no slack
#endif
#ifndef FDMI_ATTN_VLATE
#define FDMI_ATTN_VLATE 0  // 1: the V copy and the ctx store wait for barrier [A] of the next position (H1) instead of running behind [D]
#endif
  constexpr bool V_LATE = FDMI_ATTN_VLATE != 0;
  // K(p+1) is copied behind H1's first inner barrier, when every wave of the group has finished S^T(p) -- the band does not read K
  // (the relative_key_query key term does: there, and without the staggered schedule's inner barriers, the copy waits for [B])
  constexpr bool K_EARLY = FDMI_ATTN_KEARLY != 0 && STAG && !RKQ;
#ifndef FDMI_ATTN_DBG
#define FDMI_ATTN_DBG 0  // ablation builds (wrong results): linear LDS addresses for 1 the V reads, 2 the K reads; 32 no arithmetic (copies, barriers and stores only); 64 no copies / loads / stores in the item loop (arithmetic only)
#endif
#ifndef FDMI_ATTN_BP1
#define FDMI_ATTN_BP1 5   // band operations in front of the first / second barrier of H1 (see band_op); same-box, attention at C2:
#define FDMI_ATTN_BP2 8   // (5, 8) 101.2 us, (4, 10) 104.8, (5, 11) 105.1, (7, 11) 107.5; round-3 kernel 106.1 (profiles/r04_attention_ab1.log)
#endif
  constexpr int NOPS = REL ? 5 + 3 * T : 0;
  constexpr int BP1 = FDMI_ATTN_BP1 < NOPS ? FDMI_ATTN_BP1 : NOPS, BP2 = FDMI_ATTN_BP2 < NOPS ? (FDMI_ATTN_BP2 > BP1 ? FDMI_ATTN_BP2 : BP1) : NOPS;
  using G = Geo<T, ELDS>;
  constexpr int LP = G::LP;
  constexpr int GSZ = REL ? G::G_REL : G::G_ABS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, wq = wid & 3;
  const int half = lane >> 5, l31 = lane & 31;
  unsigned char* Es = smem;
  unsigned char* gbase = smem + G::E_BYTES + grp * GSZ;
  unsigned char* Ks = gbase + G::OFF_K;
  unsigned char* Vt = gbase + G::OFF_V;
  // this wave's skew scratch: two 4 KiB slots, band tile q in slot q & 1, row (band index within the tile) x 128 B, column = query
  unsigned char* Rw = gbase + G::OFF_R + wq * 8192;
  const unsigned rw_lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lds_ptr_t)(Rw));
  // MFMA row i of a band tile computes band row pi(i) of the tile, pi(8a + 4h + e) = 8a + 2e + h: the C/D layout puts row
  // 8a + 4h + e into register 4a + e of half-wave h, and ds_write_addtid_b32 of register r lands at (2r + h) * 128 + 4 * column --
  // with the rows permuted like this the scratch row IS the band index.
  const int pi31 = (l31 & 24) | ((l31 & 3) << 1) | ((l31 >> 2) & 1);
  // gather base of this lane: query l31, key kl = kl_r + 4 half (register r: kl_r = (r & 3) + 8 (r >> 2)) needs band index
  // j = l31 - kl + 31 of its tile pair, i.e. byte j * 128 + 4 l31 = gb + (27 - kl_r) * 128 when the pair's lower tile sits in slot 0
  const unsigned gb = lds_addr(Rw) + (unsigned)((l31 - 4 * half + 4) * 128 + 4 * l31);
  // ... and when it sits in slot 1 (odd pairs) the two slots are swapped: scores whose band index is >= 32 (l31 > kl) read 4096
  // bytes lower, the others 4096 bytes higher.  One address register per score register, shared by every odd pair and computed
  // once per kernel (opaque to the compiler, which otherwise keeps the sixteen +-4096 and adds gb to each at every use).
  unsigned godd[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int kl = (r & 3) + 8 * (r >> 2) + 4 * half;
    godd[r] = REL && T >= 2 ? (l31 > kl ? gb - 4096 : gb + 4096) : gb;
    if (REL && T >= 2) asm volatile("" : "+v"(godd[r]));
  }
  const int H = p.H, nqg = p.NKT;  // query groups == key tiles
  const int nitems = p.B * H * nqg;
  const int gstride = NG * gridDim.x;

  // ---- the stream of (item, key tile) positions of this group: items 2 blockIdx + grp, + 2 gridDim, ...
  struct Pos { int item, kt, b, h, qg, len, nkt; };
  auto load_item = [&](Pos& s, int item) {
    s.item = item;
    s.kt = 0;
    const int it = item < nitems ? item : nitems - 1;
    s.qg = it % nqg;
    s.h = (it / nqg) % H;
    s.b = it / (nqg * H);
    s.len = p.lens[s.b];
    s.nkt = (s.len + LP - 1) / LP;  // key tiles holding at least one unmasked key (the rest contribute exactly 0)
  };
  // item + gstride without the three integer divisions of load_item (scalar divisions are ~25 instructions each, per item and wave):
  // (b, h, qg) += (gs_b, gs_h, gs_q) with carries
  const int gs_q = gstride % nqg, gs_h = (gstride / nqg) % H, gs_b = gstride / (nqg * H);
  auto advance = [&](Pos& s) {  // next position; past the end it stays on the last one (copies are repeated, harmlessly)
    if (s.kt + 1 < s.nkt) { ++s.kt; return; }
    if (s.item + gstride < nitems) {
      s.item += gstride;
      s.kt = 0;
      int qg = s.qg + gs_q, hh = s.h + gs_h, bb = s.b + gs_b;
      if (qg >= nqg) { qg -= nqg; ++hh; }
      if (hh >= H) { hh -= H; ++bb; }
      s.qg = qg; s.h = hh; s.b = bb;
      s.len = p.lens[bb];
      s.nkt = (s.len + LP - 1) / LP;
    }
  };
  auto is_last = [&](const Pos& s) { return s.kt + 1 >= s.nkt && s.item + gstride >= nitems; };
  // positions of a group's stream (both groups loop to the longer one: the barriers are shared)
  auto stream_len = [&](int g) {
    int n = 0;
    for (int it = NG * blockIdx.x + g; it < nitems; it += gstride) {
      const int bb = it / (nqg * H);
      n += (p.lens[bb] + LP - 1) / LP;
    }
    return n;
  };
  const int my_len = stream_len(grp), other_len = NG == 2 ? stream_len(grp ^ 1) : 0;
  const int niter = my_len > other_len ? my_len : other_len;
  const bool has_work = my_len > 0;

  // linear LDS-DMA copies; wave wq of the group takes pieces wq, wq + 4, ...; indices past the end repeat the last piece
  auto copy = [&](const unsigned char* src, int bytes, int npieces, int per_wave, unsigned char* dst_base) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < per_wave; ++i) {
      int piece = wq + 4 * i;
      piece = piece < npieces ? piece : npieces - 1;
      dma16(rs, (lds_ptr_t)(dst_base) + piece * 1024, lane * 16, piece * 1024);
    }
  };
  // K tile (LP keys = LP / 32 whole groups): piece j = keys 8j .. 8j+7; wave wq takes pieces wq, wq + 4, ... (all of wq's parity;
  // KC = 4 T pieces: no clamping).  Lane -> (unit position lane / 8, key lane % 8).
  const int kvoff = (((lane >> 3) ^ (wq & 1)) * 512) + (lane & 7) * 16;
  auto issue_k = [&](const Pos& s) {
    const size_t bh = (size_t)s.b * H + s.h;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(p.kbuf + (bh * p.LTOT + (size_t)s.kt * LP) * 128), 0, G::K_BYTES, 0x00020000);
#pragma unroll
    for (int i = 0; i < G::KW; ++i) {
      const int piece = wq + 4 * i;
      dma16(rs, (lds_ptr_t)(Ks) + piece * 1024, kvoff, (piece >> 2) * 4096 + (piece & 3) * 128);
    }
  };
  auto issue_v = [&](const Pos& s) {
    const size_t bh = (size_t)s.b * H + s.h;
    copy(p.vbuf + (bh * p.LTOT + (size_t)s.kt * LP) * 128, G::V_BYTES, G::VC, G::VW, Vt);
  };
  // Q operand of this lane: query l31 of the wave's row block, d = 16c + 8 half + j: units 2c + half (hi), 4 + 2c + half (lo)
  // It is fetched straight into the operand registers qh / ql when the NEXT position starts an item: that happens behind barrier
  // [B] of an item's last position, and the S^T / band phase -- the only reader of qh / ql -- of that position is over by then.
  f16x8 qh[2], ql[2];
  auto load_q = [&](const Pos& s) {
    const size_t bh = (size_t)s.b * H + s.h;
    // the wave's 32 queries are one group of the grouped image: unit u of query l31 at (u * 32 + l31) * 16
    const u32x4* grp0 = reinterpret_cast<const u32x4*>(p.qbuf + (bh * p.LTOT + (size_t)s.qg * LP + 32 * (wq < T ? wq : 0)) * 128);
    qh[0] = __builtin_bit_cast(f16x8, grp0[half * 32 + l31]);
    qh[1] = __builtin_bit_cast(f16x8, grp0[(2 + half) * 32 + l31]);
    ql[0] = __builtin_bit_cast(f16x8, grp0[(4 + half) * 32 + l31]);
    ql[1] = __builtin_bit_cast(f16x8, grp0[(6 + half) * 32 + l31]);
  };

  Pos cur, nxt;
  load_item(cur, NG * blockIdx.x + grp);
  // ELDS: LDS row rho of the table copy holds table row clamp(rho - esh, 0, 2 maxpos - 2) for ALL 256 rows, esh = max(0, LP - maxpos):
  // every band row a wave can ask for (rho = maxpos - LP + esh + 32 wq + 32 q + row, 0 <= rho <= 255) exists in LDS, so the
  // fragment addresses need no clamp, are the same for every item, and tile q sits 4096 q bytes behind tile 0 (immediate offsets).
  // Rows outside the table are only ever paired with padding keys / queries (L <= maxpos): any finite content will do.
  const int esh = ELDS ? (LP > p.maxpos ? LP - p.maxpos : 0) : 0;
  if constexpr (REL && ELDS) {
    // distance table -> LDS once per workgroup: 32 pieces of 8 rows, unit u of LDS row rho stored at u ^ ((rho >> 1) & 7)
    const int nrow_e = 2 * p.maxpos - 1;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(p.demb)), 0, nrow_e * 128, 0x00020000);
#pragma unroll
    for (int i = 0; i < 8 / NG; ++i) {
      const int piece = wid + 4 * NG * i;  // LDS rows 8 piece .. 8 piece + 7
      const int rho = 8 * piece + (lane >> 3);
      int row = rho - esh;
      row = row < 0 ? 0 : (row > nrow_e - 1 ? nrow_e - 1 : row);
      dma16(rs, (lds_ptr_t)(Es) + piece * 1024, row * 128 + (((lane & 7) ^ ((rho >> 1) & 7)) << 4), 0);
    }
  }
  // ELDS: addresses of this lane's four operand units of band tile 0 (MFMA row l31 -> band row pi31; `eun`: un-permuted rows for the
  // relative_key_query key term); tile q is at + 4096 q
  unsigned eaddr[4] = {0, 0, 0, 0}, eun[4] = {0, 0, 0, 0};
  if constexpr (REL && ELDS) {
    const int rho0 = p.maxpos - LP + esh + 32 * wq;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rp = rho0 + pi31, ru = rho0 + l31;
      eaddr[k] = lds_addr(Es) + (unsigned)(rp * 128 + (((2 * k + half) ^ ((rp >> 1) & 7)) << 4));
      eun[k] = lds_addr(Es) + (unsigned)(ru * 128 + (((2 * k + half) ^ ((ru >> 1) & 7)) << 4));
      asm volatile("" : "+v"(eaddr[k]));
      if (RKQ) asm volatile("" : "+v"(eun[k]));
    }
  }
  issue_k(cur);
  load_q(cur);
  issue_v(cur);
  nxt = cur;
  bool done = is_last(cur) || !has_work;
  advance(nxt);

  const float s_scale = kLog2e * kInvSqrtD / (p.q_scale * p.k_scale);  // raw MFMA sums -> log2 domain
  const float mask_raw = -10000.0f * kLog2e / s_scale;                  // (1 - mask) * -10000 at the raw scale
  float m_run = -INFINITY, l_run = 0.f;
  f32x16 oacc;
  bool pend = false;         // the previous position ended an item: its ctx block is stored behind barrier [A] of this one
  unsigned pend_voff = 0;    // ... at this lane offset (beyond the buffer for rows that are no token)
  float pend_onorm = 0.f;
  int pend_hoff = 0;
  const bool rec = PROF && blockIdx.x == 0 && p.stamps != nullptr && grp == 0;
  unsigned long long* st = PROF ? p.stamps + (size_t)wq * 64 * 8 : nullptr;
  int slot = 0;
#define FD_STAMP(i) do { if (PROF) { if (rec && slot < 64 && lane == 0) st[slot * 8 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
#define FD_SB() __builtin_amdgcn_sched_barrier(0)

  bool stored_prev = false;  // (!V_LATE) the previous position ended an item: 4 ctx stores are younger than its V copy
  // ctx[token row][head block] of the item that ended = O^T[d][query] / l_run: register r = 4q + e <-> d = 8q + 4 half + e (quad
  // layout).  Grouped image (the rows of a unit are adjacent); 32-bit lane offset into a buffer descriptor over the image; lanes
  // whose row is no token have an offset beyond the descriptor's range and the hardware drops their stores (whatever they computed
  // stays in their own lane pair: the half-wave exchange pairs the two halves of ONE query)
  auto flush_ctx = [&]() {
    if (!pend) return;
    float o[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = oacc[r] * pend_onorm;
    u32x4 h0, h1, lo0, lo1;
    pack_block(o, 1.0f, h0, h1, lo0, lo1);
    const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(p.ctx, 0, 0xFFFFFF00u, 0x00020000);
    if (FDMI_EPI_DBG != 1 && !(FDMI_ATTN_DBG & 64)) {
      __builtin_amdgcn_raw_buffer_store_b128(h0, rsc, (int)pend_voff, pend_hoff, 0);
      __builtin_amdgcn_raw_buffer_store_b128(h1, rsc, (int)pend_voff, pend_hoff + 512, 0);
      __builtin_amdgcn_raw_buffer_store_b128(lo0, rsc, (int)pend_voff, pend_hoff + 4 * 512, 0);
      __builtin_amdgcn_raw_buffer_store_b128(lo1, rsc, (int)pend_voff, pend_hoff + 5 * 512, 0);
      store_guard(h0, h1);
      store_guard(lo0, lo1);
    }
    pend = false;
  };
  constexpr bool S_NEXT = FDMI_ATTN_SNEXT != 0;
  f32x16 sacc[T];
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // S^T tiles of the position whose K is in LDS and whose Q is in qh / ql: rows = keys r0 + 32 t + rowmap(r, half), cols = queries
  // l0 + l31; raw MFMA sums (scale q_scale * k_scale).  tl_s = the position's live 32-key tiles.
  // Two tiles at a time (FDMI_ATTN_ILP): their accumulators alternate on the matrix pipe
  auto s_tiles = [&](int tl_s) {
    auto kfrag = [&](int t, int c, f16x8& kh, f16x8& kl) {
      // key 32 t + l31: piece 4 t + l31 / 8, unit u at position u ^ (piece & 1)
      const unsigned char* pc = (FDMI_ATTN_DBG & 2) ? Ks + t * 4096 + lane * 16 : Ks + (size_t)(4 * t + (l31 >> 3)) * 1024 + (l31 & 7) * 16;
      const int ksz = (FDMI_ATTN_DBG & 2) ? 0 : (l31 >> 3) & 1;
      kh = *reinterpret_cast<const f16x8*>(pc + (((2 * c + half) ^ ksz) << 7));
      kl = *reinterpret_cast<const f16x8*>(pc + (((4 + 2 * c + half) ^ ksz) << 7));
    };
#pragma unroll
    for (int t = 0; t < T; t += (FDMI_ATTN_ILP != 0 ? 2 : 1)) {
      if (t >= tl_s) continue;
      if (FDMI_ATTN_ILP != 0 && t + 1 < T && t + 1 < tl_s) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          f16x8 kh0, kl0, kh1, kl1;
          kfrag(t, c, kh0, kl0);
          kfrag(t + 1, c, kh1, kl1);
          // (the first product starts from the inline constant 0: no accumulator zeroing pass)
          sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh0, qh[c], c == 0 ? zero16 : sacc[t], 0, 0, 0);
          sacc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh1, qh[c], c == 0 ? zero16 : sacc[t + 1], 0, 0, 0);
          sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh0, ql[c], sacc[t], 0, 0, 0);
          sacc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh1, ql[c], sacc[t + 1], 0, 0, 0);
          sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl0, qh[c], sacc[t], 0, 0, 0);
          sacc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl1, qh[c], sacc[t + 1], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          f16x8 kh, kl;
          kfrag(t, c, kh, kl);
          sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[c], c == 0 ? zero16 : sacc[t], 0, 0, 0);
          sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[c], sacc[t], 0, 0, 0);
          sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[c], sacc[t], 0, 0, 0);
        }
      }
    }
  };
  if constexpr (STAG) {
    if (grp == 1) {  // (its copies of the distance table must have landed before group 0 reads the table behind this barrier)
      FD_WAIT_VM(G::VW);
      barrier_keep_vm();
      barrier_keep_vm();
      barrier_keep_vm();
    }
  }
  for (int iter = 0; iter < niter; ++iter) {
    FD_STAMP(0);
    const bool live = iter < my_len;  // this group still has positions (otherwise it only keeps the barriers company)
    const int b = cur.b, h = cur.h, qg = cur.qg, kt = cur.kt, len = cur.len;
    const int row0 = p.seq_row0[b];
    const int nrows = p.seq_row0[b + 1] - row0;  // token rows of the sequence (multiple of 8, >= real rows)
    const int Lb = p.nrow[b];                    // real rows: positions >= Lb are not keys at all
    const int l0 = qg * LP + 32 * wq;
    const bool active = live && wq < T && l0 < nrows;
    const bool compute = active && !(FDMI_ATTN_DBG & 32);
    const int r0 = kt * LP;
    const bool first_tile = kt == 0;
    const bool nxt_first = nxt.kt == 0 && !done;  // the next position starts an item: its Q is fetched with its K
    // 32-key tiles of this position that hold at least one key (packed rows: Lb = the sequence's length; keys beyond it are
    // -inf scores = probability exactly 0): the S^T, band, softmax and P V work of the other tiles is skipped, bit-identically
    // (they would add exact zeros).  Padded rows: Lb = L, every tile is live.
    int tl = (Lb - r0 + 31) >> 5;
    tl = tl < 1 ? 1 : (tl > T ? T : tl);
    const int q0 = T - tl;  // first band tile any live S^T tile needs

    // ---- [A] K(p) (+ Q(p)) landed.  Younger: nothing (V_LATE), else the V(p) pieces and the ctx stores of the previous position
    if (SAFE || V_LATE) FD_WAIT_VM(0);
    else if (stored_prev) FD_WAIT_VM(G::VW + 4);
    else FD_WAIT_VM(G::VW);
    FD_STAMP(1);
    barrier_keep_vm();
    FD_STAMP(2);
    if constexpr (V_LATE) flush_ctx();
    // every wave of the group passed [D] of the previous position: its V region is free (position 0's copy is in the prologue)
    if (V_LATE && iter > 0 && !(FDMI_ATTN_DBG & 64)) issue_v(cur);
    if (first_tile) {
      m_run = -INFINITY;
      l_run = 0.f;
      if constexpr (!ELDS) {  // (ELDS: one key tile per item, the first P V product starts from the constant 0)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
      }
    }

    f32x16 racc[2];
    // ---- relative_key band.  Band origin of this wave's row block: band index x of the position <-> table row
    //   m = (maxpos-1) - (LP-1) + LP (qg-kt) + 32 wq + x,  x = 32 q + (row of tile q); S^T tile t element (key kl, query ql) needs
    //   x = 32 (T-1-t) + ql - kl + 31.  Rows outside the table (clamped) are only ever paired with padding keys / queries (L <= maxpos).
    const int mbase = (p.maxpos - 1) - (LP - 1) + LP * (qg - kt) + 32 * wq;
    auto band_rows = [&](int qq, bool permuted, u32x4 (&e)[4]) {  // units half, 2 + half (hi d 0-15 / 16-31), 4 + half, 6 + half (lo) of the lane's band row
      if constexpr (ELDS) {
        const unsigned* ea = permuted ? eaddr : eun;
        e[0] = lds_u128(ea[0] + (unsigned)(qq * 4096));
        e[1] = lds_u128(ea[1] + (unsigned)(qq * 4096));
        e[2] = lds_u128(ea[2] + (unsigned)(qq * 4096));
        e[3] = lds_u128(ea[3] + (unsigned)(qq * 4096));
      } else {
        int m = mbase + 32 * qq + (permuted ? pi31 : l31);
        m = m < 0 ? 0 : (m > 2 * (p.maxpos - 1) ? 2 * (p.maxpos - 1) : m);
        const u32x4_t* erow = p.demb + (size_t)m * 8;
        e[0] = erow[half]; e[1] = erow[2 + half]; e[2] = erow[4 + half]; e[3] = erow[6 + half];
      }
    };
    auto op_M = [&](auto QQ) {  // R^T tile qq (rows = band rows pi(i), columns = this wave's queries) -> racc[qq & 1]
      constexpr int qq = decltype(QQ)::value;
      if (qq >= q0) {
        u32x4 e[4];
        band_rows(qq, true, e);
        const f16x8 eh0 = __builtin_bit_cast(f16x8, e[0]), el0 = __builtin_bit_cast(f16x8, e[2]);
        const f16x8 eh1 = __builtin_bit_cast(f16x8, e[1]), el1 = __builtin_bit_cast(f16x8, e[3]);
        f32x16 ra;
        ra = __builtin_amdgcn_mfma_f32_32x32x16_f16(eh0, qh[0], zero16, 0, 0, 0);
        ra = __builtin_amdgcn_mfma_f32_32x32x16_f16(el0, qh[0], ra, 0, 0, 0);
        ra = __builtin_amdgcn_mfma_f32_32x32x16_f16(eh0, ql[0], ra, 0, 0, 0);
        ra = __builtin_amdgcn_mfma_f32_32x32x16_f16(eh1, qh[1], ra, 0, 0, 0);
        ra = __builtin_amdgcn_mfma_f32_32x32x16_f16(el1, qh[1], ra, 0, 0, 0);
        ra = __builtin_amdgcn_mfma_f32_32x32x16_f16(eh1, ql[1], ra, 0, 0, 0);
        racc[qq & 1] = ra;
      }
    };
    // M(0) and M(1) together: their accumulators are independent, so the twelve MFMAs alternate instead of forming two dependent
    // chains of six (H1 is a latency chain: stamps, profiles/r04_attention_stamps_v2.log)
    auto op_MM = [&]() {
      if (q0 <= 0) {
        u32x4 e0[4], e1[4];
        band_rows(0, true, e0);
        band_rows(1, true, e1);
        f32x16 ra = zero16, rb = zero16;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const f16x8 ah = __builtin_bit_cast(f16x8, e0[c]), al = __builtin_bit_cast(f16x8, e0[2 + c]);
          const f16x8 bh = __builtin_bit_cast(f16x8, e1[c]), bl = __builtin_bit_cast(f16x8, e1[2 + c]);
          ra = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[c], ra, 0, 0, 0);
          rb = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, qh[c], rb, 0, 0, 0);
          ra = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[c], ra, 0, 0, 0);
          rb = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, qh[c], rb, 0, 0, 0);
          ra = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[c], ra, 0, 0, 0);
          rb = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ql[c], rb, 0, 0, 0);
        }
        racc[0] = ra;
        racc[1] = rb;
      } else {
        op_M(IC<1>{});  // (tile 0 is not needed: q0 >= 1)
      }
    };
    // scratch slot qq & 1 <- racc[qq & 1]: register r of all 64 lanes is ONE ds_write_addtid_b32 (address = M0 + offset + 4 * lane,
    // no address VGPR: 2 LDS cycles instead of 4) and lands as the two 128-byte band rows 2r, 2r + 1 (see pi31)
    auto op_W = [&](auto QQ) {
      constexpr int qq = decltype(QQ)::value;
      if (qq >= q0) {
        const f32x16 ra = racc[qq & 1];
        const unsigned m0v = rw_lds + (unsigned)((qq & 1) * 4096);
        unsigned keep;
        asm volatile(
            // (the MFMA results need 12 wait states before a non-MFMA reader: hipcc pads nothing inside an asm statement)
            "s_nop 7\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %17\n\ts_nop 2\n\t"
            "ds_write_addtid_b32 %1 offset:0\n\tds_write_addtid_b32 %2 offset:256\n\t"
            "ds_write_addtid_b32 %3 offset:512\n\tds_write_addtid_b32 %4 offset:768\n\t"
            "ds_write_addtid_b32 %5 offset:1024\n\tds_write_addtid_b32 %6 offset:1280\n\t"
            "ds_write_addtid_b32 %7 offset:1536\n\tds_write_addtid_b32 %8 offset:1792\n\t"
            "ds_write_addtid_b32 %9 offset:2048\n\tds_write_addtid_b32 %10 offset:2304\n\t"
            "ds_write_addtid_b32 %11 offset:2560\n\tds_write_addtid_b32 %12 offset:2816\n\t"
            "ds_write_addtid_b32 %13 offset:3072\n\tds_write_addtid_b32 %14 offset:3328\n\t"
            "ds_write_addtid_b32 %15 offset:3584\n\tds_write_addtid_b32 %16 offset:3840\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(ra[0]), "v"(ra[1]), "v"(ra[2]), "v"(ra[3]), "v"(ra[4]), "v"(ra[5]), "v"(ra[6]), "v"(ra[7]),
              "v"(ra[8]), "v"(ra[9]), "v"(ra[10]), "v"(ra[11]), "v"(ra[12]), "v"(ra[13]), "v"(ra[14]), "v"(ra[15]),
              "s"(m0v)
            : "memory");
      }
    };
    // relative_key_query: the key term of band tile qq, K_t E_qq^T for the (at most two) S^T tiles t that pair it, through scratch
    // slot `sl` (free at the call sites below).  Columns = band rows in their natural order: the table rows are re-read un-permuted.
    auto rkq_tile = [&](auto QQ, int sl) {
      constexpr int qq = decltype(QQ)::value;
      if constexpr (RKQ) {
        if (qq >= q0) {
          u32x4 e[4];
          band_rows(qq, false, e);
          const f16x8 eh0 = __builtin_bit_cast(f16x8, e[0]), el0 = __builtin_bit_cast(f16x8, e[2]);
          const f16x8 eh1 = __builtin_bit_cast(f16x8, e[1]), el1 = __builtin_bit_cast(f16x8, e[3]);
          const float rk = p.r_scale_k;  // q_scale / table scale: this term at the raw scale of the scores
          float* Rk = reinterpret_cast<float*>(Rw + sl * 4096);
#pragma unroll
          for (int side = 0; side < 2; ++side) {
            const int t = side == 0 ? T - 1 - qq : T - qq;  // side 0: qq is the lower tile of t's pair, side 1: the upper one
            if (t < 0 || t >= T || t >= tl) continue;
            const unsigned char* pc = Ks + (size_t)(4 * t + (l31 >> 3)) * 1024 + (l31 & 7) * 16;
            const int ksz = (l31 >> 3) & 1;
            const f16x8 kh0 = *reinterpret_cast<const f16x8*>(pc + (((0 + half) ^ ksz) << 7));
            const f16x8 kl0 = *reinterpret_cast<const f16x8*>(pc + (((4 + half) ^ ksz) << 7));
            const f16x8 kh1 = *reinterpret_cast<const f16x8*>(pc + (((2 + half) ^ ksz) << 7));
            const f16x8 kl1 = *reinterpret_cast<const f16x8*>(pc + (((6 + half) ^ ksz) << 7));
            f32x16 kacc;
            kacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh0, eh0, zero16, 0, 0, 0);
            kacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh0, el0, kacc, 0, 0, 0);
            kacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl0, eh0, kacc, 0, 0, 0);
            kacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh1, eh1, kacc, 0, 0, 0);
            kacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh1, el1, kacc, 0, 0, 0);
            kacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl1, eh1, kacc, 0, 0, 0);
            // scratch [register r][lane]: row (key) 8q + 4 half + e of column l31 at r * 256 + lane * 4
#pragma unroll
            for (int r = 0; r < 16; ++r) Rk[r * 64 + lane] = kacc[r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int kl = (r & 3) + 8 * (r >> 2) + 4 * half;
              const float g = Rk[r * 64 + half * 32 + ((l31 - kl + 31) & 31)];
              const bool mine = side == 0 ? l31 <= kl : l31 > kl;  // band index l31 - kl + 31 < 32: the pair's lower tile
              sacc[t][r] = __builtin_fmaf(mine ? g : 0.f, rk, sacc[t][r]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the next user overwrites this scratch
            __builtin_amdgcn_wave_barrier();
          }
        }
      }
    };
    // S^T tile T-1-q += r_scale * band value: ONE ds_read_b32 (immediate offset) and ONE fma per score.  Pair q has its lower tile
    // in slot q & 1: for even q the 63 band rows are consecutive in the scratch; for odd q the two slots are swapped, so the lanes
    // whose band index is >= 32 (l31 > kl) start 4096 bytes lower and the others 4096 bytes higher (one v_cndmask per score).
    auto op_G = [&](auto Q) {
      constexpr int q = decltype(Q)::value;
      constexpr int t = T - 1 - q;
      if (q >= q0) {
        float gth[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int klr = (r & 3) + 8 * (r >> 2);
          gth[r] = lds_f32(((q & 1) ? godd[r] : gb) + (unsigned)((27 - klr) * 128));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[t][r] = __builtin_fmaf(gth[r], p.r_scale, sacc[t][r]);
      }
      if constexpr (RKQ) {  // slot q & 1 is free now (W(q+2) comes after this); the last pair also frees the other slot
        rkq_tile(IC<q>{}, q & 1);
        if constexpr (q == T - 1) rkq_tile(IC<T>{}, T & 1);
      }
    };
    auto band_ops = [&](auto LO, auto HI) {
      static_for<decltype(LO)::value, decltype(HI)::value>([&](auto I) {
        constexpr BandOp o = band_op<T>(decltype(I)::value);
        if constexpr (o.kind == 0 && o.q == 0 && FDMI_ATTN_ILP != 0) op_MM();
        else if constexpr (o.kind == 0 && o.q == 1 && FDMI_ATTN_ILP != 0) {}
        else if constexpr (o.kind == 0) op_M(IC<o.q>{});
        else if constexpr (o.kind == 1) op_W(IC<o.q>{});
        else if constexpr (o.kind == 2) op_G(IC<o.q>{});
        FD_SB();
      });
    };
    if (compute) {
      if (!S_NEXT || iter == 0) s_tiles(tl);  // (S_NEXT: done at the end of the previous position, except for the stream's first)
      if constexpr (REL) band_ops(IC<0>{}, IC<BP1>{});
    }

    if constexpr (STAG) barrier_keep_vm();  // (staggered schedule: three segments per half position, see the kernel header)
    if constexpr (K_EARLY) {
      if (!(FDMI_ATTN_DBG & 64)) issue_k(nxt);
    }
    if constexpr (REL) {
      if (compute) band_ops(IC<BP1>{}, IC<BP2>{});
    }
    if constexpr (STAG) barrier_keep_vm();
    if constexpr (REL) {
      if (compute) band_ops(IC<BP2>{}, IC<NOPS>{});
    }

    // ---- [B] every wave is done with K and with its Q operands: fetch the next item's Q (and copy the next position's K unless K_EARLY)
    FD_STAMP(3);
    barrier_keep_vm();
    if (!(FDMI_ATTN_DBG & 64)) {
      if constexpr (!K_EARLY) issue_k(nxt);
      if (nxt_first) load_q(nxt);
    }
    FD_STAMP(4);

    if (compute) {
      // mask + online softmax over keys (log2 domain): this lane + its partner (lane ^ 32) hold one query's scores.
      // Per 32-key tile: entirely below len -> no mask arithmetic at all (a wave-uniform test)
      float mt = -INFINITY;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        if (t >= tl) continue;
        if (r0 + 32 * (t + 1) <= len) {
#pragma unroll
          for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sacc[t][r]);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = r0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
            float sc = sacc[t][r];
            if (key >= len) sc += mask_raw;     // (1 - mask) * -10000   (modelling.py:452)
            if (key >= Lb) sc = -INFINITY;      // not a key at all (tile padding / rows that do not exist)
            sacc[t][r] = sc;
            mt = fmaxf(mt, sc);
          }
        }
      }
      mt = pair_max(mt);
      const float m_new = fmaxf(m_run, mt);
      const float nm = __builtin_fmaf(-m_new, s_scale, 10.0f);   // + log2(PS): p' = PS * 2^((u - m) * s_scale)
      static_assert(PS == 1024.0f, "exponent offset above is log2(PS)");
      float psum = 0.f;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        if (t >= tl) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pexp = exp2_neg(__builtin_fmaf(sacc[t][r], s_scale, nm));
          sacc[t][r] = pexp;
          psum += pexp;
        }
      }
      psum = pair_sum(psum);
      if constexpr (ELDS) {  // a single key tile per item: nothing to rescale
        l_run = psum;        // carries the factor PS
      } else {
        const float alpha = exp2_neg((m_run - m_new) * s_scale);  // first tile: 2^-inf = 0 (accumulators are 0 anyway)
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
      }
      m_run = m_new;
    }

    // ---- [C] V(p) landed.  Younger: the K(p+1) pieces (+ Q loads).
    if (SAFE) FD_WAIT_VM(0);
    else if (nxt_first) FD_WAIT_VM(G::KW + 4);
    else FD_WAIT_VM(G::KW);
    FD_STAMP(5);
    barrier_keep_vm();
    FD_STAMP(6);

    if (compute) {
      // O^T += V^T P^T :  A[i = d = l31][position (c, half, j)] = V[key(c,half,j)][d],  B = P (registers),
      //                   key(c, half, j) = 32 t + 16 c + 8 (j>>2) + 4 half + (j&3)   (the C/D row map)
      // (The per-tile liveness test keeps hipcc from merging the V fetches of two key tiles into ds_read2st64_b64: they are lone
      // ds_read_b64, which is what vt_swz is conflict free for (img_common.h).  Writing the P V loop in tile pairs to get the merged
      // instruction back is 8 % SLOWER (101 -> 109 us: more registers and both tiles' split arithmetic in front of the MFMAs).)
      const unsigned char* vrow = (FDMI_ATTN_DBG & 1) ? Vt + lane * 8 : Vt + (size_t)l31 * 128;
      const int sz = (FDMI_ATTN_DBG & 1) ? 0 : vt_swz(l31);
#pragma unroll
      for (int t = 0; t < T; ++t) {
        if (t >= tl) continue;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          // p = hi + lo, pairwise (img_common.h: split_pair)
          u32x4 phu, plu;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            unsigned hv, lv;
            split_pair(sacc[t][8 * c + 2 * j], sacc[t][8 * c + 2 * j + 1], hv, lv);
            phu[j] = hv;
            plu[j] = lv;
          }
          const f16x8 ph = __builtin_bit_cast(f16x8, phu), pl = __builtin_bit_cast(f16x8, plu);
          // V operand of key block t: keys 16c + 4 half + {0..3} = unit 4c + half, and + 8 = unit 4c + half + 2; lo plane + 8
          const unsigned char* blk = vrow + t * 4096;
          const int ua = 4 * c + half;
          const u32x2 vh0 = *reinterpret_cast<const u32x2*>(blk + ((ua ^ sz) << 3));
          const u32x2 vh1 = *reinterpret_cast<const u32x2*>(blk + (((ua + 2) ^ sz) << 3));
          const u32x2 vl0 = *reinterpret_cast<const u32x2*>(blk + (((ua + 8) ^ sz) << 3));
          const u32x2 vl1 = *reinterpret_cast<const u32x2*>(blk + (((ua + 10) ^ sz) << 3));
          const u32x4 vhu = {vh0[0], vh0[1], vh1[0], vh1[1]};
          const u32x4 vlu = {vl0[0], vl0[1], vl1[0], vl1[1]};
          const f16x8 vh = __builtin_bit_cast(f16x8, vhu), vl = __builtin_bit_cast(f16x8, vlu);
          // (tile 0 is always live; ELDS: a single key tile per item, so its first product opens the accumulator)
          oacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, (ELDS && t == 0 && c == 0) ? zero16 : oacc, 0, 0, 0);
          oacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, oacc, 0, 0, 0);
          oacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, oacc, 0, 0, 0);
        }
      }
    }

    // ---- [D] every wave is done with V: copy the next position's V (V_LATE: behind barrier [A] of the next position)
    FD_STAMP(7);
    ++slot;
    if (S_NEXT) FD_WAIT_VM(0);  // K(p+1) (+ Q) landed -- requested behind [B], a softmax and a P V product ago; nothing else is in flight
    barrier_keep_vm();
    if (!V_LATE && !(FDMI_ATTN_DBG & 64)) issue_v(nxt);

    stored_prev = kt + 1 >= cur.nkt;
    if (stored_prev) {  // the item ends: its ctx block leaves now, or (V_LATE) behind the next barrier [A] (oacc is not touched before)
      const int l = l0 + l31;
      const bool ok = active && l < nrows;
      const int row = row0 + l;
      pend = true;
      pend_onorm = p.ctx_scale / (p.v_scale * l_run);  // l_run and the accumulator both carry PS; at the ctx image's scale
      pend_voff = ok ? (unsigned)((((row >> 5) * H * 8 + 2 * half) * 32 + (row & 31)) * 16) : 0xFFFFFF00u;
      pend_hoff = h * 4096;  // block h of the row: (.. * H + h) * 8 units * 32 rows * 16 B
      if constexpr (!V_LATE) flush_ctx();
    }
    if (S_NEXT && !done) {
      // S^T of the next position: its K is in LDS behind [D], its Q in qh / ql since [B], and the score registers are free since P V
      const int nrows_n = p.seq_row0[nxt.b + 1] - p.seq_row0[nxt.b];
      int tl_n = (p.nrow[nxt.b] - nxt.kt * LP + 31) >> 5;
      tl_n = tl_n < 1 ? 1 : (tl_n > T ? T : tl_n);
      if (wq < T && nxt.qg * LP + 32 * wq < nrows_n && !(FDMI_ATTN_DBG & 32)) s_tiles(tl_n);
    }
    if (!done) {
      cur = nxt;
      done = is_last(cur);
      advance(nxt);
    }
  }
  flush_ctx();  // (V_LATE: the stream's last item)
#undef FD_STAMP
#undef FD_SB
  if constexpr (STAG) {
    if (grp == 0) {
      barrier_keep_vm();
      barrier_keep_vm();
      barrier_keep_vm();
    }
  }
  FD_WAIT_VM(0);  // nothing may land in LDS after the workgroup has exited
}

template <int T, bool REL, bool ELDS, int NG, bool RKQ>
static void launch_ng(const AttnImgArgs& p, hipStream_t s) {
  using G = Geo<T, ELDS>;
  constexpr int smem = G::E_BYTES + NG * (REL ? G::G_REL : G::G_ABS);
  static const bool safe = [] { const char* e = getenv("FDMI_ATTN_SAFE"); return e && atoi(e) != 0; }();
  static bool attr_set[64] = {false};
  static int n_cu[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_img_kernel<T, REL, ELDS, NG, false, false, RKQ>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_img_kernel<T, REL, ELDS, NG, true, false, RKQ>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_img_kernel<T, REL, ELDS, NG, false, true, RKQ>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipDeviceProp_t prop;
    n_cu[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    attr_set[dev] = true;
  }
  const int nitems = p.B * p.H * p.NKT;
  int grid = n_cu[dev] * (NG == 1 ? 2 : 1);  // 8 waves per CU either way
  if (grid > (nitems + NG - 1) / NG) grid = (nitems + NG - 1) / NG;
  if (p.stamps) hipLaunchKernelGGL((attn_img_kernel<T, REL, ELDS, NG, false, true, RKQ>), dim3(grid), dim3(256 * NG), smem, s, p);
  else if (safe) hipLaunchKernelGGL((attn_img_kernel<T, REL, ELDS, NG, true, false, RKQ>), dim3(grid), dim3(256 * NG), smem, s, p);
  else hipLaunchKernelGGL((attn_img_kernel<T, REL, ELDS, NG, false, false, RKQ>), dim3(grid), dim3(256 * NG), smem, s, p);
}

// NG = groups (item streams) per workgroup = 2: one 8-wave workgroup per CU, both groups behind the same barriers, one
// copy of the distance table.  (NG = 1 -- independent 4-wave workgroups with a table copy each, exactly 80 KiB -- was
// measured at 179 vs 131 us: two such workgroups do not become co-resident, profiles/r02_attention_notes.log.)
template <int T, bool REL, bool ELDS>
static void launch(const AttnImgArgs& p, hipStream_t s) {
  if constexpr (REL) {
    if (p.rkq) {
      launch_ng<T, REL, ELDS, 2, true>(p, s);
      return;
    }
  }
  launch_ng<T, REL, ELDS, 2, false>(p, s);
}

}  // namespace ai

bool launch_attention_img(const AttnImgArgs& p, int L, hipStream_t s) {
  if (L < 1) return false;
  const int T = L > 128 ? 4 : (L + 31) / 32;  // keys per tile = 32 T; L > 128: 128-key tiles, online softmax
  const bool rel = p.demb != nullptr;
  const bool elds = rel && p.NKT == 1 && p.maxpos <= 128;  // the whole distance table fits 32 KB of LDS
#define FD_ATTN_IMG_CASE(TT)                                   \
  case TT:                                                     \
    if (!rel) ai::launch<TT, false, false>(p, s);              \
    else if (elds) ai::launch<TT, true, true>(p, s);           \
    else ai::launch<TT, true, false>(p, s);                    \
    break;
  switch (T) {
    FD_ATTN_IMG_CASE(1)
    FD_ATTN_IMG_CASE(2)
    FD_ATTN_IMG_CASE(3)
    FD_ATTN_IMG_CASE(4)
  }
#undef FD_ATTN_IMG_CASE
  return true;
}

}  // namespace fdmi
