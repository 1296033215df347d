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
// R = Q E^T over the band, skewed through a per-wave LDS scratch (see the comments inside).
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


constexpr float PS = 1024.0f;  // probabilities are <= 1
constexpr float kLog2e = 1.44269504088896341f;
constexpr float kInvSqrtD = 0.17677669529663687f;  // 1 / sqrt(32)

__device__ __forceinline__ float exp2_neg(float x) { return __builtin_amdgcn_exp2f(x); }

constexpr int ceil_div(int a, int b) { return (a + b - 1) / b; }

template <int T, bool ELDS>
struct Geo {
  static constexpr int LP = 32 * T;
  static constexpr int K_BYTES = LP * 128, V_BYTES = LP * 128;
  static constexpr int KC = ceil_div(K_BYTES, 1024), VC = ceil_div(V_BYTES, 1024);
  static constexpr int KW = ceil_div(KC, 4), VW = ceil_div(VC, 4);  // DMA pieces per wave (4 waves per group)
  static constexpr int E_BYTES = ELDS ? 32 * 1024 : 0;              // distance table image, 255 rows x 128 B (maxpos <= 128)
  static constexpr int OFF_K = 0, OFF_V = OFF_K + KW * 4 * 1024, OFF_R = OFF_V + VW * 4 * 1024;
  static constexpr int G_REL = OFF_R + 4 * 32 * 32 * 4, G_ABS = OFF_R;  // bytes per group (skew scratch: 4 KiB per wave)
};

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
// band tiles T-1-t and T-t, skewed through the same per-wave scratch: lane (query l31) register r (key kl) reads
// scratch[r][half][(l31 - kl + 31) & 31] -- the scratch ROW is the register's own, only the column is skewed.
template <int T, bool REL, bool ELDS, int NG, bool SAFE, bool PROF = false, bool RKQ = false>
__global__ __launch_bounds__(256 * NG) void attn_img_kernel(AttnImgArgs p) {
  static_assert(REL || !RKQ, "relative_key_query is a relative position type");
  // STAG: group 1 runs HALF A POSITION behind group 0.  A position is two halves of three barrier-separated segments each:
  //     H1  [A] S^T tiles + band tiles [0, B1)  |  band tiles [B1, B2)  |  band tiles [B2, T]           (matrix heavy)
  //     H2  [B] issue K(p+1), softmax           |  [C] P V              |  [D] issue V(p+1), ctx store  (VALU heavy)
  // so on every SIMD one wave is in H1 while the other is in H2: the plain fp32 VALU instructions of the one issue beside the
  // MFMAs of the other (profiles/r03_coissue2_probe.log; this file is compiled with -fno-slp-vectorize, packed fp32 would
  // serialize with the matrix pipe).  In lockstep both waves of a SIMD ran the same phase and the matrix pipe idled through
  // every softmax (29 % MFMA utilisation, cycle stamps).  Group 1 starts with an empty half, group 0 ends with one.
  constexpr bool STAG = FDMI_ATTN_STAG != 0 && NG == 2;
#ifndef FDMI_ATTN_DBG
#define FDMI_ATTN_DBG 0  // ablation builds (wrong results): linear LDS addresses for 1 the V reads, 2 the K reads, 4 the skew gather, 8 the table reads, 16 plain scratch stores
#endif
#ifndef FDMI_ATTN_B1
#define FDMI_ATTN_B1 1
#define FDMI_ATTN_B2 3
#endif
  constexpr int B1 = T >= 3 ? FDMI_ATTN_B1 : 1, B2 = T >= 3 ? FDMI_ATTN_B2 : 2;  // band tile runs (T + 1 tiles)
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
  float* Rw = reinterpret_cast<float*>(gbase + G::OFF_R) + wq * 32 * 32;
  // this wave's skew scratch: LDS byte address for the addtid stores, and the row this lane reads back (query l31 =
  // 8q + 4h + e lives at register slot 4q + e, half h)
  const unsigned rw_lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lds_ptr_t)(reinterpret_cast<unsigned char*>(Rw)));
  const float* Rrow = Rw + ((l31 >> 3) * 4 + (l31 & 3)) * 64 + ((l31 >> 2) & 1) * 32;
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
  auto advance = [&](Pos& s) {  // next position; past the end it stays on the last one (copies are repeated, harmlessly)
    if (s.kt + 1 < s.nkt) { ++s.kt; return; }
    if (s.item + gstride < nitems) load_item(s, s.item + gstride);
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
  auto load_q = [&](const Pos& s, u32x4 (&q)[4]) {
    const size_t bh = (size_t)s.b * H + s.h;
    // the wave's 32 queries are one group of the grouped image: unit u of query l31 at (u * 32 + l31) * 16
    const u32x4* grp0 = reinterpret_cast<const u32x4*>(p.qbuf + (bh * p.LTOT + (size_t)s.qg * LP + 32 * (wq < T ? wq : 0)) * 128);
    q[0] = grp0[half * 32 + l31]; q[1] = grp0[(2 + half) * 32 + l31]; q[2] = grp0[(4 + half) * 32 + l31]; q[3] = grp0[(6 + half) * 32 + l31];
  };

  // band weights of the relative_key skew (see the S phase): r_scale where register r of this lane belongs to the lower /
  // upper S^T tile of a band tile's pair, else 0
  float bw_lo[16], bw_hi[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int kl = (r & 3) + 8 * (r >> 2) + 4 * half;
    bw_lo[r] = (REL && l31 <= kl) ? p.r_scale : 0.f;
    bw_hi[r] = (REL && l31 > kl) ? p.r_scale : 0.f;
  }
  Pos cur, nxt;
  load_item(cur, NG * blockIdx.x + grp);
  u32x4 qn[4];
  if constexpr (REL && ELDS) {
    // distance table -> LDS once per workgroup: 32 pieces of 8 rows, unit u of row m stored at u ^ ((m >> 1) & 7)
    const int nrow_e = 2 * p.maxpos - 1;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(p.demb)), 0, nrow_e * 128, 0x00020000);
#pragma unroll
    for (int i = 0; i < 8 / NG; ++i) {
      const int piece = wid + 4 * NG * i;  // rows 8 piece .. 8 piece + 7
      const int row = 8 * piece + (lane >> 3);
      dma16(rs, (lds_ptr_t)(Es) + piece * 1024, row * 128 + (((lane & 7) ^ ((row >> 1) & 7)) << 4), 0);
    }
  }
  issue_k(cur);
  load_q(cur, qn);
  issue_v(cur);
  nxt = cur;
  bool done = is_last(cur) || !has_work;
  advance(nxt);

  const float s_scale = kLog2e * kInvSqrtD / (p.q_scale * p.k_scale);  // raw MFMA sums -> log2 domain
  const float mask_raw = -10000.0f * kLog2e / s_scale;                  // (1 - mask) * -10000 at the raw scale
  f16x8 qh[2], ql[2];
  float m_run = -INFINITY, l_run = 0.f;
  f32x16 oacc;
  bool stored_prev = false;  // the previous position ended an item (4 ctx stores are younger than its V copy)
  const bool rec = PROF && blockIdx.x == 0 && p.stamps != nullptr && grp == 0;
  unsigned long long* st = PROF ? p.stamps + (size_t)wq * 64 * 8 : nullptr;
  int slot = 0;
#define FD_STAMP(i) do { if (PROF) { if (rec && slot < 64 && lane == 0) st[slot * 8 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)

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
    const int r0 = kt * LP;
    const bool first_tile = kt == 0;
    const bool nxt_first = nxt.kt == 0 && !done;  // the next position starts an item: its Q is fetched with its K

    // ---- [A] K(p) (+ Q(p)) landed.  Younger: V(p) pieces, and the ctx stores of the previous position.
    if (SAFE) FD_WAIT_VM(0);
    else if (stored_prev) FD_WAIT_VM(G::VW + 4);
    else FD_WAIT_VM(G::VW);
    FD_STAMP(1);
    barrier_keep_vm();
    FD_STAMP(2);
    if (first_tile) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        qh[c] = __builtin_bit_cast(f16x8, qn[c]);
        ql[c] = __builtin_bit_cast(f16x8, qn[2 + c]);
      }
      m_run = -INFINITY;
      l_run = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    }

    f32x16 sacc[T];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // ---- relative_key band (used inside the `active` blocks below)
    // R tile q: rows = queries rowmap(r, half), cols = band index 32 q + l31 (band origin: this wave's row
    // block).  S^T tile t element (key kl, query ql) needs band column j = ql - kl + 31 of the tile pair
    // (q = T-1-t, q+1): j < 32 -> tile q, else tile q+1 column j-32.  Band row of R tile q, column l31:
    //   m = (maxpos-1) - (LP-1) + LP (qg-kt) + 32 wq + 32 q + l31, clamped: rows outside the table are only
    // ever paired with padding keys / queries (L <= maxpos).
    auto band_tile = [&](auto QQ) {
      constexpr int qq = decltype(QQ)::value;
      int m = (p.maxpos - 1) - (LP - 1) + LP * (qg - kt) + 32 * wq + l31 + 32 * qq;
      m = m < 0 ? 0 : (m > 2 * (p.maxpos - 1) ? 2 * (p.maxpos - 1) : m);
      u32x4 e0, e1, e2, e3;  // units half, 2 + half (hi d 0-15 / 16-31), 4 + half, 6 + half (lo)
      if constexpr (ELDS) {
        const unsigned char* erow = (FDMI_ATTN_DBG & 8) ? Es + qq * 4096 + l31 * 16 : Es + m * 128;
        const int sz = (FDMI_ATTN_DBG & 8) ? 0 : (m >> 1) & 7;
        e0 = *reinterpret_cast<const u32x4*>(erow + ((half ^ sz) << 4));
        e1 = *reinterpret_cast<const u32x4*>(erow + (((2 + half) ^ sz) << 4));
        e2 = *reinterpret_cast<const u32x4*>(erow + (((4 + half) ^ sz) << 4));
        e3 = *reinterpret_cast<const u32x4*>(erow + (((6 + half) ^ sz) << 4));
      } else {
        const u32x4_t* erow = p.demb + (size_t)m * 8;
        e0 = erow[half]; e1 = erow[2 + half]; e2 = erow[4 + half]; e3 = erow[6 + half];
      }
      f32x16 racc;
      {
        const f16x8 eh0 = __builtin_bit_cast(f16x8, e0), el0 = __builtin_bit_cast(f16x8, e2);
        const f16x8 eh1 = __builtin_bit_cast(f16x8, e1), el1 = __builtin_bit_cast(f16x8, e3);
        racc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh[0], eh0, zero16, 0, 0, 0);
        racc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh[0], el0, racc, 0, 0, 0);
        racc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ql[0], eh0, racc, 0, 0, 0);
        racc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh[1], eh1, racc, 0, 0, 0);
        racc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh[1], el1, racc, 0, 0, 0);
        racc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ql[1], eh1, racc, 0, 0, 0);
      }
      // scratch layout [register r][lane] (R row 8q + 4 half + e of lane-column l31 at r * 256 + lane * 4): the 16 stores
      // are ds_write_addtid_b32 (address = M0 + offset + 4 * lane, no address VGPR: 2 LDS cycles instead of 4)
      {
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
            : "v"(racc[0]), "v"(racc[1]), "v"(racc[2]), "v"(racc[3]), "v"(racc[4]), "v"(racc[5]), "v"(racc[6]), "v"(racc[7]),
              "v"(racc[8]), "v"(racc[9]), "v"(racc[10]), "v"(racc[11]), "v"(racc[12]), "v"(racc[13]), "v"(racc[14]), "v"(racc[15]),
              "s"(rw_lds)
            : "memory");
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // scratch row = query l31 (this lane); band column j = l31 - kl + 31 of the tile PAIR lives in tile q for
      // j < 32 (l31 <= kl) and in tile q+1, column j - 32, otherwise: both read scratch column j & 31
      float gth[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kl = (r & 3) + 8 * (r >> 2) + 4 * half;
        gth[r] = (FDMI_ATTN_DBG & 4) ? Rw[r * 64 + lane] : Rrow[(l31 - kl + 31) & 31];
      }
      // band weights bw_lo / bw_hi (r_ratio where the element belongs to the lower / upper tile of the pair, else 0): one
      // fma per element and tile instead of a select + fma (band values are finite MFMA sums, so 0 * value = 0)
      if (qq < T) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[T - 1 - qq][r] = __builtin_fmaf(gth[r], bw_lo[r], sacc[T - 1 - qq][r]);
      }
      if (qq > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[T - qq][r] = __builtin_fmaf(gth[r], bw_hi[r], sacc[T - qq][r]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the next band tile overwrites this scratch
      __builtin_amdgcn_wave_barrier();
      if constexpr (RKQ) {
        // the key term against the SAME band tile: S^T tile T-1-qq takes it as its lower tile, S^T tile T-qq as its upper one
        const f16x8 eh0 = __builtin_bit_cast(f16x8, e0), el0 = __builtin_bit_cast(f16x8, e2);
        const f16x8 eh1 = __builtin_bit_cast(f16x8, e1), el1 = __builtin_bit_cast(f16x8, e3);
        const float rk = p.r_scale_k / p.r_scale;  // band weights carry r_scale (k_scale / table scale); this term needs q_scale / table scale
#pragma unroll
        for (int side = 0; side < 2; ++side) {
          const int t = side == 0 ? T - 1 - qq : T - qq;
          if (t < 0 || t >= T) continue;
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
          // scratch [register r][lane]: row (key) 8q + 4 half + e of column l31 at r * 256 + lane * 4 (as above, plain stores)
#pragma unroll
          for (int r = 0; r < 16; ++r) Rw[r * 64 + lane] = kacc[r];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kl = (r & 3) + 8 * (r >> 2) + 4 * half;
            const float g = Rw[r * 64 + half * 32 + ((l31 - kl + 31) & 31)] * rk;
            sacc[t][r] = __builtin_fmaf(g, side == 0 ? bw_lo[r] : bw_hi[r], sacc[t][r]);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
    };
    // band tiles 0 .. T in three runs [0, B1) [B1, B2) [B2, T]: the staggered schedule puts a workgroup barrier between them
    auto band_run = [&](auto LO, auto HI) {
      constexpr int lo = decltype(LO)::value, hi = decltype(HI)::value;
      if constexpr (lo < hi && lo <= T) {
        band_tile(IC<lo>{});
        if constexpr (lo + 1 < hi && lo + 1 <= T) band_tile(IC<lo + 1>{});
        if constexpr (lo + 2 < hi && lo + 2 <= T) band_tile(IC<lo + 2>{});
        if constexpr (lo + 3 < hi && lo + 3 <= T) band_tile(IC<lo + 3>{});
        if constexpr (lo + 4 < hi && lo + 4 <= T) band_tile(IC<lo + 4>{});
      }
    };
    if (active) {
      // S^T tiles: rows = keys r0 + 32 t + rowmap(r, half), cols = queries l0 + l31; raw MFMA sums (scale q_scale * k_scale)
#pragma unroll
      for (int t = 0; t < T; ++t) {
        // key 32 t + l31: piece 4 t + l31 / 8, unit u at position u ^ (piece & 1)
        const unsigned char* pc = (FDMI_ATTN_DBG & 2) ? Ks + t * 4096 + lane * 16 : Ks + (size_t)(4 * t + (l31 >> 3)) * 1024 + (l31 & 7) * 16;
        const int ksz = (FDMI_ATTN_DBG & 2) ? 0 : (l31 >> 3) & 1;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const f16x8 kh = *reinterpret_cast<const f16x8*>(pc + (((2 * c + half) ^ ksz) << 7));
          const f16x8 kl = *reinterpret_cast<const f16x8*>(pc + (((4 + 2 * c + half) ^ ksz) << 7));
          // (the first product starts from the inline constant 0: no accumulator zeroing pass)
          sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[c], c == 0 ? zero16 : sacc[t], 0, 0, 0);
          sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[c], sacc[t], 0, 0, 0);
          sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[c], sacc[t], 0, 0, 0);
        }
      }
      if constexpr (REL) band_run(IC<0>{}, IC<B1>{});
    }

    if constexpr (STAG) barrier_keep_vm();  // (staggered schedule: three segments per half position, see the kernel header)
    if constexpr (REL) {
      if (active) band_run(IC<B1>{}, IC<B2>{});
    }
    if constexpr (STAG) barrier_keep_vm();
    if constexpr (REL) {
      if (active) band_run(IC<B2>{}, IC<T + 1>{});
    }

    // ---- [B] every wave is done with K: copy the next position's K, fetch the next item's Q
    FD_STAMP(3);
    barrier_keep_vm();
    issue_k(nxt);
    if (nxt_first) load_q(nxt, qn);
    FD_STAMP(4);

    if (active) {
      // mask + online softmax over keys (log2 domain): this lane + its partner (lane ^ 32) hold one query's scores
      float mt = -INFINITY;
      if (r0 + LP <= len) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sacc[t][r]);
      } else {
#pragma unroll
        for (int t = 0; t < T; ++t)
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
      mt = fmaxf(mt, __shfl_xor(mt, 32));
      const float m_new = fmaxf(m_run, mt);
      const float alpha = exp2_neg((m_run - m_new) * s_scale);  // first tile: 2^-inf = 0 (accumulators are 0 anyway)
      const float nm = __builtin_fmaf(-m_new, s_scale, 10.0f);   // + log2(PS): p' = PS * 2^((u - m) * s_scale)
      static_assert(PS == 1024.0f, "exponent offset above is log2(PS)");
      float psum = 0.f;
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pexp = exp2_neg(__builtin_fmaf(sacc[t][r], s_scale, nm));
          sacc[t][r] = pexp;
          psum += pexp;
        }
      psum += __shfl_xor(psum, 32);
      l_run = l_run * alpha + psum;  // carries the factor PS
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
    }

    // ---- [C] V(p) landed.  Younger: the K pieces (+ Q loads) just issued.
    if (SAFE) FD_WAIT_VM(0);
    else if (nxt_first) FD_WAIT_VM(G::KW + 4);
    else FD_WAIT_VM(G::KW);
    FD_STAMP(5);
    barrier_keep_vm();
    FD_STAMP(6);

    if (active) {
      // O^T += V^T P^T :  A[i = d = l31][position (c, half, j)] = V[key(c,half,j)][d],  B = P (registers),
      //                   key(c, half, j) = 32 t + 16 c + 8 (j>>2) + 4 half + (j&3)   (the C/D row map)
      const unsigned char* vrow = (FDMI_ATTN_DBG & 1) ? Vt + lane * 8 : Vt + (size_t)l31 * 128;
      const int sz = (FDMI_ATTN_DBG & 1) ? 0 : vt_swz(l31);
#pragma unroll
      for (int t = 0; t < T; ++t)
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
          oacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, oacc, 0, 0, 0);
          oacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, oacc, 0, 0, 0);
          oacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, oacc, 0, 0, 0);
        }
    }

    // ---- [D] every wave is done with V: copy the next position's V
    FD_STAMP(7);
    ++slot;
    barrier_keep_vm();
    issue_v(nxt);

    const bool item_ends = kt + 1 >= cur.nkt;
    if (item_ends) {
      // ctx[row0 + query][head h block] = O^T[d][query] / l_run: register r = 4q + e <-> d = 8q + 4 half + e (quad layout)
      const int l = l0 + l31;
      const bool ok = active && l < nrows;
      const float onorm = ok ? 1.0f / (p.v_scale * l_run) : 0.f;  // l_run and the accumulator both carry PS
      float o[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = ok ? oacc[r] * onorm : 0.f;
      store_block_g(p.ctx, H, ok ? row0 + l : 0, h, o, p.ctx_scale, half, ok);  // grouped image: the rows of a unit are adjacent
    }
    stored_prev = item_ends;
    if (!done) {
      cur = nxt;
      done = is_last(cur);
      advance(nxt);
    }
  }
#undef FD_STAMP
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
