// Token GEMMs of the encoder on row images (img_common.h): fp32-class accuracy on the fp16 matrix cores.
//
//   C[M,N] = A[M,K] * W[N,K]^T + bias[N]   then one of these epilogues
//     EPI_QK    q | k of HF BertSelfAttention (transformers 4.11.3, called from foldingdiff/modelling.py:473-480), stored as
//               grouped images per (sequence, head) that the attention kernel reads with contiguous accesses
//     EPI_VT    v, written transposed per (sequence, head, 32-key block): [d][hi keys | lo keys], swizzled
//     EPI_QKV   both in ONE launch (n_heads % 6 == 0: v starts on a 384-column tile)
//     EPI_GELU  BertIntermediate.dense / AnglesPredictor.dense1 + exact-erf GELU (modelling.py:195-196, :203-205)
//     EPI_LN    BertSelfOutput / BertOutput: LayerNorm(dense(x) + residual)
//
// Arithmetic: a product a*w is  a_hi*w_hi + a_hi*w_lo + a_lo*w_hi  on three v_mfma_f32_32x32x16_f16 into one
// fp32 accumulator (fp16 x fp16 products are exact in fp32; the dropped lo*lo term is 2^-22 relative).
//
// Structure (one persistent workgroup per CU: 8 compute waves + NL loader waves):
//  * BOTH operands live in HBM as hi|lo images, so a k-tile (32 k = one 128-byte block per row) is staged with LDS-DMA
//    (buffer_load_dwordx4 ... lds): no VGPR round trip, no split arithmetic, no ds_write in the k-loop.  A stage is a
//    sequence of 1 KiB pieces of 8 rows, each piece UNIT-major ([position p][row % 8][16 B], p holding unit p ^ (piece & 1)):
//    eight neighbouring lanes copy the eight rows of one unit, which are one 128-byte line of the grouped activation
//    image, and the fragment reads (ds_read_b128) are bank-conflict free.  The weight image in HBM is the stage byte for
//    byte ([384-row tile][k-tile][48 KiB], api.hip: pack_weight_tiles).
//  * loader waves issue every piece, paced by counted s_waitcnt vmcnt and a raw s_barrier that leaves the vector-memory
//    queue alone (one wave sustains ~22 B/clk out of L2, two ~40: scripts/probes/dma_probe.hip).  W ring 2 stages (L2
//    hits), A ring 3 stages (the HBM stream, two k-tiles ahead), ONE workgroup barrier per k-tile.
//  * 128 (M) x 384 (N) tile, 8 waves as 2 (M) x 4 (N), wave tile 64 x 96 = 2 x 3 MFMA tiles (96 accumulators).  A k-tile is
//    six groups of six MFMAs; the operands of a group are fetched while the previous group runs, and the barrier of the
//    NEXT k-tile sits before the last group, so the first fragments of the next k-tile fly while this one finishes.
//  * persistent: an XCD-aware tile list, the (tile, k-tile) pairs form one continuous stream; the column tiles of one
//    A panel run side by side on one XCD (one HBM read).
//  * "swapped" MFMA form (D^T = W A^T; everything but the v tiles): a lane owns ONE token row and 16 columns, so LayerNorm
//    statistics are in-lane sums, and a block is written with four 16-byte stores per lane after a half-wave register
//    exchange (img_common.h); the grouped image layout makes each of those stores 512 contiguous bytes per half-wave.
//  What bounds it (profiles/r02_gemm_ablation.log, r02_probes.log, r03_coissue2_probe.log, r03_pingpong_*.log): the k-loop runs
//  at ~2900 cycles per k-tile for 2304 matrix cycles -- the loader stream alone (64 KiB per k-tile through the CU's L2 -> LDS
//  path) needs 2200-2900; the tile epilogues run with the matrix pipe idle because BOTH waves of a SIMD are in the epilogue
//  together (plain fp32 VALU of one wave does issue beside another wave's MFMAs; only PACKED fp32 serializes with the matrix
//  pipe); and the outputs' HBM writes cost ~1.8 ms of a 7.7 ms timestep.  That last cost is the traffic itself, not a
//  chip-wide store burst: with the workgroups' start times spread over a tile period a tile costs the same (de-phase probe,
//  third part of the r02 ablation log).
#include <cstdlib>
#include <type_traits>

#include "fdmi_kernels.h"
#include "img_common.h"

namespace fdmi {
namespace gi {

template <int V> using IC = std::integral_constant<int, V>;


#ifndef FDMI_NL
#define FDMI_NL 2
#endif
#ifndef FDMI_LN_RD
#define FDMI_LN_RD 1  // LayerNorm epilogue: residual half-blocks requested ahead
#endif
#ifndef FDMI_GEMM_NOBAR
#define FDMI_GEMM_NOBAR 0  // ablation build (WRONG results): no workgroup barrier in the k-loop -- what would a barrier-free hand-off gain?
#endif
#define FD_KBAR() do { if (!FDMI_GEMM_NOBAR) barrier_keep_vm(); } while (0)
#ifndef FDMI_G6FIRST
#define FDMI_G6FIRST 1  // the wm 1 waves run MFMA group 6 before their first-fragment reads (0: every wave reads first); -0.6 % per timestep
#endif
constexpr int NL = FDMI_NL;                                      // loader waves (1, 2 or 4)
constexpr int BM = 128, BN = 384, NTHR = 64 * (8 + NL);         // 8 compute waves + NL loader waves
constexpr int W_STAGE = BN * 128, A_STAGE = BM * 128;           // bytes per k-tile stage
constexpr int NWS = 2, NAS = 3;
constexpr int OFF_A = NWS * W_STAGE;                            //  98,304
constexpr int OFF_PAR = OFF_A + NAS * A_STAGE;                  // 147,456: bias | gamma | beta (EPI_LN)
constexpr int OFF_RED = OFF_PAR + 3 * BN * 4;                   // 152,064: 2 x part[128][4]
constexpr int OFF_RI = OFF_RED + 2 * BM * 4 * 4;                // 156,160: (sequence, position) of each compute wave's 64 token rows, 8 x 512 B
constexpr int SMEM = OFF_RI + 8 * 512;                          // 160,256 B

// PROF: workgroup 0 records s_memtime stamps of its waves (debug instrumentation, FDMI_STAMPS=1):
//   stamps[EPI][wave][slot][6] = {k-tile top, after MFMA group 5, after the barrier of the next position, after issuing its
//   first fragment reads, after group 6, after the epilogue (last k-tile of a tile only)}
// DBG (ablation builds, FDMI_GEMM_DBG, wrong results by design): 1 = no DMA pieces inside the k-loop, 2 = no MFMAs,
// 3 = no fragment reads + no MFMAs (DMA only)
// TAIL: 1 = the tile list may end in row slices (below).  Its own instantiation: the slice-capable code is ~1 % slower on its
// whole-tile path for no visible reason in the loops (same instruction streams; code placement), so launches whose tile count is
// known on the host to fill whole rounds (BASELINE C2) keep the round-3 code byte for byte.
template <int EPI, bool SWAP, bool PROF, int TAIL>
__global__ __launch_bounds__(NTHR) void gemm_img_kernel(GemmImgArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wid & 3, wm = wid >> 2;
  const int half = lane >> 5, l31 = lane & 31;
  const int nk = p.K >> 5, rb = nk * 128;                 // k-tiles; bytes per image row (A and W share K)
  const int Mp = p.dims[1];
  const int tiles_n = (p.N + BN - 1) / BN, ntiles = (Mp / BM) * tiles_n;
  // tiles are dealt XCD-aware, n fastest: the workgroups of one XCD (blockIdx % 8) take neighbouring tiles at
  // the same time, so the column tiles of one A panel meet in that XCD's L2.
  // Tail (round 4): an XCD's tiles rarely fill whole rounds of its workgroups (BASELINE C3, chunk 0: 315 row panels = 39.4
  // per XCD for 32 workgroups -- the last round used to keep 7 or 8 of them busy for a whole tile time).  The tiles of an
  // incomplete last round are cut into S = 2 or 4 ROW SLICES of 64 / 32 rows (as many as the XCD's workgroups can take at
  // once); a slice runs on the wm 0 waves (the other four only keep the barriers company), streams the whole weight tile but only
  // its own rows of A, and its epilogue is a half / a quarter of a tile's.  Every epilogue is row-local (LayerNorm included), and
  // a row's arithmetic does not depend on the slice it is computed in: results are bit-identical with and without the tail.
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int tlo = (int)((long long)ntiles * xcd / 8), thi = (int)((long long)ntiles * (xcd + 1) / 8);
  const int first = tlo + jx, stride = per;
  const int nx = thi - tlo, nfull = nx / per, nrem = nx - nfull * per;
  const int tsplit = (TAIL && nrem > 0) ? (4 * nrem <= per ? 4 : (2 * nrem <= per ? 2 : 1)) : 1;  // slices per tail tile
  const bool has_tail = jx < nrem * tsplit;
  const int cnt = nfull + (has_tail ? 1 : 0);
  const int tail_tile = tlo + nfull * per + jx / tsplit, tail_slice = jx - (jx / tsplit) * tsplit;
  if (cnt == 0) return;
  const int G = cnt * nk;  // stream positions
  {  // bias (all N <= 3 BN columns) or bias | gamma | beta (EPI_LN, N <= BN) -> LDS, published by the first barrier
    float* par = reinterpret_cast<float*>(smem + OFF_PAR);
    if constexpr (EPI == EPI_IMG_LN) {
      for (int i = tid; i < BN; i += NTHR) {
        const bool ok = i < p.N;
        par[i] = ok ? p.bias[i] : 0.f;
        par[BN + i] = ok ? p.gamma[i] : 0.f;
        par[2 * BN + i] = ok ? p.beta[i] * p.out_scale : 0.f;  // beta at the OUTPUT image's scale (a power of two: exact)
      }
    } else if constexpr (EPI == EPI_IMG_QK || EPI == EPI_IMG_VT || EPI == EPI_IMG_QKV) {
      // bias at the scale of the image its column is written to: (acc os + b) sc == fma(acc, os sc, b sc) exactly for the
      // power-of-two sc, and the split then needs no multiply per element
      const int nq = p.H * 32;
      for (int i = tid; i < 3 * BN; i += NTHR) {
        const float sc = EPI == EPI_IMG_VT ? p.v_scale : (i < nq ? p.q_scale : (i < 2 * nq ? p.k_scale : p.v_scale));
        par[i] = i < p.N ? p.bias[i] * sc : 0.f;
      }
    } else {
      for (int i = tid; i < 3 * BN; i += NTHR) par[i] = i < p.N ? p.bias[i] : 0.f;
    }
  }

  auto tile_mn = [&](int ti, int& m0, int& n0) {
    const bool tail = ti >= nfull;
    const int tile = tail ? tail_tile : first + ti * stride;
    m0 = (tile / tiles_n) * BM + (tail ? tail_slice * (BM / tsplit) : 0);
    n0 = (tile - (tile / tiles_n) * tiles_n) * BN;
  };
  auto tile_rows = [&](int ti) { return ti >= nfull ? BM / tsplit : BM; };  // 128, or 64 / 32 (tail slices)

  // ================================================================ the loader waves
  // They issue every LDS-DMA piece of the workgroup: per k-tile 48 W pieces + 16 A pieces of 1 KiB (8 rows x 8 units,
  // unit-major).  With the pieces spread over the compute waves the issue stalls sat between their MFMAs; here the compute
  // waves only read fragments and issue MFMAs, and the copy stream runs beside them at its own pace.
  //   prologue A(0) W(0) A(1);  iteration g:  [vmcnt: W(g), A(g) landed] [barrier g] W(g+1) A(g+2)
  // The barrier publishes k-tile g to the compute waves and tells the loaders that compute(g-1) is over, which frees W slot
  // (g+1) & 1 and A slot (g+2) % 3.  The compute waves execute the same barriers and nothing else of this protocol.
  if (wid >= 8) {
    const int li = wid - 8;  // pieces j = li, li + NL, ...: all of one parity
    const int vw = lane * 16;  // W: the HBM image of a (tile, k-tile) IS the LDS stage (api.hip: pack_weight_tiles)
    // A: grouped image [row / 32][k-tile][unit][row % 32][16 B].  A piece = rows 8j..8j+7 x 8 units, read unit-major (eight
    // neighbouring lanes fetch the eight rows of one unit = one 128-byte line) and therefore ALSO unit-major in LDS:
    // [piece j][unit position p][row % 8][16 B], position p holding unit p ^ (j & 1) (bank-conflict-free fragment reads)
    const int va_e = (lane >> 3) * 512 + (lane & 7) * 16, va_o = ((lane >> 3) ^ 1) * 512 + (lane & 7) * 16;
    const int va_l = (li & 1) ? va_o : va_e;  // even NL: a loader's pieces share one parity
    int w_ti = 0, w_kt = 0, a_ti = 0, a_kt = 0, w_slot = 0, a_slot = 0, w_n0, a_m0;
    tile_mn(0, a_m0, w_n0);
    auto issue_w = [&]() {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<unsigned char*>(p.W) + (size_t)w_n0 * rb, 0, BN * rb, 0x00020000);
      lds_ptr_t dst = (lds_ptr_t)(smem) + w_slot * W_STAGE;
      const int so = w_kt * W_STAGE;
#pragma unroll
      for (int i = 0; i < 48 / NL; ++i) {
        const int j = li + i * NL;
        dma16(rs, dst + j * 1024, vw, so + j * 1024);
      }
      w_slot ^= 1;
      if (w_ti * nk + w_kt + 1 < G) {  // past the end: re-issue the last position (lands in a free slot, never read)
        if (++w_kt == nk) {
          w_kt = 0;
          ++w_ti;
          int mm;
          tile_mn(w_ti, mm, w_n0);
        }
      }
    };
    // The loaders' instruction stream IS the k-loop's critical path (ten more scalar instructions per stage cost 1 % of a q | k | v
    // launch): the stages of whole tiles keep the straight-line issue and the constant wait of rounds 2-3 (main loop); only the
    // last nk + 2 stages of a stream that ends in a tail slice go through the general path.
    int a_last = 16 / NL;  // pieces this wave issued for the most recent A stage (a tail slice has 8 or 4 pieces instead of 16)
    auto issue_a = [&](auto FULL) {  // FULL: the stage belongs to a whole tile (16 pieces)
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<unsigned char*>(p.A) + (size_t)a_m0 * rb, 0, BM * rb, 0x00020000);  // the tile's (up to) four 32-row groups
      lds_ptr_t dst = (lds_ptr_t)(smem) + OFF_A + a_slot * A_STAGE;
      const int so = a_kt * 4096;
      if constexpr (decltype(FULL)::value) {
#pragma unroll
        for (int i = 0; i < 16 / NL; ++i) {
          const int j = li + i * NL;
          dma16(rs, dst + j * 1024, (NL & 1) ? ((j & 1) ? va_o : va_e) : va_l, so + (j >> 2) * 32 * rb + (j & 3) * 128);
        }
      } else {
        const int np = tile_rows(a_ti) >> 3;
#pragma unroll
        for (int i = 0; i < 16 / NL; ++i) {
          const int j = li + i * NL;
          if (j < np) dma16(rs, dst + j * 1024, (NL & 1) ? ((j & 1) ? va_o : va_e) : va_l, so + (j >> 2) * 32 * rb + (j & 3) * 128);
        }
        a_last = np / NL;
      }
      a_slot = a_slot == NAS - 1 ? 0 : a_slot + 1;
      if (a_ti * nk + a_kt + 1 < G) {
        if (++a_kt == nk) {
          a_kt = 0;
          ++a_ti;
          int nn;
          tile_mn(a_ti, a_m0, nn);
        }
      }
    };
    issue_a(IC<0>{});
    issue_w();
    issue_a(IC<0>{});
    // The compute waves pass barrier g + 1 BEFORE the last MFMA group of position g (they prefetch the first fragments of
    // g + 1 behind it), so the two barriers of a LayerNorm epilogue follow the barrier of the next tile's first position.
    const bool sliced = has_tail && tsplit > 1;
    const int g_main = sliced ? (nfull * nk > 2 ? nfull * nk - 2 : 0) : G;  // iteration g issues A(g + 2): whole-tile stages only
    int g = 0, kt = 0;
    for (; g < g_main; ++g) {
      FD_WAIT_VM(16 / NL);
      FD_KBAR();
      issue_w();
      issue_a(IC<1>{});
      if constexpr (EPI == EPI_IMG_LN) {
        if (g > 0 && kt == 0) {
          barrier_keep_vm();
          barrier_keep_vm();
        }
      }
      if (++kt == nk) kt = 0;
    }
    for (; g < G; ++g) {
      // everything but the most recent A stage has landed (counted wait: its immediate is an instruction field)
      if (a_last == 16 / NL) FD_WAIT_VM(16 / NL);
      else if (a_last == 8 / NL) FD_WAIT_VM(8 / NL);
      else FD_WAIT_VM(4 / NL);
      FD_KBAR();
      issue_w();
      issue_a(IC<0>{});
      if constexpr (EPI == EPI_IMG_LN) {
        if (g > 0 && kt == 0) {
          barrier_keep_vm();
          barrier_keep_vm();
        }
      }
      if (++kt == nk) kt = 0;
    }
    if constexpr (EPI == EPI_IMG_LN) {
      barrier_keep_vm();
      barrier_keep_vm();
    }
    FD_WAIT_VM(0);  // nothing may land in LDS after the workgroup has exited
    return;
  }

  // ================================================================ the compute waves
  // ---- fragment reads: rows wn*96 + 32 jn + l31 (W) / wm*64 + 32 im + l31 (A); every such row has
  // swizzle (l31 >> 1) & 7, so a lane needs four unit offsets per operand: [k16 step c][plane]
  // (one set of per-lane offsets serves both operands: the wave's row bases are wave-uniform and ride in the slot base)
  int rd[2][2];  // both stages are unit-major pieces (see the loader): row l31 -> piece l31 / 8, unit u at position u ^ (piece & 1)
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
      rd[c][pl] = (l31 >> 3) * 1024 + ((((2 * c + half + 4 * pl) ^ ((l31 >> 3) & 1)) * 8 + (l31 & 7)) << 4);  // ([0][0]: see first_fragments)
  const int wbase = wn * 96 * 128, abase = OFF_A + wm * 64 * 128;

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

  // One k-tile = six groups of six MFMAs (k16 step c: wh ah | wh al | wl ah; consecutive MFMAs never share an accumulator).
  // The operands of a group are fetched while the previous group runs, and the barrier of the NEXT position sits before the
  // last group, so the first fragments of the next k-tile are on their way while this one finishes: the matrix pipe
  // never waits for a whole fragment set behind a barrier.
  // NI: 32-row blocks of the tile this wave computes (2; 1 in the 32-row tail slices)
  auto mm6 = [&](auto SW, auto NI, const f16x8 (&wf)[3], const f16x8 (&af)[2]) {  // SW: swapped form (D^T = W A^T)
#pragma unroll
    for (int jn = 0; jn < 3; ++jn)
#pragma unroll
      for (int im = 0; im < decltype(NI)::value; ++im)
        acc[jn][im] = decltype(SW)::value ? __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[jn], af[im], acc[jn][im], 0, 0, 0)
                                          : __builtin_amdgcn_mfma_f32_32x32x16_f16(af[im], wf[jn], acc[jn][im], 0, 0, 0);
  };
  auto ldw = [&](f16x8 (&d)[3], const unsigned char* wb, int off) {
#pragma unroll
    for (int jn = 0; jn < 3; ++jn) d[jn] = *reinterpret_cast<const f16x8*>(wb + off + jn * 4096);
  };
  auto lda = [&](auto NI, f16x8 (&d)[2], const unsigned char* ab, int off) {
#pragma unroll
    for (int im = 0; im < decltype(NI)::value; ++im) d[im] = *reinterpret_cast<const f16x8*>(ab + off + im * 4096);
  };

  // ---- epilogues.  SWAP form: lane (l31, half) owns token row  m0 + wm*64 + 32 im + l31  and, per MFMA tile jn,
  // the columns  n0 + wn*96 + 32 jn + 8q + 4 half + e  (register r = 4q + e): the quad layout of img_common.h.
  // Returns the wave's number of column blocks inside N (every such block issues exactly 8 store instructions:
  // pad rows are redirected to a scratch line instead of being predicated off, so the count is exact).
  // (sequence, position) of the lane's token rows, fetched one k-tile before the epilogue needs them
  // (sequence, position) of the wave's 64 token rows, landed in LDS by LDS-DMA one k-tile before the epilogue reads them:
  // held in registers across the tile's last k-tile they were spilled the moment they arrived (load, s_waitcnt vmcnt(0),
  // scratch store: three exposed round trips per v tile)
  constexpr int RD = FDMI_LN_RD;  // EPI_LN: residual half-blocks in flight
  u32x4 rres[RD][2];  // [slot][hi | lo]; see the epilogue
  auto prefetch_resid = [&](int ti) {
    if constexpr (EPI == EPI_IMG_LN) {
      int m0, n0, ln;
      tile_mn(ti, m0, n0);
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
      const int nb = p.N >> 5;
      const unsigned roff = (unsigned)((ln & 31) * 16 + (ln >> 5) * 1024);
#pragma unroll
      for (int i = 0; i < RD; ++i) {  // half-blocks 0, 1, 2 = (im 0, jn 0, hb 0 / 1), (im 0, jn 1, hb 0)
        const int jn = i / 2, hb = i % 2;
        int cb = wn * 3 + jn;
        cb = cb < nb ? cb : 0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned char*>(p.resid + ((size_t)((m0 + wm * 64) >> 5) * nb + cb) * 4096), 0, 4096, 0x00020000);
        rres[i][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(roff + hb * 512), 0, 0));
        rres[i][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(roff + 2048 + hb * 512), 0, 0));
      }
    }
  };
  auto prefetch_rowinfo = [&](auto SW, int ti) {
    constexpr bool kQK = EPI == EPI_IMG_QK || (EPI == EPI_IMG_QKV && decltype(SW)::value);
    constexpr bool kVT = EPI == EPI_IMG_VT || (EPI == EPI_IMG_QKV && !decltype(SW)::value);
    if constexpr (kQK || kVT) {
      int m0, n0, ln;
      tile_mn(ti, m0, n0);
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
      // (64 rows from the wave's first row; a 32-row tail slice at the very end of the row table has fewer behind it: the range
      // of the descriptor stops at the table's end -- reading on was a memory fault waiting for an unlucky allocation)
      const int left = (Mp - (m0 + wm * 64)) * 8;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<int2*>(p.rowinfo + m0 + wm * 64), 0, left < 512 ? (left > 0 ? left : 0) : 512, 0x00020000);
      const lds_ptr_t dst = (lds_ptr_t)(smem) + OFF_RI + wid * 512;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 4, ln * 4, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst + 256, 4, ln * 4, 256, 0, 0);
    }
  };
  // (the lane indices are re-derived inside the epilogue from an opaque copy: as values that live across the whole tile
  // loop they and everything computed from them get spilled, and a scratch reload behind stores waits for those stores)
  auto epilogue = [&](auto SW, auto NI, int ti) {
    constexpr int NIM = decltype(NI)::value;  // 32-row blocks of this wave's rows that exist (tail slices of 32 rows: 1)
    constexpr bool kQK = EPI == EPI_IMG_QK || (EPI == EPI_IMG_QKV && decltype(SW)::value);
    constexpr bool kVT = EPI == EPI_IMG_VT || (EPI == EPI_IMG_QKV && !decltype(SW)::value);
    int m0, n0, ln;
    tile_mn(ti, m0, n0);
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const int l31 = ln & 31, half = ln >> 5;
    const float os = p.acc_scale;
    const float* par0 = reinterpret_cast<const float*>(smem + OFF_PAR);
    // bias of columns 32 cbg + 8 q + 4 half .. + 3 at scale sc: the LDS image holds the first 3 BN columns (already scaled);
    // wider projections (d_model > 384) read the rest from global memory (cbg is wave-uniform)
    auto bias4 = [&](int cbg, int q, float sc) -> float4 {
      if (cbg * 32 < 3 * BN) return *reinterpret_cast<const float4*>(par0 + cbg * 32 + 8 * q + 4 * half);
      float4 b = *reinterpret_cast<const float4*>(p.bias + cbg * 32 + 8 * q + 4 * half);
      b.x *= sc; b.y *= sc; b.z *= sc; b.w *= sc;
      return b;
    };
    if constexpr (EPI == EPI_IMG_GELU || EPI == EPI_IMG_BIAS) {
      const int nb = p.N >> 5;  // blocks per output row
#pragma unroll
      for (int jn = 0; jn < 3; ++jn) {
        const int cb = (n0 >> 5) + wn * 3 + jn;  // wave-uniform
        if (cb >= nb) continue;
        float4 b4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b4[q] = bias4(cb, q, 1.0f);
#pragma unroll
        for (int im = 0; im < NIM; ++im) {
          float o[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            o[4 * q + 0] = __builtin_fmaf(acc[jn][im][4 * q + 0], os, b4[q].x);
            o[4 * q + 1] = __builtin_fmaf(acc[jn][im][4 * q + 1], os, b4[q].y);
            o[4 * q + 2] = __builtin_fmaf(acc[jn][im][4 * q + 2], os, b4[q].z);
            o[4 * q + 3] = __builtin_fmaf(acc[jn][im][4 * q + 3], os, b4[q].w);
          }
          if constexpr (EPI == EPI_IMG_BIAS) {
            if (p.out_f32) {
              // un-fused LayerNorm path (d_model > 384): dense + bias + residual leaves as fp32 rows for launch_ln_f32_img
              if (p.resid) {
                float rv[16];
                u32x4 raw[4];
                load_group_block_raw(p.resid + ((size_t)((m0 + wm * 64 + im * 32) >> 5) * nb + cb) * 4096, raw, l31, half);
                unpack_block(raw, rv, p.resid_inv);
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] += rv[r];
              }
              float* dst = p.out_f32 + (size_t)(m0 + wm * 64 + im * 32 + l31) * p.N + cb * 32 + 4 * half;
#pragma unroll
              for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(dst + 8 * q) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
              continue;
            }
          }
          if constexpr (EPI == EPI_IMG_GELU) {
#ifdef FDMI_GELU_PAIRS  // A/B build: the pair form of rounds 2-3 (the scale is then applied here instead of inside)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              const gf2 g = gelu_erf2(gf2{o[r], o[r + 1]}) * p.out_scale;
              o[r] = g[0];
              o[r + 1] = g[1];
            }
#else
            const float hs = 0.5f * p.out_scale;  // the GELU leaves at the output image's scale
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
              const gf4 g = gelu_erf4_scaled(gf4{o[r], o[r + 1], o[r + 2], o[r + 3]}, hs);
              o[r] = g[0];
              o[r + 1] = g[1];
              o[r + 2] = g[2];
              o[r + 3] = g[3];
            }
#endif
          }
          store_group_block(p.out + ((size_t)((m0 + wm * 64 + im * 32) >> 5) * nb + cb) * 4096, o, EPI == EPI_IMG_GELU ? 1.0f : p.out_scale,
                            l31, half);
        }
      }
    } else if constexpr (kQK) {
      const int H = p.H;
      FD_WAIT_VM(0);  // the row info landed long ago; the wave has nothing else in flight
      const int2* rinfo = reinterpret_cast<const int2*>(smem + OFF_RI + wid * 512);
      // q and k are grouped images per (sequence, head): [position / 32][unit][position % 32][16 B].  The lane's byte offset
      // inside the image of head 0 is computed ONCE per tile and row; the head's share (h LTOT 128 B) rides in the scalar offset
      // of the buffer stores.  Rows that are no token (alignment / tail rows) get an offset beyond the buffer: the hardware drops
      // their stores -- no predication, no 64-bit address arithmetic per block (it was ~40 of a block's ~190 VALU instructions).
      unsigned roff[2];
#pragma unroll
      for (int im = 0; im < NIM; ++im) {
        const int2 ri = rinfo[im * 32 + l31];
        roff[im] = ri.x >= 0 ? (unsigned)ri.x * (unsigned)(H * p.LTOT * 128) + (unsigned)((ri.y >> 5) * 4096 + (ri.y & 31) * 16 + half * 1024)
                             : 0xFFFFF000u;
      }
      const __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc(p.qbuf, 0, p.qkv_bytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc(p.kbuf, 0, p.qkv_bytes, 0x00020000);
#pragma unroll
      for (int jn = 0; jn < 3; ++jn) {
        const int cb = (n0 >> 5) + wn * 3 + jn;  // wave-uniform: block of the [q | k] column space
        if (cb >= 2 * H) continue;
        const int isk = cb >= H ? 1 : 0, h = cb - isk * H;
        float4 b4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b4[q] = bias4(cb, q, isk ? p.k_scale : p.q_scale);
        const float oss = os * (isk ? p.k_scale : p.q_scale);  // (the bias in LDS already carries the image's scale)
        const int hoff = h * p.LTOT * 128;
#pragma unroll
        for (int im = 0; im < NIM; ++im) {
          float o[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            o[4 * q + 0] = __builtin_fmaf(acc[jn][im][4 * q + 0], oss, b4[q].x);
            o[4 * q + 1] = __builtin_fmaf(acc[jn][im][4 * q + 1], oss, b4[q].y);
            o[4 * q + 2] = __builtin_fmaf(acc[jn][im][4 * q + 2], oss, b4[q].z);
            o[4 * q + 3] = __builtin_fmaf(acc[jn][im][4 * q + 3], oss, b4[q].w);
          }
          u32x4 h0, h1, l0, l1;
          pack_block(o, 1.0f, h0, h1, l0, l1);
          if (FDMI_EPI_DBG == 1) {
            asm volatile("" ::"v"(h0), "v"(h1), "v"(l0), "v"(l1));
            continue;
          }
          // (plain stores: write-through ones made the q | k | v projection 13 % slower)
          if (isk) {
            __builtin_amdgcn_raw_buffer_store_b128(h0, rsk, (int)roff[im], hoff, 0);
            __builtin_amdgcn_raw_buffer_store_b128(h1, rsk, (int)roff[im], hoff + 512, 0);
            __builtin_amdgcn_raw_buffer_store_b128(l0, rsk, (int)roff[im], hoff + 2048, 0);
            __builtin_amdgcn_raw_buffer_store_b128(l1, rsk, (int)roff[im], hoff + 2560, 0);
          } else {
            __builtin_amdgcn_raw_buffer_store_b128(h0, rsq, (int)roff[im], hoff, 0);
            __builtin_amdgcn_raw_buffer_store_b128(h1, rsq, (int)roff[im], hoff + 512, 0);
            __builtin_amdgcn_raw_buffer_store_b128(l0, rsq, (int)roff[im], hoff + 2048, 0);
            __builtin_amdgcn_raw_buffer_store_b128(l1, rsq, (int)roff[im], hoff + 2560, 0);
          }
          store_guard(h0, h1);
          store_guard(l0, l1);
        }
      }
    } else if constexpr (kVT) {
      // normal MFMA form: lane = column (d = l31 of head cb), register r = 4q + e <-> token row 8q + 4 half + e of the
      // 32-row MFMA tile.  After the exchange the lower lane holds token octets 0, 1 and the upper lane 2, 3 of the
      // tile; sequences start at multiples of 8 rows, so an octet never straddles two sequences.
      // V^T block layout: [b][h][key block l/32][d][128 B = sixteen 8-byte units: hi keys 4u..4u+3 (u < 8) | lo],
      // unit u stored at position u ^ vt_swz(d) (img_common.h: conflict-free 8-byte LDS fetches in the attention kernel
      // after a linear LDS-DMA copy); an octet = one aligned 16-byte pair of units, halves swapped when the
      // swizzle is odd -- so every store is an aligned 16-byte piece of a 128-byte line, as in the q / k epilogue.
      const int H = p.H, nkb = p.LTOT >> 5;
      FD_WAIT_VM(0);  // the row info landed long ago; the wave has nothing else in flight
      const int2* rinfo = reinterpret_cast<const int2*>(smem + OFF_RI + wid * 512);
      const int sz = vt_swz(l31);
      const float osv = os * p.v_scale;  // (the bias in LDS already carries the image's scale)
      constexpr bool merged = EPI == EPI_IMG_QKV;  // v columns follow the 2 H blocks of q | k
      // byte offset of the lane's hi octet inside the V^T image of head 0, once per tile and token octet (im, u); the head's share
      // rides in the scalar offset; octets of rows that are no token get an offset beyond the buffer (dropped by the hardware)
      unsigned voff[2][2], voffl[2][2];  // (hi octet, lo octet = four 16-byte pairs further: the hi offset ^ 64, oc < 4)
#pragma unroll
      for (int im = 0; im < NIM; ++im)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int2 ri = rinfo[im * 32 + 16 * half + 8 * u];
          const int kb = ri.y >> 5, oc = (ri.y & 31) >> 3;
          voff[im][u] = ri.x >= 0 ? ((unsigned)ri.x * (unsigned)(H * nkb) + (unsigned)kb) * 4096u + (unsigned)(l31 * 128 + ((oc ^ (sz >> 1)) << 4))
                                  : 0xFFFFF000u;
          voffl[im][u] = voff[im][u] ^ 64u;
        }
      const __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc(p.vbuf, 0, p.qkv_bytes, 0x00020000);
#pragma unroll
      for (int jn = 0; jn < 3; ++jn) {
        const int cb = (n0 >> 5) + wn * 3 + jn - (merged ? 2 * H : 0);
        if (cb >= H) continue;
        const int cbg = cb + (merged ? 2 * H : 0);
        const float bz = cbg * 32 < 3 * BN ? par0[cbg * 32 + l31] : p.bias[cbg * 32 + l31] * p.v_scale;
        const int hoff = cb * nkb * 4096;
#pragma unroll
        for (int im = 0; im < NIM; ++im) {
          float o[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) o[r] = __builtin_fmaf(acc[jn][im][r], osv, bz);
          u32x4 hh[2], ll[2];
          pack_block(o, 1.0f, hh[0], hh[1], ll[0], ll[1]);
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            u32x4 vh = hh[u], vl = ll[u];
            if (sz & 1) {
              vh = u32x4{vh[2], vh[3], vh[0], vh[1]};
              vl = u32x4{vl[2], vl[3], vl[0], vl[1]};
            }
            if (FDMI_EPI_DBG == 1) {
              asm volatile("" ::"v"(vh), "v"(vl));
              continue;
            }
            __builtin_amdgcn_raw_buffer_store_b128(vh, rsv, (int)voff[im][u], hoff, 0);
            __builtin_amdgcn_raw_buffer_store_b128(vl, rsv, (int)voffl[im][u], hoff, 0);
            store_guard(vh, vl);
          }
        }
      }
    } else if constexpr (EPI == EPI_IMG_LN) {
      const int N = p.N, nb = N >> 5;
      const float* par = reinterpret_cast<const float*>(smem + OFF_PAR);
      float* red = reinterpret_cast<float*>(smem + OFF_RED);
      // (wave-uniform: kept in an SGPR -- as a VGPR it is hoisted out of the tile loop and spilled)
      const float inv_n = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 1.0f / (float)N)));
      float s[2] = {0.f, 0.f};
      // v = acc / (a_scale w_scale) + bias + residual.  The residual image is read by HALF-blocks (one hi and one lo 16-byte
      // unit per lane: the quad pairs {0,2} / {1,3} of a 32 x 32 MFMA tile), three half-blocks ahead: the first three were
      // requested before the tile's last k-tile (prefetch_resid), each one's registers are re-used for the request three
      // half-blocks later.  (The old loop fetched a whole block at its point of use: six exposed round trips per wave.)
      const unsigned char* rbase = p.resid + (size_t)((m0 + wm * 64) >> 5) * nb * 4096;
      const unsigned roff = (unsigned)(l31 * 16 + half * 1024);
      auto pass1_half = [&](auto JN, auto IM, auto HB, const u32x4& rh, const u32x4& rl) {
        constexpr int jn = decltype(JN)::value, im = decltype(IM)::value, hb = decltype(HB)::value;
        const int cb = wn * 3 + jn;
        if (cb >= nb) {  // columns beyond N (d_model < 384): contribute nothing
#pragma unroll
          for (int qi = 0; qi < 2; ++qi)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[jn][im][4 * (hb + 2 * qi) + e] = 0.f;
          return;
        }
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
            // + (hi + lo) / s_r as two fused multiply-adds on the fp16 halves (v_fma_mix_f32): 1 / s_r is a power of two
            float v0 = __builtin_fmaf(acc[jn][im][4 * q + 2 * dd], os, bb[2 * dd]);
            float v1 = __builtin_fmaf(acc[jn][im][4 * q + 2 * dd + 1], os, bb[2 * dd + 1]);
            v0 = fma_mix_lo(H[2 * qi + dd], ri, v0);
            v1 = fma_mix_hi(H[2 * qi + dd], ri, v1);
            v0 = fma_mix_lo(Lo[2 * qi + dd], ri, v0);
            v1 = fma_mix_hi(Lo[2 * qi + dd], ri, v1);
            acc[jn][im][4 * q + 2 * dd] = v0;
            acc[jn][im][4 * q + 2 * dd + 1] = v1;
            s[im] += v0;
            s[im] += v1;
          }
        }
      };
      auto request = [&](auto HBI, u32x4& rh, u32x4& rl) {  // half-block index 0..11 -> (im, jn, hb)
        constexpr int hbi = decltype(HBI)::value, im = hbi / 6, jn = (hbi % 6) / 2, hb = hbi % 2;
        int cb = wn * 3 + jn;
        cb = cb < nb ? cb : 0;
        // (buffer loads: wave-uniform block base in SGPRs + one 32-bit lane offset; with flat loads hipcc computes the nine
        // 64-bit lane addresses up front and spills them)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned char*>(rbase + ((size_t)im * nb + cb) * 4096), 0, 4096, 0x00020000);
        rh = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(roff + hb * 512), 0, 0));
        rl = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(roff + 2048 + hb * 512), 0, 0));
      };
#define FD_P1(i)                                                                                                         \
  do {                                                                                                                   \
    const u32x4 rh_ = rres[(i) % RD][0], rl_ = rres[(i) % RD][1];                                                        \
    if constexpr ((i) + RD < 6 * NIM) request(IC<((i) + RD < 12 ? (i) + RD : 0)>{}, rres[(i) % RD][0], rres[(i) % RD][1]); \
    pass1_half(IC<((i) % 6) / 2>{}, IC<(i) / 6>{}, IC<(i) % 2>{}, rh_, rl_);                                             \
  } while (0)
      FD_P1(0); FD_P1(1); FD_P1(2); FD_P1(3); FD_P1(4); FD_P1(5);
      if constexpr (NIM == 2) { FD_P1(6); FD_P1(7); FD_P1(8); FD_P1(9); FD_P1(10); FD_P1(11); }
#undef FD_P1
      // row sums: in-lane (48 columns) + the other half-wave + the four N-waves through LDS, fixed order
      auto block_sum = [&](float (&t)[2], float* part) {
#pragma unroll
        for (int im = 0; im < NIM; ++im) {
          {  // t += the other half-wave's t: both halves end up with lower + upper (no lane-index register as __shfl_xor needs)
            unsigned lo_side = __builtin_bit_cast(unsigned, t[im]), hi_side = lo_side;
            swap32(lo_side, hi_side);
            t[im] = __builtin_bit_cast(float, lo_side) + __builtin_bit_cast(float, hi_side);
          }
          if (half == 0) part[(wm * 64 + im * 32 + l31) * 4 + wn] = t[im];
        }
        barrier_keep_vm();
#pragma unroll
        for (int im = 0; im < NIM; ++im) {
          const float4 q4 = *reinterpret_cast<const float4*>(part + (wm * 64 + im * 32 + l31) * 4);
          t[im] = (q4.x + q4.y) + (q4.z + q4.w);
        }
      };
      block_sum(s, red);
      float t2[2] = {0.f, 0.f};
#pragma unroll
      for (int im = 0; im < NIM; ++im) {
        float mean = s[im] * inv_n;
        asm volatile("" : "+v"(mean));  // (a value, not a product: `acc - mean` must not become an fma in one kernel and not in the other)
#pragma unroll
        for (int jn = 0; jn < 3; ++jn) {
          if (wn * 3 + jn >= nb) continue;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float dl = acc[jn][im][r] - mean;
            acc[jn][im][r] = dl;
            t2[im] = __builtin_fmaf(dl, dl, t2[im]);  // (spelled out: the few-rows kernel, gemm_ln_rows.hip, must round exactly alike)
          }
        }
      }
      block_sum(t2, red + BM * 4);
#pragma unroll
      for (int im = 0; im < NIM; ++im) {
        const float rstd = (1.0f / sqrtf(__builtin_fmaf(t2[im], inv_n, p.eps))) * p.out_scale;  // at the output image's scale (beta in LDS too)
#pragma unroll
        for (int jn = 0; jn < 3; ++jn) {
          const int cb = wn * 3 + jn;
          if (cb >= nb) continue;
          float o[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 g4 = *reinterpret_cast<const float4*>(par + BN + cb * 32 + 8 * q + 4 * half);
            const float4 e4 = *reinterpret_cast<const float4*>(par + 2 * BN + cb * 32 + 8 * q + 4 * half);
            o[4 * q + 0] = __builtin_fmaf(acc[jn][im][4 * q + 0] * rstd, g4.x, e4.x);
            o[4 * q + 1] = __builtin_fmaf(acc[jn][im][4 * q + 1] * rstd, g4.y, e4.y);
            o[4 * q + 2] = __builtin_fmaf(acc[jn][im][4 * q + 2] * rstd, g4.z, e4.z);
            o[4 * q + 3] = __builtin_fmaf(acc[jn][im][4 * q + 3] * rstd, g4.w, e4.w);
          }
          store_group_block(p.out + ((size_t)((m0 + wm * 64 + im * 32) >> 5) * nb + cb) * 4096, o, 1.0f, l31, half);
        }
      }
    }
  };

  // ---- the stream of the compute waves.  Their only vector-memory work is the epilogue's loads and stores.
  int cw = 0, ca = 0;  // slots of the position being computed
  const bool rec = PROF && blockIdx.x == 0 && p.stamps != nullptr;
  unsigned long long* st = PROF ? p.stamps + ((size_t)(EPI == EPI_IMG_QKV ? (int)EPI_IMG_QK : EPI) * 8 + wid) * 64 * 6 : nullptr;
  int slot = 0;
#define FD_STAMP(i) do { if (PROF) { if (rec && slot < 64 && lane == 0) st[slot * 6 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
#define FD_SB() __builtin_amdgcn_sched_barrier(0)
  // Fragment registers: FOUR buffers (40 VGPRs) serve the eight fragment sets of a k-tile.  Every set is requested at the
  // START of the MFMA group before the one that consumes it, into the buffer whose last reader was the group before that:
  //     group            1: wh0 ah0   2: wh0 al0   3: wl0 ah0   4: wh1 ah1   5: wh1 al1   6: wl1 ah1
  //     requested during    al0 -> Yb    wl0 -> Xb    wh1 -> Xa    al1 -> Ya    wl1 -> Xb    wh0' -> Xa
  //                                                   ah1 -> Yb                              ah0' -> Ya
  // With one variable per set hipcc reused the registers of the set still being read and sank every request to one MFMA
  // (32 cycles) before its consumer; the explicit buffers also free ~10 VGPRs of the 168 a 10-wave workgroup may use.
  f16x8 Xa[3], Xb[3], Ya[2], Yb[2];
  // first fragments (k16 step 0, hi planes) of the position in slots (cw, ca).  Their per-lane offset is re-derived from the
  // lane id here: as a loop invariant it is the allocator's favourite spill victim (its only use follows a barrier), and a
  // scratch reload is ~500 cycles during which no wave of the workgroup has a matrix instruction to issue.
  auto first_fragments = [&]() {
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const int r00 = ((ln & 31) >> 3) * 1024 + ((((ln >> 5) ^ ((ln >> 3) & 1)) * 8 + (ln & 7)) << 4);
    lda(IC<2>{}, Ya, smem + abase + ca * A_STAGE, r00);
    ldw(Xa, smem + wbase + cw * W_STAGE, r00);
  };
  auto groups_1_to_5 = [&](auto SW, auto NI) {
    const unsigned char* wb = smem + wbase + cw * W_STAGE;
    const unsigned char* ab = smem + abase + ca * A_STAGE;
    // the four fragment offsets, re-derived from the lane id per k-tile (8 VALU instructions against ~2900 cycles): as loop
    // invariants one of them is spilled, and its reload at every tile boundary waits for the previous epilogue's stores
    int rd[2][2];
    {
      int ln;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
      const int l31_ = ln & 31, half_ = ln >> 5;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          rd[c][pl] = (l31_ >> 3) * 1024 + ((((2 * c + half_ + 4 * pl) ^ ((l31_ >> 3) & 1)) * 8 + (l31_ & 7)) << 4);
    }
    FD_SB();
    lda(NI, Yb, ab, rd[0][1]);         // al0
    FD_SB();
    mm6(SW, NI, Xa, Ya);               // 1: wh0 ah0
    FD_SB();
    ldw(Xb, wb, rd[0][1]);             // wl0
    FD_SB();
    mm6(SW, NI, Xa, Yb);               // 2: wh0 al0
    FD_SB();
    ldw(Xa, wb, rd[1][0]);             // wh1
    lda(NI, Yb, ab, rd[1][0]);         // ah1
    FD_SB();
    mm6(SW, NI, Xb, Ya);               // 3: wl0 ah0
    FD_SB();
    lda(NI, Ya, ab, rd[1][1]);         // al1
    FD_SB();
    mm6(SW, NI, Xa, Yb);               // 4: wh1 ah1
    FD_SB();
    ldw(Xb, wb, rd[1][1]);             // wl1
    FD_SB();
    mm6(SW, NI, Xa, Ya);               // 5: wh1 al1
    FD_SB();
    cw ^= 1;
    ca = ca == NAS - 1 ? 0 : ca + 1;
  };
  auto run_tile = [&](auto SW, auto NI, int ti) {
    for (int kt = 0; kt + 1 < nk; ++kt) {
      FD_STAMP(0);
      groups_1_to_5(SW, NI);
      FD_STAMP(1);
#if FDMI_G6FIRST
      FD_WAIT_LGKM0();  // (the same wait as inside the barrier, but visible to the compiler's counter model: no waits in group 6)
#endif
      FD_KBAR();  // every fragment of this position is in registers: its slots are free; the next position landed
      FD_STAMP(2);
#if FDMI_G6FIRST
      // Behind the barrier all eight waves want the LDS for their first fragments at once (~300 cycles during which no
      // wave issues a matrix instruction).  The two waves of a SIMD are (wm 0, wm 1): the wm 1 wave runs group 6 first --
      // it needs registers only -- and fetches afterwards, under the other wave's group 6.
      if (wm == 0) first_fragments();
      FD_SB();
      FD_STAMP(3);
      mm6(SW, NI, Xb, Yb);  // 6: wl1 ah1
      FD_SB();
      if (wm != 0) first_fragments();
      FD_SB();
#else
      first_fragments();
      FD_SB();  // reads first: they fly while group 6 runs
      FD_STAMP(3);
      mm6(SW, NI, Xb, Yb);  // 6: wl1 ah1
      FD_SB();
#endif
      FD_STAMP(4);
      ++slot;
    }
    // the tile's last k-tile: the epilogue sits between group 6 and the next tile's first fragments
    FD_STAMP(0);
    prefetch_rowinfo(SW, ti);
    prefetch_resid(ti);
    groups_1_to_5(SW, NI);
    FD_STAMP(1);
    const bool stream_end = ti + 1 == cnt;
    if (!stream_end) FD_KBAR();
    FD_STAMP(2);
    FD_STAMP(3);
    mm6(SW, NI, Xb, Yb);  // 6: wl1 ah1
    FD_SB();
    FD_STAMP(4);
    epilogue(SW, NI, ti);
    FD_STAMP(5);
    zero_acc();
    if (!stream_end) first_fragments();
    ++slot;
  };
  // a tail slice (always the LAST tile of a workgroup's stream) is computed by the wm 0 waves; the wm 1 waves execute the tile's
  // barriers and nothing else: nk - 1 k-loop barriers (the stream's last position has none) and the LayerNorm epilogue's two
  auto idle_tile = [&]() {
    for (int kt = 0; kt + 1 < nk; ++kt) FD_KBAR();
    if constexpr (EPI == EPI_IMG_LN) {
      barrier_keep_vm();
      barrier_keep_vm();
    }
  };
  // (one call site per instantiation of run_tile: a second one would turn the k-loop into an out-of-line function)
  auto do_tile = [&](auto SW, int ti) {
    const int rows = tile_rows(ti);
    if (rows != BM && wm != 0) idle_tile();
    else if (rows == BM / 4) run_tile(SW, IC<1>{}, ti);
    else run_tile(SW, IC<2>{}, ti);  // 128 rows, or the 64 rows of a half slice (wm 0)
  };
  barrier_keep_vm();  // position 0 landed (also publishes the parameter image)
  first_fragments();
  for (int ti = 0; ti < cnt; ++ti) {
    if constexpr (EPI == EPI_IMG_QKV) {  // the v tiles run the normal MFMA form (lane = feature), the q | k tiles the swapped one
      int m0, n0;
      tile_mn(ti, m0, n0);
      if (n0 >= 2 * p.H * 32) do_tile(IC<0>{}, ti);
      else do_tile(IC<1>{}, ti);
    } else {
      do_tile(IC<SWAP ? 1 : 0>{}, ti);
    }
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

template <int EPI, bool SWAP>
static void launch(const GemmImgArgs& p, int max_rows, hipStream_t s) {
  static bool attr_set[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    for (const void* f : {reinterpret_cast<const void*>(&gemm_img_kernel<EPI, SWAP, false, 0>),
                          reinterpret_cast<const void*>(&gemm_img_kernel<EPI, SWAP, false, 1>),
                          reinterpret_cast<const void*>(&gemm_img_kernel<EPI, SWAP, true, 1>)})
      (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr_set[dev] = true;
  }
  const int ntiles_max = ((max_rows + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  int grid = n_cu_of_current_device() / 8 * 8;
  if (grid > ntiles_max) grid = (ntiles_max + 7) / 8 * 8;
  if (grid < 8) grid = 8;
  static const int force_tail = [] { const char* e = getenv("FDMI_GEMM_TAIL"); return e ? atoi(e) : -1; }();  // A/B: 0 / 1 override the host's choice
  const bool tail = force_tail >= 0 ? force_tail != 0 : p.tail != 0;
  if (p.stamps) hipLaunchKernelGGL((gemm_img_kernel<EPI, SWAP, true, 1>), dim3(grid), dim3(NTHR), SMEM, s, p);
  else if (tail) hipLaunchKernelGGL((gemm_img_kernel<EPI, SWAP, false, 1>), dim3(grid), dim3(NTHR), SMEM, s, p);
  else hipLaunchKernelGGL((gemm_img_kernel<EPI, SWAP, false, 0>), dim3(grid), dim3(NTHR), SMEM, s, p);
}

}  // namespace gi

// workgroups launch_gemm_img starts for a problem of up to max_rows rows and N columns (the host decides with it whether a
// launch's tiles fill whole rounds: GemmImgArgs::tail)
int gemm_img_grid(int max_rows, int N) {
  const int ntiles_max = ((max_rows + gi::BM - 1) / gi::BM) * ((N + gi::BN - 1) / gi::BN);
  int grid = gi::n_cu_of_current_device() / 8 * 8;
  if (grid > ntiles_max) grid = (ntiles_max + 7) / 8 * 8;
  return grid < 8 ? 8 : grid;
}

void launch_gemm_img(int epilogue, const GemmImgArgs& p, int max_rows, hipStream_t s) {
  // Few rows: the weight-stationary kernel (gemm_ws.hip) has no 128-row tile to fill -- a launch costs >= 19 us here (one whole
  // tile's k-loop per CU) against 8 us there; it wins up to ~12 k rows and ties or loses above (scripts/ws_sweep.py,
  // profiles/r04_ws_gemm.log).  FDMI_GEMM_WS = 0 never, 1 wherever it applies, unset: by row count.  Same bits either way.
  static const int ws_mode = [] { const char* e = getenv("FDMI_GEMM_WS"); return e ? atoi(e) : -1; }();
  if ((ws_mode > 0 || (ws_mode < 0 && max_rows <= 12288)) && !p.stamps && gemm_ws_supported(epilogue, p)) return launch_gemm_ws(epilogue, p, s);
  // ... and the LayerNorm GEMMs of few rows on gemm_ln_rows.hip (a workgroup per 32-row group, weights straight from L2 to registers)
  static const int ln_rows_max = [] { const char* e = getenv("FDMI_LN_ROWS_MAX"); return e ? atoi(e) : 8192; }();
  if (epilogue == EPI_IMG_LN && ws_mode != 0 && max_rows <= (ws_mode > 0 ? (1 << 30) : ln_rows_max) && !p.stamps && gemm_ln_rows_supported(p))
    return launch_gemm_ln_rows(p, max_rows, s);
  switch (epilogue) {
    case EPI_IMG_GELU: gi::launch<EPI_IMG_GELU, true>(p, max_rows, s); break;
    case EPI_IMG_LN: gi::launch<EPI_IMG_LN, true>(p, max_rows, s); break;
    case EPI_IMG_QK: gi::launch<EPI_IMG_QK, true>(p, max_rows, s); break;
    case EPI_IMG_BIAS: gi::launch<EPI_IMG_BIAS, true>(p, max_rows, s); break;
    case EPI_IMG_QKV: gi::launch<EPI_IMG_QKV, true>(p, max_rows, s); break;
    default: gi::launch<EPI_IMG_VT, false>(p, max_rows, s); break;
  }
}

}  // namespace fdmi
