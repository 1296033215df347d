// Weight-stationary token GEMM on row images (round 4): C[M,N] = A[M,384] W[N,384]^T + bias, epilogues GELU / plain bias / q | k | v
// (BertIntermediate.dense, AnglesPredictor.dense1: foldingdiff/modelling.py:195-196, :203-205; BertSelfAttention.query / key / value of
// HF 4.11.3, called at modelling.py:473-480), the same split
// arithmetic and the same image layouts as gemm_img.hip.  The product path for launches of FEW ROWS (<= 12,288: gemm_img.hip,
// launch_gemm_img), where it is up to 2.4 x faster than the tile kernel; an experiment above that (FDMI_GEMM_WS=1), where it ties.
//
// The tile kernel (gemm_img.hip) streams a 576 KiB weight tile through LDS for every 128 rows: a launch costs one whole tile's k-loop
// per CU (>= 19 us) however few rows there are, and at full size its k-loop is bound by the bytes the workgroup moves.
// Here the WEIGHTS stay in registers for the whole launch and only the activations stream:
//   * a workgroup (8 waves, one per CU) owns a 256-column slice of W; wave w holds the 32 rows (output columns) 32 w .. 32 w + 31 of
//     the slice as MFMA A-operand fragments: 12 k-tiles x 2 k16 steps x (hi, lo) x 4 registers = 192 VGPRs, loaded once;
//   * the A image is consumed one 32-row GROUP at a time: a group is 48 KiB CONTIGUOUS in HBM ([row / 32][K / 32][unit][row % 32]
//     [16 B]) and already in fragment order, so it is a linear LDS-DMA copy into a ring of three 48 KiB slots, and a lane's B operand
//     of k16 step (kt, c) is the 16 bytes at kt * 4096 + unit * 512 + l31 * 16: conflict-free ds_read_b128, no swizzle;
//   * per group and wave 72 MFMAs into ONE 32 x 32 accumulator (D^T = W A^T: a lane owns a token row, the epilogues of gemm_img.hip
//     apply unchanged), then the epilogue and four 16-byte stores per lane; one barrier per group.
// The accumulation order per output is the tile kernel's (k ascending, wh ah | wh al | wl ah per k16 step), so results are
// BIT-IDENTICAL to gemm_img.hip (tests/test_gpu_parity.py: test_gemm_ws_bit_identical_to_tile_kernel).
//
// What bounds it (s_memtime stamps of workgroup 0, FDMI_WS_STAMPS=1; ablations FDMI_WS_DBG; profiles/r04_ws_gemm.log).  Per group and
// CU the model said 48 KiB loaded + 32 KiB stored (4.9 k cycles by the law of DESIGN.md 4.1) beside 4608 matrix cycles per SIMD.
// Measured: a group takes 6.8 k cycles, and the limit is each WAVE'S OWN serial chain -- wait + barrier 0.4 k, its six copy pieces
// 0.4-0.9 k, its 72 MFMAs 2.6 k (36 cycles each, alone on the pipe: the SIMD's other wave is in its epilogue), epilogue 2.7-3.3 k of
// which 1.7 k is the wave stalled in its four stores (four waves store 16 KiB at once: 9.4 B/clk) -- not the SIMD's matrix pipe
// (76 % busy).  Without MFMAs 77 us, without epilogue 89 us, copies alone 23 us, everything 118-129 us for the BertIntermediate
// shape (tile kernel 118-123 us).  Shortening the chain means issuing a group's stores (or its whole epilogue) between the NEXT
// group's MFMAs, i.e. 16-36 more live registers, and the kernel sits at 250-256 of the 256 a two-wave SIMD allows (W alone is 192):
// the design is register-starved, and K = 384 in hi + lo fragments is what starves it.  Tried, no gain: operand prefetch two k16
// steps ahead (the MFMA phase is hand-placed below: hipcc clusters four reads and waits on the spot, and
// __builtin_amdgcn_sched_group_barrier spills the weights), the late-epilogue arrangement (waves 4-7 running their epilogue half a
// group behind, so that it issues beside the SIMD's other wave's MFMAs: the stamps showed exactly that complementarity and the same
// period; removed), the epilogue in halves (kept, it needs fewer registers), plain / non-temporal stores.
// Limits: K = 384 (the registers hold K), column-local epilogues (a LayerNorm row does not fit a 256-column slice), image output,
// q | k | v of head size 32.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fdmi_kernels.h"
#include "img_common.h"

#ifndef FDMI_WS_DBG
#define FDMI_WS_DBG 0  // ablations (timing only, results wrong): 1 no epilogue, 2 no MFMA phase, 4 no copies, 8 no stores
#endif

namespace fdmi {
namespace ws {

constexpr int NKT = 12;                      // K = 384
constexpr int A_TILE = NKT * 4096;           // one 32-row group of the A image
constexpr int NSLOT = 3;
constexpr int OFF_PAR = NSLOT * A_TILE;      // 147,456: bias of the slice (256 floats)
constexpr int OFF_RI = OFF_PAR + 256 * 4;    // 148,480: (sequence, position) of a group's 32 rows, ring of three (q | k | v only)
constexpr int SMEM = OFF_RI + NSLOT * 256;   // 149,248 B
constexpr int SLICE = 256;                   // columns per workgroup = 8 waves x 32

// ---- the MFMA phase of one group, written out: four k16 steps per block, operands requested two steps ahead (left to the compiler
// the 48 operand reads are clustered four at a time and waited for on the spot).  Three operand buffers rotate: on entry X holds (in flight) step s0, Y step s0 + 1, Z is
// free; on exit Y holds s0 + 4, Z s0 + 5, X is free -- the caller rotates the names.  lgkmcnt is counted: LDS returns in order, two
// reads per step, so "at most 4 outstanding" means the step about to be multiplied has landed.
template <int V> struct IC { static constexpr int value = V; };
constexpr int step_off(int s) { return (s >> 1) * 4096 + (s & 1) * 1024; }
// D^T = W A^T (weights first: a lane owns a token row; the "_N" products) -- or D = A W^T (activations first: a lane owns an output
// column, the form the V^T epilogue wants; "_T").  The same three products in the same order either way.
#define FD_WS_MM_N(W_H, W_L, B_H, B_L, C0)                               \
  "v_mfma_f32_32x32x16_f16 %[acc], %[" W_H "], %[" B_H "], " C0 "\n\t"     \
  "v_mfma_f32_32x32x16_f16 %[acc], %[" W_H "], %[" B_L "], %[acc]\n\t"    \
  "v_mfma_f32_32x32x16_f16 %[acc], %[" W_L "], %[" B_H "], %[acc]\n\t"
#define FD_WS_MM_T(W_H, W_L, B_H, B_L, C0)                               \
  "v_mfma_f32_32x32x16_f16 %[acc], %[" B_H "], %[" W_H "], " C0 "\n\t"     \
  "v_mfma_f32_32x32x16_f16 %[acc], %[" B_L "], %[" W_H "], %[acc]\n\t"    \
  "v_mfma_f32_32x32x16_f16 %[acc], %[" B_H "], %[" W_L "], %[acc]\n\t"
#define FD_WS_RD(B_H, B_L, O)                                            \
  "ds_read_b128 %[" B_H "], %[a] offset:%[" O "]\n\t"                     \
  "ds_read_b128 %[" B_L "], %[a] offset:%[" O "l]\n\t"
#define FD_WS_BODY(MM, C0)                                                                        \
  FD_WS_RD("zh", "zl", "o2") "s_waitcnt lgkmcnt(4)\n\t" MM("wh0", "wl0", "xh", "xl", C0)          \
  FD_WS_RD("xh", "xl", "o3") "s_waitcnt lgkmcnt(4)\n\t" MM("wh1", "wl1", "yh", "yl", "%[acc]")    \
  FD_WS_RD("yh", "yl", "o4") "s_waitcnt lgkmcnt(4)\n\t" MM("wh2", "wl2", "zh", "zl", "%[acc]")    \
  FD_WS_RD("zh", "zl", "o5") "s_waitcnt lgkmcnt(4)\n\t" MM("wh3", "wl3", "xh", "xl", "%[acc]")
// the last four steps: nothing left to request after step S0 + 3, and nothing may be in flight at the end.  (The compiler cannot see
// the MFMA inside the block: the wait states before a VALU may read its result -- 11 for an 8-pass MFMA, 19 for a 16-pass one -- are
// spelled out.)
#define FD_WS_BODY_LAST(MM)                                                                       \
  FD_WS_RD("zh", "zl", "o2") "s_waitcnt lgkmcnt(4)\n\t" MM("wh0", "wl0", "xh", "xl", "%[acc]")    \
  FD_WS_RD("xh", "xl", "o3") "s_waitcnt lgkmcnt(4)\n\t" MM("wh1", "wl1", "yh", "yl", "%[acc]")    \
  "s_waitcnt lgkmcnt(2)\n\t" MM("wh2", "wl2", "zh", "zl", "%[acc]")                               \
  "s_waitcnt lgkmcnt(0)\n\t" MM("wh3", "wl3", "xh", "xl", "%[acc]") "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
#define FD_WS_BUFS [xh] "+v"(xh), [xl] "+v"(xl), [yh] "+v"(yh), [yl] "+v"(yl), [zh] "+v"(zh), [zl] "+v"(zl)
#define FD_WS_INS                                                                                                               \
  [a] "v"(addr), [wh0] "v"(wh0), [wl0] "v"(wl0), [wh1] "v"(wh1), [wl1] "v"(wl1), [wh2] "v"(wh2), [wl2] "v"(wl2), [wh3] "v"(wh3),  \
      [wl3] "v"(wl3), [o2] "n"(step_off(S0 + 2)), [o2l] "n"(step_off(S0 + 2) + 2048), [o3] "n"(step_off(S0 + 3)),               \
      [o3l] "n"(step_off(S0 + 3) + 2048)
#define FD_WS_INS_MID                                                                                                           \
  FD_WS_INS, [o4] "n"(step_off(S0 + 4)), [o4l] "n"(step_off(S0 + 4) + 2048), [o5] "n"(step_off(S0 + 5)), [o5l] "n"(step_off(S0 + 5) + 2048)
template <int S0, bool FIRST, bool LAST, bool VT>
__device__ __forceinline__ void mfma_block(f32x16& acc, unsigned addr, f16x8& xh, f16x8& xl, f16x8& yh, f16x8& yl, f16x8& zh, f16x8& zl,
                                           const f16x8& wh0, const f16x8& wl0, const f16x8& wh1, const f16x8& wl1, const f16x8& wh2,
                                           const f16x8& wl2, const f16x8& wh3, const f16x8& wl3) {
  if constexpr (LAST) {
    if constexpr (VT) asm volatile(FD_WS_BODY_LAST(FD_WS_MM_T) : [acc] "+v"(acc), FD_WS_BUFS : FD_WS_INS);
    else asm volatile(FD_WS_BODY_LAST(FD_WS_MM_N) : [acc] "+v"(acc), FD_WS_BUFS : FD_WS_INS);
  } else if constexpr (FIRST) {
    if constexpr (VT) asm volatile(FD_WS_BODY(FD_WS_MM_T, "0") : [acc] "=&v"(acc), FD_WS_BUFS : FD_WS_INS_MID);
    else asm volatile(FD_WS_BODY(FD_WS_MM_N, "0") : [acc] "=&v"(acc), FD_WS_BUFS : FD_WS_INS_MID);
  } else {
    if constexpr (VT) asm volatile(FD_WS_BODY(FD_WS_MM_T, "%[acc]") : [acc] "+v"(acc), FD_WS_BUFS : FD_WS_INS_MID);
    else asm volatile(FD_WS_BODY(FD_WS_MM_N, "%[acc]") : [acc] "+v"(acc), FD_WS_BUFS : FD_WS_INS_MID);
  }
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_ws_kernel(GemmImgArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int nb_out = p.N >> 5;                         // 32-column blocks of the output
  const int nslice = (p.N + SLICE - 1) / SLICE;
  const int Mp = p.dims[1], NT = Mp >> 5;              // 32-row groups
  // XCD-aware deal: the workgroups of one XCD (blockIdx % 8) form S streams of `nslice` neighbours; the neighbours of a stream hold
  // the slices of W and walk the SAME groups of A at the same time (one HBM read, the rest L2 hits)
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int S = per / nslice;
  if (S == 0 || jx >= S * nslice) return;              // (workgroups that do not fill a stream stay idle: 2 of 32 for 3 or 5 slices)
  const int slice = jx % nslice, j = jx / nslice;
  const int tlo = (int)((long long)NT * xcd / 8), thi = (int)((long long)NT * (xcd + 1) / 8);
  const int cnt = tlo + j < thi ? (thi - tlo - j + S - 1) / S : 0;
  if (cnt == 0) return;
  const int cb = slice * 8 + wid;                      // this wave's 32-column block
  const bool has_cols = cb < nb_out;

  // q | k | v projection (EPI_IMG_QKV): column blocks [0, H) are q, [H, 2H) k (a lane owns a token row, as for GELU / bias),
  // [2H, 3H) v -- computed in the OTHER operand order, a lane owns an output column: what the V^T image wants (gemm_img.hip)
  constexpr bool kQKV = EPI == EPI_IMG_QKV;
  const int H = p.H;
  const bool vt = kQKV && cb >= 2 * H;                 // (wave-uniform)
  {  // bias of the slice -> LDS, q | k | v: times the image's scale (published by the first barrier)
    float* par = reinterpret_cast<float*>(smem + OFF_PAR);
    for (int i = tid; i < SLICE; i += 512) {
      const int c = slice * SLICE + i;
      float b = c < p.N ? p.bias[c] : 0.f;
      if (kQKV) b *= c < 32 * H ? p.q_scale : (c < 64 * H ? p.k_scale : p.v_scale);
      par[i] = b;
    }
  }

  // ---- the wave's weights: row R of W (an output column) as A-operand fragments.  Weight image (api.hip: pack_weight_tiles):
  // [384-row tile][k-tile][48 KiB stage], a stage = pieces of 8 rows, unit u of row r at ((u ^ (piece & 1)) * 8 + r % 8) * 16
  f16x8 Wh[NKT][2], Wl[NKT][2];
  {
    const int R = (has_cols ? cb : 0) * 32 + l31, piece = (R % 384) >> 3;
    const unsigned char* wrow = p.W + (size_t)(R / 384) * NKT * 49152 + piece * 1024 + (R & 7) * 16;
    const int sw = piece & 1;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        Wh[kt][c] = *reinterpret_cast<const f16x8*>(wrow + (size_t)kt * 49152 + (((2 * c + half) ^ sw) << 7));
        Wl[kt][c] = *reinterpret_cast<const f16x8*>(wrow + (size_t)kt * 49152 + (((4 + 2 * c + half) ^ sw) << 7));
      }
  }

  // ---- the A ring: group g is 48 contiguous KiB; wave w copies pieces w, w + 8, ... (6 of the 48)
  auto issue = [&](int i) {
    if (FDMI_WS_DBG & 4) return;
    const int ii = i < cnt ? i : cnt - 1;              // past the end: repeat the last group (lands in a free slot, never read)
    const int g = tlo + j + ii * S;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(p.A) + (size_t)g * A_TILE, 0, A_TILE, 0x00020000);
    lds_ptr_t dst = (lds_ptr_t)(smem) + (i % NSLOT) * A_TILE;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int piece = wid + 8 * k;
      dma16(rs, dst + piece * 1024, lane * 16, piece * 1024);
    }
    if (EPI == EPI_IMG_QKV && wid == 0) {  // the group's row info rides along: 32 x (sequence, position) = 256 B, 4 bytes per lane
      const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<int2*>(p.rowinfo) + (size_t)g * 32, 0, 256, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (lds_ptr_t)(smem) + OFF_RI + (i % NSLOT) * 256, 4, lane * 4, 0, 0, 0);
    }
  };
  issue(0);
  issue(1);

  const float os = p.acc_scale;
  const float* par = reinterpret_cast<const float*>(smem + OFF_PAR) + wid * 32;
  const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)(smem);
  const int aoff = l31 * 16 + half * 512;              // unit 2c + half of k-tile kt: kt * 4096 + (2c + half) * 512 + l31 * 16
  // ---- epilogue.  Lane (l31, half) owns token row 32 g + l31 and columns 32 cb + 8 q + 4 half + e (register 4 q + e: the quad layout
  // of img_common.h) -- V^T waves: column 32 cb + l31 and token rows 32 g + 8 q + 4 half + e.  Two halves: quads {0, 2} make the
  // lane's first hi / lo unit, quads {1, 3} its second (quad_oct_exchange); the first half's two stores are on their way while the
  // second half is computed (and the halves need fewer live registers than sixteen outputs at once).
  // (q | k | v: the (sequence, position) of the group's rows landed in LDS with the group.)
  // MODE 0: row form (bias per column quad; units at o0, + 512, lo 2048 behind); MODE 1: V^T form (one bias per lane; token octet u at
  // o0 / o1, lo at ^ 64; the 8-byte halves of a pair change places where the swizzle is odd)
  auto epi_core = [&](auto MODE, const f32x16& acc, float oss, __amdgpu_buffer_rsrc_t rs, int soff, unsigned o0, unsigned o1) {
    constexpr int mode = decltype(MODE)::value;
    const float hs = 0.5f * p.out_scale, ps = EPI == EPI_IMG_BIAS ? p.out_scale : 1.0f;
    const float bz = mode == 1 ? par[l31] : 0.f;
    const bool flip = mode == 1 && (vt_swz(l31) & 1);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      unsigned Hh[4], Ll[4];
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const int q = hf + 2 * qq;
        float4 b4;
        if constexpr (mode == 1) b4 = make_float4(bz, bz, bz, bz);
        else b4 = *reinterpret_cast<const float4*>(par + 8 * q + 4 * half);
        gf4 v = {__builtin_fmaf(acc[4 * q + 0], oss, b4.x), __builtin_fmaf(acc[4 * q + 1], oss, b4.y),
                 __builtin_fmaf(acc[4 * q + 2], oss, b4.z), __builtin_fmaf(acc[4 * q + 3], oss, b4.w)};
        if constexpr (EPI == EPI_IMG_GELU) v = gelu_erf4_scaled(v, hs);  // the GELU leaves at the output image's scale
        split_pair(v[0] * ps, v[1] * ps, Hh[2 * qq], Ll[2 * qq]);
        split_pair(v[2] * ps, v[3] * ps, Hh[2 * qq + 1], Ll[2 * qq + 1]);
      }
      swap32(Hh[0], Hh[2]);
      swap32(Hh[1], Hh[3]);
      swap32(Ll[0], Ll[2]);
      swap32(Ll[1], Ll[3]);
      u32x4 hv = {Hh[0], Hh[1], Hh[2], Hh[3]}, lv = {Ll[0], Ll[1], Ll[2], Ll[3]};
      if (mode == 1 && flip) {
        hv = u32x4{hv[2], hv[3], hv[0], hv[1]};
        lv = u32x4{lv[2], lv[3], lv[0], lv[1]};
      }
      if (FDMI_WS_DBG & 8) {
        asm volatile("" ::"v"(hv), "v"(lv));
        continue;
      }
      const unsigned oh = mode == 1 ? (hf ? o1 : o0) : o0 + hf * 512, ol = mode == 1 ? oh ^ 64u : oh + 2048;
      // (GELU / bias: write-through stores, as store_group_block; q | k | v: plain ones -- write-through made that projection slower)
      __builtin_amdgcn_raw_buffer_store_b128(hv, rs, (int)oh, soff, kQKV ? 0 : FD_STORE_AUX);
      __builtin_amdgcn_raw_buffer_store_b128(lv, rs, (int)ol, soff, kQKV ? 0 : FD_STORE_AUX);
      store_guard(hv, lv);
    }
  };
  auto epilogue = [&](auto VTK, const f32x16& acc, int i) {
    constexpr bool kVT = decltype(VTK)::value != 0;
    if (FDMI_WS_DBG & 1) {
      asm volatile("" ::"v"(acc));
      return;
    }
    if constexpr (!kQKV) {
      const int g = tlo + j + i * S;
      epi_core(IC<0>{}, acc, os, __builtin_amdgcn_make_buffer_rsrc(p.out + ((size_t)g * nb_out + cb) * 4096, 0, 4096, 0x00020000), 0,
               (unsigned)(l31 * 16 + half * 1024), 0u);
    } else {
      const int2* rinfo = reinterpret_cast<const int2*>(smem + OFF_RI + (i % NSLOT) * 256);
      if constexpr (!kVT) {
        // q and k: grouped images per (sequence, head), [position / 32][unit][position % 32][16 B]; the head's share rides in the
        // scalar offset; rows that are no token get an offset beyond the buffer -- the hardware drops their stores (gemm_img.hip)
        const int isk = cb >= H ? 1 : 0, h = cb - isk * H;
        const int2 ri = rinfo[l31];
        const unsigned o0 = ri.x >= 0 ? (unsigned)ri.x * (unsigned)(H * p.LTOT * 128) + (unsigned)((ri.y >> 5) * 4096 + (ri.y & 31) * 16 + half * 1024)
                                      : 0xFFFFF000u;
        epi_core(IC<0>{}, acc, os * (isk ? p.k_scale : p.q_scale),
                 __builtin_amdgcn_make_buffer_rsrc(isk ? p.kbuf : p.qbuf, 0, p.qkv_bytes, 0x00020000), h * p.LTOT * 128, o0, 0u);
      } else {
        // V^T: [b][h][key block l / 32][d][128 B = sixteen 8-byte units: hi keys 4u..4u+3 (u < 8) | lo], unit u at u ^ vt_swz(d); the
        // lane's two token octets (rows 16 half + 8 u) are aligned 16-byte pairs of units, lo four pairs behind hi (gemm_img.hip)
        const int nkb = p.LTOT >> 5, sz = vt_swz(l31);
        unsigned o[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int2 ri = rinfo[16 * half + 8 * u];
          const int kb = ri.y >> 5, oc = (ri.y & 31) >> 3;
          o[u] = ri.x >= 0 ? ((unsigned)ri.x * (unsigned)(H * nkb) + (unsigned)kb) * 4096u + (unsigned)(l31 * 128 + ((oc ^ (sz >> 1)) << 4))
                           : 0xFFFFF000u;
        }
        epi_core(IC<1>{}, acc, os * p.v_scale, __builtin_amdgcn_make_buffer_rsrc(p.vbuf, 0, p.qkv_bytes, 0x00020000),
                 (cb - 2 * H) * nkb * 4096, o[0], o[1]);
      }
    }
  };
  // debug instrumentation (FDMI_WS_STAMPS=1): workgroup 0 records s_memtime at five points of its first 16 groups, per wave
  const bool rec = p.stamps != nullptr && blockIdx.x == 0 && lane == 0;
#define FD_WS_STAMP(k) do { if (rec && i < 16) p.stamps[(wid * 16 + i) * 6 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
  // (the whole loop exists once per operand order: with a wave-level branch around the MFMA phase only, hipcc spilled the weights)
  auto run = [&](auto VTK) {
  constexpr bool kVT = decltype(VTK)::value != 0;
  for (int i = 0; i < cnt; ++i) {
    // group i landed.  Vector-memory operations of this wave in program order (P = its 6 copy pieces -- 7 for wave 0 of the q | k | v
    // projection, which also copies the row info; S = 4 stores, only for waves that own columns):  P0 P1 | P2 S0 | P3 S1 | ...
    // At the top of iteration i >= 1 the operations younger than P(i) include P(i+1) and S(i-1); at i = 0 only P1.
    FD_WS_STAMP(0);
    if (FDMI_WS_DBG & 13) FD_WAIT_VM(0);
    else if (kQKV && wid == 0) {
      if (i == 0 || !has_cols) FD_WAIT_VM(7);
      else FD_WAIT_VM(11);
    } else if (i == 0 || !has_cols) FD_WAIT_VM(6);
    else FD_WAIT_VM(10);
    FD_WS_STAMP(1);
    barrier_keep_vm();   // every wave's pieces of group i are in LDS, and every wave is done reading group i - 1: its slot is free
    FD_WS_STAMP(2);
    issue(i + 2);
    FD_WS_STAMP(3);
    FD_WS_STAMP(4);
    const unsigned ab = lds_base + (unsigned)((i % NSLOT) * A_TILE + aoff);
    f32x16 acc;
    f16x8 xh, xl, yh, yl, zh, zl;
    if (FDMI_WS_DBG & 2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = (float)(i + r);
    } else {
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:2048\n\tds_read_b128 %2, %4 offset:1024\n\t"
                   "ds_read_b128 %3, %4 offset:3072"
                   : "=&v"(xh), "=&v"(xl), "=&v"(yh), "=&v"(yl) : "v"(ab));
      asm volatile("" : "=v"(zh), "=v"(zl));
#define FD_WS_W(b) Wh[2 * (b)][0], Wl[2 * (b)][0], Wh[2 * (b)][1], Wl[2 * (b)][1], Wh[2 * (b) + 1][0], Wl[2 * (b) + 1][0], Wh[2 * (b) + 1][1], Wl[2 * (b) + 1][1]
#define FD_WS_PHASE(VT)                                                              \
  mfma_block<0, true, false, VT>(acc, ab, xh, xl, yh, yl, zh, zl, FD_WS_W(0));       \
  mfma_block<4, false, false, VT>(acc, ab, yh, yl, zh, zl, xh, xl, FD_WS_W(1));      \
  mfma_block<8, false, false, VT>(acc, ab, zh, zl, xh, xl, yh, yl, FD_WS_W(2));      \
  mfma_block<12, false, false, VT>(acc, ab, xh, xl, yh, yl, zh, zl, FD_WS_W(3));     \
  mfma_block<16, false, false, VT>(acc, ab, yh, yl, zh, zl, xh, xl, FD_WS_W(4));     \
  mfma_block<20, false, true, VT>(acc, ab, zh, zl, xh, xl, yh, yl, FD_WS_W(5));
      FD_WS_PHASE(kVT)
#undef FD_WS_PHASE
#undef FD_WS_W
    }
    FD_WS_STAMP(5);
    if (has_cols) epilogue(VTK, acc, i);
  }
  };
  if (kQKV && vt) run(IC<1>{});
  else run(IC<0>{});
#undef FD_WS_STAMP
  FD_WAIT_VM(0);  // nothing may land in LDS after the workgroup has exited
}

static int current_device() {
  int dev = 0;
  return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) ? dev : 0;
}
static int n_cu() {  // per device (a process may hold models on several GPUs: fd_create takes any device_id)
  static int cached[64] = {0};
  const int dev = current_device();
  if (!cached[dev]) {
    hipDeviceProp_t prop;
    cached[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return cached[dev];
}

template <int EPI>
static void launch(const GemmImgArgs& p, hipStream_t s) {
  static bool attr_set[64] = {false};  // (the attribute is per device)
  const int dev = current_device();
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ws_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr_set[dev] = true;
  }
  static const bool want_stamps = [] { const char* e = getenv("FDMI_WS_STAMPS"); return e && atoi(e) != 0; }();
  static int dumped = 0;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (want_stamps) (void)hipStreamIsCapturing(s, &cap);  // (its allocations and synchronisation would invalidate a capture in progress)
  if (want_stamps && cap == hipStreamCaptureStatusNone && dumped < 2 && p.N >= 512) {  // debug: the 4th launch of a wide shape is stamped and printed (stderr)
    static int calls = 0;
    if (++calls == 4) {
      unsigned long long* d = nullptr;
      const size_t n = 8 * 16 * 6;
      if (hipMalloc(&d, n * 8) == hipSuccess && hipMemset(d, 0, n * 8) == hipSuccess) {
        GemmImgArgs q = p;
        q.stamps = d;
        hipLaunchKernelGGL((gemm_ws_kernel<EPI>), dim3(n_cu() / 8 * 8), dim3(512), SMEM, s, q);
        std::vector<unsigned long long> h(n);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
        (void)hipFree(d);
        fprintf(stderr, "ws stamps N=%d (cycles: wait | barrier | row info + copy issue | - | mfma | epilogue to the next top)\n", p.N);
        for (int w = 0; w < 8; ++w)
          for (int i = 2; i < 12; ++i) {
            const unsigned long long* t = &h[(w * 16 + i) * 6];
            const unsigned long long* tn = &h[(w * 16 + i + 1) * 6];
            fprintf(stderr, "  wave %d group %2d: top@%8llu  %6lld %6lld %6lld %6lld %6lld %6lld\n", w, i, t[0] - h[(0 * 16 + 2) * 6],
                    (long long)(t[1] - t[0]), (long long)(t[2] - t[1]), (long long)(t[3] - t[2]), (long long)(t[4] - t[3]),
                    (long long)(t[5] - t[4]), (long long)(tn[0] - t[5]));
          }
        ++dumped;
        calls = 0;
        return;
      }
    }
  }
  hipLaunchKernelGGL((gemm_ws_kernel<EPI>), dim3(n_cu() / 8 * 8), dim3(512), SMEM, s, p);
}

}  // namespace ws

// true if this shape runs on the weight-stationary kernel (K = 384, N a multiple of 32, column-local epilogue, image output)
bool gemm_ws_supported(int epilogue, const GemmImgArgs& p) {
  if (p.K != 32 * ws::NKT || p.N % 32 != 0) return false;
  if ((p.N + ws::SLICE - 1) / ws::SLICE > ws::n_cu() / 8) return false;  // an XCD's workgroups must hold at least one stream of slices
  if (epilogue == EPI_IMG_QKV) return p.N == 96 * p.H && p.LTOT % 32 == 0;  // q | k | v of head size 32 in one launch
  return (epilogue == EPI_IMG_GELU || epilogue == EPI_IMG_BIAS) && p.out_f32 == nullptr && p.resid == nullptr;
}

void launch_gemm_ws(int epilogue, const GemmImgArgs& p, hipStream_t s) {
  if (epilogue == EPI_IMG_GELU) ws::launch<EPI_IMG_GELU>(p, s);
  else if (epilogue == EPI_IMG_QKV) ws::launch<EPI_IMG_QKV>(p, s);
  else ws::launch<EPI_IMG_BIAS>(p, s);
}

}  // namespace fdmi
