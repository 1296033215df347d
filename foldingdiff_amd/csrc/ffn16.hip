// The tail of a BertLayer for 128 token rows in ONE kernel (HF 4.11.3 BertLayer.forward behind the attention: BertSelfOutput,
// BertIntermediate, BertOutput; the encoder is constructed at foldingdiff/modelling.py:271 and called at :473-480):
//   TAIL   attention.output.dense + bias + residual (the layer's input rows) + LayerNorm                 (BertSelfOutput)
//          intermediate.dense + bias + exact-erf GELU                                                     (BertIntermediate)
//          output.dense + bias + residual (BertSelfOutput's output) + LayerNorm                           (BertOutput)
// Neither BertSelfOutput's output nor the 2 d wide intermediate reaches HBM: 604 MB less traffic per layer at BASELINE C2.  The step is
// bound by what the chip can store and, with the shader clock at 1.8-2.1 of 2.4 GHz under the 1400 W cap, by what it burns
// (profiles/r06_power_probe.log); same box, sustained: 74.7 -> 83.4 backbones/s (profiles/r06_ffn16_notes.log).
//
// The structure is seq_attn16.hip's projection, three times over: a wave owns SIXTEEN token rows, two waves per SIMD, 256 registers each,
// every contraction on v_mfma_f32_16x16x32_f16 in the swapped form D^T = W x^T (a lane owns a token):
//   * the rows' operand image (hi / lo fp16) is stationary: the hi plane in 48 registers, the lo plane in LDS (96 KiB per 128 rows, each
//     wave its own rows, lane-linear: read back as the B operand w_hi x_lo needs, one 16-byte read per step -- with both planes in
//     registers hipcc spills the image and reloads it behind vmcnt(0) in every group); B operand of a dense AND the residual of the next
//     one; a dense's 16 x d output accumulates in 96 registers;
//   * the weight rows are PERMUTED in the stream (tile j of a pair, row i = feature 8 (i / 4) + 4 j + (i % 4)) so that the C/D layout hands a
//     lane eight consecutive features of its token = one 16-byte B-operand unit of the next dense, and one unit of the row images: no
//     cross-lane traffic anywhere.  BertSelfOutput's LayerNorm output therefore IS the stationary operand of the first dense (hi halves
//     into the context's registers, lo halves into its LDS rows); the intermediate is produced 64 features (four 16 x 16 tiles) at a
//     time -- bias, GELU and the hi / lo split in registers -- as the B operand pairs of the second dense;
//   * the three weight matrices arrive as ONE linear stream of 16 KiB stages in consumption order -- attention.output.dense (36 stages at
//     d_model 384), then per 64-feature group 6 stages of the first dense and 6 of the second, the first dense one group ahead (its
//     neighbour's GELU runs inside it): 180 stages = 2.8 MiB per pass -- through a 3-slot LDS ring: LDS-DMA, two 1 KiB pieces per wave and
//     stage, one workgroup barrier per stage, counted s_waitcnt vmcnt (the stream never drains inside a pass and does not stop at a
//     pass's end); every block is a whole number of ring turns, so every fragment address is an immediate;
//   * a step = four weight tiles against one B operand pair: 8 fragment reads and 12 MFMAs (w_hi x_hi | w_hi x_lo | w_lo x_hi), the
//     next plane requested while the current one multiplies.
// LayerNorm: a row's 384 values sit in the four lanes (c, g = 0..3): in-lane sums + two lane-group swaps.
// Arithmetic: the fp16 hi / lo split triples of gemm_img.hip with K = 32 MFMAs: fp32-class results, not the bits of the GEMM path (the
// oracle gates of tests/test_gpu_parity.py are the contract; tests/test_host.py emulates the layouts lane by lane).
#include <cstdlib>
#include <type_traits>

#include "fdmi_kernels.h"
#include "img_common.h"

namespace fdmi {
namespace ffn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) u32x4* lds_cu128_t;
// LDS accesses through INTEGER addresses (behind a pointer derived from the LDS array hipcc assumes an alias with the LDS-DMA writes
// in flight and waits for the whole weight stream: profiles/r05_seq_attn_notes.log)
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long long)(lds_ptr_t)(const_cast<void*>(p)); }
__device__ __forceinline__ u32x4 lds_u128(unsigned a) { return *(lds_cu128_t)(unsigned long long)a; }
__device__ __forceinline__ f16x8 lds_f16x8(unsigned a) { return __builtin_bit_cast(f16x8, lds_u128(a)); }

__device__ __forceinline__ void swap16(unsigned& vdst, unsigned& src) {  // rows 1, 3 of vdst <-> rows 0, 2 of src (v_permlane16_swap_b32)
  const auto r = __builtin_amdgcn_permlane16_swap(vdst, src, false, false);
  vdst = r[0];
  src = r[1];
}
// sum over the four lanes (c, g = 0..3) that hold one token row: lanes c, c + 16, c + 32, c + 48 (every lane ends up with it)
__device__ __forceinline__ float quad_sum(float x) {
  unsigned a = __builtin_bit_cast(unsigned, x), b = a;
  swap32(a, b);
  float m = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
  a = __builtin_bit_cast(unsigned, m);
  b = a;
  swap16(a, b);
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}

template <int V> using IC = std::integral_constant<int, V>;
template <int LO, int HI, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (LO < HI) {
    f(IC<LO>{});
    static_for<LO + 1, HI>(f);
  }
}

#ifndef FDMI_FFN_DBG
#define FDMI_FFN_DBG 0  // ablation builds (WRONG results): 1 = no GELU arithmetic, 2 = no MFMAs of the second dense, 4 = none of the first,
                        // 8 = no lo-plane fragment reads (half the LDS traffic), 16 = no output stores
#endif
#ifndef FDMI_FFN_ST_AUX
#define FDMI_FFN_ST_AUX 0  // cache policy bits of the output stores (A/B builds: 2 = nt)
#endif
#ifndef FDMI_FFN_PRIO
#define FDMI_FFN_PRIO 2
#endif

__device__ __forceinline__ f32x4 mfma16(const f16x8& a, const f16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

constexpr int NW = 8;                 // waves per workgroup: 16 token rows each
constexpr int TILE = 2048;            // a 16 x 32 weight tile, hi and lo: [unit 0-7][row 0-15][16 B]
constexpr int STEP = 4 * TILE;        // four tiles against one B operand pair
constexpr int SPS = 2;                // steps per ring stage
constexpr int STAGE = SPS * STEP;     // a ring stage: 16 KiB = 16 LDS-DMA pieces = 2 per wave
constexpr int NST = 3;                // ring slots
constexpr int PPW = STAGE / 1024 / NW;  // pieces per wave and stage
constexpr int OFF_W = 0;
constexpr int OFF_X = NST * STAGE;    // lo plane of the input rows: [wave][k32 step][lane][16 B], 8 x NKT KiB
__host__ __device__ constexpr int off_p(int nkt) { return OFF_X + NW * nkt * 1024; }  // parameters: b_i [2 d] | b_d [d] | gamma [d] | beta at the output image's scale [d]

// PROF: workgroup 0 records s_memtime stamps (FDMI_STAMPS=1): stamps[wave][pass, 16][16] = 0 pass top | 1..12 after group G | 13 after
// the LayerNorm and its stores | 14 after the next rows landed
// TAIL: BertSelfOutput (attention.output.dense + residual + LayerNorm) runs in front, on the attention context; its output never
// reaches HBM either: in the C/D layout of the swapped form it IS the stationary B operand of the first dense (the same lanes hold the
// same eight features) -- the hi halves go to the registers the context's hi plane occupied, the lo halves to its LDS rows.
template <int NKT, bool TAIL, bool PROF>
__global__ __launch_bounds__(64 * NW) void ffn16_kernel(FfnArgs p) {
  static_assert(NKT == 12 || NKT == 6, "d_model 384 or 192, intermediate size 2 d_model");
  constexpr int D = 32 * NKT;          // d_model
  constexpr int NG = NKT;              // groups of 64 intermediate features
  constexpr int SPG = 2 * NKT;         // steps per group: NKT of the first dense (k32 steps), NKT of the second (2 pairs x 2 NKT tiles / 4)
  constexpr int STG = SPG / SPS;       // stages per group
  static_assert(STG % NST == 0, "a group is a whole number of ring turns: its first stage always sits in slot 0");
  constexpr int NSA = TAIL ? NKT * NKT / 2 : 0;  // steps of attention.output.dense: k32 step kt = s / (NKT / 2), output tiles 4 q .. 4 q + 3
  static_assert((NSA / SPS) % NST == 0 && NSA % SPS == 0, "whole ring turns");
  constexpr int W_BYTES = (NSA / SPS + NG * STG) * STAGE;
  constexpr int OFF_P = off_p(NKT);
  constexpr int OFF_P1 = OFF_P + 5 * D * 4;    // TAIL: b_o / os [d] | gamma_1 [d] | beta_1 at the scale of its image [d]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wq = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wq >> 2;
  unsigned smem0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  asm volatile("" : "+s"(smem0));
  // (nothing derived from the lane index lives across the loops: seq_attn16.hip)
  auto lane_id = [&]() __attribute__((always_inline)) {
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    return ln;
  };
  auto lane_off_of = [](int ln) __attribute__((always_inline)) { return (unsigned)((ln >> 4) * 256 + (ln & 15) * 16); };

  {
    float* par = reinterpret_cast<float*>(smem + OFF_P);
    for (int i = tid; i < 2 * D; i += 64 * NW) par[i] = p.bi[i];
    for (int i = tid; i < D; i += 64 * NW) {
      par[2 * D + i] = p.bd[i];
      par[3 * D + i] = p.gamma[i];
      par[4 * D + i] = p.beta[i] * p.out_scale;  // (a power of two: exact)
      if constexpr (TAIL) {
        par[5 * D + i] = p.bo[i] * (1.0f / p.ao_scale);  // (1 / os = the product of two image scales, a power of two)
        par[6 * D + i] = p.g1[i];
        par[7 * D + i] = p.b1[i] * p.a_scale;
      }
    }
  }

  // ---- the weight stream (it simply runs on past the workgroup's last stage: what it requests there lands in free slots)
  int w_src = 0, w_slot = 0;
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.wimg), 0, W_BYTES, 0x00020000);
  auto issue_w = [&]() __attribute__((always_inline)) {
    const lds_ptr_t dst = (lds_ptr_t)(unsigned long long)(smem0 + OFF_W + (unsigned)__builtin_amdgcn_readfirstlane(w_slot));
    const int so = __builtin_amdgcn_readfirstlane(w_src);
    const int vo = lane_id() * 16;
#pragma unroll
    for (int k = 0; k < PPW; ++k) dma16(rs_w, dst + (wq + NW * k) * 1024, vo, so + (wq + NW * k) * 1024);
    w_src = w_src + STAGE == W_BYTES ? 0 : w_src + STAGE;
    w_slot = w_slot + STAGE == NST * STAGE ? 0 : w_slot + STAGE;
  };

  // ---- the wave's sixteen rows of the input image: k32 step kt, lane (c, g): row c, features 32 kt + 8 g .. + 7; the hi plane in
  // registers, the lo plane in this wave's 12 KiB of LDS ([kt][lane][16 B]: nobody else touches them, no barrier involved)
  f16x8 ah[NKT];
  const __amdgpu_buffer_rsrc_t rs_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(TAIL ? p.cimg : p.aimg), 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(TAIL ? p.hres : p.aimg), 0, p.a_bytes, 0x00020000);
  auto row_off = [&](int r0) __attribute__((always_inline)) {
    const int ln = lane_id();
    const int row = r0 + 16 * wq + (ln & 15);
    return (unsigned)(((row >> 5) * (NKT * 256) + (row & 31)) * 16 + (ln >> 4) * 512);
  };
  // (The lo plane goes through registers.  Fetched by LDS-DMA with these per-lane offsets it arrived wrong now and then -- behind
  // vmcnt(0), a barrier and a long sleep: stale LDS contents in ~40 % of the runs of a 6-layer d_model-192 model, never with the
  // two wave groups' stage tops at the same place; scripts/ffn16_stress.py, profiles/r06_ffn16_notes.log.  Twelve loads and twelve
  // LDS writes per wave and pass cost nothing.)
  u32x4 tl[NKT];
  auto load_hi = [&](int r0) __attribute__((always_inline)) {  // rows beyond the image read as zeros
    const unsigned hoff = row_off(r0);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
      ah[kt] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)hoff, kt * 8 * 512, 0));
  };
  auto load_lo = [&](auto KT, unsigned hoff) __attribute__((always_inline)) {
    constexpr int kt = decltype(KT)::value;
    tl[kt] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)hoff, (kt * 8 + 4) * 512, 0));
  };
  auto commit_lo = [&]() __attribute__((always_inline)) {
    const unsigned ax = smem0 + OFF_X + (unsigned)(wq * NKT * 1024 + lane_id() * 16);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) *(__attribute__((address_space(3))) u32x4*)(unsigned long long)(ax + (unsigned)(kt * 1024)) = tl[kt];
  };
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  // TAIL: the residual of BertSelfOutput (the layer's input rows) is loaded straight into the accumulators of attention.output.dense
  // -- raw hi / lo words first (tile 2 kt: hi, 2 kt + 1: lo), turned into (b_o + h) / os at the pass top: os is a power of two, so the
  // accumulators then simply start from the bias and the residual instead of zero
  auto load_res = [&](auto KT, unsigned hoff, f32x4 (&Yr)[2 * NKT]) __attribute__((always_inline)) {
    constexpr int kt = decltype(KT)::value;
    Yr[2 * kt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_h, (int)hoff, kt * 8 * 512, 0));
    Yr[2 * kt + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_h, (int)hoff, (kt * 8 + 4) * 512, 0));
  };

  const bool rec = PROF && blockIdx.x == 0 && p.stamps != nullptr;
  unsigned long long* stp = PROF ? p.stamps + (size_t)wq * 16 * 16 : nullptr;
  int slot = 0;
#define FD_STAMP(i) do { if (PROF) { if (rec && slot < 16 && lane == 0) stp[slot * 16 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
#define FD_SB() __builtin_amdgcn_sched_barrier(0)
// (inside iteration 1 of the first pass, into slot 15: 0 top | 2 after the next group's first dense with this group's GELU inside | 3 after
// this group's second dense)
#define FD_STAMP_G(i) do { if (PROF) { if (rec && slot == 0 && G == 1 && lane == 0) stp[15 * 16 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)

  // (packed rows: the host knows an upper bound of the row count only; the count itself is in device memory)
  const int n_pass = p.dims ? min(p.panels, __builtin_amdgcn_readfirstlane(p.dims[1]) >> 7) : p.panels;
  if ((int)blockIdx.x >= n_pass) return;
  const float os_up = p.up_scale, os_dn = p.down_scale, hs = 0.5f * p.g_scale;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 Y[2 * NKT];  // second dense: out tile T = 2 kt + j, row i = output feature 32 kt + 8 (i / 4) + 4 j + (i % 4)
  f32x4 U[4];        // first dense: tile t = 2 pair + j, row i = intermediate feature 64 G + 32 pair + 8 (i / 4) + 4 j + (i % 4)
  f32x4 Up[4];       // ... of the group before (its GELU runs beside this group's matrix instructions)
  u32x4 uh[2], ul[2];  // GELU output of a group as B operand pairs: this lane's token, features 64 G + 32 pair + 8 g .. + 7

  load_hi((int)blockIdx.x * 128);
  {
    const unsigned hoff = row_off((int)blockIdx.x * 128);
    static_for<0, NKT>([&](auto KT) __attribute__((always_inline)) { load_lo(KT, hoff); });
    if constexpr (TAIL) static_for<0, NKT>([&](auto KT) __attribute__((always_inline)) { load_res(KT, hoff, Y); });
  }
  issue_w();
  issue_w();
  issue_w();
  FD_WAIT_VM(3 * PPW);  // the rows landed
  commit_lo();
  FD_WAIT_VM(2 * PPW);  // ... and stage 0 (requested in this order; stages 1, 2 may still be in flight)
  barrier_keep_vm();

  // ---- steps [S0, S1) of a group (its first stage sits in ring slot 0).  Step s < NKT: first dense, k32 step s (B = the input
  // image); s >= NKT: second dense, pair (s - NKT) / (NKT / 2), output tiles 4 q .. 4 q + 3, q = (s - NKT) % (NKT / 2) (B = the
  // GELU output).  A software pipeline over the planes hi(s) lo(s) hi(s + 1) ...: while a plane multiplies (hi: 8 MFMAs, lo: 4) the next
  // one is requested into the other buffer.  The last step of every stage (odd s) carries the stage top: the barrier that
  // publishes the NEXT stage sits behind that step's lo-plane reads (the last reads of the stage, whose ring slot the request at that
  // top overwrites) and in front of the reads of hi(s + 1).  The two waves of a SIMD have it at different places: group 0 in front of
  // the step's eight hi-plane MFMAs, group 1 behind them.  vmcnt retires in order: when a stage is published the only younger
  // requests are the two pieces of the stage after it.
  // ---- one eighth of a group's GELU: bias, 0.5 s_g x (1 + erf(x / sqrt 2)) at the scale of the intermediate's image (the operations of
  // gemm_img.hip's gelu_erf4_scaled), hi / lo split of TWO values = word j of pair pr's B operands.  abq: LDS address of this lane's
  // eight bias values of pair 0.  Plain fp32 VALU: it issues beside the SIMD's other wave's matrix instructions.
  auto gelu_piece = [&](auto P, unsigned abq) __attribute__((always_inline)) {
    constexpr int pc = decltype(P)::value, pr = pc >> 2, j = pc & 3;
    const u32x2 b = *(const __attribute__((address_space(3))) u32x2*)(unsigned long long)(abq + (unsigned)((32 * pr + 2 * j) * 4));
    float o0 = __builtin_fmaf(Up[2 * pr + (j >> 1)][2 * (j & 1)], os_up, __builtin_bit_cast(float, (unsigned)b[0]));
    float o1 = __builtin_fmaf(Up[2 * pr + (j >> 1)][2 * (j & 1) + 1], os_up, __builtin_bit_cast(float, (unsigned)b[1]));
#if !(FDMI_FFN_DBG & 1)
    const float h0 = o0 * hs, h1 = o1 * hs;
    o0 = __builtin_fmaf(h0, erf_rational(o0 * 0.70710678118654752440f), h0);
    o1 = __builtin_fmaf(h1, erf_rational(o1 * 0.70710678118654752440f), h1);
#endif
    unsigned a, c;
    split_pair(o0, o1, a, c);
    uh[pr][j] = a;
    ul[pr][j] = c;
  };

  // GEL = 2: the steps of attention.output.dense (TAIL): k32 step kt = s / (NKT / 2) of the context (B operand), output tiles
  // 4 q .. 4 q + 3, q = s % (NKT / 2)
  auto steps = [&](auto S0, auto S1, auto GEL, unsigned abq) __attribute__((always_inline)) {
    constexpr int s0 = decltype(S0)::value, s1 = decltype(S1)::value;
    constexpr bool gel = decltype(GEL)::value == 1, pha = decltype(GEL)::value == 2;
    auto stage_top = [&]() __attribute__((always_inline)) {
      FD_WAIT_VM(PPW);
      barrier_keep_vm();
      issue_w();
    };
    f16x8 fx[4], fy[4], xl;
    unsigned a_W, a_X;
    {
      const int ln = lane_id();
      a_W = smem0 + OFF_W + lane_off_of(ln);
      a_X = smem0 + OFF_X + (unsigned)(wq * NKT * 1024 + ln * 16);
      asm volatile("" : "+v"(a_W), "+v"(a_X));
    }
    auto plane_reads = [&](auto S, auto LO, f16x8 (&f)[4]) __attribute__((always_inline)) {
      constexpr int s = decltype(S)::value, lo = decltype(LO)::value;
      constexpr unsigned off = (unsigned)(((s / SPS) % NST) * STAGE + (s % SPS) * STEP + lo * 1024);
#if FDMI_FFN_DBG & 8
      if constexpr (lo == 1) {
#pragma unroll
        for (int t = 0; t < 4; ++t) f[t] = fx[t];
      } else
#endif
#pragma unroll
      for (int t = 0; t < 4; ++t) f[t] = lds_f16x8(a_W + off + (unsigned)(t * TILE));
      // (the step's x_lo operand rides with its hi plane: the previous step's w_hi x_lo is done)
      if constexpr (pha) {
        if constexpr (lo == 0 && s % (NKT / 2) == 0) xl = lds_f16x8(a_X + (unsigned)((s / (NKT / 2)) * 1024));
      } else {
        if constexpr (lo == 0 && s < NKT) xl = lds_f16x8(a_X + (unsigned)(s * 1024));
      }
    };
    // which: 0 = w_hi x_hi, 1 = w_hi x_lo, 2 = w_lo x_hi; consecutive MFMAs never share an accumulator
    auto mm = [&](auto S, auto WHICH, const f16x8 (&f)[4]) __attribute__((always_inline)) {
      constexpr int s = decltype(S)::value, which = decltype(WHICH)::value;
      if constexpr (pha) {
        constexpr int kt = s / (NKT / 2), q = s % (NKT / 2);
#pragma unroll
        for (int t = 0; t < 4; ++t) Y[4 * q + t] = mfma16(f[t], which == 1 ? xl : ah[kt], Y[4 * q + t]);
      } else if constexpr (s < NKT) {
#if !(FDMI_FFN_DBG & 4)
#pragma unroll
        for (int t = 0; t < 4; ++t) U[t] = mfma16(f[t], which == 1 ? xl : ah[s], U[t]);
#endif
      } else {
#if !(FDMI_FFN_DBG & 2)
        constexpr int pr = (s - NKT) / (NKT / 2), q = (s - NKT) % (NKT / 2);
#pragma unroll
        for (int t = 0; t < 4; ++t) Y[4 * q + t] = mfma16(f[t], __builtin_bit_cast(f16x8, which == 1 ? ul[pr] : uh[pr]), Y[4 * q + t]);
#endif
      }
    };
    plane_reads(S0, IC<0>{}, fx);
    FD_SB();
    static_for<s0, s1>([&](auto S) __attribute__((always_inline)) {
      constexpr int s = decltype(S)::value;
      constexpr bool top = s % SPS == SPS - 1;
      plane_reads(S, IC<1>{}, fy);
      FD_SB();
      if constexpr (top) {
        if (grp == 0) stage_top();
      }
#if FDMI_FFN_PRIO
      __builtin_amdgcn_s_setprio(FDMI_FFN_PRIO);
#endif
      mm(S, IC<0>{}, fx);
      mm(S, IC<1>{}, fx);
#if FDMI_FFN_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      FD_SB();
      if constexpr (top) {
        if (grp != 0) stage_top();
      }
      if constexpr (s + 1 < s1) plane_reads(IC<s + 1>{}, IC<0>{}, fx);
      FD_SB();
      if constexpr (gel) {
        // the previous group's GELU, an eighth at a time in the steps 1 .. NKT - 2 of this group's first dense; group 0 here (beside
        // the lo-plane instructions of group 1), group 1 at the step's end.  (Measured: at the same place or at different ones, inside the
        // first dense or on its own in front of it, a group costs the same 15.4-15.6 k cycles -- with 16 x 16 x 32 MFMAs the vector time
        // adds to the matrix time, profiles/r06_ffn16_notes.log.  The arrangement stays for the day fragments can be requested further ahead.)
        if (grp == 0) {
          static_for<0, 8>([&](auto P) __attribute__((always_inline)) {
            if constexpr (1 + decltype(P)::value * (NKT - 2) / 8 == s) gelu_piece(P, abq);
          });
        }
        FD_SB();
      }
      mm(S, IC<2>{}, fy);
      FD_SB();
      if constexpr (gel) {  // (group 1: at the end of the step, beside the hi-plane instructions of group 0's next step)
        if (grp != 0) {
          static_for<0, 8>([&](auto P) __attribute__((always_inline)) {
            if constexpr (1 + decltype(P)::value * (NKT - 2) / 8 == s) gelu_piece(P, abq);
          });
        }
        FD_SB();
      }
    });
  };

  for (int panel = blockIdx.x; panel < n_pass; panel += (int)gridDim.x) {
    FD_STAMP(0);
    if constexpr (TAIL) {
      // ================================================ BertSelfOutput: dense on the context + bias + residual, LayerNorm
      const int ln = lane_id();
      const int g = ln >> 4;
      const unsigned pb1 = smem0 + OFF_P1 + (unsigned)(8 * g * 4);
      const unsigned ax = smem0 + OFF_X + (unsigned)(wq * NKT * 1024 + ln * 16);
      {
        const float c = p.hres_inv * (1.0f / p.ao_scale);  // (both powers of two)
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          const u32x4 rh = __builtin_bit_cast(u32x4, Y[2 * kt]), rl = __builtin_bit_cast(u32x4, Y[2 * kt + 1]);
          const u32x4 b0 = lds_u128(pb1 + (unsigned)(kt * 128)), b1 = lds_u128(pb1 + (unsigned)(kt * 128 + 16));
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
              const unsigned hw = rh[2 * j + e / 2], lw = rl[2 * j + e / 2];  // features 8 g + 4 j + e, + 1 of block kt
              const unsigned bw0 = j == 0 ? b0[e] : b1[e], bw1 = j == 0 ? b0[e + 1] : b1[e + 1];
              float v0 = fma_mix_lo(hw, c, __builtin_bit_cast(float, bw0));
              float v1 = fma_mix_hi(hw, c, __builtin_bit_cast(float, bw1));
              Y[2 * kt + j][e] = fma_mix_lo(lw, c, v0);
              Y[2 * kt + j][e + 1] = fma_mix_hi(lw, c, v1);
            }
        }
      }
      steps(IC<0>{}, IC<NSA>{}, IC<2>{}, 0u);
      // LayerNorm; its output at the scale of the image the two-kernel path writes (s_a): hi halves -> the stationary B operand,
      // lo halves -> this wave's LDS rows (the context's planes are dead)
      const float os_ao = p.ao_scale;
      float sum = 0.f;
#pragma unroll
      for (int t = 0; t < 2 * NKT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          Y[t][e] *= os_ao;
          sum += Y[t][e];
        }
      const float inv_n = 1.0f / (float)D;
      float mean = quad_sum(sum) * inv_n;
      asm volatile("" : "+v"(mean));
      float t2 = 0.f;
#pragma unroll
      for (int t = 0; t < 2 * NKT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dl = Y[t][e] - mean;
          Y[t][e] = dl;
          t2 = __builtin_fmaf(dl, dl, t2);
        }
      const float rstd = (1.0f / sqrtf(__builtin_fmaf(quad_sum(t2), inv_n, p.eps1))) * p.a_scale;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const u32x4 g0 = lds_u128(pb1 + (unsigned)(D * 4 + kt * 128)), g1 = lds_u128(pb1 + (unsigned)(D * 4 + kt * 128 + 16));
        const u32x4 e0 = lds_u128(pb1 + (unsigned)(2 * D * 4 + kt * 128)), e1 = lds_u128(pb1 + (unsigned)(2 * D * 4 + kt * 128 + 16));
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = __builtin_fmaf(Y[2 * kt][e] * rstd, __builtin_bit_cast(float, (unsigned)g0[e]), __builtin_bit_cast(float, (unsigned)e0[e]));
          o[4 + e] = __builtin_fmaf(Y[2 * kt + 1][e] * rstd, __builtin_bit_cast(float, (unsigned)g1[e]), __builtin_bit_cast(float, (unsigned)e1[e]));
        }
        u32x4 hv, lv;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned a, b;
          split_pair(o[2 * j], o[2 * j + 1], a, b);
          hv[j] = a;
          lv[j] = b;
        }
        ah[kt] = __builtin_bit_cast(f16x8, hv);
        *(__attribute__((address_space(3))) u32x4*)(unsigned long long)(ax + (unsigned)(kt * 1024)) = lv;
      }
    }
#pragma unroll
    for (int t = 0; t < 2 * NKT; ++t) Y[t] = zero4;
    // Software pipeline over the groups: the first dense of group G + 1 runs BEFORE the second dense of group G, and the GELU of group G
    // sits inside it (the weight stream is ordered that way: up(0) | up(1) down(0) | up(2) down(1) | ... | up(NG - 1) down(NG - 2) |
    // down(NG - 1)).  In sequence (first dense, GELU, second dense) both waves of a SIMD would run their GELU at the same time with
    // the matrix pipe idle: 3.4 k of a group's 15.9 k cycles (profiles/r06_ffn16_notes.log).
#pragma unroll
    for (int t = 0; t < 4; ++t) U[t] = zero4;
    steps(IC<0>{}, IC<NKT>{}, IC<0>{}, 0u);
    for (int G = 0; G < NG; ++G) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        Up[t] = U[t];
        U[t] = zero4;
      }
      unsigned abq = smem0 + OFF_P + (unsigned)((64 * G + 8 * (lane_id() >> 4)) * 4);
      asm volatile("" : "+v"(abq));
      FD_STAMP_G(0);
      FD_STAMP_G(1);
      if (G + 1 < NG) {
#ifdef FDMI_FFN_NOPIPE  // A/B build: the GELU on its own, in front of the next group's first dense
        static_for<0, 8>([&](auto P) __attribute__((always_inline)) { gelu_piece(P, abq); });
        steps(IC<0>{}, IC<NKT>{}, IC<0>{}, abq);
#else
        steps(IC<0>{}, IC<NKT>{}, IC<1>{}, abq);
#endif
      } else {
        static_for<0, 8>([&](auto P) __attribute__((always_inline)) { gelu_piece(P, abq); });
      }
      FD_STAMP_G(2);
      steps(IC<NKT>{}, IC<SPG>{}, IC<0>{}, 0u);
      FD_STAMP_G(3);
      if (slot == 0) FD_STAMP(1 + (G < 12 ? G : 11));
    }
    // ---- dense + bias + residual, LayerNorm, the output image (BertOutput): v = acc / (s_g s_w) + b_d + (hi + lo) / s_a
    {
      const int ln = lane_id();
      const int g = ln >> 4;
      const unsigned pb = smem0 + OFF_P + (unsigned)((2 * D + 8 * g) * 4);
      const unsigned ax = smem0 + OFF_X + (unsigned)(wq * NKT * 1024 + ln * 16);
      const float ri = p.resid_inv;
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const u32x4 b0 = lds_u128(pb + (unsigned)(kt * 128)), b1 = lds_u128(pb + (unsigned)(kt * 128 + 16));
        const u32x4 rh = __builtin_bit_cast(u32x4, ah[kt]), rl = lds_u128(ax + (unsigned)(kt * 1024));
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const unsigned bw0 = j == 0 ? b0[e] : b1[e], bw1 = j == 0 ? b0[e + 1] : b1[e + 1];
            float v0 = __builtin_fmaf(Y[2 * kt + j][e], os_dn, __builtin_bit_cast(float, bw0));
            float v1 = __builtin_fmaf(Y[2 * kt + j][e + 1], os_dn, __builtin_bit_cast(float, bw1));
            const unsigned hw = rh[2 * j + e / 2], lw = rl[2 * j + e / 2];  // features 8 g + 4 j + e, + 1 of block kt
            v0 = fma_mix_lo(hw, ri, v0);
            v1 = fma_mix_hi(hw, ri, v1);
            v0 = fma_mix_lo(lw, ri, v0);
            v1 = fma_mix_hi(lw, ri, v1);
            Y[2 * kt + j][e] = v0;
            Y[2 * kt + j][e + 1] = v1;
            sum += v0;
            sum += v1;
          }
      }
      // the rows are done with the input image: the next pass's rows replace it under the LayerNorm
      const int next = panel + (int)gridDim.x;
      load_hi(next * 128);  // (the lo plane: block by block behind the stores below, into the registers they free)
      const unsigned noff = row_off(next * 128);
      const float inv_n = 1.0f / (float)D;
      float mean = quad_sum(sum) * inv_n;
      asm volatile("" : "+v"(mean));
      float t2 = 0.f;
#pragma unroll
      for (int t = 0; t < 2 * NKT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dl = Y[t][e] - mean;
          Y[t][e] = dl;
          t2 = __builtin_fmaf(dl, dl, t2);
        }
      const float rstd = (1.0f / sqrtf(__builtin_fmaf(quad_sum(t2), inv_n, p.eps))) * p.out_scale;
      const unsigned ooff = row_off(panel * 128);
      static_for<0, NKT>([&](auto KT) __attribute__((always_inline)) {
        constexpr int kt = decltype(KT)::value;
        const u32x4 g0 = lds_u128(pb + (unsigned)(D * 4 + kt * 128)), g1 = lds_u128(pb + (unsigned)(D * 4 + kt * 128 + 16));
        const u32x4 e0 = lds_u128(pb + (unsigned)(2 * D * 4 + kt * 128)), e1 = lds_u128(pb + (unsigned)(2 * D * 4 + kt * 128 + 16));
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = __builtin_fmaf(Y[2 * kt][e] * rstd, __builtin_bit_cast(float, (unsigned)g0[e]), __builtin_bit_cast(float, (unsigned)e0[e]));
          o[4 + e] = __builtin_fmaf(Y[2 * kt + 1][e] * rstd, __builtin_bit_cast(float, (unsigned)g1[e]), __builtin_bit_cast(float, (unsigned)e1[e]));
        }
        u32x4 hv, lv;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned a, b;
          split_pair(o[2 * j], o[2 * j + 1], a, b);
          hv[j] = a;
          lv[j] = b;
        }
#if FDMI_FFN_DBG & 16
        if (hv[0] == 0x12345678u && lv[1] == 0x9abcdef0u)   // (the values stay needed; practically never stored)
#endif
        {
          __builtin_amdgcn_raw_buffer_store_b128(hv, rs_o, (int)ooff, kt * 8 * 512, FDMI_FFN_ST_AUX);
          __builtin_amdgcn_raw_buffer_store_b128(lv, rs_o, (int)ooff, (kt * 8 + 4) * 512, FDMI_FFN_ST_AUX);
        }
        load_lo(KT, noff);  // (unconditionally: beyond the last pass the rows read as zeros and are never used)
        if constexpr (TAIL) load_res(KT, noff, Y);
        FD_SB();
      });
    }
    FD_STAMP(13);
    FD_WAIT_VM(0);  // the next rows landed, the stores left (and with them the two stages in flight: once per pass)
    commit_lo();
    FD_STAMP(14);
    ++slot;
  }
  FD_WAIT_VM(0);  // nothing may land in LDS after the workgroup has exited
#undef FD_STAMP
#undef FD_STAMP_G
#undef FD_SB
}

static int n_cu_of(int dev) {
  static int cached[64] = {0};
  if (dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    hipDeviceProp_t prop;
    cached[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return cached[dev];
}

template <int NKT, bool TAIL>
static bool launch(const FfnArgs& p, hipStream_t s) {
  constexpr int SMEM = off_p(NKT) + (TAIL ? 8 : 5) * 32 * NKT * 4;
  static int attr_state[64] = {0};  // 0 unknown, 1 set, -1 refused
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  if (attr_state[dev] == 0) {
    bool ok = true;
    for (const void* f : {reinterpret_cast<const void*>(&ffn16_kernel<NKT, TAIL, false>), reinterpret_cast<const void*>(&ffn16_kernel<NKT, TAIL, true>)})
      ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) == hipSuccess;
    attr_state[dev] = ok ? 1 : -1;
  }
  if (attr_state[dev] < 0) return false;
  int grid = n_cu_of(dev);
  if (grid > p.panels) grid = p.panels;
  if (grid <= 0) return true;
  if (p.stamps) hipLaunchKernelGGL((ffn16_kernel<NKT, TAIL, true>), dim3(grid), dim3(64 * NW), SMEM, s, p);
  else hipLaunchKernelGGL((ffn16_kernel<NKT, TAIL, false>), dim3(grid), dim3(64 * NW), SMEM, s, p);
  return hipGetLastError() == hipSuccess;
}

}  // namespace ffn

// d_model 384 / 192 (every released configuration / the reference's test fixture) with the intermediate size 2 d_model they all have
bool ffn16_supported(int d_model, int d_ff) { return (d_model == 384 || d_model == 192) && d_ff == 2 * d_model; }

bool launch_ffn16(const FfnArgs& p, int d_model, hipStream_t s) {
  if (p.cimg) return d_model == 384 ? ffn::launch<12, true>(p, s) : ffn::launch<6, true>(p, s);
  return d_model == 384 ? ffn::launch<12, false>(p, s) : ffn::launch<6, false>(p, s);
}

}  // namespace fdmi
