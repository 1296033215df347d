// BertSelfAttention of a whole sequence in ONE kernel, two waves per SIMD: the q | k | v projection (HF 4.11.3
// BertSelfAttention.query / key / value, constructed at foldingdiff/modelling.py:271, called :473-480) AND the attention with the
// relative_key term, the additive -10000 key mask and the softmax -- q, k and v never reach HBM.
//
// Round 6.  seq_attn.hip runs one 512-register wave per SIMD (a wave owns 32 token rows; its slice of the hidden
// state alone is 192 registers), and one in-order wave cannot overlap its own vector / LDS work with its own matrix instructions:
// 45 % of the matrix pipe.  Here a wave owns SIXTEEN token rows and every contraction runs on v_mfma_f32_16x16x32_f16:
//   * 8 waves per workgroup = 2 per SIMD, 256 registers each: hidden state 96 (12 k32 steps x (hi, lo) x 4), projection
//     accumulators 24 (q_h | k_h | v_h: six 16 x 16 tiles), weight fragments 60, the attention's state ~100;
//   * the two waves of a SIMD fill each other's issue gaps: in the projection both stream MFMAs from register fragments (the pipe is
//     saturated: 72 MFMAs per stage and SIMD) while the LDS-DMA requests and the fragment reads of one hide behind the matrix
//     instructions of the other (the two groups of waves request their pieces at opposite ends of a stage); in the attention the
//     LDS round trips, the skew and the softmax of one wave sit beside the other's;
//   * no half-wave exchanges at all: the weight rows of a head are PERMUTED in the weight image (tile j, row i = head feature
//     8 (i / 4) + 4 j + (i % 4)), so that the 16 x 16 C/D layout (lane (c, g) holds rows 4 g .. 4 g + 3 of column c) hands every
//     lane the eight consecutive features 8 g .. 8 g + 7 of its token -- exactly one 16-byte operand unit of the next contraction
//     (Q^T as B operand, K rows as A operand, the ctx image's units); V^T and P^T share a permuted key order inside a 32-key step
//     (kappa = 8 g + 4 p + e  <->  key 16 p + 4 g + e) for the same reason.
// Arithmetic: the fp16 hi / lo split triples of gemm_img.hip / attention_img.hip (a product = hi hi + hi lo + lo hi, fp32
// accumulate); the MFMA's K is 32 instead of 16, so sums are associated differently: fp32-class results, NOT the bits of the
// two-kernel path (the oracle gates of tests/test_gpu_parity.py are the contract).
//
// Structure.  One workgroup per CU, persistent over sequences of <= 128 rows.  Wave (group, r) = wave 4 group + r owns the 16-row
// block r + 4 ((r & 1) ^ group) of the sequence (blocks 0-3 of a short sequence land on four different SIMDs).
//   per (sequence, head), all eight waves in step:
//     projection   NS stages of two k32 steps; the head's 96 weight rows stream through a 3-slot LDS ring (24 KiB stages, LDS-DMA,
//                  three 1 KiB pieces per wave, ONE workgroup barrier per stage, counted s_waitcnt vmcnt: the stream never drains);
//                  per wave and stage 24 fragment reads (requested ahead of their MFMAs) and 36 MFMAs
//     epilogue     bias, scale, hi / lo split: q_h -> this wave's operand registers, k_h -> LDS (A operand tiles), v_h -> LDS (V^T)
//     barrier      (K and V^T of all eight waves are in place)
//     attention    S^T = K Q^T (8 key tiles), the relative_key band R^T = E Q^T (9 band tiles of 16 distances, skewed through a 2 KiB
//                  per-wave LDS scratch), mask, softmax in the log2 domain (row statistics: in-lane + two lane-group swaps),
//                  O^T = V^T P^T, ctx block -> HBM (two 16-byte stores per lane) -- every piece an explicit software pipeline
//                  (operands of group n + 1 requested while group n multiplies, sched_barriers between the steps): left alone hipcc
//                  serialises the tiles to save registers, and the wave waits out every LDS round trip
//   Other structures measured on the way (scripts/round6/README.md, profiles/r06_seq_attn16_notes.log): the two groups half a period
//   apart (one projects while the other attends: the attending wave becomes the SIMD's critical path, every weight stage is streamed
//   twice, and an LDS-DMA piece costs its issuer ~70 cycles -- 273-287 us against this file's figure).
//   Key-range halves / waves beyond the sequence's rows are skipped (wave-uniform branches), so short sequences (BASELINE C3, packed
//   rows) cost what their rows cost.
#include <cstdlib>
#include <type_traits>

#include "fdmi_kernels.h"
#include "img_common.h"

namespace fdmi {
namespace s16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) float* lds_cf32_t;
typedef const __attribute__((address_space(3))) u32x4* lds_cu128_t;
typedef __attribute__((address_space(3))) u32x4* lds_u128_t;
typedef __attribute__((address_space(3))) u32x2* lds_u64_t;
typedef __attribute__((address_space(3))) float* lds_f32_t;
// LDS accesses through INTEGER addresses: behind a pointer derived from the LDS array hipcc assumes that the access may alias the
// LDS-DMA writes in flight and waits for the whole weight stream to land (profiles/r05_seq_attn_notes.log)
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long long)(lds_ptr_t)(const_cast<void*>(p)); }
__device__ __forceinline__ float lds_f32(unsigned a) { return *(lds_cf32_t)(unsigned long long)a; }
__device__ __forceinline__ u32x4 lds_u128(unsigned a) { return *(lds_cu128_t)(unsigned long long)a; }
__device__ __forceinline__ f16x8 lds_f16x8(unsigned a) { return __builtin_bit_cast(f16x8, lds_u128(a)); }

constexpr float PS = 1024.0f;  // probabilities are <= 1: p' = PS p keeps the lo halves of small probabilities normal
constexpr float kLog2e = 1.44269504088896341f;
constexpr float kInvSqrtD = 0.17677669529663687f;  // 1 / sqrt(32)

// one int of a small device table as a SCALAR load (hipcc reads such tables with global_load_dword inside the loop, and the
// vmcnt(0) it then puts in front of the first use drains the weight stream)
__device__ __forceinline__ void sload4(const int* a, int ia, const int* b, int ib, const int* c, int ic, const int* d, int id, int& x,
                                       int& y, int& z, int& w) {
  asm volatile(
      "s_load_dword %0, %4, %5\n\ts_load_dword %1, %6, %7\n\ts_load_dword %2, %8, %9\n\ts_load_dword %3, %10, %11\n\ts_waitcnt lgkmcnt(0)"
      : "=&s"(x), "=&s"(y), "=&s"(z), "=&s"(w)
      : "s"(a), "s"(__builtin_amdgcn_readfirstlane(ia * 4)), "s"(b), "s"(__builtin_amdgcn_readfirstlane(ib * 4)), "s"(c),
        "s"(__builtin_amdgcn_readfirstlane(ic * 4)), "s"(d), "s"(__builtin_amdgcn_readfirstlane(id * 4))
      : "memory");
}

// max / sum over the four lanes (c, g = 0..3) that hold one query's scores: lanes c, c + 16, c + 32, c + 48
__device__ __forceinline__ void swap16(unsigned& vdst, unsigned& src) {  // rows 1, 3 of vdst <-> rows 0, 2 of src (v_permlane16_swap_b32)
  const auto r = __builtin_amdgcn_permlane16_swap(vdst, src, false, false);
  vdst = r[0];
  src = r[1];
}
__device__ __forceinline__ float quad_max(float x) {
  unsigned a = __builtin_bit_cast(unsigned, x), b = a;
  swap32(a, b);
  float m = fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
  a = __builtin_bit_cast(unsigned, m);
  b = a;
  swap16(a, b);
  return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
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

#ifndef FDMI_S16_DBG
#define FDMI_S16_DBG 0  // ablation builds (WRONG results): 1 = the projection does not wait for its weight stages, 2 = no attention work,
                        // 4 = no projection MFMAs
#endif
#ifndef FDMI_S16_PRIO
#define FDMI_S16_PRIO 2  // > 0: s_setprio of a wave around the twelve hi-plane MFMAs of a projection step (0: off; +1.2 % same box)
#endif

__device__ __forceinline__ f32x4 mfma16(const f16x8& a, const f16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

constexpr int LP = 128;                   // keys per sequence tile
constexpr int NW = 8;                     // waves per workgroup
constexpr int KS_BYTES = 96 * 128;        // one k32 step of a head's 96 weight rows: six tiles of 2 KiB ([unit 0-7][row 0-15][16 B])
constexpr int STAGE = 2 * KS_BYTES;       // a ring stage: two k32 steps, 24 KiB = 24 LDS-DMA pieces = 3 per wave
constexpr int NST = 3;                    // ring slots
constexpr int OFF_E = 0;                  // distance table: 16 groups of 16 rows, [group][unit][row][16 B], 32 KiB
constexpr int OFF_K = 32768;              // K of the head: 8 key tiles, [tile][unit][key][16 B], 16 KiB
constexpr int OFF_V = OFF_K + 16384;      // V^T of the head: [32-key step][d tile][unit][d][16 B], 16 KiB
constexpr int OFF_R = OFF_V + 16384;      // skew scratch: 8 waves x two 1 KiB band tile slots
constexpr int OFF_W = OFF_R + NW * 2048;  // weight ring
constexpr int OFF_B = OFF_W + NST * STAGE;       // bias q | k | v at the images' scales, 3 x 384 floats
constexpr int SMEM = OFF_B + 3 * 384 * 4;        // 160,256 B

// PROF: workgroup 0 records s_memtime stamps (debug instrumentation, FDMI_STAMPS=1): stamps[wave][slot = head iteration, 32][16] =
//   0 head top | 1..NS after each projection stage | 7 after the epilogue | 8 after the K / V barrier | 9 after S^T | 10 after the band
//   | 11 after mask, maximum, exponentials, sum | 12 after P V | 13 after the ctx stores
template <int NKT, bool PROF>
__global__ __launch_bounds__(64 * NW) void seq_attn16_kernel(SeqAttnArgs p) {
  static_assert(NKT == 12 || NKT == 6, "d_model 384 or 192");
  constexpr int H = NKT;         // heads of size 32
  constexpr int NS = NKT / 2;    // ring stages per head
  constexpr int D = 32 * NKT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wq = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wq >> 2, wr = wq & 3;
  const int rb = wr + 4 * ((wr & 1) ^ grp);  // the 16-row block of the sequence this wave owns
  unsigned smem0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  asm volatile("" : "+s"(smem0));
  // Nothing derived from the lane index lives across the loops: every phase re-derives its lane coordinates (c = lane & 15: row /
  // column of a 16 x 16 tile, g = lane >> 4: k group) and LDS addresses from an OPAQUE copy of the lane index.  As loop invariants
  // hipcc keeps ~14 such values in scratch, and every reload sits behind a vmcnt(0) that drains the weight stream.
  auto lane_id = [&]() __attribute__((always_inline)) {
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    return ln;
  };
  // operand tile address inside a 2 KiB tile: unit g (hi; + 1024: lo unit 4 + g), row c
  auto lane_off_of = [](int ln) __attribute__((always_inline)) { return (unsigned)((ln >> 4) * 256 + (ln & 15) * 16); };
  static_assert(OFF_R % 2048 == 0, "the skew scratch of a wave must be 2 KiB aligned (slot swap = address ^ 1024)");

  // ---- once per workgroup: bias at the scale of the image its column feeds ((acc os + b) sc == fma(acc, os sc, b sc) exactly for the
  // power-of-two sc), the distance table (LDS row e holds table row clamp(e - esh): e = l - r + 127), zeros in K and V^T (a skipped key
  // range's probabilities are exact zeros, and 0 x whatever the LDS held at power-on must not be NaN)
  {
    float* par = reinterpret_cast<float*>(smem + OFF_B);
    for (int i = tid; i < 3 * D; i += 64 * NW) {
      const float sc = i < D ? p.q_scale : (i < 2 * D ? p.k_scale : p.v_scale);
      par[i] = p.bias[i] * sc;
    }
    const int esh = LP > p.maxpos ? LP - p.maxpos : 0;
    const int nrow_e = 2 * p.maxpos - 1;
    u32x4* Es = reinterpret_cast<u32x4*>(smem + OFF_E);
    for (int i = tid; i < 256 * 8; i += 64 * NW) {
      const int e = i >> 3, u = i & 7;
      int row = e - esh;
      row = row < 0 ? 0 : (row > nrow_e - 1 ? nrow_e - 1 : row);
      Es[(e >> 4) * 128 + u * 16 + (e & 15)] = p.demb[row * 8 + u];
    }
    u32x4* Vz = reinterpret_cast<u32x4*>(smem + OFF_K);
    for (int i = tid; i < 2048; i += 64 * NW) Vz[i] = u32x4{0u, 0u, 0u, 0u};
  }

  const float s_scale = kLog2e * kInvSqrtD / (p.q_scale * p.k_scale);  // raw MFMA sums -> log2 domain
  const float mask_raw = -10000.0f * kLog2e / s_scale;                  // (1 - mask) * -10000 at the raw scale (modelling.py:452)
  const float oss_q = p.acc_scale * p.q_scale, oss_k = p.acc_scale * p.k_scale, oss_v = p.acc_scale * p.v_scale;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // ---- the weight stream: stages (sequence, head, two k32 steps) of this workgroup; it does not stop at a sequence's end, and it
  // simply runs on past the workgroup's last stage (what it requests there lands in free slots and is never read)
  int w_src = 0, w_slot = 0;
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.wimg), 0, H * NS * STAGE, 0x00020000);
  auto issue_w = [&]() __attribute__((always_inline)) {
    const lds_ptr_t dst = (lds_ptr_t)(unsigned long long)(smem0 + OFF_W + (unsigned)__builtin_amdgcn_readfirstlane(w_slot));
    const int so = __builtin_amdgcn_readfirstlane(w_src);
    const int vo = lane_id() * 16;
#pragma unroll
    for (int k = 0; k < 3; ++k) dma16(rs_w, dst + (wq + NW * k) * 1024, vo, so + (wq + NW * k) * 1024);
    w_src = w_src + STAGE == H * NS * STAGE ? 0 : w_src + STAGE;
    w_slot = w_slot + STAGE == NST * STAGE ? 0 : w_slot + STAGE;
  };

  // ---- the wave's rows of the hidden state: k32 step kt, planes hi / lo -- B operand of the swapped form D^T = W h^T (q, k: a lane
  // owns a token) and A operand of the normal form (v: a lane owns a feature, which is what O^T = V^T P^T wants)
  f16x8 hh[NKT], hl[NKT];
  const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.himg), 0, p.himg_bytes, 0x00020000);
  auto load_h = [&](int r0) __attribute__((always_inline)) {  // rows beyond the image read as zeros
    const int ln = lane_id();
    const int row = r0 + 16 * rb + (ln & 15);
    const unsigned hoff = (unsigned)(((row >> 5) * (NKT * 256) + (row & 31)) * 16 + (ln >> 4) * 512);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      hh[kt] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_h, (int)hoff, kt * 8 * 512, 0));
      hl[kt] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_h, (int)hoff, (kt * 8 + 4) * 512, 0));
    }
  };

  const bool rec = PROF && blockIdx.x == 0 && p.stamps != nullptr;
  unsigned long long* stp = PROF ? p.stamps + (size_t)wq * 32 * 16 : nullptr;
  int slot = 0;
#define FD_STAMP(i) do { if (PROF) { if (rec && slot < 32 && lane == 0) stp[slot * 16 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
#define FD_SB() __builtin_amdgcn_sched_barrier(0)

  if ((int)blockIdx.x >= p.B) return;
  const int nseq = (p.B - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int N = nseq * H;  // items (= heads of sequences) of this workgroup
  // An ITEM is projected in two parts: its EARLY stages (0 .. NE-1) inside the previous item's attention region, its LATE stages
  // (NE .. NS-1) by all waves in step.  The region is where the two waves of a SIMD part ways: group 0 runs the attention of item i
  // and then the early stages of item i + 1, group 1 the other way round -- the attention's LDS round trips, skew and softmax
  // (VALU / latency bound, few MFMAs) sit beside the partner's projection steps (MFMA bound) instead of beside the partner's
  // softmax.  In step (everything else) both waves of a SIMD want the same unit at the same time.
  constexpr int NE = NS == 6 ? 2 : 1;
  // sequence of the item whose late stages / epilogue / attention run (a_*) and of the item after it (n_*: early stages)
  int n_seq = blockIdx.x, n_head = 0, n_row0, n_row1, n_Lb, n_len;
  sload4(p.seq_row0, n_seq, p.seq_row0, n_seq + 1, p.nrow, n_seq, p.lens, n_seq, n_row0, n_row1, n_Lb, n_len);
  int a_head = 0, a_row0 = 0, a_row1 = 0, a_Lb = 0, a_len = 0;
  load_h(n_row0);
  issue_w();
  issue_w();
  issue_w();
  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(p.ctx, 0, 0xFFFFFF00u, 0x00020000);
  int pos = 0;  // ring slot (byte offset) of stage 0 of the item whose stages are read
  f32x4 acc[6];  // q tile 0, 1 | k tile 0, 1 (swapped form: lane = token) | v tile 0, 1 (normal form: lane = feature)

  // ---- projection steps [KS0, KS1) of the item whose first stage sits in ring slot `pos`: a software pipeline over k32 steps -- two
  // fragment buffers of six tiles (48 registers) alternate over the planes hi(ks) lo(ks) hi(ks + 1) ...: while a plane multiplies
  // (hi: 12 MFMAs, lo: 6) the next one is requested into the other buffer, across stage boundaries too.  TOPS: the steps of this
  // range carry the stage tops of the late stages: the barrier that publishes stage j = steps 2 j, 2 j + 1 sits inside step 2 j - 1,
  // behind that step's lo-plane reads (the last reads of stage j - 1, whose ring slot the request at that top overwrites) and in
  // front of the reads of hi(2 j).  The two waves of a SIMD have the top at different places: group 0 in front of the step's twelve
  // hi-plane MFMAs, group 1 behind them (one wave's requests, ~210 cycles, beside the other's MFMAs).
  // Stage-top waits: vmcnt retires in issue order; when stage G is published the only younger requests are those of stage G + 1
  // (d_model 192, three stages per head: none).
  auto proj_steps = [&](auto KS0, auto KS1, auto TOPS, bool act) __attribute__((always_inline)) {
    constexpr int ks0 = decltype(KS0)::value, ks1 = decltype(KS1)::value;
    constexpr bool tops = decltype(TOPS)::value;
    auto stage_top = [&]() __attribute__((always_inline)) {
      if constexpr (NS == 6) FD_WAIT_VM(3);
      else FD_WAIT_VM(0);
      barrier_keep_vm();
      issue_w();
    };
    f16x8 fx[6], fy[6];
    unsigned a_W = smem0 + OFF_W + lane_off_of(lane_id());
    auto plane_reads = [&](auto KS, auto LO, f16x8 (&f)[6]) __attribute__((always_inline)) {
      constexpr int ks = decltype(KS)::value, lo = decltype(LO)::value;
      int off = pos + (ks >> 1) * STAGE;
      off = off >= NST * STAGE ? off - NST * STAGE : off;
      unsigned sb = a_W + (unsigned)(off + (ks & 1) * KS_BYTES + lo * 1024);
      asm volatile("" : "+v"(sb));
#pragma unroll
      for (int t = 0; t < 6; ++t) f[t] = lds_f16x8(sb + (unsigned)(t * 2048));
    };
    // per accumulator: wh hh | wh hl | wl hh (gemm_img.hip's order); consecutive MFMAs never share an accumulator
    auto mm_hh = [&](auto KS, const f16x8 (&fh)[6]) __attribute__((always_inline)) {
      constexpr int kt = decltype(KS)::value;
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mfma16(fh[t], hh[kt], acc[t]);
#pragma unroll
      for (int t = 4; t < 6; ++t) acc[t] = mfma16(hh[kt], fh[t], acc[t]);
    };
    auto mm_hl = [&](auto KS, const f16x8 (&fh)[6]) __attribute__((always_inline)) {
      constexpr int kt = decltype(KS)::value;
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mfma16(fh[t], hl[kt], acc[t]);
#pragma unroll
      for (int t = 4; t < 6; ++t) acc[t] = mfma16(hl[kt], fh[t], acc[t]);
    };
    auto mm_lh = [&](auto KS, const f16x8 (&fl)[6]) __attribute__((always_inline)) {
      constexpr int kt = decltype(KS)::value;
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mfma16(fl[t], hh[kt], acc[t]);
#pragma unroll
      for (int t = 4; t < 6; ++t) acc[t] = mfma16(hh[kt], fl[t], acc[t]);
    };
    // (activity tests around the pieces of a step make hipcc spill: a wave without rows takes a path of its own, stage tops only)
    if (act) {
      plane_reads(KS0, IC<0>{}, fx);
      FD_SB();
      static_for<ks0, ks1>([&](auto KS) __attribute__((always_inline)) {
        constexpr int ks = decltype(KS)::value;
        constexpr bool top = tops && (ks & 1) == 1 && ks + 1 < ks1;  // publishes the stage of steps ks + 1, ks + 2
        plane_reads(KS, IC<1>{}, fy);
        FD_SB();
        if constexpr (top) {
          if (grp == 0) stage_top();
        }
#if FDMI_S16_PRIO
        __builtin_amdgcn_s_setprio(FDMI_S16_PRIO);  // MFMAs of ONE wave back to back: alternating between the SIMD's two waves they issue slower
#endif
        mm_hh(KS, fx);
        mm_hl(KS, fx);
#if FDMI_S16_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        FD_SB();
        if constexpr (top) {
          if (grp != 0) stage_top();
        }
        if constexpr (ks + 1 < ks1) plane_reads(IC<ks + 1>{}, IC<0>{}, fx);
        FD_SB();
        mm_lh(KS, fy);
        FD_SB();
      });
    } else if constexpr (tops) {
      static_for<ks0, ks1>([&](auto KS) __attribute__((always_inline)) {
        constexpr int ks = decltype(KS)::value;
        if constexpr ((ks & 1) == 1 && ks + 1 < ks1) stage_top();
      });
    }
  };

  // ---- prologue: the early stages of item 0 by every wave, then the first "closing" barrier (see the loop)
  FD_WAIT_VM(3);  // the hidden state and stages 0, 1 landed (requested in this order; stage 2 may still be in flight)
  barrier_keep_vm();
#pragma unroll
  for (int t = 0; t < 6; ++t) acc[t] = zero4;
  bool n_act = 16 * rb < n_Lb;  // this wave owns rows of the item's sequence (wave-uniform)
  proj_steps(IC<0>{}, IC<2 * NE>{}, IC<0>{}, n_act);
  FD_WAIT_VM(0);
  barrier_keep_vm();  // stage NE landed for every wave; every wave is done with the early stages: their ring slots are free
#pragma unroll
  for (int k = 0; k < NE; ++k) issue_w();

  for (int item = 0; item < N; ++item) {
    // ---- the item's late stages, all waves in step
    FD_STAMP(0);
    const bool act = n_act;
    proj_steps(IC<2 * NE>{}, IC<NKT>{}, IC<1>{}, act);
    FD_STAMP(1);
    // this item becomes the attended one; the next item (if any) is the one whose early stages run in the region
    a_head = n_head; a_row0 = n_row0; a_row1 = n_row1; a_Lb = n_Lb; a_len = n_len;
    const bool has_next = item + 1 < N;
    if (has_next) {
      if (n_head == H - 1) {
        // the hidden state is dead: the next sequence's replaces it under the epilogue (waited for in front of the K / V barrier)
        n_head = 0;
        n_seq += (int)gridDim.x;
        sload4(p.seq_row0, n_seq, p.seq_row0, n_seq + 1, p.nrow, n_seq, p.lens, n_seq, n_row0, n_row1, n_Lb, n_len);
        load_h(n_row0);
        n_act = 16 * rb < n_Lb;
      } else {
        ++n_head;
      }
    }
    const int head = a_head, row0 = a_row0, Lb = a_Lb, len = a_len;
    const int nrows = a_row1 - a_row0;
    const int nkt = (Lb + 15) >> 4;         // key tiles that hold a key at all
    const bool upper = nkt > 4;             // the sequence has keys beyond 64: work is skipped in halves of the key range
    const bool active = act;
    f16x8 qh, ql;  // Q^T operand of the head: this lane's token, features 8 g .. 8 g + 7
    qh = ql = __builtin_bit_cast(f16x8, u32x4{0u, 0u, 0u, 0u});
    if (active) {
        // ================================================================ epilogue: q_h -> registers, k_h -> LDS, v_h -> LDS
        // (same arithmetic as gemm_img.hip's q | k and v^T epilogues: fma(acc, os sc, b sc), then the split)
        const int ln = lane_id();
        const int c = ln & 15, g = ln >> 4;
        const unsigned a_B = smem0 + OFF_B;
        unsigned bq = a_B + (unsigned)((head * 32 + 8 * g) * 4);
        unsigned aKw = smem0 + OFF_K + lane_off_of(ln) + (unsigned)(rb * 2048),
                 aVw = smem0 + OFF_V + lane_off_of(ln) + (unsigned)((rb >> 1) * 4096 + (rb & 1) * 8);
        float o[8];
        {
          const u32x4 b0 = lds_u128(bq), b1 = lds_u128(bq + 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = __builtin_fmaf(acc[0][e], oss_q, __builtin_bit_cast(float, (unsigned)b0[e]));
            o[4 + e] = __builtin_fmaf(acc[1][e], oss_q, __builtin_bit_cast(float, (unsigned)b1[e]));
          }
          u32x4 hv, lv;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            unsigned a, b;
            split_pair(o[2 * j], o[2 * j + 1], a, b);
            hv[j] = a;
            lv[j] = b;
          }
          qh = __builtin_bit_cast(f16x8, hv);
          ql = __builtin_bit_cast(f16x8, lv);
        }
        {
          const u32x4 b0 = lds_u128(bq + (unsigned)(D * 4)), b1 = lds_u128(bq + (unsigned)(D * 4 + 16));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = __builtin_fmaf(acc[2][e], oss_k, __builtin_bit_cast(float, (unsigned)b0[e]));
            o[4 + e] = __builtin_fmaf(acc[3][e], oss_k, __builtin_bit_cast(float, (unsigned)b1[e]));
          }
          u32x4 hv, lv;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            unsigned a, b;
            split_pair(o[2 * j], o[2 * j + 1], a, b);
            hv[j] = a;
            lv[j] = b;
          }
          // this lane's key is row c of key tile rb; unit g (hi) / 4 + g (lo)
          *(lds_u128_t)(unsigned long long)(aKw) = hv;
          *(lds_u128_t)(unsigned long long)(aKw + 1024u) = lv;
        }
        // v (normal form): lane c = feature 8 (c / 4) + 4 jv + (c % 4) of the head, registers e = tokens 4 g + e of this wave's 16:
        // in the 32-key step rb / 2 they are kappa = 8 g + 4 (rb & 1) + e: one 8-byte half of unit g (hi) / 4 + g (lo) of feature row c
#pragma unroll
        for (int jv = 0; jv < 2; ++jv) {
          const float bz = lds_f32(a_B + (unsigned)((2 * D + head * 32 + 8 * (c >> 2) + 4 * jv + (c & 3)) * 4));
          unsigned h0, l0, h1, l1;
          split_pair(__builtin_fmaf(acc[4 + jv][0], oss_v, bz), __builtin_fmaf(acc[4 + jv][1], oss_v, bz), h0, l0);
          split_pair(__builtin_fmaf(acc[4 + jv][2], oss_v, bz), __builtin_fmaf(acc[4 + jv][3], oss_v, bz), h1, l1);
          const unsigned va = aVw + (unsigned)(jv * 2048);
          *(lds_u64_t)(unsigned long long)(va) = u32x2{h0, h1};
          *(lds_u64_t)(unsigned long long)(va + 1024) = u32x2{l0, l1};
        }
    }
    FD_STAMP(2);
    // K and V^T of the item are complete; the early stages of the next item (requested at the last two stage tops) and, at a
    // sequence's end, the next hidden state landed: this wave's every vector-memory operation is behind it
    FD_WAIT_VM(0);
    barrier_keep_vm();
    issue_w();  // (the ring slot of the item's last stage is free: the next item's stage NE)
    pos = pos + NS * STAGE;
    pos = pos >= 2 * NST * STAGE ? pos - 2 * NST * STAGE : (pos >= NST * STAGE ? pos - NST * STAGE : pos);
    FD_STAMP(3);
    // ================================================================ the region
    auto attention = [&]() __attribute__((always_inline)) {
      if (active) {
        // ================================================================ attention of this wave's 16 queries
        // LDS bases of this phase as opaque values: every operand address below is one of them + an immediate offset of the instruction.
        // (Left to itself hipcc forms ~60 distinct addresses once, in front of the loops, and spills them.)
        const int ln = lane_id();
        const int c = ln & 15, g = ln >> 4;
        const unsigned a_R = smem0 + OFF_R + (unsigned)(wq * 2048);
        const unsigned aKv = smem0 + OFF_K + lane_off_of(ln), aVv = smem0 + OFF_V + lane_off_of(ln),
                       aEr = smem0 + OFF_E + lane_off_of(ln) + (unsigned)(rb * 2048), aRw = a_R + (unsigned)(ln * 4);
        // skew gather addresses (see the band below): score (query c, key 16 t + 4 g + e) of S^T tile t takes band row
        // j = c - (4 g + e) + 15 of the tile pair (t, t - 1): rows 0-15 are tile t (scratch slot t & 1), rows 16-30 tile t - 1 (the other
        // slot).  A band tile in the scratch: row rho = 4 gg + ee of query cc at float index ee * 64 + gg * 16 + cc (= 4 x lane + 256 ee:
        // the layout the C/D registers are written with).  ga[e]: the address for EVEN t; odd t: the slots are swapped, address ^ 1024
        unsigned ga[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int j = c - (4 * g + e) + 15, rho = j & 15;
          ga[e] = a_R + (unsigned)((j >= 16 ? 1024 : 0) + ((rho & 3) * 64 + (rho >> 2) * 16 + c) * 4);
        }
        // (every tile starts as zeros: tiles of a skipped half are never computed and must be zero probabilities for P V)
        f32x4 sacc[8], oacc[2], oacc2[2];
#pragma unroll
        for (int t = 0; t < 8; ++t) sacc[t] = zero4;
        oacc[0] = oacc[1] = oacc2[0] = oacc2[1] = zero4;
        float bvv[8][4];     // gathered band values of the S^T tiles (each lives from its gather to its add, one pipeline step later)

        // ---- S^T tile t = K_t Q^T: lane (query c, g) holds keys 16 t + 4 g + e.  kh qh | kh ql | kl qh; groups of two tiles, the next
        // group's fragments requested while this one multiplies (two buffers of 16 registers)
        {
          auto k_reads = [&](auto T0, f16x8 (&kh)[2], f16x8 (&kl)[2]) __attribute__((always_inline)) {
            constexpr int t0 = decltype(T0)::value;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              kh[t] = lds_f16x8(aKv + (unsigned)((t0 + t) * 2048));
              kl[t] = lds_f16x8(aKv + (unsigned)((t0 + t) * 2048 + 1024));
            }
          };
          auto s_mm = [&](auto T0, const f16x8 (&kh)[2], const f16x8 (&kl)[2]) __attribute__((always_inline)) {
            constexpr int t0 = decltype(T0)::value;
#pragma unroll
            for (int t = 0; t < 2; ++t) sacc[t0 + t] = mfma16(kh[t], qh, zero4);
#pragma unroll
            for (int t = 0; t < 2; ++t) sacc[t0 + t] = mfma16(kh[t], ql, sacc[t0 + t]);
#pragma unroll
            for (int t = 0; t < 2; ++t) sacc[t0 + t] = mfma16(kl[t], qh, sacc[t0 + t]);
          };
          f16x8 kha[2], kla[2], khb[2], klb[2];
          k_reads(IC<0>{}, kha, kla);
          FD_SB();
          k_reads(IC<2>{}, khb, klb);
          s_mm(IC<0>{}, kha, kla);
          FD_SB();
          if (upper) k_reads(IC<4>{}, kha, kla);
          s_mm(IC<2>{}, khb, klb);
          FD_SB();
          if (upper) {
            k_reads(IC<6>{}, khb, klb);
            s_mm(IC<4>{}, kha, kla);
            FD_SB();
            s_mm(IC<6>{}, khb, klb);
            FD_SB();
          }
        }
        FD_STAMP(9);
        // ---- relative_key (HF BertSelfAttention 4.11.3): S[l][r] += q_l . E[l - r + maxpos - 1].  Band tile u, u = -1 .. 7: R^T = E_u Q^T
        // for the 16 distances l - r = 16 (rb - u) - 15 + rho, rho = 0..15 (LDS table rows 16 (rb - u + 7) + rho: one 16-row group);
        // S^T tile t needs tiles t (band rows 0-15 of the pair) and t - 1 (rows 16-30).  eh qh | el qh | eh ql, the tiles of a group side
        // by side.  The tiles then go through the wave's scratch (slot u & 1) in C/D register order and come back skewed, one
        // ds_read_b32 per score, in the order write(u) [gather(u)] write(u + 1) gather(u + 1) ...: tile u + 1 replaces tile u - 1, which
        // gather(u) reads (one wave's LDS operations execute in order; nothing waits for a round trip before the values are added)
        auto b_reads = [&](auto U0, auto NT, f16x8* eh, f16x8* el) __attribute__((always_inline)) {
          constexpr int u0 = decltype(U0)::value, nt = decltype(NT)::value;
#pragma unroll
          for (int i = 0; i < nt; ++i) {
            const unsigned ea = aEr + (unsigned)((7 - (u0 + i)) * 2048);
            eh[i] = lds_f16x8(ea);
            el[i] = lds_f16x8(ea + 1024);
          }
        };
        auto b_mm = [&](auto NT, const f16x8* eh, const f16x8* el, f32x4* r) __attribute__((always_inline)) {
          constexpr int nt = decltype(NT)::value;
#pragma unroll
          for (int i = 0; i < nt; ++i) r[i] = mfma16(eh[i], qh, zero4);
#pragma unroll
          for (int i = 0; i < nt; ++i) r[i] = mfma16(el[i], qh, r[i]);
#pragma unroll
          for (int i = 0; i < nt; ++i) r[i] = mfma16(eh[i], ql, r[i]);
        };
        auto b_skew = [&](auto U0, auto NT, const f32x4* r) __attribute__((always_inline)) {
          constexpr int u0 = decltype(U0)::value, nt = decltype(NT)::value;
#pragma unroll
          for (int i = 0; i < nt; ++i) {
            const int u = u0 + i;
            const unsigned wa = aRw + (unsigned)((u & 1) * 1024);
#pragma unroll
            for (int e = 0; e < 4; ++e) *(lds_f32_t)(unsigned long long)(wa + (unsigned)(e * 256)) = r[i][e];
            if (u >= 0) {
#pragma unroll
              for (int e = 0; e < 4; ++e) bvv[u < 0 ? 0 : u][e] = lds_f32((u & 1) ? (ga[e] ^ 1024u) : ga[e]);
            }
          }
        };
        auto b_add = [&](auto T0, auto NT) __attribute__((always_inline)) {
          constexpr int t0 = decltype(T0)::value, nt = decltype(NT)::value;
#pragma unroll
          for (int t = t0; t < t0 + nt; ++t) {
#pragma unroll
            for (int e = 0; e < 4; ++e) sacc[t][e] = __builtin_fmaf(bvv[t][e], p.r_scale, sacc[t][e]);
          }
        };
        {
          // groups: (-1, 0) (1, 2) (3) | (4, 5) (6, 7); a group's fragments are requested while the group before multiplies, its tiles
          // pass the scratch while the next group multiplies, its gathered values are added a step later
          f16x8 eha[2], ela[2], ehb[2], elb[2];
          f32x4 ra[2], rbb[2];
          b_reads(IC<-1>{}, IC<2>{}, eha, ela);
          FD_SB();
          b_reads(IC<1>{}, IC<2>{}, ehb, elb);
          b_mm(IC<2>{}, eha, ela, ra);
          FD_SB();
          b_reads(IC<3>{}, IC<1>{}, eha, ela);
          b_mm(IC<2>{}, ehb, elb, rbb);
          b_skew(IC<-1>{}, IC<2>{}, ra);
          FD_SB();
          if (upper) b_reads(IC<4>{}, IC<2>{}, ehb, elb);
          b_mm(IC<1>{}, eha, ela, ra);
          b_skew(IC<1>{}, IC<2>{}, rbb);
          b_add(IC<0>{}, IC<1>{});
          FD_SB();
          b_skew(IC<3>{}, IC<1>{}, ra);
          b_add(IC<1>{}, IC<2>{});
          if (upper) {
            b_reads(IC<6>{}, IC<2>{}, eha, ela);
            b_mm(IC<2>{}, ehb, elb, rbb);
          }
          FD_SB();
          b_add(IC<3>{}, IC<1>{});
          if (upper) {
            b_mm(IC<2>{}, eha, ela, ra);
            b_skew(IC<4>{}, IC<2>{}, rbb);
            FD_SB();
            b_skew(IC<6>{}, IC<2>{}, ra);
            b_add(IC<4>{}, IC<2>{});
            FD_SB();
            b_add(IC<6>{}, IC<2>{});
          }
          FD_SB();
        }
        FD_STAMP(10);
        // ---- key mask (this lane + lanes c + 16 g' hold one query's scores), the row maximum, the exponentials, the row sum
        float mt = -INFINITY;
        {
          int g4 = 4 * g;
          asm volatile("" : "+v"(g4));
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            if (t < 4 || upper) {
              if (len < 16 * (t + 1)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  // key = 16 t + 4 g + e, compared as 4 g against a scalar (32 loop-invariant key indices would be hoisted and spilled)
                  float sc = sacc[t][e];
                  if (g4 >= len - 16 * t - e) sc += mask_raw;  // key >= len: (1 - mask) * -10000   (modelling.py:452)
                  if (g4 >= Lb - 16 * t - e) sc = -INFINITY;   // key >= Lb: not a key at all (rows that do not exist)
                  sacc[t][e] = sc;
                }
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) mt = fmaxf(mt, sacc[t][e]);
            }
          }
        }
        mt = quad_max(mt);
        const float nm = __builtin_fmaf(-mt, s_scale, 10.0f);  // + log2(PS): p' = PS * 2^((u - m) * s_scale)
        static_assert(PS == 1024.0f, "exponent offset above is log2(PS)");
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if (t < 4 || upper) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[t][e], s_scale, nm));
              sacc[t][e] = pe;
              psum += pe;
            }
          }
        }
        const float l_run = quad_sum(psum);  // carries the factor PS
        FD_STAMP(11);
        // ---- O^T = V^T P^T over 32-key steps: kappa = 8 g + 4 p + e <-> key 16 (2 s4 + p) + 4 g + e.  vh ph | vl ph | vh pl; even steps
        // accumulate into oacc, odd ones into oacc2 (four independent MFMA chains per round of two steps), added at the end
        auto pv_round = [&](auto S0) __attribute__((always_inline)) {  // steps s0, s0 + 1
          constexpr int s0 = decltype(S0)::value;
          f16x8 vh[2][2], vl[2][2], ph[2], pl[2];
#pragma unroll
          for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int jv = 0; jv < 2; ++jv) {
              const unsigned va = aVv + (unsigned)((s0 + k) * 4096 + jv * 2048);
              vh[k][jv] = lds_f16x8(va);
              vl[k][jv] = lds_f16x8(va + 1024);
            }
          }
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            u32x4 phu, plu;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              unsigned a, b;
              split_pair(sacc[2 * (s0 + k) + (j >> 1)][2 * (j & 1)], sacc[2 * (s0 + k) + (j >> 1)][2 * (j & 1) + 1], a, b);
              phu[j] = a;
              plu[j] = b;
            }
            ph[k] = __builtin_bit_cast(f16x8, phu);
            pl[k] = __builtin_bit_cast(f16x8, plu);
          }
          FD_SB();
#pragma unroll
          for (int jv = 0; jv < 2; ++jv) {
            oacc[jv] = mfma16(vh[0][jv], ph[0], oacc[jv]);
            oacc2[jv] = mfma16(vh[1][jv], ph[1], oacc2[jv]);
          }
#pragma unroll
          for (int jv = 0; jv < 2; ++jv) {
            oacc[jv] = mfma16(vl[0][jv], ph[0], oacc[jv]);
            oacc2[jv] = mfma16(vl[1][jv], ph[1], oacc2[jv]);
          }
#pragma unroll
          for (int jv = 0; jv < 2; ++jv) {
            oacc[jv] = mfma16(vh[0][jv], pl[0], oacc[jv]);
            oacc2[jv] = mfma16(vh[1][jv], pl[1], oacc2[jv]);
          }
          FD_SB();
        };
        pv_round(IC<0>{});
        if (upper) pv_round(IC<2>{});
        FD_STAMP(12);
        // ---- ctx[token row][head block] = O^T / l_run at the ctx image's scale: lane (query c, g) holds features 8 g + 4 jv + e = unit g
        {
          const float onorm = p.ctx_scale / (p.v_scale * l_run);
#pragma unroll
          for (int e = 0; e < 4; ++e) {  // (element by element: a vector add becomes v_pk_add_f32, which serializes with the partner's MFMAs)
            oacc[0][e] = oacc[0][e] + oacc2[0][e];
            oacc[1][e] = oacc[1][e] + oacc2[1][e];
          }
          u32x4 hv, lv;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            unsigned a, b;
            split_pair(oacc[j >> 1][2 * (j & 1)] * onorm, oacc[j >> 1][2 * (j & 1) + 1] * onorm, a, b);
            hv[j] = a;
            lv[j] = b;
          }
          const int l = 16 * rb + c, row = row0 + l;
          unsigned voff = (unsigned)((((row >> 5) * H * 8 + g) * 32 + (row & 31)) * 16);
          voff = l < nrows ? voff : 0xFFFFFF00u;  // rows that are no rows of the sequence: dropped by the range check
          __builtin_amdgcn_raw_buffer_store_b128(hv, rsc, (int)voff, head * 4096, 0);
          __builtin_amdgcn_raw_buffer_store_b128(lv, rsc, (int)voff, head * 4096 + 2048, 0);
          store_guard(hv, lv);
        }
      } else {
        // a wave without rows issues the same number of vector-memory operations per item (the waits count them)
        u32x4 z = {0u, 0u, 0u, 0u};
        __builtin_amdgcn_raw_buffer_store_b128(z, rsc, (int)0xFFFFFF00u, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(z, rsc, (int)0xFFFFFF00u, 0, 0);
        store_guard(z, z);
      }
    };
    auto early = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < 6; ++t) acc[t] = zero4;
      if (has_next) proj_steps(IC<0>{}, IC<2 * NE>{}, IC<0>{}, n_act);
    };
    if (grp == 0) {
      attention();
      FD_STAMP(4);
      early();
    } else {
      early();
      FD_STAMP(4);
      attention();
    }
    FD_STAMP(5);
    // the next item's stage NE landed (requested at the K / V barrier; only this wave's two ctx stores are younger); every wave is
    // done with the early stages: their ring slots take the stages after it
    FD_WAIT_VM(2);
    barrier_keep_vm();
#pragma unroll
    for (int k = 0; k < NE; ++k) issue_w();
    FD_STAMP(6);
    ++slot;
  }
  FD_WAIT_VM(0);  // nothing may land in LDS after the workgroup has exited
#undef FD_STAMP
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

template <int NKT>
static bool launch(const SeqAttnArgs& p, hipStream_t s) {
  static int attr_state[64] = {0};  // 0 unknown, 1 set, -1 refused
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  if (attr_state[dev] == 0) {
    bool ok = true;
    for (const void* f : {reinterpret_cast<const void*>(&seq_attn16_kernel<NKT, false>), reinterpret_cast<const void*>(&seq_attn16_kernel<NKT, true>)})
      ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) == hipSuccess;
    attr_state[dev] = ok ? 1 : -1;
  }
  if (attr_state[dev] < 0) return false;
  int grid = n_cu_of(dev);
  if (grid > p.B) grid = p.B;
  if (p.stamps) hipLaunchKernelGGL((seq_attn16_kernel<NKT, true>), dim3(grid), dim3(64 * NW), SMEM, s, p);
  else hipLaunchKernelGGL((seq_attn16_kernel<NKT, false>), dim3(grid), dim3(64 * NW), SMEM, s, p);
  return hipGetLastError() == hipSuccess;
}

}  // namespace s16

// head size 32, d_model 384 / 192 (every released configuration / the reference's test fixture), sequences of up to 128 rows, the
// distance table in 32 KiB of LDS
bool seq_attn16_supported(int d_model, int n_heads, int L, int maxpos) {
  return (d_model == 384 || d_model == 192) && n_heads * 32 == d_model && L >= 1 && L <= 128 && maxpos <= 128 && maxpos >= L;
}

bool launch_seq_attn16(const SeqAttnArgs& p, hipStream_t s) {
  return p.H == 12 ? s16::launch<12>(p, s) : s16::launch<6>(p, s);
}

}  // namespace fdmi
