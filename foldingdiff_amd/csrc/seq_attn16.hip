// BertSelfAttention of a whole sequence in ONE kernel, two waves per SIMD: the q | k | v projection (HF 4.11.3
// BertSelfAttention.query / key / value, constructed at foldingdiff/modelling.py:271, called :473-480) AND the attention with the
// relative_key term, the additive -10000 key mask and the softmax -- q, k and v never reach HBM.
//
// Round 6.  seq_attn.hip runs one 512-register wave per SIMD (a wave owns 32 token rows; its slice of the hidden
// state alone is 192 registers), and one in-order wave cannot overlap its own vector / LDS work with its own matrix instructions:
// 45 % of the matrix pipe.  Here a wave owns SIXTEEN token rows and every contraction runs on v_mfma_f32_16x16x32_f16:
//   * 8 waves per workgroup = 2 per SIMD, 256 registers each: hidden state 96 (12 k32 steps x (hi, lo) x 4), projection
//     accumulators 24 (q_h | k_h | v_h: six 16 x 16 tiles), weight fragments 48, the attention's state ~100;
//   * the two waves of a SIMD fill each other's issue gaps (LDS round trips, dependent VALU chains, the softmax), which is what the
//     one-wave kernel had to do by hand with a slot schedule;
//   * no half-wave exchanges at all: the weight rows of a head are PERMUTED in the weight image (tile j, row i = head feature
//     8 (i / 4) + 4 j + (i % 4)), so that the 16 x 16 C/D layout (lane (c, g) holds rows 4 g .. 4 g + 3 of column c) hands every
//     lane the eight consecutive features 8 g .. 8 g + 7 of its token -- exactly one 16-byte operand unit of the next contraction
//     (Q^T as B operand, K rows as A operand, the ctx image's units); V^T and P^T share a permuted key order inside a 32-key step
//     (kappa = 8 g + 4 p + e  <->  key 16 p + 4 g + e) for the same reason.
// Arithmetic: the fp16 hi / lo split triples of gemm_img.hip / attention_img.hip (a product = hi hi + hi lo + lo hi, fp32
// accumulate), in the same order per accumulator; the MFMA's K is 32 instead of 16, so sums are associated differently:
// fp32-class results, NOT the bits of the two-kernel path (the oracle gates of tests/test_gpu_parity.py are the contract).
//
// Structure.  One workgroup per CU, persistent over sequences of <= 128 rows; wave w owns token rows 16 w .. 16 w + 15.
//   per (sequence, head):
//     projection   NKT k32 steps; the head's 96 weight rows stream through a 3-slot LDS ring in stages of two k32 steps (24 KiB,
//                  LDS-DMA, three 1 KiB pieces per wave, ONE workgroup barrier per stage, counted s_waitcnt vmcnt: the stream
//                  never drains); 36 MFMAs per wave and stage
//     epilogue     bias, scale, hi / lo split: q_h -> this wave's operand registers, k_h -> LDS (A operand tiles), v_h -> LDS (V^T)
//     barrier      (K and V^T of all eight waves are in place)
//     attention    S^T = K Q^T (8 key tiles), the relative_key band R^T = E Q^T (9 band tiles of 16 distances, skewed through a 2 KiB
//                  per-wave LDS scratch), mask, softmax in the log2 domain (row statistics: in-lane + two lane-group swaps),
//                  O^T = V^T P^T, ctx block -> HBM (two 16-byte stores per lane)
//   Key tiles / waves beyond the sequence's rows are skipped (wave-uniform branches: a second wave covers the bubbles), so short
//   sequences (BASELINE C3, packed rows) cost what their rows cost.
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
      : "s"(a), "s"(ia * 4), "s"(b), "s"(ib * 4), "s"(c), "s"(ic * 4), "s"(d), "s"(id * 4)
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
  const int c = lane & 15, g = lane >> 4;
  unsigned smem0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  asm volatile("" : "+s"(smem0));
  const unsigned lane_off = (unsigned)(g * 256 + c * 16);  // unit g (hi; + 1024: lo unit 4 + g), row c of a 2 KiB operand tile
  const unsigned a_E = smem0 + OFF_E + lane_off, a_K = smem0 + OFF_K + lane_off, a_V = smem0 + OFF_V + lane_off,
                 a_W = smem0 + OFF_W + lane_off, a_R = smem0 + OFF_R + (unsigned)(wq * 2048), a_B = smem0 + OFF_B;

  // ---- once per workgroup: bias at the scale of the image its column feeds ((acc os + b) sc == fma(acc, os sc, b sc) exactly for the
  // power-of-two sc), the distance table (LDS row e holds table row clamp(e - esh): e = l - r + 127), zeros in V^T (a skipped key
  // tile's probabilities are exact zeros, and 0 x whatever the LDS held at power-on must not be NaN)
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

  // skew gather addresses (see the band below): score (query c, key 16 t + 4 g + e) of S^T tile t takes band row j = c - (4 g + e) + 15
  // of the tile pair (t, t - 1): rows 0-15 are tile t (scratch slot t & 1), rows 16-30 tile t - 1 (the other slot).  A band tile in the
  // scratch: row rho = 4 gg + ee of query cc at float index ee * 64 + gg * 16 + cc (= 4 x lane + 256 ee: the layout the C/D
  // registers are written with).  gad[e]: the address for EVEN t; odd t: the slots are swapped, address ^ 1024
  unsigned gad[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = c - (4 * g + e) + 15, rho = j & 15;
    gad[e] = a_R + (unsigned)((j >= 16 ? 1024 : 0) + ((rho & 3) * 64 + (rho >> 2) * 16 + c) * 4);
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
#pragma unroll
    for (int k = 0; k < 3; ++k) dma16(rs_w, dst + (wq + NW * k) * 1024, lane * 16, so + (wq + NW * k) * 1024);
    w_src = w_src + STAGE == H * NS * STAGE ? 0 : w_src + STAGE;
    w_slot = w_slot + STAGE == NST * STAGE ? 0 : w_slot + STAGE;
  };

  // ---- the wave's rows of the hidden state: k32 step kt, planes hi / lo -- B operand of the swapped form D^T = W h^T (q, k: a lane
  // owns a token) and A operand of the normal form (v: a lane owns a feature, which is what O^T = V^T P^T wants)
  f16x8 hh[NKT], hl[NKT];
  const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.himg), 0, p.himg_bytes, 0x00020000);
  auto load_h = [&](int r0) __attribute__((always_inline)) {  // rows beyond the image read as zeros
    const int row = r0 + 16 * wq + c;
    const unsigned hoff = (unsigned)(((row >> 5) * (NKT * 256) + (row & 31)) * 16 + g * 512);
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

  int seq = blockIdx.x;
  if (seq >= p.B) return;
  int row0, row1, Lb, len;  // the sequence's first row, the next sequence's, rows that exist (keys), unmasked keys
  sload4(p.seq_row0, seq, p.seq_row0, seq + 1, p.nrow, seq, p.lens, seq, row0, row1, Lb, len);
  load_h(row0);
  issue_w();
  issue_w();
  FD_WAIT_VM(3);  // stage 0 (and the hidden state, requested before it) landed
  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(p.ctx, 0, 0xFFFFFF00u, 0x00020000);
  int pos = 0;  // ring slot of the stage being computed (byte offset)

  for (;;) {
    const int nrows = row1 - row0;
    const int nkt = (Lb + 15) >> 4;         // key tiles that hold a key at all
    const bool active = 16 * wq < Lb;       // this wave owns rows of the sequence (wave-uniform)
    const int next_seq = seq + (int)gridDim.x;
    for (int head = 0; head < H; ++head) {
      // ================================================================ projection of head `head`
      FD_STAMP(0);
      f32x4 acc[6];  // q tile 0, 1 | k tile 0, 1 (swapped form: lane = token) | v tile 0, 1 (normal form: lane = feature)
#pragma unroll
      for (int t = 0; t < 6; ++t) acc[t] = zero4;
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        // this stage landed (its three pieces of this wave; every wave says so at the barrier).  vmcnt retires in issue order: at
        // most the pieces of the stage after this one may be outstanding -- and, at the first two tops of a head, the two ctx stores
        // of the head before, which are younger than this stage's pieces
        if (st < 2) FD_WAIT_VM(5);
        else FD_WAIT_VM(3);
        barrier_keep_vm();  // ... for every wave; every wave is done with the stage before: its slot is free
        issue_w();
        if (active) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const int kt = 2 * st + ks;
            const unsigned wb = a_W + (unsigned)pos + (unsigned)(ks * KS_BYTES);
            f16x8 wh[6], wl[6];
#pragma unroll
            for (int t = 0; t < 6; ++t) wh[t] = lds_f16x8(wb + (unsigned)(t * 2048));
#pragma unroll
            for (int t = 0; t < 6; ++t) wl[t] = lds_f16x8(wb + (unsigned)(t * 2048 + 1024));
            // per accumulator: wh hh | wh hl | wl hh (gemm_img.hip's order); consecutive MFMAs never share an accumulator
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma16(wh[t], hh[kt], acc[t]);
#pragma unroll
            for (int t = 4; t < 6; ++t) acc[t] = mfma16(hh[kt], wh[t], acc[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma16(wh[t], hl[kt], acc[t]);
#pragma unroll
            for (int t = 4; t < 6; ++t) acc[t] = mfma16(hl[kt], wh[t], acc[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma16(wl[t], hh[kt], acc[t]);
#pragma unroll
            for (int t = 4; t < 6; ++t) acc[t] = mfma16(hh[kt], wl[t], acc[t]);
          }
        }
        pos = pos + STAGE == NST * STAGE ? 0 : pos + STAGE;
        FD_STAMP(1 + st);
      }
      // the hidden state is dead after the sequence's last projection: the next sequence's replaces it while this head's epilogue and
      // attention run (the first row alone is read here; the other parameters of that sequence at the loop's end)
      if (head == H - 1 && next_seq < p.B) {
        int nr0, d0, d1, d2;
        sload4(p.seq_row0, next_seq, p.seq_row0, next_seq, p.seq_row0, next_seq, p.seq_row0, next_seq, nr0, d0, d1, d2);
        load_h(nr0);
      }

      f16x8 qh, ql;  // Q^T operand of the head: this lane's token, features 8 g .. 8 g + 7
      if (active) {
        // ================================================================ epilogue: q_h -> registers, k_h -> LDS, v_h -> LDS
        // (same arithmetic as gemm_img.hip's q | k and v^T epilogues: fma(acc, os sc, b sc), then the split)
        const unsigned bq = a_B + (unsigned)((head * 32 + 8 * g) * 4);
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
          // this lane's key is row c of key tile wq; unit g (hi) / 4 + g (lo)
          *(lds_u128_t)(unsigned long long)(a_K + (unsigned)(wq * 2048)) = hv;
          *(lds_u128_t)(unsigned long long)(a_K + (unsigned)(wq * 2048 + 1024)) = lv;
        }
        // v (normal form): lane c = feature 8 (c / 4) + 4 jv + (c % 4) of the head, registers e = tokens 4 g + e of this wave's 16:
        // in the 32-key step wq / 2 they are kappa = 8 g + 4 (wq & 1) + e: one 8-byte half of unit g (hi) / 4 + g (lo) of feature row c
#pragma unroll
        for (int jv = 0; jv < 2; ++jv) {
          const float bz = lds_f32(a_B + (unsigned)((2 * D + head * 32 + 8 * (c >> 2) + 4 * jv + (c & 3)) * 4));
          unsigned h0, l0, h1, l1;
          split_pair(__builtin_fmaf(acc[4 + jv][0], oss_v, bz), __builtin_fmaf(acc[4 + jv][1], oss_v, bz), h0, l0);
          split_pair(__builtin_fmaf(acc[4 + jv][2], oss_v, bz), __builtin_fmaf(acc[4 + jv][3], oss_v, bz), h1, l1);
          const unsigned va = a_V + (unsigned)((wq >> 1) * 4096 + jv * 2048 + (wq & 1) * 8);
          *(lds_u64_t)(unsigned long long)(va) = u32x2{h0, h1};
          *(lds_u64_t)(unsigned long long)(va + 1024) = u32x2{l0, l1};
        }
      }
      FD_STAMP(7);
      barrier_keep_vm();  // K and V^T of the head are complete
      FD_STAMP(8);

      if (active) {
        // ================================================================ attention of this wave's 16 queries
        // S^T tile t = K_t Q^T: lane (query c, g) holds keys 16 t + 4 g + e.  kh qh | kh ql | kl qh
        f32x4 sacc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if (t < nkt) {
            const f16x8 kh = lds_f16x8(a_K + (unsigned)(t * 2048)), kl = lds_f16x8(a_K + (unsigned)(t * 2048 + 1024));
            f32x4 s = mfma16(kh, qh, zero4);
            s = mfma16(kh, ql, s);
            sacc[t] = mfma16(kl, qh, s);
          }
        }
        FD_STAMP(9);
        // relative_key (HF BertSelfAttention 4.11.3): S[l][r] += q_l . E[l - r + maxpos - 1].  Band tile u, u = -1 .. 7: R^T = E_u Q^T
        // for the 16 distances l - r = 16 (wq - u) - 15 + rho, rho = 0..15 (LDS table rows 16 (wq - u + 7) + rho: one 16-row group);
        // S^T tile t needs tiles t (band rows 0-15 of the pair) and t - 1 (rows 16-30).  The tiles go through the wave's scratch
        // (slot u & 1) in C/D register order and come back skewed, one ds_read_b32 and one fma per score.  eh qh | el qh | eh ql
#pragma unroll
        for (int u = -1; u < 8; ++u) {
          if (u < nkt) {
            const unsigned ea = a_E + (unsigned)((wq - u + 7) * 2048);
            const f16x8 eh = lds_f16x8(ea), el = lds_f16x8(ea + 1024);
            f32x4 r = mfma16(eh, qh, zero4);
            r = mfma16(el, qh, r);
            r = mfma16(eh, ql, r);
            const unsigned wa = a_R + (unsigned)((u & 1) * 1024 + lane * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) *(lds_f32_t)(unsigned long long)(wa + (unsigned)(e * 256)) = r[e];
            if (u >= 0) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float bv = lds_f32((u & 1) ? (gad[e] ^ 1024u) : gad[e]);
                sacc[u][e] = __builtin_fmaf(bv, p.r_scale, sacc[u][e]);
              }
            }
          }
        }
        FD_STAMP(10);
        // key mask (this lane + lanes c + 16 g' hold one query's scores) and the row maximum
        float mt = -INFINITY;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if (t < nkt) {
            if (len < 16 * (t + 1)) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int key = 16 * t + 4 * g + e;
                float sc = sacc[t][e];
                if (key >= len) sc += mask_raw;  // (1 - mask) * -10000   (modelling.py:452)
                if (key >= Lb) sc = -INFINITY;   // not a key at all (rows that do not exist)
                sacc[t][e] = sc;
              }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) mt = fmaxf(mt, sacc[t][e]);
          }
        }
        mt = quad_max(mt);
        const float nm = __builtin_fmaf(-mt, s_scale, 10.0f);  // + log2(PS): p' = PS * 2^((u - m) * s_scale)
        static_assert(PS == 1024.0f, "exponent offset above is log2(PS)");
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if (t < nkt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[t][e], s_scale, nm));
              sacc[t][e] = pe;
              psum += pe;
            }
          } else {
            sacc[t] = zero4;
          }
        }
        const float l_run = quad_sum(psum);  // carries the factor PS
        FD_STAMP(11);
        // O^T = V^T P^T over 32-key steps: kappa = 8 g + 4 p + e <-> key 16 (2 s4 + p) + 4 g + e.  vh ph | vl ph | vh pl
        f32x4 oacc[2] = {zero4, zero4};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          if (2 * s4 < nkt) {
            u32x4 phu, plu;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              unsigned a, b;
              split_pair(sacc[2 * s4 + (j >> 1)][2 * (j & 1)], sacc[2 * s4 + (j >> 1)][2 * (j & 1) + 1], a, b);
              phu[j] = a;
              plu[j] = b;
            }
            const f16x8 ph = __builtin_bit_cast(f16x8, phu), pl = __builtin_bit_cast(f16x8, plu);
#pragma unroll
            for (int jv = 0; jv < 2; ++jv) {
              const unsigned va = a_V + (unsigned)(s4 * 4096 + jv * 2048);
              const f16x8 vh = lds_f16x8(va), vl = lds_f16x8(va + 1024);
              f32x4 o2 = mfma16(vh, ph, oacc[jv]);
              o2 = mfma16(vl, ph, o2);
              oacc[jv] = mfma16(vh, pl, o2);
            }
          }
        }
        FD_STAMP(12);
        // ctx[token row][head block] = O^T / l_run at the ctx image's scale: lane (query c, g) holds features 8 g + 4 jv + e = unit g
        {
          const float onorm = p.ctx_scale / (p.v_scale * l_run);
          u32x4 hv, lv;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            unsigned a, b;
            split_pair(oacc[j >> 1][2 * (j & 1)] * onorm, oacc[j >> 1][2 * (j & 1) + 1] * onorm, a, b);
            hv[j] = a;
            lv[j] = b;
          }
          const int l = 16 * wq + c, row = row0 + l;
          unsigned voff = (unsigned)((((row >> 5) * H * 8 + g) * 32 + (row & 31)) * 16);
          voff = l < nrows ? voff : 0xFFFFFF00u;  // rows that are no rows of the sequence: dropped by the range check
          __builtin_amdgcn_raw_buffer_store_b128(hv, rsc, (int)voff, head * 4096, 0);
          __builtin_amdgcn_raw_buffer_store_b128(lv, rsc, (int)voff, head * 4096 + 2048, 0);
          store_guard(hv, lv);
        }
      } else {
        // a wave without rows issues the same number of vector-memory operations per head (the stage-top waits count them)
        u32x4 z = {0u, 0u, 0u, 0u};
        __builtin_amdgcn_raw_buffer_store_b128(z, rsc, (int)0xFFFFFF00u, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(z, rsc, (int)0xFFFFFF00u, 0, 0);
        store_guard(z, z);
      }
      FD_STAMP(13);
      ++slot;
    }
    if (next_seq >= p.B) break;
    seq = next_seq;
    sload4(p.seq_row0, seq, p.seq_row0, seq + 1, p.nrow, seq, p.lens, seq, row0, row1, Lb, len);
  }
  FD_WAIT_VM(0);  // nothing may land in LDS after the workgroup has exited
#undef FD_STAMP
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
