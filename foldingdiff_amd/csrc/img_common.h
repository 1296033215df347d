// Device helpers shared by the "row image" kernels (gemm_img.hip, attention_img.hip, rowwise_img.hip).
//
// Row image: an activation (or weight) matrix X[R][K], K % 32 == 0, is kept in HBM already split for
// the fp16 matrix cores:   X * s = hi + lo   (s a power of two, hi = fp16(X s), lo = fp16(X s - hi)),
// as 128-byte blocks  [ hi(k0..k31) fp16 x32 | lo(k0..k31) fp16 x32 ]  per (row, 32 columns) -- 4 bytes per element,
// the same footprint as fp32, 22 significant bits.  A block is eight 16-byte units: 0-3 hi, 4-7 lo;
// unit u of the hi plane holds k = 8u .. 8u+7, which is exactly one lane's A/B operand slice of
// v_mfma_f32_32x32x16_f16 (k16 step c, half-wave h  ->  unit 2c + h).
// Consumers stage blocks with LDS-DMA (no VGPR round trip, no conversion); producers write them from
// the MFMA C/D layout with the helpers below.
//
// Activation images are stored GROUPED:  [row / 32][K / 32][unit 0..7][row % 32][16 B]  (img_unit_offset).  The producers'
// lanes own token ROWS (swapped MFMA form, one 16-byte unit per lane and store), so with the rows of a unit adjacent a
// store instruction writes 512 contiguous bytes per half-wave; with row-major blocks the same instruction scattered
// sixteen-byte pieces over 64 lines (same-box: 8.74 -> 8.25 ms per timestep, profiles/r02_gemm_ablation.log).  A consumer's
// LDS-DMA piece (8 rows x 8 units) still reads whole 128-byte lines: eight rows of one unit are contiguous.
// Weight images and the attention operands (q, k, v^T, distance table) keep their own layouts.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fdmi {

#ifndef FD_STORE_AUX
#define FD_STORE_AUX 16  // cache policy of the producers' 16-byte stores (buffer aux bits: 1 = sc0, 2 = nt, 16 = sc1)
#endif

#ifndef FDMI_EPI_DBG
#define FDMI_EPI_DBG 0  // ablation build (wrong results): 1 = the epilogues compute but do not store (profiles/r02_gemm_ablation.log)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

typedef __attribute__((address_space(3))) unsigned char* lds_ptr_t;

// s_waitcnt vmcnt(n) only (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] at [15:14])
__device__ __forceinline__ constexpr int waitcnt_vm_imm(int n) { return ((n >> 4) << 14) | 0x0F70 | (n & 15); }
#define FD_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(::fdmi::waitcnt_vm_imm(n))
// s_waitcnt lgkmcnt(0) only
#define FD_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)

// workgroup barrier that does NOT drain the vector-memory queue (LDS-DMA stays in flight across it);
// LDS accesses of this wave are complete first.
__device__ __forceinline__ void barrier_keep_vm() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// 16 bytes per lane global -> LDS, destination = wave-uniform `dst` + lane * 16
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, lds_ptr_t dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, voff, soff, 0, 0);
}

__device__ __forceinline__ unsigned pack_h2(_Float16 a, _Float16 b) {
  f16x2 v;
  v[0] = a;
  v[1] = b;
  return __builtin_bit_cast(unsigned, v);
}

__device__ __forceinline__ void swap32(unsigned& vdst, unsigned& src) {
  // lanes 32-63 of vdst <-> lanes 0-31 of src (v_permlane32_swap_b32)
#ifdef FDMI_NOSWAP  // ablation build (wrong results): what do the half-wave exchanges cost?
  asm volatile("" : "+v"(vdst), "+v"(src));
#else
  const auto r = __builtin_amdgcn_permlane32_swap(vdst, src, false, false);
  vdst = r[0];
  src = r[1];
#endif
}

// V^T image rows ([b][h][key block][d][128 B = sixteen 8-byte units]): unit u of feature row d sits at position u ^ vt_swz(d).
// The attention kernel fetches its P V operand from LDS with 8-byte reads, lane = d.  Which swizzle is conflict free depends on the
// INSTRUCTION hipcc picks (SQ_LDS_BANK_CONFLICT, both measured):
//   * a lone ds_read_b64 is served in 32-lane groups against 64 banks: the 16 even rows of a group share banks 0-31, the 16 odd
//     ones 32-63, so rows of one parity must name 16 different units -> (d >> 1) & 15.  This is what the kernel emits since round 4
//     (the per-tile liveness test keeps the fetches of two key tiles apart): 0 conflict cycles against 1.57 M per launch with d & 15
//     (profiles/r04_attention_lds_conflicts.log; the time is the same: the conflicts hid behind the other wave group);
//   * ds_read2st64_b64 (two key tiles' fetches merged, rounds 2-3) is served in 16-lane groups against 32 banks: 16 consecutive rows
//     must name 16 different units -> d & 15 (round 3: 3.1 M conflict cycles -> 0, profiles/r03_attention_notes.log).
#ifdef FDMI_VT_SWZ_R3  // A/B build: the round-3 swizzle (for merged ds_read2st64_b64 fetches)
__device__ __forceinline__ constexpr int vt_swz(int d) { return d & 15; }
#else
__device__ __forceinline__ constexpr int vt_swz(int d) { return (d >> 1) & 15; }
#endif

// byte offset of 16-byte unit u (0-3 hi, 4-7 lo) of column block cb of token row `row`; nb = blocks per row
__device__ __forceinline__ size_t img_unit_offset(long long row, int nb, int cb, int u) {
  return ((((size_t)(row >> 5) * nb + cb) * 8 + u) * 32 + (size_t)(row & 31)) * 16;
}

// MFMA 32x32 C/D layout along the register axis: register r = 4q + e of lane (l31, half) is element
// n = 8q + 4*half + e of a 32-wide block ("quad layout": four runs of 4).  The HBM block wants 16-byte
// units of 8 consecutive elements ("oct layout").  With the two half-waves exchanging half of their
// registers (4 v_permlane32_swap per plane) the lane pair ends up holding
//     lower lane (half 0): unit 0 (n 0-7),   unit 1 (n 8-15)
//     upper lane (half 1): unit 2 (n 16-23), unit 3 (n 24-31)
// so a block is written / read with 16-byte accesses only.  The permutation is an involution: the
// same swaps convert oct -> quad.
//   X[0..1] = quad 0, X[2..3] = quad 2, X[4..5] = quad 1, X[6..7] = quad 3   <-- quad side
//   X[0..3] = this lane's first unit, X[4..7] = its second unit              <-- oct side
__device__ __forceinline__ void quad_oct_exchange(unsigned (&X)[8]) {
  swap32(X[0], X[2]);
  swap32(X[1], X[3]);
  swap32(X[4], X[6]);
  swap32(X[5], X[7]);
}

// Behind a group of 16-byte buffer stores: keeps their data registers alive (and one wait state away from the next VALU write)
// until the stores have been issued.  A VALU write of a store's data VGPR in the very next instruction corrupted stored
// dwords on gfx950 (seen once with `v_xor` recomputing an address INTO a data register right behind buffer_store_dwordx4,
// profiles/r03_gemm_notes.log); hipcc only pads that hazard for stores without an SGPR offset.
__device__ __forceinline__ void store_guard(const u32x4& a, const u32x4& b) { asm volatile("s_nop 0" ::"v"(a), "v"(b)); }

// (x0, x1) -> hi = fp16 pair of (x0, x1), lo = fp16 pair of (x - hi).  v_fma_mixlo/mixhi_f16 take the fp32 x and the fp16 hi
// operand directly: x * 1.0 - hi is exact in fp32 and rounded once -- the same values as convert / subtract / convert, in
// 1.5 instead of 2.5 VALU instructions per element (VALU time adds to the matrix time on a SIMD: profiles/r02_probes.log)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
  const f32x2 xv = {x0, x1};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(xv, f16x2));
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(hi));
  asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(hi));
  lo = l;
}

// v[r] (quad layout, fp32) -> the lane's two hi units and two lo units of the block, split at scale s
__device__ __forceinline__ void pack_block(const float (&v)[16], float s, u32x4& h0, u32x4& h1, u32x4& l0, u32x4& l1) {
  unsigned H[8], Lo[8];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int slot = (q == 0) ? 0 : (q == 2) ? 2 : (q == 1) ? 4 : 6;
#pragma unroll
    for (int dd = 0; dd < 2; ++dd) {
      split_pair(v[4 * q + 2 * dd] * s, v[4 * q + 2 * dd + 1] * s, H[slot + dd], Lo[slot + dd]);
    }
  }
  quad_oct_exchange(H);
  quad_oct_exchange(Lo);
  h0 = u32x4{H[0], H[1], H[2], H[3]};
  h1 = u32x4{H[4], H[5], H[6], H[7]};
  l0 = u32x4{Lo[0], Lo[1], Lo[2], Lo[3]};
  l1 = u32x4{Lo[4], Lo[5], Lo[6], Lo[7]};
}

// Store v (quad layout) as block cb of token row `row` (this lane pair's row) of a GROUPED image.  Every lane takes part in
// the exchange; only lanes with `pred` store.
__device__ __forceinline__ void store_block_g(unsigned char* img, int nb, long long row, int cb, const float (&v)[16], float s, int half,
                                              bool pred) {
  u32x4 h0, h1, l0, l1;
  pack_block(v, s, h0, h1, l0, l1);
  if (FDMI_EPI_DBG == 1) {
    asm volatile("" ::"v"(h0), "v"(h1), "v"(l0), "v"(l1));
    return;
  }
  if (pred) {  // (plain stores: write-through ones made the q | k | v projection 13 % slower)
    unsigned char* u0 = img + img_unit_offset(row, nb, cb, 2 * half);
    *reinterpret_cast<u32x4*>(u0) = h0;
    *reinterpret_cast<u32x4*>(u0 + 512) = h1;
    *reinterpret_cast<u32x4*>(u0 + 4 * 512) = l0;
    *reinterpret_cast<u32x4*>(u0 + 5 * 512) = l1;
  }
}

// fp16 half of a packed pair times an fp32 factor plus an fp32 addend in ONE instruction (v_fma_mix_f32): exact product, one rounding
__device__ __forceinline__ float fma_mix_lo(unsigned hpair, float b, float c) {
  float d;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hpair), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float fma_mix_hi(unsigned hpair, float b, float c) {
  float d;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hpair), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float h2f_lo(unsigned w) { return (float)__builtin_bit_cast(f16x2, w)[0]; }
__device__ __forceinline__ float h2f_hi(unsigned w) { return (float)__builtin_bit_cast(f16x2, w)[1]; }

// raw = the lane's {hi unit 0, hi unit 1, lo unit 0, lo unit 1} of a block  ->  quad layout: v[r] = (hi + lo) * inv_s
__device__ __forceinline__ void unpack_block(const u32x4 (&raw)[4], float (&v)[16], float inv_s) {
  const u32x4 h0 = raw[0], h1 = raw[1], l0 = raw[2], l1 = raw[3];
  unsigned H[8] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
  unsigned Lo[8] = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
  quad_oct_exchange(H);
  quad_oct_exchange(Lo);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int slot = (q == 0) ? 0 : (q == 2) ? 2 : (q == 1) ? 4 : 6;
#pragma unroll
    for (int dd = 0; dd < 2; ++dd) {
      v[4 * q + 2 * dd] = (h2f_lo(H[slot + dd]) + h2f_lo(Lo[slot + dd])) * inv_s;
      v[4 * q + 2 * dd + 1] = (h2f_hi(H[slot + dd]) + h2f_hi(Lo[slot + dd])) * inv_s;
    }
  }
}

// The producers' case: the 32 rows of a wave's MFMA tile are one whole group.  blk0 = unit 0, row 0 of (group, block) --
// wave-uniform, so the address is an SGPR base plus the 32-bit lane offset (l31 * 16 + unit * 512): no 64-bit address
// pair per block in VGPRs.
__device__ __forceinline__ void store_group_block(unsigned char* blk0, const float (&v)[16], float s, int l31, int half) {
  u32x4 h0, h1, l0, l1;
  pack_block(v, s, h0, h1, l0, l1);
  if (FDMI_EPI_DBG == 1) {
    asm volatile("" ::"v"(h0), "v"(h1), "v"(l0), "v"(l1));
    return;
  }
  const unsigned off = (unsigned)(l31 * 16 + half * 1024);
  // write-through stores (sc1): nothing is left dirty in L2 for the end of the kernel, and the lines do not displace the
  // weight tile that every workgroup keeps re-reading (FFN-up -3 %, head GEMM -4 % against plain stores, same box)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(blk0, 0, 4096, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(h0, rs, (int)off, 0, FD_STORE_AUX);
  __builtin_amdgcn_raw_buffer_store_b128(h1, rs, (int)off + 512, 0, FD_STORE_AUX);
  __builtin_amdgcn_raw_buffer_store_b128(l0, rs, (int)off + 2048, 0, FD_STORE_AUX);
  __builtin_amdgcn_raw_buffer_store_b128(l1, rs, (int)off + 2560, 0, FD_STORE_AUX);
  store_guard(h0, h1);
  store_guard(l0, l1);
}
__device__ __forceinline__ void load_group_block_raw(const unsigned char* blk0, u32x4 (&raw)[4], int l31, int half) {
  const unsigned off = (unsigned)(l31 * 16 + half * 1024);
  raw[0] = *reinterpret_cast<const u32x4*>(blk0 + (size_t)off);
  raw[1] = *reinterpret_cast<const u32x4*>(blk0 + (size_t)(off + 512));
  raw[2] = *reinterpret_cast<const u32x4*>(blk0 + (size_t)(off + 2048));
  raw[3] = *reinterpret_cast<const u32x4*>(blk0 + (size_t)(off + 2560));
}

// ---- exact-erf GELU of the GEMM epilogues (HF "gelu", modelling.py:195-196 / BertIntermediate)
__device__ __forceinline__ float erf_rational(float x) {  // (13,8) rational minimax on [-4,4], |err| <= 4.5e-7 (tests/test_host.py)
  x = __builtin_fminf(__builtin_fmaxf(x, -4.0f), 4.0f);
  const float x2 = x * x;
  float p = -2.72614225801306e-10f;
  p = __builtin_fmaf(p, x2, 2.77068142495902e-08f);
  p = __builtin_fmaf(p, x2, -2.10102402082508e-06f);
  p = __builtin_fmaf(p, x2, -5.69250639462346e-05f);
  p = __builtin_fmaf(p, x2, -7.34990630326855e-04f);
  p = __builtin_fmaf(p, x2, -2.95459980854025e-03f);
  p = __builtin_fmaf(p, x2, -1.60960333262415e-02f);
  float q = -1.45660718464996e-05f;
  q = __builtin_fmaf(q, x2, -2.13374055278905e-04f);
  q = __builtin_fmaf(q, x2, -1.68282697438203e-03f);
  q = __builtin_fmaf(q, x2, -7.37332916720468e-03f);
  q = __builtin_fmaf(q, x2, -1.42647390514189e-02f);
  return (p * x) * __builtin_amdgcn_rcpf(q);
}
__device__ __forceinline__ float gelu_erf(float x) {  // HF "gelu": 0.5 x (1 + erf(x / sqrt 2))
  const float h = 0.5f * x;
  return __builtin_fmaf(h, erf_rational(x * 0.70710678118654752440f), h);
}

// the same arithmetic on element pairs (v_pk_fma_f32 / v_pk_mul_f32 for every step but the clamp and the reciprocal): the same
// operations in the same order per element, hence the same bits.  Only for epilogues that run with the matrix pipe idle
// (gemm_img.hip): packed fp32 instructions serialize with another wave's MFMAs (profiles/r03_coissue2_probe.log)
typedef float gf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gf2 gelu_erf2(gf2 x) {
  const gf2 h = x * 0.5f;
  gf2 y = x * 0.70710678118654752440f;
  y = __builtin_elementwise_min(__builtin_elementwise_max(y, gf2{-4.0f, -4.0f}), gf2{4.0f, 4.0f});
  const gf2 y2 = y * y;
  auto c = [](float v) { return gf2{v, v}; };
  gf2 p = c(-2.72614225801306e-10f);
  p = __builtin_elementwise_fma(p, y2, c(2.77068142495902e-08f));
  p = __builtin_elementwise_fma(p, y2, c(-2.10102402082508e-06f));
  p = __builtin_elementwise_fma(p, y2, c(-5.69250639462346e-05f));
  p = __builtin_elementwise_fma(p, y2, c(-7.34990630326855e-04f));
  p = __builtin_elementwise_fma(p, y2, c(-2.95459980854025e-03f));
  p = __builtin_elementwise_fma(p, y2, c(-1.60960333262415e-02f));
  gf2 q = c(-1.45660718464996e-05f);
  q = __builtin_elementwise_fma(q, y2, c(-2.13374055278905e-04f));
  q = __builtin_elementwise_fma(q, y2, c(-1.68282697438203e-03f));
  q = __builtin_elementwise_fma(q, y2, c(-7.37332916720468e-03f));
  q = __builtin_elementwise_fma(q, y2, c(-1.42647390514189e-02f));
  const gf2 r = {__builtin_amdgcn_rcpf(q[0]), __builtin_amdgcn_rcpf(q[1])};
  const gf2 e = (p * y) * r;
  return __builtin_elementwise_fma(h, e, h);
}
// four elements at a time, result already multiplied by the output image's scale (hs = 0.5 * scale, a power of two: the same bits
// as scale * gelu).  Two independent element pairs per step: consecutive packed instructions do not depend on each other, so the
// one wait state a dependent v_pk_* needs costs no s_nop (39 of them per 32 x 32 block in the pair form, ISA listing).
typedef float gf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ gf4 gelu_erf4_scaled(gf4 x, float hs) {
  const gf4 h = x * hs;
  gf4 y = x * 0.70710678118654752440f;
  auto c = [](float v) { return gf4{v, v, v, v}; };
  y = __builtin_elementwise_min(__builtin_elementwise_max(y, c(-4.0f)), c(4.0f));
  const gf4 y2 = y * y;
  gf4 p = c(-2.72614225801306e-10f);
  p = __builtin_elementwise_fma(p, y2, c(2.77068142495902e-08f));
  p = __builtin_elementwise_fma(p, y2, c(-2.10102402082508e-06f));
  p = __builtin_elementwise_fma(p, y2, c(-5.69250639462346e-05f));
  p = __builtin_elementwise_fma(p, y2, c(-7.34990630326855e-04f));
  p = __builtin_elementwise_fma(p, y2, c(-2.95459980854025e-03f));
  p = __builtin_elementwise_fma(p, y2, c(-1.60960333262415e-02f));
  gf4 q = c(-1.45660718464996e-05f);
  q = __builtin_elementwise_fma(q, y2, c(-2.13374055278905e-04f));
  q = __builtin_elementwise_fma(q, y2, c(-1.68282697438203e-03f));
  q = __builtin_elementwise_fma(q, y2, c(-7.37332916720468e-03f));
  q = __builtin_elementwise_fma(q, y2, c(-1.42647390514189e-02f));
  const gf4 r = {__builtin_amdgcn_rcpf(q[0]), __builtin_amdgcn_rcpf(q[1]), __builtin_amdgcn_rcpf(q[2]), __builtin_amdgcn_rcpf(q[3])};
  const gf4 e = (p * y) * r;
  return __builtin_elementwise_fma(h, e, h);
}
}  // namespace fdmi
