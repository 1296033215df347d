// Device helpers shared by the "row image" kernels (gemm_img.hip, attention_img.hip, rowwise_img.hip).
//
// Row image: an activation (or weight) matrix X[R][K], K % 32 == 0, is kept in HBM already split for
// the fp16 matrix cores:   X * s = hi + lo   (s a power of two, hi = fp16(X s), lo = fp16(X s - hi)),
// as 128-byte blocks  [row][K/32][ hi(k0..k31) fp16 x32 | lo(k0..k31) fp16 x32 ]  -- 4 bytes per element,
// the same footprint as fp32, 22 significant bits.  A block is eight 16-byte units: 0-3 hi, 4-7 lo;
// unit u of the hi plane holds k = 8u .. 8u+7, which is exactly one lane's A/B operand slice of
// v_mfma_f32_32x32x16_f16 (k16 step c, half-wave h  ->  unit 2c + h).
// Consumers stage blocks with LDS-DMA (no VGPR round trip, no conversion); producers write them from
// the MFMA C/D layout with the helpers below.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fdmi {

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
  const auto r = __builtin_amdgcn_permlane32_swap(vdst, src, false, false);
  vdst = r[0];
  src = r[1];
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

// v[r] (quad layout, fp32) -> the lane's two hi units and two lo units of the block, split at scale s
__device__ __forceinline__ void pack_block(const float (&v)[16], float s, u32x4& h0, u32x4& h1, u32x4& l0, u32x4& l1) {
  unsigned H[8], Lo[8];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int slot = (q == 0) ? 0 : (q == 2) ? 2 : (q == 1) ? 4 : 6;
#pragma unroll
    for (int dd = 0; dd < 2; ++dd) {
      const float x0 = v[4 * q + 2 * dd] * s, x1 = v[4 * q + 2 * dd + 1] * s;
      const _Float16 a0 = (_Float16)x0, a1 = (_Float16)x1;
      const _Float16 b0 = (_Float16)(x0 - (float)a0), b1 = (_Float16)(x1 - (float)a1);
      H[slot + dd] = pack_h2(a0, a1);
      Lo[slot + dd] = pack_h2(b0, b1);
    }
  }
  quad_oct_exchange(H);
  quad_oct_exchange(Lo);
  h0 = u32x4{H[0], H[1], H[2], H[3]};
  h1 = u32x4{H[4], H[5], H[6], H[7]};
  l0 = u32x4{Lo[0], Lo[1], Lo[2], Lo[3]};
  l1 = u32x4{Lo[4], Lo[5], Lo[6], Lo[7]};
}

// Store v (quad layout) as one 128-byte block at `blk` (this lane pair's row).  Every lane takes part in
// the exchange; only lanes with `pred` store.
__device__ __forceinline__ void store_block(unsigned char* blk, const float (&v)[16], float s, int half, bool pred) {
  u32x4 h0, h1, l0, l1;
  pack_block(v, s, h0, h1, l0, l1);
  if (pred) {
    u32x4* u = reinterpret_cast<u32x4*>(blk) + 2 * half;
    u[0] = h0;
    u[1] = h1;
    u[4] = l0;
    u[5] = l1;
  }
}

// The same with the block's eight 16-byte units permuted: unit u is stored at u ^ sz (sz = (row >> 1) & 7 for the k rows,
// which the attention kernel copies linearly into LDS and reads with 16-byte operand fetches: conflict free without
// padding the rows to 144 bytes)
__device__ __forceinline__ void store_block_swz(unsigned char* blk, const float (&v)[16], float s, int half, int sz) {
  u32x4 h0, h1, l0, l1;
  pack_block(v, s, h0, h1, l0, l1);
  u32x4* u = reinterpret_cast<u32x4*>(blk);
  u[(2 * half) ^ sz] = h0;
  u[(2 * half + 1) ^ sz] = h1;
  u[(4 + 2 * half) ^ sz] = l0;
  u[(5 + 2 * half) ^ sz] = l1;
}

__device__ __forceinline__ float h2f_lo(unsigned w) { return (float)__builtin_bit_cast(f16x2, w)[0]; }
__device__ __forceinline__ float h2f_hi(unsigned w) { return (float)__builtin_bit_cast(f16x2, w)[1]; }

// Load one block into quad layout: v[r] = (hi + lo) * inv_s
__device__ __forceinline__ void load_block(const unsigned char* blk, float (&v)[16], float inv_s, int half) {
  const u32x4* u = reinterpret_cast<const u32x4*>(blk) + 2 * half;
  const u32x4 h0 = u[0], h1 = u[1], l0 = u[4], l1 = u[5];
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

}  // namespace fdmi
