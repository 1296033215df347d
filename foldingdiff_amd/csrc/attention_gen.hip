// Multi-head self-attention on row images for attention head sizes 64 / 96 / 128 (= 32 NB): HF BertSelfAttention 4.11.3 semantics
// as called from foldingdiff/modelling.py:473-480 (absolute, relative_key and relative_key_query position types, additive -10000
// key mask).  The reference's own unit-test model is the HuggingFace default BertConfig -- hidden 768 / 12 heads = head size 64
// (/root/reference/tests/test_transformer.py:21-24) -- and bin/train.py:301-307 lets a user train any hidden_size / num_heads.
// Every released configuration has head size 32 and runs attention_img.hip; this kernel is the general, un-tuned companion.
//
// A head of size 32 NB is NB consecutive 32-column "sub-heads" of the q | k | v projections: the GEMM epilogues (gemm_img.hip)
// write per-(sequence, sub-head) images exactly as for head size 32, so nothing changes on their side.  Here
//     S   = sum over the NB sub-heads of  Q_j K_j^T  (+ Q_j E_j^T skewed, + K_j E_j^T skewed)      one score tile, 6 NB MFMAs
//     O_j = P V_j                                                                                   NB output blocks
// with the same arithmetic as attention_img.hip: fp16 hi/lo split triples on v_mfma_f32_32x32x16_f16 (fp32-class accuracy),
// S^T layout (keys x queries: a query's scores live in one lane pair), online softmax over 32-key tiles in the log2 domain,
// the relative_key band as R^T = E Q^T tiles skewed through a per-wave LDS scratch with one gather per score.
// One wave per (sequence, head, 32-query block), four independent waves per workgroup (no workgroup barrier); the operands are
// already in MFMA fragment layout in HBM (img_common.h), so K, Q, V and the distance table are read straight into registers.
#include <cstdlib>
#include <type_traits>

#include "fdmi_kernels.h"
#include "img_common.h"

namespace fdmi {
namespace ag {

template <int V> using IC = std::integral_constant<int, V>;
typedef const __attribute__((address_space(3))) float* lds_cf32_t;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long long)(lds_ptr_t)(const_cast<void*>(p)); }
__device__ __forceinline__ float lds_f32(unsigned a) { return *(lds_cf32_t)(unsigned long long)a; }

constexpr float PS = 1024.0f;  // probabilities are <= 1
constexpr float kLog2e = 1.44269504088896341f;

template <int NB, bool REL, bool RKQ>
__global__ __launch_bounds__(256) void attn_gen_kernel(AttnImgArgs p) {
  static_assert(REL || !RKQ, "relative_key_query is a relative position type");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int H = p.H, HS = H * NB;             // heads; 32-column sub-heads (blocks of the q / k / v^T images and of ctx)
  const int nqb = p.LTOT >> 5;                // 32-query blocks per sequence
  const long long nitems = (long long)p.B * H * nqb;
  const long long item = (long long)blockIdx.x * 4 + wid;
  if (item >= nitems) return;                 // (no workgroup barrier anywhere below: the four waves are independent)
  const int qb = (int)(item % nqb), h = (int)((item / nqb) % H), b = (int)(item / ((long long)nqb * H));
  const int row0 = p.seq_row0[b], nrows = p.seq_row0[b + 1] - row0;  // token rows of the sequence (multiple of 8, >= real rows)
  const int len = p.lens[b], Lb = p.nrow[b];  // unmasked keys; positions that are keys at all
  const int l0 = 32 * qb;
  if (l0 >= nrows) return;

  unsigned char* Rw = smem + wid * 8192;      // skew scratch: two 4 KiB tile slots (attention_img.hip)
  const unsigned rw_lds = __builtin_amdgcn_readfirstlane(lds_addr(Rw));
  const int pi31 = (l31 & 24) | ((l31 & 3) << 1) | ((l31 >> 2) & 1);
  const unsigned gb = lds_addr(Rw) + (unsigned)((l31 - 4 * half + 4) * 128 + 4 * l31);
  const int nkb = p.LTOT >> 5;

  // ---- Q operands of the NB sub-heads: query l31 of block qb, units 2c + half (hi), 4 + 2c + half (lo)
  f16x8 qh[NB][2], ql[NB][2];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const u32x4* g0 = reinterpret_cast<const u32x4*>(p.qbuf + (((size_t)b * HS + h * NB + j) * p.LTOT + (size_t)l0) * 128);
    qh[j][0] = __builtin_bit_cast(f16x8, g0[half * 32 + l31]);
    qh[j][1] = __builtin_bit_cast(f16x8, g0[(2 + half) * 32 + l31]);
    ql[j][0] = __builtin_bit_cast(f16x8, g0[(4 + half) * 32 + l31]);
    ql[j][1] = __builtin_bit_cast(f16x8, g0[(6 + half) * 32 + l31]);
  }
  const float inv_sqrt_d = 1.0f / sqrtf((float)(32 * NB));
  const float s_scale = kLog2e * inv_sqrt_d / (p.q_scale * p.k_scale);  // raw MFMA sums -> log2 domain
  const float mask_raw = -10000.0f * kLog2e / s_scale;                  // (1 - mask) * -10000 at the raw scale
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 oacc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) oacc[j] = zero16;
  float m_run = -INFINITY, l_run = 0.f;

  // distance-table fragments of sub-head j for band rows mb + row (clamped: rows outside the table are only ever paired with
  // padding keys / queries, L <= maxpos)
  auto table_frag = [&](int mb, int row, int j, f16x8 (&eh)[2], f16x8 (&el)[2]) {
    int m = mb + row;
    m = m < 0 ? 0 : (m > 2 * (p.maxpos - 1) ? 2 * (p.maxpos - 1) : m);
    const u32x4_t* e = p.demb + ((size_t)m * NB + j) * 8;
    eh[0] = __builtin_bit_cast(f16x8, e[half]);
    eh[1] = __builtin_bit_cast(f16x8, e[2 + half]);
    el[0] = __builtin_bit_cast(f16x8, e[4 + half]);
    el[1] = __builtin_bit_cast(f16x8, e[6 + half]);
  };
  auto write_slot = [&](const f32x16& ra, int slot) {  // 16 x ds_write_addtid_b32: register r of lane L -> slot + 256 r + 4 L
    const unsigned m0v = rw_lds + (unsigned)(slot * 4096);
    unsigned keep;
    asm volatile(
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
  };

  // 32-key tiles holding an unmasked key: the others contribute exactly 0 (exp2 underflows).  With an explicit key mask (any pattern,
  // fd_forward_ex) every tile of the Lb positions is visited
  const unsigned char* km = p.kmask ? p.kmask + (size_t)b * p.L : nullptr;
  const int nkt = ((km ? Lb : len) + 31) >> 5;
  for (int kt = 0; kt < nkt; ++kt) {
    // ---- S^T tile: rows = keys 32 kt + rowmap(r, half), columns = queries l0 + l31; raw sums at scale q_scale * k_scale
    f32x16 sacc = zero16;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const u32x4* kg = reinterpret_cast<const u32x4*>(p.kbuf + (((size_t)b * HS + h * NB + j) * p.LTOT + (size_t)kt * 32) * 128);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const f16x8 kh = __builtin_bit_cast(f16x8, kg[(2 * c + half) * 32 + l31]);
        const f16x8 kl = __builtin_bit_cast(f16x8, kg[(4 + 2 * c + half) * 32 + l31]);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[j][c], sacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[j][c], sacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[j][c], sacc, 0, 0, 0);
      }
    }
    if constexpr (REL) {
      // band index x = ql - kl + 31 of (key kl, query ql) <-> table row mb + x; band tile tau = rows [32 tau, 32 tau + 32)
      const int mb = 32 * (qb - kt) + (p.maxpos - 1) - 31;
#pragma unroll
      for (int tau = 0; tau < 2; ++tau) {  // R^T tile tau = E_tile Q^T (MFMA row i -> band row pi(i)) -> scratch slot tau
        f32x16 ra = zero16;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          f16x8 eh[2], el[2];
          table_frag(mb + 32 * tau, pi31, j, eh, el);
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            ra = __builtin_amdgcn_mfma_f32_32x32x16_f16(eh[c], qh[j][c], ra, 0, 0, 0);
            ra = __builtin_amdgcn_mfma_f32_32x32x16_f16(el[c], qh[j][c], ra, 0, 0, 0);
            ra = __builtin_amdgcn_mfma_f32_32x32x16_f16(eh[c], ql[j][c], ra, 0, 0, 0);
          }
        }
        write_slot(ra, tau);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {  // one gather per score: the 63 band rows of the pair are consecutive 128-byte scratch rows
        const int klr = (r & 3) + 8 * (r >> 2);
        sacc[r] = __builtin_fmaf(lds_f32(gb + (unsigned)((27 - klr) * 128)), p.r_scale, sacc[r]);
      }
      if constexpr (RKQ) {
        // key term k_r . E[l - r + maxpos - 1]: K_t E_tau^T (rows = keys = the S^T tile's own rows, columns = band rows in natural
        // order); lane (query l31) register r (key kl) reads scratch[r][half][(l31 - kl + 31) & 31] of the tile its band index is in
        float* Rk = reinterpret_cast<float*>(Rw);
#pragma unroll
        for (int tau = 0; tau < 2; ++tau) {
          f32x16 ka = zero16;
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            f16x8 eh[2], el[2];
            table_frag(mb + 32 * tau, l31, j, eh, el);
            const u32x4* kg = reinterpret_cast<const u32x4*>(p.kbuf + (((size_t)b * HS + h * NB + j) * p.LTOT + (size_t)kt * 32) * 128);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const f16x8 kh = __builtin_bit_cast(f16x8, kg[(2 * c + half) * 32 + l31]);
              const f16x8 kl = __builtin_bit_cast(f16x8, kg[(4 + 2 * c + half) * 32 + l31]);
              ka = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, eh[c], ka, 0, 0, 0);
              ka = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, el[c], ka, 0, 0, 0);
              ka = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, eh[c], ka, 0, 0, 0);
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the gathers above / of the previous tile are done)
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int r = 0; r < 16; ++r) Rk[r * 64 + lane] = ka[r];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kl = (r & 3) + 8 * (r >> 2) + 4 * half;
            const float g = Rk[r * 64 + half * 32 + ((l31 - kl + 31) & 31)];
            const bool mine = tau == 0 ? l31 <= kl : l31 > kl;  // band index < 32: the lower tile
            sacc[r] = __builtin_fmaf(mine ? g : 0.f, p.r_scale_k, sacc[r]);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the next key tile overwrites the scratch
        __builtin_amdgcn_wave_barrier();
      }
    }

    // ---- mask + online softmax over the keys (log2 domain): this lane and lane ^ 32 hold one query's scores
    float mt = -INFINITY;
    if (!km && 32 * (kt + 1) <= len) {
#pragma unroll
      for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sacc[r]);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * half;
        float sc = sacc[r];
        if (km ? (key < Lb && km[key] == 0) : key >= len) sc += mask_raw;  // (1 - mask) * -10000   (modelling.py:452)
        if (key >= Lb) sc = -INFINITY;   // not a key at all (rows that do not exist)
        sacc[r] = sc;
        mt = fmaxf(mt, sc);
      }
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * s_scale);  // first tile: 2^-inf = 0 (the accumulators are 0 anyway)
    const float nm = __builtin_fmaf(-m_new, s_scale, 10.0f);                 // + log2(PS)
    static_assert(PS == 1024.0f, "exponent offset above is log2(PS)");
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], s_scale, nm));
      sacc[r] = pe;
      psum += pe;
    }
    psum += __shfl_xor(psum, 32);
    l_run = l_run * alpha + psum;  // carries the factor PS
    m_run = m_new;

    // ---- O_j^T += V_j^T P^T: A[i = d = l31][(c, half, e)] = V[key][d], B = P (this lane's registers)
    u32x4 phu[2], plu[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned hv, lv;
        split_pair(sacc[8 * c + 2 * e], sacc[8 * c + 2 * e + 1], hv, lv);
        phu[c][e] = hv;
        plu[c][e] = lv;
      }
    const int sz = vt_swz(l31);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[j][r] *= alpha;
      const unsigned char* blk = p.vbuf + ((((size_t)b * HS + h * NB + j) * nkb + kt) * 32 + l31) * 128;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int ua = 4 * c + half;
        const u32x2 vh0 = *reinterpret_cast<const u32x2*>(blk + ((ua ^ sz) << 3));
        const u32x2 vh1 = *reinterpret_cast<const u32x2*>(blk + (((ua + 2) ^ sz) << 3));
        const u32x2 vl0 = *reinterpret_cast<const u32x2*>(blk + (((ua + 8) ^ sz) << 3));
        const u32x2 vl1 = *reinterpret_cast<const u32x2*>(blk + (((ua + 10) ^ sz) << 3));
        const f16x8 vh = __builtin_bit_cast(f16x8, u32x4{vh0[0], vh0[1], vh1[0], vh1[1]});
        const f16x8 vl = __builtin_bit_cast(f16x8, u32x4{vl0[0], vl0[1], vl1[0], vl1[1]});
        const f16x8 ph = __builtin_bit_cast(f16x8, phu[c]), pl = __builtin_bit_cast(f16x8, plu[c]);
        oacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, oacc[j], 0, 0, 0);
        oacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, oacc[j], 0, 0, 0);
        oacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, oacc[j], 0, 0, 0);
      }
    }
  }

  // ---- ctx[row0 + query][sub-head block] = O_j^T[d][query] / l_run (register r = 4q + e <-> d = 8q + 4 half + e: quad layout)
  const int l = l0 + l31;
  const bool ok = l < nrows;
  const int row = row0 + l;
  const float onorm = p.ctx_scale / (p.v_scale * l_run);  // l_run and the accumulators both carry PS
  const unsigned voff = ok ? (unsigned)((((row >> 5) * HS * 8 + 2 * half) * 32 + (row & 31)) * 16) : 0xFFFFFF00u;
  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(p.ctx, 0, 0xFFFFFF00u, 0x00020000);
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    float o[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = oacc[j][r] * onorm;
    u32x4 h0, h1, lo0, lo1;
    pack_block(o, 1.0f, h0, h1, lo0, lo1);
    const int hoff = (h * NB + j) * 4096;
    __builtin_amdgcn_raw_buffer_store_b128(h0, rsc, (int)voff, hoff, 0);
    __builtin_amdgcn_raw_buffer_store_b128(h1, rsc, (int)voff, hoff + 512, 0);
    __builtin_amdgcn_raw_buffer_store_b128(lo0, rsc, (int)voff, hoff + 4 * 512, 0);
    __builtin_amdgcn_raw_buffer_store_b128(lo1, rsc, (int)voff, hoff + 5 * 512, 0);
    store_guard(h0, h1);
    store_guard(lo0, lo1);
  }
}

template <int NB>
static void launch(const AttnImgArgs& p, hipStream_t s) {
  const long long nitems = (long long)p.B * p.H * (p.LTOT >> 5);
  const unsigned grid = (unsigned)((nitems + 3) / 4);
  if (p.demb == nullptr) hipLaunchKernelGGL((attn_gen_kernel<NB, false, false>), dim3(grid), dim3(256), 0, s, p);
  else if (p.rkq) hipLaunchKernelGGL((attn_gen_kernel<NB, true, true>), dim3(grid), dim3(256), 4 * 8192, s, p);
  else hipLaunchKernelGGL((attn_gen_kernel<NB, true, false>), dim3(grid), dim3(256), 4 * 8192, s, p);
}

}  // namespace ag

// p.H = heads of size 32 * nb; the q / k / v^T images and ctx are indexed by 32-column sub-head (nb per head)
bool launch_attention_gen(const AttnImgArgs& p, int nb, hipStream_t s) {
  switch (nb) {
    case 1: ag::launch<1>(p, s); return true;
    case 2: ag::launch<2>(p, s); return true;
    case 3: ag::launch<3>(p, s); return true;
    case 4: ag::launch<4>(p, s); return true;
    default: return false;
  }
}

}  // namespace fdmi
