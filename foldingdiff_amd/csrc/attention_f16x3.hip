// Multi-head self-attention (HF BertSelfAttention 4.11.3 semantics incl. relative_key and the
// additive -10000 key mask; see attention_f32.hip for the math and citations) with the three
// contractions on the fp16 matrix cores at fp32-class accuracy: every operand x is split
// x*s = hi + lo (fp16 each, s a power of two) and a.b = a_hi.b_hi + a_hi.b_lo + a_lo.b_hi is
// issued as three v_mfma_f32_32x32x16_f16 into one fp32 accumulator (see gemm_f16x3.hip).
//
// One 4-wave workgroup per (sequence, head, group of 128 queries); a wave owns one 32-query row block
// and walks the key tiles (128 keys each) with a flash-style online softmax, so any
// L <= max_position_embeddings fits (L <= 128: a single tile).  53 KB of LDS per workgroup: two to
// three workgroups share a CU, so one's LDS fill overlaps the others' MFMA / softmax phases.
//   LDS (fp16 hi|lo row images, rows padded so the 16/8-byte operand fetches are conflict free):
//     K   [LP keys][hi d0-31 | lo d0-31]      144 B rows
//     Vt  [32 d][hi key0..LP-1 | lo ...]       transposed while filling: keys contiguous
//     Rw  [4 waves][32 queries][32 fp32 (+4 pad)]       scratch for the relative-key skew
//   The distance table E is split once on the host (fd_finalize) into 128-byte row images
//   [m][hi d0-31 | lo d0-31] and its band rows are fetched as MFMA operands straight from L2
//   (32 KB per layer, shared by every workgroup).
//   * S^T tile (keys x queries) = K . Q^T : A = K rows (LDS), B = Q (registers).  The transposed
//     form puts a query's scores in ONE lane pair: softmax max/sum are in-register reductions
//     plus a single cross-half shuffle, and P is already laid out as the A operand of P.V
//     (the k <-> (step, half, j) assignment of an MFMA is free as long as A and B agree:
//     key(c, half, j) = 16c + 8(j>>2) + 4*half + (j&3), which is exactly the C/D row map).
//   * R tile (queries x band) = Q . E^T, same Q registers as A operand.  The Toeplitz skew
//     S^T[r, l] += R[l, l - r + c] transposes queries from registers to lanes, so R goes through
//     the wave's LDS scratch one 32x32 tile at a time and is read back with per-lane addresses
//     (row l = lane, column l - r + const: bank = 5*lane mod 32, conflict free).
//   * O^T = V^T . P^T : A = Vt (LDS), B = P (registers, split to fp16 hi/lo after the softmax).  In the
//     transposed form queries stay in lanes, so the online-softmax rescale is a per-lane scalar.
// Scores, probabilities and the context accumulate in fp32; nothing but q|k|v and ctx touches HBM.
#include <cstdlib>

#include "fdmi_kernels.h"

namespace fdmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace a16 {

constexpr int HPB = 1;          // heads per workgroup
constexpr int KROW = 144;       // bytes per K / E row image: 64 B hi + 64 B lo + 16 B pad
constexpr int RLD = 36;         // floats per row of the R scratch
constexpr float QS = 16.0f;     // power-of-two operand scales (exact); undone in fp32 after the MFMAs
constexpr float KS = 16.0f;
constexpr float PS = 1024.0f;   // probabilities are <= 1
constexpr float VS = 16.0f;
constexpr float kLog2eOverSqrtD = 1.44269504088896341f * 0.17677669529663687f;  // log2(e) / sqrt(32)
constexpr float kMaskLog2 = -10000.0f * 1.44269504088896341f;                     // (1 - mask) * -10000, log2 domain

// 2^x for x <= 0 on v_exp_f32 (1 ulp).  x = -inf (padding keys, first-tile rescale) gives exactly 0;
// masked keys (about -14427 in the log2 domain) underflow to exactly 0 as exp(-10000) does in the reference.
__device__ __forceinline__ float exp2_neg(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ void split1(float x, float s, _Float16& hi, _Float16& lo) {
  const float xs = x * s;
  hi = (_Float16)xs;
  lo = (_Float16)(xs - (float)hi);
}

__device__ __forceinline__ void split8(const float4& p, const float4& q, float s, f16x8& hi, f16x8& lo) {
  const float x[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    _Float16 h, l;
    split1(x[i], s, h, l);
    hi[i] = h;
    lo[i] = l;
  }
}

// OCC = workgroups per CU the register allocation is capped for (2: up to 256 VGPRs, no spills;
// 3: 168 VGPRs, T = 4 spills ~100 dwords -- experiment knob FDMI_ATTN_OCC, 53.5 KB of LDS per workgroup
// allows three).
// TC = S^T tiles (32 keys each) per softmax chunk, T % TC == 0 (TC = T: the whole key tile at once).
// RB = scratch buffers per wave for the relative-key skew (2: consecutive band tiles alternate buffers, which
// removes the write-after-read barrier between them -- untimed experiment, FDMI_ATTN_RBUF=2).
template <int T, bool REL, int OCC, int TC = T, int RB = 1>
__global__ __launch_bounds__(256 * HPB, OCC) void attn_f16x3_kernel(const float* __restrict__ qkv,
                                                               const u32x4* __restrict__ demb, float r_scale,
                                                               const int* __restrict__ lens, float* __restrict__ ctx,
                                                               int L, int H, int maxpos) {
  constexpr int LP = 32 * T;        // keys per key tile = queries per query group
  constexpr int NT = 256 * HPB;
  constexpr int VROW = 4 * LP + 8;  // bytes per Vt row: LP hi + LP lo halves + 8 B pad (b64 reads conflict free)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
  unsigned char* Ks = smem16;                               // [HPB][LP] rows of KROW bytes
  unsigned char* Vt = Ks + HPB * LP * KROW;                 // [HPB][32] rows of VROW bytes
  float* Rs = reinterpret_cast<float*>(Vt + HPB * 32 * VROW);  // [4*HPB waves][32][RLD]   (REL only)

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int hgroups = (H + HPB - 1) / HPB;
  const int nqg = (L + LP - 1) / LP;  // query groups == key tiles (1 when L <= 128)
  const int qg = blockIdx.x % nqg;
  const int b = (blockIdx.x / nqg) / hgroups, h0 = ((blockIdx.x / nqg) % hgroups) * HPB;
  const int d = H * 32, ld = 3 * d;
  const int len = lens[b];
  const float* seq = qkv + (size_t)b * L * ld;

  const int hh = wid >> 2, wq = wid & 3, h = h0 + hh;
  const int l0 = qg * LP + 32 * wq;                       // first query of this wave's row block
  const bool active = h < H && wq < T && l0 < L;          // inactive waves only help filling LDS
  const float* base = seq + (h < H ? h : 0) * 32;
  const unsigned char* Kh = Ks + (size_t)hh * LP * KROW;
  const unsigned char* Vh = Vt + (size_t)hh * 32 * VROW;
  float* Rw0 = Rs + wid * RB * 32 * RLD;

  // Q operand: lane (query l31, half) holds d = 16c + 8*half + j   (B of K.Q^T and A of Q.E^T alike)
  f16x8 qh[2], ql[2];
  {
    const int l = l0 + l31;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float4 p = make_float4(0.f, 0.f, 0.f, 0.f), q = p;
      if (active && l < L) {
        const float* src = base + (size_t)l * ld + 16 * c + 8 * half;
        p = *reinterpret_cast<const float4*>(src);
        q = *reinterpret_cast<const float4*>(src + 4);
      }
      split8(p, q, QS, qh[c], ql[c]);
    }
  }
  // running softmax state of this lane's query (identical in both half-waves) and O^T accumulator:
  // rows = d (C/D row map), cols = queries (lanes) -> the per-query rescale is a per-lane scalar
  float m_run = -INFINITY, l_run = 0.f;
  f32x16 oacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.f;

  for (int kt = 0; kt < nqg; ++kt) {
    const int r0 = kt * LP;  // first key of the tile
    // keys >= len carry the additive -10000 and underflow to exactly 0 after the softmax (there is
    // always an unmasked key in tile 0), so tiles made only of such keys are skipped
    if (r0 >= len) break;
    if (kt > 0) __syncthreads();  // everyone is done reading the previous tile
    // ---- fill K (row images) : one thread per (head, key, 8-wide d octet)
    for (int idx = tid; idx < HPB * LP * 4; idx += NT) {
      const int fh = idx / (LP * 4), rem = idx % (LP * 4);
      const int r = rem >> 2, oct = rem & 3;
      // unconditional loads (row clamped), zeroed by select: no branch around a load
      const bool ok = r0 + r < L && h0 + fh < H;
      const float* src = seq + (size_t)(r0 + r < L ? r0 + r : L - 1) * ld + d + (h0 + fh < H ? h0 + fh : H - 1) * 32 + oct * 8;
      float4 p = *reinterpret_cast<const float4*>(src);
      float4 q = *reinterpret_cast<const float4*>(src + 4);
      if (!ok) p = q = make_float4(0.f, 0.f, 0.f, 0.f);
      f16x8 hi, lo;
      split8(p, q, KS, hi, lo);
      unsigned char* row = Ks + (size_t)(fh * LP + r) * KROW;
      *reinterpret_cast<u32x4*>(row + oct * 16) = __builtin_bit_cast(u32x4, hi);
      *reinterpret_cast<u32x4*>(row + 64 + oct * 16) = __builtin_bit_cast(u32x4, lo);
    }
    // ---- fill Vt (transposed): one thread per (head, key pair, 4-wide d group)
    for (int idx = tid; idx < HPB * (LP / 2) * 8; idx += NT) {
      const int fh = idx / ((LP / 2) * 8), rem = idx % ((LP / 2) * 8);
      const int kp = rem >> 3, c4 = rem & 7;
      const float* src = seq + 2 * d + (h0 + fh < H ? h0 + fh : H - 1) * 32 + c4 * 4;
      const int k0 = r0 + 2 * kp, k1 = k0 + 1;
      float4 v0 = *reinterpret_cast<const float4*>(src + (size_t)(k0 < L ? k0 : L - 1) * ld);
      float4 v1 = *reinterpret_cast<const float4*>(src + (size_t)(k1 < L ? k1 : L - 1) * ld);
      if (k0 >= L || h0 + fh >= H) v0 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k1 >= L || h0 + fh >= H) v1 = make_float4(0.f, 0.f, 0.f, 0.f);
      const float a0[4] = {v0.x, v0.y, v0.z, v0.w}, a1[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f16x2 hi, lo;
        _Float16 hv, lv;
        split1(a0[i], VS, hv, lv); hi[0] = hv; lo[0] = lv;
        split1(a1[i], VS, hv, lv); hi[1] = hv; lo[1] = lv;
        unsigned char* row = Vt + (size_t)(fh * 32 + c4 * 4 + i) * VROW;
        *reinterpret_cast<unsigned*>(row + 4 * kp) = __builtin_bit_cast(unsigned, hi);
        *reinterpret_cast<unsigned*>(row + 2 * LP + 4 * kp) = __builtin_bit_cast(unsigned, lo);
      }
    }
    __syncthreads();
    if (!active) continue;

    // The 32*T keys of the tile are processed TC S^T tiles (32*TC keys) at a time with the same online
    // softmax that links key tiles: TC < T trades a little recomputation (one band tile per extra chunk,
    // one rescale of the O accumulator) for 16*(T-TC) fewer live score registers.
    constexpr float S_SCALE = kLog2eOverSqrtD / (QS * KS);
    const unsigned char* vrow = Vh + (size_t)l31 * VROW;
#pragma unroll
    for (int t0 = 0; t0 < T; t0 += TC) {
      // chunks made only of keys >= len contribute exactly 0 (see above); chunk 0 always has key 0 < len
      if (r0 + 32 * t0 >= len) break;
      // band row of R tile q, column l31:  m = (maxpos-1) - (LP-1) + LP*(qg-kt) + 32*wq + 32q + l31.
      // Rows outside the table are only ever paired with padding keys / queries (L <= maxpos), so
      // the index is clamped instead of predicated (keeps the loads unconditional).  (Prefetching
      // one band tile ahead was measured: no gain, and the extra registers push the kernel into spills.)
      auto load_band = [&](u32x4 (&e)[4], int q) {
        int m = (maxpos - 1) - (LP - 1) + LP * (qg - kt) + 32 * wq + l31 + 32 * q;
        m = m < 0 ? 0 : (m > 2 * (maxpos - 1) ? 2 * (maxpos - 1) : m);
        const u32x4* row = demb + (size_t)m * 8;  // 128-byte image: units 0-3 hi d0-31, 4-7 lo
        e[0] = row[half]; e[1] = row[2 + half]; e[2] = row[4 + half]; e[3] = row[6 + half];
      };
      // S^T tiles of this chunk: rows = keys r0 + 32(t0+t) + rowmap(r, half), cols = queries l0 + l31.
      // Scores stay RAW MFMA sums u (scaled by QS*KS): the factor log2(e) / sqrt(head size) / (QS*KS)
      // that takes them to the log2 domain is applied inside the exponent fma of the softmax, and the
      // band values are brought to the same raw scale by the (power-of-two, exact) ratio of the scales
      // inside the accumulate fma -- no separate scaling pass over scores or band tiles.
      f32x16 sacc[TC];
#pragma unroll
      for (int t = 0; t < TC; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[t][r] = 0.f;
        const unsigned char* row = Kh + (size_t)(32 * (t0 + t) + l31) * KROW;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const f16x8 kh = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(row + 32 * c + 16 * half));
          const f16x8 kl = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(row + 64 + 32 * c + 16 * half));
          sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[c], sacc[t], 0, 0, 0);
          sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[c], sacc[t], 0, 0, 0);
          sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[c], sacc[t], 0, 0, 0);
        }
      }
      if constexpr (REL) {
        // R tile q: rows = queries rowmap(r, half), cols = band index 32q + l31 (band origin: this
        // wave's row block).  S^T tile t (of the key tile) element (key kl, query ql) needs band column
        // j = ql - kl + 31 of the tile pair (q = T-1-t, q+1):  j < 32 -> tile q, else tile q+1 column j-32.
        // The chunk's tiles t0 .. t0+TC-1 therefore use band tiles q = T-t0-TC .. T-t0.
        const float R_RATIO = r_scale;  // KS / table scale: band sums -> the raw scale of the scores
#pragma unroll
        for (int qq = 0; qq <= TC; ++qq) {
          u32x4 ecur[4];
          load_band(ecur, T - t0 - TC + qq);
          f32x16 racc;
#pragma unroll
          for (int r = 0; r < 16; ++r) racc[r] = 0.f;
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const f16x8 eh = __builtin_bit_cast(f16x8, ecur[c]);
            const f16x8 el = __builtin_bit_cast(f16x8, ecur[2 + c]);
            racc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh[c], eh, racc, 0, 0, 0);
            racc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh[c], el, racc, 0, 0, 0);
            racc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ql[c], eh, racc, 0, 0, 0);
          }
          float* Rw = Rw0 + (RB == 2 ? (qq & 1) * 32 * RLD : 0);
#pragma unroll
          for (int r = 0; r < 16; ++r) Rw[((r & 3) + 8 * (r >> 2) + 4 * half) * RLD + l31] = racc[r];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          // scratch row = query l31 (this lane).  Band column j = l31 - kl + 31 of the tile PAIR
          // (q, q+1) lives in tile q for j < 32 (l31 <= kl) and in tile q+1, column j - 32, otherwise:
          // both cases read scratch column j & 31, so one branch-free set of 16 reads serves the two
          // S^T tiles that use band tile q (selects, no exec-mask branches around LDS reads).
          float gth[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kl = (r & 3) + 8 * (r >> 2) + 4 * half;
            gth[r] = Rw[l31 * RLD + ((l31 - kl + 31) & 31)];
          }
          if (qq < TC) {  // low part: chunk tile TC-1-qq
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int kl = (r & 3) + 8 * (r >> 2) + 4 * half;
              sacc[TC - 1 - qq][r] = __builtin_fmaf((l31 <= kl) ? gth[r] : 0.f, R_RATIO, sacc[TC - 1 - qq][r]);
            }
          }
          if (qq > 0) {  // high part: chunk tile TC-qq
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int kl = (r & 3) + 8 * (r >> 2) + 4 * half;
              sacc[TC - qq][r] = __builtin_fmaf((l31 > kl) ? gth[r] : 0.f, R_RATIO, sacc[TC - qq][r]);
            }
          }
          if constexpr (RB == 1) {  // the next band tile overwrites this scratch: reads first
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
          }
          // RB == 2: tile qq+1 goes to the other buffer; tile qq+2 is ordered behind these reads by the
          // release/acquire pair of tile qq+1
        }
        if constexpr (RB == 2) {  // next chunk / key tile starts at buffer 0 again
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
      // mask + online softmax over keys (log2 domain): this lane + its partner (lane ^ 32) hold one
      // query's scores.  Chunks without masked / padding keys (wave-uniform test) skip the mask ops.
      float mt = -INFINITY;
      if (r0 + 32 * (t0 + TC) <= len) {
#pragma unroll
        for (int t = 0; t < TC; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sacc[t][r]);
      } else {
#pragma unroll
        for (int t = 0; t < TC; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = r0 + 32 * (t0 + t) + (r & 3) + 8 * (r >> 2) + 4 * half;
            float sc = sacc[t][r];
            if (key >= len) sc += kMaskLog2 / S_SCALE;   // (1 - mask) * -10000   (modelling.py:452), raw scale
            if (key >= L) sc = -INFINITY;                // tile padding: not a key at all
            sacc[t][r] = sc;
            mt = fmaxf(mt, sc);
          }
      }
      mt = fmaxf(mt, __shfl_xor(mt, 32));
      const float m_new = fmaxf(m_run, mt);                    // running maximum, raw scale
      const float alpha = exp2_neg((m_run - m_new) * S_SCALE);  // first chunk: 2^-inf = 0 (accumulators are 0 anyway)
      // p' = PS * 2^((u - m) * S_SCALE): the fp16-split scale PS = 2^10 rides in the exponent
      const float nm = __builtin_fmaf(-m_new, S_SCALE, 10.0f);
      static_assert(PS == 1024.0f, "exponent offset above is log2(PS)");
      float psum = 0.f;
#pragma unroll
      for (int t = 0; t < TC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pexp = exp2_neg(__builtin_fmaf(sacc[t][r], S_SCALE, nm));
          sacc[t][r] = pexp;
          psum += pexp;
        }
      psum += __shfl_xor(psum, 32);
      l_run = l_run * alpha + psum;   // carries the factor PS
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
      // O^T += V^T P^T :  A[i = d = l31][position (c, half, j)] = V[key(c,half,j)][d],
      //                   B[position][n = query l31] = P[query][key(c,half,j)] = sacc[t][8c + j],
      //                   key(c, half, j) = 32(t0+t) + 16c + 8(j>>2) + 4*half + (j&3)   (the C/D row map)
#pragma unroll
      for (int t = 0; t < TC; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          f16x8 ph, pl;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float xs = sacc[t][8 * c + j];  // PS * unnormalised probability (<= PS): ready for the fp16 split
            const _Float16 hv = (_Float16)xs;
            ph[j] = hv;
            pl[j] = (_Float16)(xs - (float)hv);
          }
          // V operand: keys 32(t0+t) + 16c + 4half + {0..3} and + 8 : two 8-byte reads per plane
          const int kb = 2 * (32 * (t0 + t) + 16 * c + 4 * half);
          const u32x2 vh0 = *reinterpret_cast<const u32x2*>(vrow + kb);
          const u32x2 vh1 = *reinterpret_cast<const u32x2*>(vrow + kb + 16);
          const u32x2 vl0 = *reinterpret_cast<const u32x2*>(vrow + 2 * LP + kb);
          const u32x2 vl1 = *reinterpret_cast<const u32x2*>(vrow + 2 * LP + kb + 16);
          const u32x4 vhu = {vh0[0], vh0[1], vh1[0], vh1[1]};
          const u32x4 vlu = {vl0[0], vl0[1], vl1[0], vl1[1]};
          const f16x8 vh = __builtin_bit_cast(f16x8, vhu), vl = __builtin_bit_cast(f16x8, vlu);
          oacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, oacc, 0, 0, 0);
          oacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, oacc, 0, 0, 0);
          oacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, oacc, 0, 0, 0);
        }
    }
  }
  if (!active) return;
  // ctx[query][h*32 + d] = O^T[d][query] / l_run : this lane owns query l0 + l31 and, per register
  // quad, four consecutive d = 8*(r>>2) + 4*half + (r&3)
  const int l = l0 + l31;
  if (l < L) {
    const float onorm = 1.0f / (VS * l_run);  // l_run and the accumulator both carry PS
    float* dst = ctx + ((size_t)b * L + l) * d + h * 32 + 4 * half;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(dst + 8 * g) =
          make_float4(oacc[4 * g] * onorm, oacc[4 * g + 1] * onorm, oacc[4 * g + 2] * onorm, oacc[4 * g + 3] * onorm);
  }
}

template <int T, bool REL, int OCC, int TC = T, int RB = 1>
static void launch_occ(const float* qkv, const void* demb, float r_scale, const int* lens, float* ctx, int B, int L, int H,
                     int maxpos, hipStream_t s) {
  constexpr int LP = 32 * T;
  const size_t smem = (size_t)HPB * LP * KROW + (size_t)HPB * 32 * (4 * LP + 8) +
                      (REL ? sizeof(float) * 4 * HPB * RB * 32 * RLD : 0);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f16x3_kernel<T, REL, OCC, TC, RB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  const int hgroups = (H + HPB - 1) / HPB;
  const int nqg = (L + LP - 1) / LP;
  hipLaunchKernelGGL((attn_f16x3_kernel<T, REL, OCC, TC, RB>), dim3(B * hgroups * nqg), dim3(256 * HPB), smem, s, qkv,
                     static_cast<const u32x4*>(demb), r_scale, lens, ctx, L, H, maxpos);
}

template <int T, bool REL>
static void launch_t(const float* qkv, const void* demb, float r_scale, const int* lens, float* ctx, int B, int L, int H,
                     int maxpos, hipStream_t s) {
  static const int occ = [] { const char* e = getenv("FDMI_ATTN_OCC"); return e ? atoi(e) : 2; }();
  static const int chunk = [] { const char* e = getenv("FDMI_ATTN_CHUNK"); return e ? atoi(e) : 0; }();
  static const int rbuf = [] { const char* e = getenv("FDMI_ATTN_RBUF"); return e ? atoi(e) : 1; }();
  if constexpr (T == 4 && REL) {
    if (rbuf == 2) {  // double-buffered skew scratch (72 KB of LDS per workgroup, still two per CU)
      launch_occ<T, REL, 2, T, 2>(qkv, demb, r_scale, lens, ctx, B, L, H, maxpos, s);
      return;
    }
  }
  if constexpr (T == 4) {
    if (chunk == 2) {  // 64-key softmax chunks: 32 fewer live registers (experiment)
      if (occ == 3) launch_occ<T, REL, 3, 2>(qkv, demb, r_scale, lens, ctx, B, L, H, maxpos, s);
      else launch_occ<T, REL, 2, 2>(qkv, demb, r_scale, lens, ctx, B, L, H, maxpos, s);
      return;
    }
  }
  if (occ == 3) launch_occ<T, REL, 3>(qkv, demb, r_scale, lens, ctx, B, L, H, maxpos, s);
  else launch_occ<T, REL, 2>(qkv, demb, r_scale, lens, ctx, B, L, H, maxpos, s);
}

}  // namespace a16

bool launch_attention_f16x3(const float* qkv, const void* dist_emb_split, float table_scale, const int* lens, float* ctx,
                            int B, int L, int H, int maxpos, hipStream_t s) {
  if (L < 1) return false;
  const int T = L > 128 ? 4 : (L + 31) / 32;  // long sequences: 128-key tiles x 128-query groups, online softmax
  const bool rel = dist_emb_split != nullptr;
  const float r_scale = a16::KS / table_scale;  // band sums (scaled QS * table_scale) -> raw score scale QS * KS; a power of two
#define FD_ATTN16_CASE(TT)                                                                              \
  case TT:                                                                                              \
    if (rel) a16::launch_t<TT, true>(qkv, dist_emb_split, r_scale, lens, ctx, B, L, H, maxpos, s);      \
    else a16::launch_t<TT, false>(qkv, nullptr, 1.f, lens, ctx, B, L, H, maxpos, s);                    \
    break;
  switch (T) {
    FD_ATTN16_CASE(1)
    FD_ATTN16_CASE(2)
    FD_ATTN16_CASE(3)
    FD_ATTN16_CASE(4)
  }
#undef FD_ATTN16_CASE
  return true;
}

}  // namespace fdmi
