// Multi-head self-attention of HF BertSelfAttention (transformers 4.11.3; called
// from foldingdiff/modelling.py:473-480) with the additive key mask of
// modelling.py:450-452 and the relative_key score term the released model uses
// (config_jsons/cath_full_angles_cosine.json:10):
//
//   S[l,r] = ( q_l . k_r + q_l . E[l - r + maxpos - 1] ) / sqrt(32) + (r >= len ? -10000 : 0)
//   P = softmax_r(S);   ctx[l,:] = sum_r P[l,r] v_r
//
// One 8-wave workgroup per (sequence, pair of heads): each head is handled by 4 waves
// (32-query row blocks), so every SIMD hosts two waves whose MFMA and VALU/LDS phases
// overlap; K, V of both heads and the shared band of the distance table E live in LDS.  All three
// contractions run on v_mfma_f32_32x32x2_f32 (exact fp32):
//   * S tiles   32 x 32 :  A = Q rows (registers), B = K rows (LDS)
//   * R tiles   32 x 32 :  R[l, m] = q_l . E[m] over the 32*(T+1)-wide band of m the
//                          row block can reach; the Toeplitz skew  S[l,r] += R[l, l-r+c]
//                          is a same-row cross-lane gather (ds_bpermute), two candidate
//                          tiles per S tile  => (T+1)/T extra MFMA work instead of 2x.
//   * softmax in registers: a row lives in one 32-lane half -> xor-shuffle butterflies.
//   * P goes tile by tile through a per-wave 32x36 LDS scratch to become the A operand of P.V.
// The score matrix never touches HBM (it would be B*H*L^2*4 = 403 MB per layer at C2).
#include "fdmi_kernels.h"

namespace fdmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KLD = 36;  // padded row stride (floats) of K / E / P tiles: conflict-free ds_read_b128
constexpr int HPB = 2;   // heads per workgroup (4 waves each): 2 waves per SIMD, they share the E band

// exp(x) for x <= 0 on the v_exp_f32 path: exp2(x*log2e) with the rounding error of the
// product carried into a first-order correction (accuracy ~1 ulp, ~6 VALU ops instead of
// libm's range-reduced expf).  Inputs far below -87 (masked keys, -10000) give exactly 0.
__device__ __forceinline__ float exp_neg(float x) {
  const float LOG2E_HI = 1.44269502162933349609375f, LOG2E_LO = 1.925963033500011e-08f;
  x = fmaxf(x, -200.0f);  // exp(-200) == 0 in fp32; keeps -inf (tile padding) away from inf - inf below
  const float t = x * LOG2E_HI;
  const float err = fmaf(x, LOG2E_HI, -t) + x * LOG2E_LO;
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e, err * 0.693147180559945f, e);
}

// RKQ: relative_key_query -- the score also gets k_r . E[l - r + maxpos - 1] (HF BertSelfAttention 4.11.3; bin/train.py:305-307)
template <int T, bool REL, bool RKQ = false>
__global__ __launch_bounds__(256 * HPB) void attn_f32_kernel(const float* __restrict__ qkv,
                                                             const float* __restrict__ demb,
                                                             const int* __restrict__ lens, float* __restrict__ ctx,
                                                             int L, int H, int maxpos) {
  constexpr int LP = 32 * T;   // padded sequence length
  constexpr int NT = 256 * HPB;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;                              // [HPB][LP][36]
  float* Vs = Ks + HPB * LP * KLD;               // [HPB][LP][32]
  float* Es = Vs + HPB * LP * 32;                // [2*LP][36]   (REL only)
  float* Ps = Es + (REL ? 2 * LP * KLD : 0);     // [4*HPB waves][32][36]  one P tile per wave

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int hgroups = (H + HPB - 1) / HPB;
  const int b = blockIdx.x / hgroups, h0 = (blockIdx.x % hgroups) * HPB;
  const int d = H * 32, ld = 3 * d;
  const int len = lens[b];
  const float* seq = qkv + (size_t)b * L * ld;

  for (int idx = tid; idx < HPB * LP * 8; idx += NT) {
    const int hh = idx / (LP * 8), rem = idx % (LP * 8);
    const int r = rem >> 3, c4 = rem & 7;
    float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
    if (r < L && h0 + hh < H) {
      const float* p = seq + (size_t)r * ld + (h0 + hh) * 32 + c4 * 4;
      kv = *reinterpret_cast<const float4*>(p + d);
      vv = *reinterpret_cast<const float4*>(p + 2 * d);
    }
    *reinterpret_cast<float4*>(&Ks[(hh * LP + r) * KLD + c4 * 4]) = kv;
    *reinterpret_cast<float4*>(&Vs[(hh * LP + r) * 32 + c4 * 4]) = vv;
  }
  if constexpr (REL) {
    // Es[e] = E[m_min + e],  m_min = (maxpos-1) - (LP-1); rows outside the table are never
    // selected for a valid (l, r) pair and are zero filled.
    const int m_min = (maxpos - 1) - (LP - 1);
    for (int idx = tid; idx < 2 * LP * 8; idx += NT) {
      const int e = idx >> 3, c4 = idx & 7, m = e + m_min;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m >= 0 && m <= 2 * (maxpos - 1)) v = *reinterpret_cast<const float4*>(demb + (size_t)m * 32 + c4 * 4);
      *reinterpret_cast<float4*>(&Es[e * KLD + c4 * 4]) = v;
    }
  }
  __syncthreads();

  const int hh = wid >> 2, wq = wid & 3, h = h0 + hh;
  if (h >= H) return;  // odd head count: the second wave group of the last workgroup has no head
  const float* base = seq + h * 32;
  const float* Kh = Ks + hh * LP * KLD;
  const float* Vh = Vs + hh * LP * 32;
  float* Pw = Ps + wid * 32 * KLD;
  const int nrb = (L + 31) >> 5;
  for (int rb = wq; rb < nrb; rb += 4) {
    const int l0 = rb * 32;
    // Q fragment: A[i = l31][k = 8g + 4*half + s]
    f32x4 qa[4];
    {
      const int l = l0 + l31;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (l < L) v = *reinterpret_cast<const float4*>(base + (size_t)l * ld + 8 * g + 4 * half);
        qa[g][0] = v.x; qa[g][1] = v.y; qa[g][2] = v.z; qa[g][3] = v.w;
      }
    }
    f32x16 sacc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[t][r] = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 kb = *reinterpret_cast<const f32x4*>(&Kh[(32 * t + l31) * KLD + 8 * g + 4 * half]);
#pragma unroll
        for (int s = 0; s < 4; ++s) sacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[g][s], kb[s], sacc[t], 0, 0, 0);
      }
    }
    if constexpr (REL) {
      f32x16 racc[T + 1];
#pragma unroll
      for (int q = 0; q <= T; ++q) {
#pragma unroll
        for (int r = 0; r < 16; ++r) racc[q][r] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 eb = *reinterpret_cast<const f32x4*>(&Es[(l0 + 32 * q + l31) * KLD + 8 * g + 4 * half]);
#pragma unroll
          for (int s = 0; s < 4; ++s) racc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[g][s], eb[s], racc[q], 0, 0, 0);
        }
      }
      // R tile q, column j  <->  m = c + l0 - 32*(T-1-q) - 31 + j.  For S tile t (q = T-1-t)
      // element (li, rj) needs j = li - rj + 31 in [0, 62]: tile q (j < 32) or q+1 (j - 32).
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int q = T - 1 - t;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int li = (r & 3) + 8 * (r >> 2) + 4 * half;
          const int delta = li - l31 + 31;
          const int src = (delta & 31) + 32 * half;
          const float lo = __shfl(racc[q][r], src);
          const float hi = __shfl(racc[q + 1][r], src);
          sacc[t][r] += (delta >= 32) ? hi : lo;
        }
      }
    }
    if constexpr (RKQ) {
      // key term: tile K_t E_q^T (rows = keys rowmap(r, half), columns = band index, lane l31) for the band tiles q = T-1-t
      // and q+1 of S tile t; element (query li, key rj = this lane) needs row rj, column (li - rj + 31) & 31 of the lower
      // tile (li <= rj) or the upper one: a transposing gather through the wave's [32][36] scratch
#pragma unroll
      for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int side = 0; side < 2; ++side) {
          const int q = T - 1 - t + side;
          f32x16 kacc;
#pragma unroll
          for (int r = 0; r < 16; ++r) kacc[r] = 0.f;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 ka = *reinterpret_cast<const f32x4*>(&Kh[(32 * t + l31) * KLD + 8 * g + 4 * half]);
            const f32x4 eb = *reinterpret_cast<const f32x4*>(&Es[(l0 + 32 * q + l31) * KLD + 8 * g + 4 * half]);
#pragma unroll
            for (int s = 0; s < 4; ++s) kacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[s], eb[s], kacc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) Pw[((r & 3) + 8 * (r >> 2) + 4 * half) * KLD + l31] = kacc[r];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int li = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int delta = li - l31 + 31;   // band column of the tile pair for (query li, key l31)
            const float v = Pw[l31 * KLD + (delta & 31)];
            if ((delta >= 32) == (side == 1)) sacc[t][r] += v;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    // scale, mask, softmax over r (row li lives in this lane's 32-lane half)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int key = 32 * t + l31;
        float s = sacc[t][r] * 0.17677669529663687f;  // / sqrt(attention_head_size = 32)
        if (key >= len) s += -10000.0f;               // (1 - mask) * -10000   (modelling.py:452)
        if (key >= L) s = -INFINITY;                  // tile padding: not a key at all
        sacc[t][r] = s;
        mx = fmaxf(mx, s);
      }
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
      float sum = 0.f;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float pexp = exp_neg(sacc[t][r] - mx);
        sacc[t][r] = pexp;
        sum += pexp;
      }
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
      const float inv = 1.0f / sum;
#pragma unroll
      for (int t = 0; t < T; ++t) sacc[t][r] *= inv;
    }
    // ctx = P V, one key tile at a time through the wave's [32][36] LDS scratch (C-layout ->
    // A-operand layout):  A[i = l31][k] = P[l0 + l31][32t + 8*g8 + 4*half + s],  B[k][j = l31] = V[k][j]
    f32x16 oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) Pw[((r & 3) + 8 * (r >> 2) + 4 * half) * KLD + l31] = sacc[t][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int g8 = 0; g8 < 4; ++g8) {
        const f32x4 pa = *reinterpret_cast<const f32x4*>(&Pw[l31 * KLD + 8 * g8 + 4 * half]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float vb = Vh[(32 * t + 8 * g8 + 4 * half + s) * 32 + l31];
          oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[s], vb, oacc, 0, 0, 0);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int l = l0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (l < L) ctx[((size_t)b * L + l) * d + h * 32 + l31] = oacc[r];
    }
  }
}

template <int T, bool REL, bool RKQ = false>
static void launch_t(const float* qkv, const float* demb, const int* lens, float* ctx, int B, int L, int H, int maxpos,
                     hipStream_t s) {
  constexpr int LP = 32 * T;
  const size_t smem = sizeof(float) * (HPB * LP * KLD + HPB * LP * 32 + (REL ? 2 * LP * KLD : 0) + 4 * HPB * 32 * KLD);
  static bool attr_set[64] = {false};  // the attribute is per device (one model per device in a multi-GPU process)
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f32_kernel<T, REL, RKQ>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  const int hgroups = (H + HPB - 1) / HPB;
  hipLaunchKernelGGL((attn_f32_kernel<T, REL, RKQ>), dim3(B * hgroups), dim3(256 * HPB), smem, s, qkv, demb, lens, ctx, L, H,
                     maxpos);
}

bool launch_attention_f32(const float* qkv, const float* dist_emb, const int* lens, float* ctx, int B, int L, int H,
                          int maxpos, hipStream_t s, int rkq) {
  if (L < 1 || L > 128) return false;
  const int T = (L + 31) / 32;
  const bool rel = dist_emb != nullptr;
#define FD_ATTN_CASE(TT)                                                          \
  case TT:                                                                        \
    if (rel && rkq) launch_t<TT, true, true>(qkv, dist_emb, lens, ctx, B, L, H, maxpos, s); \
    else if (rel) launch_t<TT, true>(qkv, dist_emb, lens, ctx, B, L, H, maxpos, s); \
    else launch_t<TT, false>(qkv, dist_emb, lens, ctx, B, L, H, maxpos, s);       \
    break;
  switch (T) {
    FD_ATTN_CASE(1)
    FD_ATTN_CASE(2)
    FD_ATTN_CASE(3)
    FD_ATTN_CASE(4)
  }
#undef FD_ATTN_CASE
  return true;
}

}  // namespace fdmi
