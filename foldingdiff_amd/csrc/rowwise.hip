// HBM-bound, one-wave-per-token kernels of the reverse-diffusion step:
//   embed        K1  Linear(F->d) + BertEmbeddings LayerNorm + time embedding
//                    (foldingdiff/modelling.py:464-472, :157-170)
//   layernorm        standalone residual LayerNorm (unfused fallback of the GEMM epilogue)
//   head_update  K8 tail + K9: AnglesPredictor.layer_norm + dense2 (modelling.py:206-207),
//                    the p_sample update (foldingdiff/sampling.py:62-75) and the per-feature
//                    wrap to [-pi, pi) (sampling.py:119-130, utils.py:100-106)
//   philox_fill      the perf-mode N(0,1) stream, exposed for tests
// A token's d_model row is spread over the 64 lanes (column = lane + 64*j, coalesced
// 256-byte wave accesses); reductions are xor-shuffle butterflies over the wavefront.
#include <cstdlib>

#include "fdmi_kernels.h"

namespace fdmi {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// sum over the 16 lanes of a DPP row (result in all 16): quad butterflies + the two mirror steps
__device__ __forceinline__ float row16_sum(float v) {
#define FD_DPP_ADD(ctrl) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
  FD_DPP_ADD(0xB1);   // quad_perm [1,0,3,2]
  FD_DPP_ADD(0x4E);   // quad_perm [2,3,0,1]
  FD_DPP_ADD(0x141);  // row_half_mirror
  FD_DPP_ADD(0x140);  // row_mirror
#undef FD_DPP_ADD
  return v;
}

static int rowwise_env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// ---- 16-lanes-per-token layout (d = 64 * NV, the released d = 384 -> NV = 6) ----
// Lane k of a 16-lane group owns the float4 columns k + 16 j (j < NV): 256-byte coalesced group
// accesses, row reductions are 4 DPP steps (no LDS round trips), 16 tokens per workgroup pass and a
// grid-stride loop, so the per-column parameters (LayerNorm gamma / beta, bias, time embedding)
// live in registers and the small weight matrix in LDS for the whole launch.
template <int NV>
__device__ __forceinline__ void row16_layernorm(float4 (&v)[NV], const float4 (&gm)[NV], const float4 (&bt)[NV], int d,
                                                float eps) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  const float mean = row16_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    v[j].x -= mean; v[j].y -= mean; v[j].z -= mean; v[j].w -= mean;
    q += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
  }
  const float rstd = 1.0f / sqrtf(row16_sum(q) / (float)d + eps);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    v[j].x = v[j].x * rstd * gm[j].x + bt[j].x;
    v[j].y = v[j].y * rstd * gm[j].y + bt[j].y;
    v[j].z = v[j].z * rstd * gm[j].z + bt[j].z;
    v[j].w = v[j].w * rstd * gm[j].w + bt[j].w;
  }
}

// LayerNorm of a row held as v[j] = row[lane + 64 j]  (biased variance, rstd = 1/sqrt(var+eps))
template <int NJ>
__device__ __forceinline__ void row_layernorm(float (&v)[NJ], int lane, int d, const float* gamma, const float* beta,
                                              float eps) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) s += (lane + 64 * j < d) ? v[j] : 0.f;
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const float dl = (lane + 64 * j < d) ? v[j] - mean : 0.f;
    q += dl * dl;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    if (c < d) v[j] = (v[j] - mean) * rstd * gamma[c] + beta[c];
  }
}

// ------------------------------------------------------------------ embed (K1)
template <int NJ>
__global__ __launch_bounds__(256) void embed_kernel(const float* __restrict__ x, const float* __restrict__ w_in,
                                                    const float* __restrict__ b_in, const float* __restrict__ pos_emb,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float eps, const float* __restrict__ time_table,
                                                    const int* __restrict__ t_dev, float* __restrict__ h, int M, int L,
                                                    int F, int d) {
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= M) return;
  const int t = *t_dev;
  float xin[kMaxFeat];
#pragma unroll
  for (int f = 0; f < kMaxFeat; ++f) xin[f] = f < F ? x[(size_t)tok * F + f] : 0.f;
  float v[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    float a = 0.f;
    if (c < d) {
      a = b_in[c];
      for (int f = 0; f < F; ++f) a += xin[f] * w_in[c * F + f];
      if (pos_emb) a += pos_emb[(size_t)(tok % L) * d + c];  // absolute positions only (modelling.py:164-166)
    }
    v[j] = a;
  }
  row_layernorm<NJ>(v, lane, d, gamma, beta, eps);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    if (c < d) h[(size_t)tok * d + c] = v[j] + time_table[(size_t)t * d + c];  // added AFTER the LayerNorm (:472)
  }
}

template <int NV>
__global__ __launch_bounds__(256) void embed16_kernel(const float* __restrict__ x, const float* __restrict__ w_in,
                                                      const float* __restrict__ b_in, const float* __restrict__ pos_emb,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float eps, const float* __restrict__ time_table,
                                                      const int* __restrict__ t_dev, float* __restrict__ h, int M, int L,
                                                      int F) {
  constexpr int d = 64 * NV;
  extern __shared__ __attribute__((aligned(16))) float wT[];  // [F][d]: w_in transposed
  for (int i = threadIdx.x; i < F * d; i += 256) {
    const int f = i / d, c = i - f * d;
    wT[i] = w_in[c * F + f];
  }
  const int k = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int t = *t_dev;
  float4 bi[NV], gm[NV], bt[NV], tt[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = 4 * (k + 16 * j);
    bi[j] = *reinterpret_cast<const float4*>(b_in + c);
    gm[j] = *reinterpret_cast<const float4*>(gamma + c);
    bt[j] = *reinterpret_cast<const float4*>(beta + c);
    tt[j] = *reinterpret_cast<const float4*>(time_table + (size_t)t * d + c);
  }
  __syncthreads();
  for (int tg = blockIdx.x; tg * 16 < M; tg += gridDim.x) {
    const int tok = tg * 16 + g;
    const int tc = tok < M ? tok : M - 1;
    float4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = bi[j];
    for (int f = 0; f < F; ++f) {
      const float xf = x[(size_t)tc * F + f];
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const float4 w = *reinterpret_cast<const float4*>(wT + f * d + 4 * (k + 16 * j));
        v[j].x += xf * w.x; v[j].y += xf * w.y; v[j].z += xf * w.z; v[j].w += xf * w.w;
      }
    }
    if (pos_emb) {  // absolute positions only (modelling.py:164-166)
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const float4 pe = *reinterpret_cast<const float4*>(pos_emb + (size_t)(tc % L) * d + 4 * (k + 16 * j));
        v[j].x += pe.x; v[j].y += pe.y; v[j].z += pe.z; v[j].w += pe.w;
      }
    }
    row16_layernorm<NV>(v, gm, bt, d, eps);
    if (tok < M) {
#pragma unroll
      for (int j = 0; j < NV; ++j)  // time embedding added AFTER the LayerNorm (modelling.py:472)
        *reinterpret_cast<float4*>(h + (size_t)tok * d + 4 * (k + 16 * j)) =
            make_float4(v[j].x + tt[j].x, v[j].y + tt[j].y, v[j].z + tt[j].z, v[j].w + tt[j].w);
    }
  }
}

template <int NV>
static void launch_embed16(const float* x, const float* w_in, const float* b_in, const float* pos_emb, const float* gamma,
                           const float* beta, float eps, const float* time_table, const int* t_dev, float* h, int M, int L,
                           int F, hipStream_t s) {
  int grid = (M + 15) / 16;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL((embed16_kernel<NV>), dim3(grid), dim3(256), (size_t)F * 64 * NV * 4, s, x, w_in, b_in, pos_emb, gamma,
                     beta, eps, time_table, t_dev, h, M, L, F);
}

// FDMI_ROWWISE16=0 keeps the one-wave-per-token kernels (also the generic path for d % 64 != 0 or d > 512)
static bool use_row16(int d) {
  static const int on = rowwise_env_int("FDMI_ROWWISE16", 1);
  return on && d % 64 == 0 && d >= 64 && d <= 512;
}

void launch_embed(const float* x, const float* w_in, const float* b_in, const float* pos_emb, const float* gamma,
                  const float* beta, float eps, const float* time_table, const int* t_dev, float* h, int B, int L, int F,
                  int d, hipStream_t s) {
  const int M = B * L;
  if (use_row16(d)) {
#define FD_E16(NV) case NV: launch_embed16<NV>(x, w_in, b_in, pos_emb, gamma, beta, eps, time_table, t_dev, h, M, L, F, s); return;
    switch (d / 64) { FD_E16(1) FD_E16(2) FD_E16(3) FD_E16(4) FD_E16(5) FD_E16(6) FD_E16(7) FD_E16(8) }
#undef FD_E16
  }
  const dim3 grid((M + 3) / 4), block(256);
  const int nj = (d + 63) / 64;
#define FD_EMBED(NJ)                                                                                             \
  hipLaunchKernelGGL((embed_kernel<NJ>), grid, block, 0, s, x, w_in, b_in, pos_emb, gamma, beta, eps, time_table, \
                     t_dev, h, M, L, F, d)
  if (nj <= 1) FD_EMBED(1);
  else if (nj <= 3) FD_EMBED(3);
  else if (nj <= 6) FD_EMBED(6);
  else if (nj <= 12) FD_EMBED(12);
  else FD_EMBED(16);
#undef FD_EMBED
}

// ------------------------------------------------------------- standalone LN
template <int NJ>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        float* __restrict__ y, int rows, int d) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    v[j] = c < d ? x[(size_t)row * d + c] : 0.f;
  }
  row_layernorm<NJ>(v, lane, d, gamma, beta, eps);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    if (c < d) y[(size_t)row * d + c] = v[j];
  }
}

void launch_layernorm(const float* x, const float* gamma, const float* beta, float eps, float* y, int rows, int d,
                      hipStream_t s) {
  const dim3 grid((rows + 3) / 4), block(256);
  const int nj = (d + 63) / 64;
#define FD_LN(NJ) hipLaunchKernelGGL((layernorm_kernel<NJ>), grid, block, 0, s, x, gamma, beta, eps, y, rows, d)
  if (nj <= 1) FD_LN(1);
  else if (nj <= 3) FD_LN(3);
  else if (nj <= 6) FD_LN(6);
  else if (nj <= 12) FD_LN(12);
  else FD_LN(16);
#undef FD_LN
}

// ---------------------------------------------------------- Philox4x32-10
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    const unsigned n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    const unsigned n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// N(0,1) for feature f of token (global sequence `seq`, position l) at step t.
// Counter = (seq lo, seq hi | l << 8 ... ) keeps the stream independent of how the
// batch is sharded across GPUs (the key is the user seed).
__device__ __forceinline__ float philox_normal(unsigned long long seed, int t, long long seq, int l, int f) {
  unsigned o[4];
  philox4x32_10((unsigned)seq, (unsigned)((unsigned long long)seq >> 32), (unsigned)l | ((unsigned)(f >> 2) << 24),
                (unsigned)t, (unsigned)seed, (unsigned)(seed >> 32), o);
  const int pair = (f >> 1) & 1;
  const float u1 = ((float)o[2 * pair] + 0.5f) * 2.3283064365386963e-10f;      // (0, 1]
  const float u2 = ((float)o[2 * pair + 1] + 0.5f) * 2.3283064365386963e-10f;
  const float rad = sqrtf(-2.0f * logf(u1));
  float sn, cs;
  sincosf(6.283185307179586f * u2, &sn, &cs);
  return (f & 1) ? rad * sn : rad * cs;
}

__global__ void philox_fill_kernel(float* __restrict__ out, unsigned long long seed, int t, long long seq_offset, int B,
                                   int L, int F) {
  const long long n = (long long)B * L * F;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int f = (int)(i % F);
    const long long tok = i / F;
    out[i] = philox_normal(seed, t, seq_offset + tok / L, (int)(tok % L), f);
  }
}

void launch_philox_fill(float* out, unsigned long long seed, int t, long long seq_offset, int B, int L, int F,
                        hipStream_t s) {
  const long long n = (long long)B * L * F;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(philox_fill_kernel, dim3(blocks), dim3(256), 0, s, out, seed, t, seq_offset, B, L, F);
}

// --------------------------------------------- head tail + p_sample update (K8/K9)
// wrap: ((v - lo) % (hi - lo)) + lo with lo = -pi, hi = pi evaluated as torch does on a
// float32 tensor with python-float bounds: v + f32(pi); torch.remainder(., f32(2 pi));
// + f32(-pi).  Explicit __f*_rn keeps the compiler from contracting into FMAs, so for
// identical inputs the result is bit-identical to the reference's CPU arithmetic.
__device__ __forceinline__ float wrap_pi(float v) {
  const float PI_F = 3.14159274101257324f, TWO_PI_F = 6.28318548202514648f;
  const float sft = __fadd_rn(v, PI_F);
  float m = fmodf(sft, TWO_PI_F);
  if (m != 0.f && m < 0.f) m = __fadd_rn(m, TWO_PI_F);
  return __fadd_rn(m, -PI_F);
}

__global__ void wrap_test_f32_kernel(const float* in, float* out, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = wrap_pi(in[i]);
}

template <int NJ>
__global__ __launch_bounds__(256) void head_update_kernel(UpdateArgs a) {
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= a.M) return;
  const int d = a.d, F = a.F;
  float v[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    v[j] = c < d ? a.g[(size_t)tok * d + c] : 0.f;
  }
  if (a.do_ln) row_layernorm<NJ>(v, lane, d, a.gamma, a.beta, a.ln_eps);
  // dense2: F dot products of length d, reduced over the wave; lane f keeps result f
  float mine = 0.f;
  for (int f = 0; f < F; ++f) {
    float partial = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      if (c < d) partial += v[j] * a.w2[(size_t)f * d + c];
    }
    partial = wave_sum(partial);
    if (lane == f) mine = partial + a.b2[f];
  }
  if (lane >= F) return;
  const size_t o = (size_t)tok * F + lane;
  if (a.eps_out) a.eps_out[o] = mine;
  if (!a.x_out) return;
  const int t = *a.t_dev;
  const float* noise = a.noise;
  float* hist = a.hist;
  unsigned long long seed = a.seed;
  long long seq_offset = a.seq_offset;
  int t_start = a.t_start, hist_every = 1;
  if (a.dyn) {
    noise = a.dyn->noise; hist = a.dyn->hist; seed = a.dyn->seed; seq_offset = a.dyn->seq_offset; t_start = a.dyn->t_start;
    hist_every = a.dyn->hist_every > 1 ? a.dyn->hist_every : 1;
  }
  const float c1 = a.coef[t], bt = a.coef[a.T + t], c3 = a.coef[2 * a.T + t], sg = a.coef[3 * a.T + t];
  // model_mean = sqrt_recip_alphas_t * (x - betas_t * eps / sqrt_one_minus_alphas_cumprod_t)   (sampling.py:62-67)
  float xn = __fmul_rn(c1, __fsub_rn(a.x[o], __fdiv_rn(__fmul_rn(bt, mine), c3)));
  if (t > 0) {  // sampling.py:69-75
    const float z = noise ? noise[(size_t)t * a.noise_stride + o]
                          : philox_normal(seed, t, seq_offset + tok / a.L, tok % a.L, lane);
    xn = __fadd_rn(xn, __fmul_rn(sg, z));
  }
  if ((a.angle_mask >> lane) & 1u) xn = wrap_pi(xn);
  a.x_out[o] = xn;
  if (hist && ((t_start - t + 1) % hist_every == 0 || t == 0))  // state j = t_start - t goes to row j / hist_every
    hist[(size_t)((t_start - t) / hist_every) * a.M * F + o] = xn;
}

template <int NV>
__global__ __launch_bounds__(256) void head_update16_kernel(UpdateArgs a) {
  constexpr int d = 64 * NV;
  extern __shared__ __attribute__((aligned(16))) float w2s[];  // [F][d]
  const int F = a.F;
  for (int i = threadIdx.x; i < F * d; i += 256) w2s[i] = a.w2[i];
  const int k = threadIdx.x & 15, g = threadIdx.x >> 4;
  float4 gm[NV], bt[NV];
  if (a.do_ln) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      gm[j] = *reinterpret_cast<const float4*>(a.gamma + 4 * (k + 16 * j));
      bt[j] = *reinterpret_cast<const float4*>(a.beta + 4 * (k + 16 * j));
    }
  }
  const float b2k = a.b2[k < F ? k : 0];
  // per-call values (see UpdateDyn)
  const int t = a.x_out ? *a.t_dev : 0;
  const float* noise = a.noise;
  float* hist = a.hist;
  unsigned long long seed = a.seed;
  long long seq_offset = a.seq_offset;
  int t_start = a.t_start, hist_every = 1;
  if (a.dyn) {
    noise = a.dyn->noise; hist = a.dyn->hist; seed = a.dyn->seed; seq_offset = a.dyn->seq_offset; t_start = a.dyn->t_start;
    hist_every = a.dyn->hist_every > 1 ? a.dyn->hist_every : 1;
  }
  float c1 = 0.f, btc = 0.f, c3 = 1.f, sg = 0.f;
  if (a.x_out) { c1 = a.coef[t]; btc = a.coef[a.T + t]; c3 = a.coef[2 * a.T + t]; sg = a.coef[3 * a.T + t]; }
  __syncthreads();
  for (int tg = blockIdx.x; tg * 16 < a.M; tg += gridDim.x) {
    const int tok = tg * 16 + g;
    const int tc = tok < a.M ? tok : a.M - 1;
    float4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = *reinterpret_cast<const float4*>(a.g + (size_t)tc * d + 4 * (k + 16 * j));
    if (a.do_ln) row16_layernorm<NV>(v, gm, bt, d, a.ln_eps);
    // dense2: F dot products of length d; lane f of the group keeps result f
    float mine = 0.f;
    for (int f = 0; f < F; ++f) {
      float partial = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const float4 w = *reinterpret_cast<const float4*>(w2s + f * d + 4 * (k + 16 * j));
        partial += (v[j].x * w.x + v[j].y * w.y) + (v[j].z * w.z + v[j].w * w.w);
      }
      partial = row16_sum(partial);
      mine = (k == f) ? partial + b2k : mine;
    }
    if (k < F && tok < a.M) {
      const size_t o = (size_t)tok * F + k;
      if (a.eps_out) a.eps_out[o] = mine;
      if (a.x_out) {
        // model_mean = sqrt_recip_alphas_t * (x - betas_t * eps / sqrt_one_minus_alphas_cumprod_t)   (sampling.py:62-67)
        float xn = __fmul_rn(c1, __fsub_rn(a.x[o], __fdiv_rn(__fmul_rn(btc, mine), c3)));
        if (t > 0) {  // sampling.py:69-75
          const float z = noise ? noise[(size_t)t * a.noise_stride + o]
                                : philox_normal(seed, t, seq_offset + tok / a.L, tok % a.L, k);
          xn = __fadd_rn(xn, __fmul_rn(sg, z));
        }
        if ((a.angle_mask >> k) & 1u) xn = wrap_pi(xn);
        a.x_out[o] = xn;
        if (hist && ((t_start - t + 1) % hist_every == 0 || t == 0))  // state j = t_start - t goes to row j / hist_every
    hist[(size_t)((t_start - t) / hist_every) * a.M * F + o] = xn;
      }
    }
  }
}

template <int NV>
static void launch_head_update16(const UpdateArgs& a, hipStream_t s) {
  int grid = (a.M + 15) / 16;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL((head_update16_kernel<NV>), dim3(grid), dim3(256), (size_t)a.F * 64 * NV * 4, s, a);
}

void launch_head_update(const UpdateArgs& a, hipStream_t s) {
  if (use_row16(a.d) && a.F <= 16) {
#define FD_H16(NV) case NV: launch_head_update16<NV>(a, s); return;
    switch (a.d / 64) { FD_H16(1) FD_H16(2) FD_H16(3) FD_H16(4) FD_H16(5) FD_H16(6) FD_H16(7) FD_H16(8) }
#undef FD_H16
  }
  const dim3 grid((a.M + 3) / 4), block(256);
  const int nj = (a.d + 63) / 64;
#define FD_HU(NJ) hipLaunchKernelGGL((head_update_kernel<NJ>), grid, block, 0, s, a)
  if (nj <= 1) FD_HU(1);
  else if (nj <= 3) FD_HU(3);
  else if (nj <= 6) FD_HU(6);
  else if (nj <= 12) FD_HU(12);
  else FD_HU(16);
#undef FD_HU
}

__global__ void step_advance_kernel(int* t_dev) { *t_dev -= 1; }
void launch_step_advance(int* t_dev, hipStream_t s) { hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, s, t_dev); }

void launch_wrap_test_f32(const float* in, float* out, long long n, hipStream_t s) {
  hipLaunchKernelGGL(wrap_test_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n);
}

}  // namespace fdmi
