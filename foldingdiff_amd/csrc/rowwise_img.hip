// HBM-bound row kernels of the row-image path (img_common.h):
//   build_rows     token-row table of a batch: sequences start at multiples of 8 rows; with `packed` only the
//                  first lens[b] positions of a sequence are rows at all (positions the reference computes and then
//                  cuts away, foldingdiff/sampling.py:56-58, :201-203)
//   embed_img      K1  Linear(F->d) + BertEmbeddings LayerNorm + time embedding (modelling.py:464-472, :157-170),
//                  written as the hi|lo row image the QK / V GEMMs stage with LDS-DMA
//   head_update_img  K8 tail + K9: AnglesPredictor.layer_norm + dense2 (modelling.py:206-207), the p_sample update
//                  (sampling.py:62-75), the per-feature wrap (sampling.py:119-130, utils.py:100-106), the history
//                  row, the non-finite guard and the step counter
//   f32 <-> image converters for the test hooks
// 16 lanes per token: lane k owns the 16-byte units k + 16 j of an image row (8 columns each), row reductions are four DPP steps.
#include <cstdlib>

#include "fdmi_kernels.h"
#include "img_common.h"

namespace fdmi {

namespace {

template <int LPT = 16>
__device__ __forceinline__ float row16_sum(float v) {  // sum over the LPT (8 or 16) neighbouring lanes of a token
#define FD_DPP_ADD(ctrl) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
  FD_DPP_ADD(0xB1);   // quad_perm [1,0,3,2]
  FD_DPP_ADD(0x4E);   // quad_perm [2,3,0,1]
  FD_DPP_ADD(0x141);  // row_half_mirror
  if constexpr (LPT == 16) FD_DPP_ADD(0x140);  // row_mirror
#undef FD_DPP_ADD
  return v;
}

// ---- LPT (8 or 16) lanes per token, a lane owns whole 16-byte UNITS (8 columns: unit k + LPT j of the row): every image access is
// a 16-byte hi piece + a 16-byte lo piece.  LPT = 8 (d_model <= 512): a wave holds 8 consecutive rows x 8 units, so each of
// its store / load instructions touches whole 128-byte lines of the grouped image; LPT = 16 covers d_model <= 1024.
// (Rounds 1-2 gave a lane 4-column groups = 8-byte pieces: 50 / 58 us for the two kernels against 32 / 42 us in fp32.)
struct Unit8 {
  float v[8];
};
// LayerNorm over the valid units of a row spread over 16 lanes (biased variance, two passes)
template <int NV, int LPT>
__device__ __forceinline__ void row16_layernorm(Unit8 (&x)[NV], const bool (&ok)[NV], int d, float eps, float& mean_out, float& rstd_out) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    float t = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) t += x[j].v[e];
    s += ok[j] ? t : 0.f;
  }
  const float mean = row16_sum<LPT>(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    float t = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      x[j].v[e] -= mean;
      t += x[j].v[e] * x[j].v[e];
    }
    q += ok[j] ? t : 0.f;
  }
  mean_out = mean;
  rstd_out = 1.0f / sqrtf(row16_sum<LPT>(q) / (float)d + eps);
}
__device__ __forceinline__ void load8(const float* p, float (&o)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
// 8 values -> the unit's hi and lo 16-byte pieces (value * s = hi + lo)
__device__ __forceinline__ void split8(const float (&v)[8], float s, u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned h, l;
    split_pair(v[2 * i] * s, v[2 * i + 1] * s, h, l);
    hi[i] = h;
    lo[i] = l;
  }
}
__device__ __forceinline__ void join8(const u32x4& hi, const u32x4& lo, float inv_s, float (&v)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = (h2f_lo(hi[i]) + h2f_lo(lo[i])) * inv_s;
    v[2 * i + 1] = (h2f_hi(hi[i]) + h2f_hi(lo[i])) * inv_s;
  }
}

__device__ __forceinline__ void split4(const float4& v, float s, u32x2& hi, u32x2& lo) {
  const float x[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
  _Float16 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = (_Float16)x[i];
    b[i] = (_Float16)(x[i] - (float)a[i]);
  }
  hi = u32x2{pack_h2(a[0], a[1]), pack_h2(a[2], a[3])};
  lo = u32x2{pack_h2(b[0], b[1]), pack_h2(b[2], b[3])};
}

__device__ __forceinline__ float4 join4(const u32x2& hi, const u32x2& lo, float inv_s) {
  return make_float4((h2f_lo(hi[0]) + h2f_lo(lo[0])) * inv_s, (h2f_hi(hi[0]) + h2f_hi(lo[0])) * inv_s,
                     (h2f_lo(hi[1]) + h2f_lo(lo[1])) * inv_s, (h2f_hi(hi[1]) + h2f_hi(lo[1])) * inv_s);
}

// ---------------------------------------------------------------- token-row table
// One block.  seq_row0[b] = sum_{b' < b} ceil8(rows(b')), rows(b) = packed ? lens[b] : L.
__global__ __launch_bounds__(1024) void build_rows_kernel(const int* __restrict__ lens, int B, int L, int packed,
                                                          int* __restrict__ seq_row0, int* __restrict__ nrow,
                                                          int* __restrict__ dims) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int per = (B + 1023) / 1024;
  const int b0 = tid * per, b1 = b0 + per < B ? b0 + per : B;
  int s = 0;
  for (int b = b0; b < b1; ++b) {
    const int n = packed ? min(max(lens[b], 1), L) : L;  // (device-resident lengths are not validated by the host)
    s += (n + 7) & ~7;
  }
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i < 1024; ++i) {
      const int v = part[i];
      part[i] = run;
      run += v;
    }
    seq_row0[B] = run;
    dims[0] = run;
    dims[1] = (run + 127) & ~127;
  }
  __syncthreads();
  int run = part[tid];
  for (int b = b0; b < b1; ++b) {
    const int n = packed ? min(max(lens[b], 1), L) : L;  // (device-resident lengths are not validated by the host)
    seq_row0[b] = run;
    nrow[b] = n;
    run += (n + 7) & ~7;
  }
}

__global__ __launch_bounds__(256) void fill_rows_kernel(const int* __restrict__ seq_row0, int B, int cap,
                                                        int2* __restrict__ rowinfo) {
  const int total = seq_row0[B];
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const int r0 = seq_row0[b], r1 = seq_row0[b + 1];
    for (int r = r0 + threadIdx.x; r < r1; r += blockDim.x) rowinfo[r] = make_int2(b, r - r0);
  }
  for (int r = total + blockIdx.x * blockDim.x + threadIdx.x; r < cap; r += gridDim.x * blockDim.x)
    rowinfo[r] = make_int2(-1, -1);
}

// ---------------------------------------------------------------- embed (K1)
// LDS: w_in transposed [F][d], then b_in | gamma | beta | time embedding of this step ([4][d]): a lane reads its units' 8-column
// slices from there per token (NV up to 8 units per lane: d <= 1024)
template <int NV, int LPT>
__global__ __launch_bounds__(256) void embed_img_kernel(EmbedImgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wT[];  // [F][d] + [4][d]
  const int d = a.d, F = a.F, nu = d >> 3;
  float* par = wT + F * d;
  const int t = a.tslot[0];
  for (int i = threadIdx.x; i < F * d; i += 256) {
    const int f = i / d, c = i - f * d;
    wT[i] = a.w_in[c * F + f];
  }
  for (int i = threadIdx.x; i < d; i += 256) {
    par[i] = a.b_in[i];
    par[d + i] = a.gamma[i];
    par[2 * d + i] = a.beta[i];
    par[3 * d + i] = a.time_table[(size_t)t * d + i];
  }
  const int k = threadIdx.x & (LPT - 1), g = threadIdx.x / LPT;
  if (blockIdx.x == 0 && threadIdx.x == 0) a.tslot[1] = t;  // the step's other kernels read slot 1 (see head_update_img)
  bool ok[NV];
  int col[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    ok[j] = k + LPT * j < nu;
    col[j] = ok[j] ? 8 * (k + LPT * j) : 0;
  }
  __syncthreads();
  const int rows = a.dims[1];  // every row of the padded range is written (pad rows: zeros)
  const int nb = d >> 5;
  for (int tg = blockIdx.x; tg * (256 / LPT) < rows; tg += gridDim.x) {
    const int row = tg * (256 / LPT) + g;
    if (row >= rows) continue;  // (no barriers below)
    const int2 ri = a.rowinfo[row];
    const bool real = ri.x >= 0 && ri.y < a.nrow[ri.x >= 0 ? ri.x : 0];
    const size_t xo = real ? ((size_t)ri.x * a.L + ri.y) * F : 0;
    Unit8 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) load8(par + col[j], v[j].v);
    for (int f = 0; f < F; ++f) {
      const float xf = a.x[xo + f];
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        float w[8];
        load8(wT + f * d + col[j], w);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j].v[e] += xf * w[e];
      }
    }
    if (a.pos_emb) {  // absolute positions only (modelling.py:164-166)
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        float pe[8];
        const int pid = !real ? 0 : (a.pos_ids ? a.pos_ids[(size_t)ri.x * a.L + ri.y] : ri.y);
        load8(a.pos_emb + (size_t)pid * d + col[j], pe);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j].v[e] += pe[e];
      }
    }
    float mean, rstd;
    row16_layernorm<NV, LPT>(v, ok, d, a.eps, mean, rstd);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      if (!ok[j]) continue;
      float gm[8], bt[8], tt[8], o[8];
      load8(par + d + col[j], gm);
      load8(par + 2 * d + col[j], bt);
      load8(par + 3 * d + col[j], tt);
      // time embedding added AFTER the LayerNorm (modelling.py:472); pad rows are zero rows
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = real ? (v[j].v[e] * rstd * gm[e] + bt[e]) + tt[e] : 0.f;
      u32x4 hi, lo;
      split8(o, a.out_scale, hi, lo);
      const int u = k + LPT * j;
      unsigned char* blk = a.h + img_unit_offset(row, nb, u >> 2, u & 3);
      *reinterpret_cast<u32x4*>(blk) = hi;
      *reinterpret_cast<u32x4*>(blk + 4 * 512) = lo;
    }
  }
}

// LayerNorm of fp32 rows -> image (models with d_model > 384: the LayerNorm row does not fit one 384-column GEMM tile, so the
// projection writes dense + bias + residual as fp32 rows and this kernel normalises them; BertSelfOutput / BertOutput)
template <int NV, int LPT>
__global__ __launch_bounds__(256) void ln_f32_img_kernel(const float* __restrict__ src, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, const int* __restrict__ dims,
                                                          unsigned char* __restrict__ out, int d, float out_scale) {
  const int nu = d >> 3, nb = d >> 5;
  const int k = threadIdx.x & (LPT - 1), g = threadIdx.x / LPT;
  bool ok[NV];
  int col[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    ok[j] = k + LPT * j < nu;
    col[j] = ok[j] ? 8 * (k + LPT * j) : 0;
  }
  const int rows = dims[1];
  for (int tg = blockIdx.x; tg * (256 / LPT) < rows; tg += gridDim.x) {
    const int row = tg * (256 / LPT) + g;
    if (row >= rows) continue;
    Unit8 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) load8(src + (size_t)row * d + col[j], v[j].v);
    float mean, rstd;
    row16_layernorm<NV, LPT>(v, ok, d, eps, mean, rstd);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      if (!ok[j]) continue;
      float gm[8], bt[8], o[8];
      load8(gamma + col[j], gm);
      load8(beta + col[j], bt);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = v[j].v[e] * rstd * gm[e] + bt[e];
      u32x4 hi, lo;
      split8(o, out_scale, hi, lo);
      const int u = k + LPT * j;
      unsigned char* blk = out + img_unit_offset(row, nb, u >> 2, u & 3);
      *reinterpret_cast<u32x4*>(blk) = hi;
      *reinterpret_cast<u32x4*>(blk + 4 * 512) = lo;
    }
  }
}

// ---------------------------------------------------------------- Philox4x32-10 (same stream as rowwise.hip)
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

// wrap to [-pi, pi) exactly as torch evaluates modulo_with_wrapped_range on fp32 (utils.py:100-106)
__device__ __forceinline__ float wrap_pi(float v) {
  const float PI_F = 3.14159274101257324f, TWO_PI_F = 6.28318548202514648f;
  const float sft = __fadd_rn(v, PI_F);
  float m = fmodf(sft, TWO_PI_F);
  if (m != 0.f && m < 0.f) m = __fadd_rn(m, TWO_PI_F);
  return __fadd_rn(m, -PI_F);
}

// ---------------------------------------------------------------- head tail + p_sample update (K8/K9)
template <int NV, int LPT>
__global__ __launch_bounds__(256) void head_update_img_kernel(UpdateArgs a, HeadImgArgs ia) {
  extern __shared__ __attribute__((aligned(16))) float w2s[];  // [F][d]
  const int d = a.d, F = a.F, nu = d >> 3, nb = d >> 5;
  float* par = w2s + F * d;  // gamma | beta
  for (int i = threadIdx.x; i < F * d; i += 256) w2s[i] = a.w2[i];
  for (int i = threadIdx.x; i < d; i += 256) {
    par[i] = a.do_ln ? a.gamma[i] : 1.f;
    par[d + i] = a.do_ln ? a.beta[i] : 0.f;
  }
  const int k = threadIdx.x & (LPT - 1), g = threadIdx.x / LPT;
  bool ok[NV];
  int col[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    ok[j] = k + LPT * j < nu;
    col[j] = ok[j] ? 8 * (k + LPT * j) : 0;
  }
  const float b2k = a.b2[k < F ? k : 0];
  // per-call values (see UpdateDyn)
  const int t = a.x_out ? ia.tslot[1] : 0;
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
  const int rows = ia.dims[0];
  const size_t BLF = (size_t)a.M * F;  // elements of one [B][L][F] state (a.M = B * L)
  bool bad = false;
  for (int tg = blockIdx.x; tg * (256 / LPT) < rows; tg += gridDim.x) {
    const int row = tg * (256 / LPT) + g;
    const int2 ri = row < rows ? ia.rowinfo[row] : make_int2(-1, -1);
    const bool real = ri.x >= 0 && ri.y < ia.nrow[ri.x >= 0 ? ri.x : 0];
    const int grow = row < rows ? row : 0;
    Unit8 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int u = ok[j] ? k + LPT * j : 0;
      const unsigned char* blk = ia.g + img_unit_offset(grow, nb, u >> 2, u & 3);
      join8(*reinterpret_cast<const u32x4*>(blk), *reinterpret_cast<const u32x4*>(blk + 4 * 512), ia.g_inv, v[j].v);
    }
    if (a.do_ln) {
      float mean, rstd;
      row16_layernorm<NV, LPT>(v, ok, d, a.ln_eps, mean, rstd);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        float gm[8], bt[8];
        load8(par + col[j], gm);
        load8(par + d + col[j], bt);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j].v[e] = v[j].v[e] * rstd * gm[e] + bt[e];
      }
    }
    float mine = 0.f;
    for (int f = 0; f < F; ++f) {
      float partial = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        float w[8];
        load8(w2s + f * d + col[j], w);
        float pj = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e += 2) pj += v[j].v[e] * w[e] + v[j].v[e + 1] * w[e + 1];
        partial += ok[j] ? pj : 0.f;
      }
      partial = row16_sum<LPT>(partial);
      mine = (k == f) ? partial + b2k : mine;
    }
    if (k < F && real) {
      const size_t o = ((size_t)ri.x * a.L + ri.y) * F + k;
      if (a.eps_out) a.eps_out[o] = mine;
      bad |= !(__builtin_fabsf(mine) <= 3.0e38f);  // inf or NaN
      if (a.x_out) {
        // model_mean = sqrt_recip_alphas_t * (x - betas_t * eps / sqrt_one_minus_alphas_cumprod_t)   (sampling.py:62-67)
        float xn = __fmul_rn(c1, __fsub_rn(a.x[o], __fdiv_rn(__fmul_rn(btc, mine), c3)));
        if (t > 0) {  // sampling.py:69-75
          const float z = noise ? noise[(size_t)t * a.noise_stride + o] : philox_normal(seed, t, seq_offset + ri.x, ri.y, k);
          xn = __fadd_rn(xn, __fmul_rn(sg, z));
        }
        if ((a.angle_mask >> k) & 1u) xn = wrap_pi(xn);
        a.x_out[o] = xn;
        if (hist && ((t_start - t + 1) % hist_every == 0 || t == 0))  // state j = t_start - t goes to row j / hist_every
          hist[(size_t)((t_start - t) / hist_every) * BLF + o] = xn;
      }
    }
  }
  if (bad) atomicOr(ia.flag, 1);  // SURVEY 5 failure detection: a non-finite prediction poisons everything after it
  if (ia.advance && blockIdx.x == 0 && threadIdx.x == 0) ia.tslot[0] = t - 1;  // next step (read by embed_img only)
}

// ---------------------------------------------------------------- converters (test hooks, debug dumps)
__global__ void f32_to_img_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, long long rows, int K,
                                  long long src_rows, float s) {
  const long long n4 = rows * (K >> 2);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / (K >> 2);
    const int grp = (int)(i - r * (K >> 2));
    const float4 v = r < src_rows ? *reinterpret_cast<const float4*>(src + r * K + 4 * grp) : make_float4(0.f, 0.f, 0.f, 0.f);
    u32x2 hi, lo;
    split4(v, s, hi, lo);
    unsigned char* blk = dst + img_unit_offset(r, K >> 5, grp >> 3, (grp & 7) >> 1) + 8 * (grp & 1);
    *reinterpret_cast<u32x2*>(blk) = hi;
    *reinterpret_cast<u32x2*>(blk + 4 * 512) = lo;
  }
}

__global__ void img_to_f32_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, long long rows, int K,
                                  float inv_s) {
  const long long n4 = rows * (K >> 2);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / (K >> 2);
    const int grp = (int)(i - r * (K >> 2));
    const unsigned char* blk = src + img_unit_offset(r, K >> 5, grp >> 3, (grp & 7) >> 1) + 8 * (grp & 1);
    *reinterpret_cast<float4*>(dst + r * K + 4 * grp) =
        join4(*reinterpret_cast<const u32x2*>(blk), *reinterpret_cast<const u32x2*>(blk + 4 * 512), inv_s);
  }
}

// debug: q / k (grouped images per (sequence, head)) or v^T blocks -> fp32 [BH][LTOT][32]
__global__ void qkv_unpack_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, long long BH, int LTOT,
                                  int LP, int rowbytes, int is_vt, float inv_s) {
  const long long n = BH * LTOT * 32;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int dd = (int)(i & 31);
    const long long row = i >> 5;  // bh * LTOT + l
    const long long bh = row / LTOT;
    const int l = (int)(row - bh * LTOT);
    const _Float16 *hp, *lp;
    if (is_vt == 1) {  // [bh][l / 32][d][128 B], 8-byte unit u at position u ^ vt_swz(d)
      const int kb = l >> 5, kl = l & 31, sz = vt_swz(dd);
      const unsigned char* r = src + (((bh * (LTOT >> 5) + kb) * 32) + dd) * (size_t)128;
      hp = reinterpret_cast<const _Float16*>(r + (((kl >> 2) ^ sz) << 3)) + (kl & 3);
      lp = reinterpret_cast<const _Float16*>(r + (((8 + (kl >> 2)) ^ sz) << 3)) + (kl & 3);
    } else {  // q / k: grouped image per (sequence, head), [l / 32][unit][l % 32][16 B]
      const unsigned char* g = src + bh * (size_t)LTOT * rowbytes;
      hp = reinterpret_cast<const _Float16*>(g + img_unit_offset(l, 1, 0, dd >> 3)) + (dd & 7);
      lp = reinterpret_cast<const _Float16*>(g + img_unit_offset(l, 1, 0, 4 + (dd >> 3))) + (dd & 7);
    }
    dst[i] = ((float)*hp + (float)*lp) * inv_s;
  }
}

int blocks_for(long long n) {
  long long b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

void launch_build_rows(const int* lens, int B, int L, int packed, int cap, int* seq_row0, int* nrow, int2* rowinfo,
                       int* dims, hipStream_t s) {
  hipLaunchKernelGGL(build_rows_kernel, dim3(1), dim3(1024), 0, s, lens, B, L, packed, seq_row0, nrow, dims);
  int grid = B < 1024 ? B : 1024;
  hipLaunchKernelGGL(fill_rows_kernel, dim3(grid), dim3(256), 0, s, seq_row0, B, cap, rowinfo);
}

// lanes per token: 8 when the row fits 8 lanes x 8 units (and, for the head kernel, a lane per output feature), else 16
#define FD_ROW_SWITCH(KERNEL, ...)                                                                              \
  do {                                                                                                          \
    if (lpt == 8) {                                                                                             \
      switch (nv) {                                                                                             \
        case 1: hipLaunchKernelGGL((KERNEL<1, 8>), dim3(grid), dim3(256), smem, s, __VA_ARGS__); return;        \
        case 2: hipLaunchKernelGGL((KERNEL<2, 8>), dim3(grid), dim3(256), smem, s, __VA_ARGS__); return;        \
        case 3: hipLaunchKernelGGL((KERNEL<3, 8>), dim3(grid), dim3(256), smem, s, __VA_ARGS__); return;        \
        case 4: hipLaunchKernelGGL((KERNEL<4, 8>), dim3(grid), dim3(256), smem, s, __VA_ARGS__); return;        \
        case 5: hipLaunchKernelGGL((KERNEL<5, 8>), dim3(grid), dim3(256), smem, s, __VA_ARGS__); return;        \
        case 6: hipLaunchKernelGGL((KERNEL<6, 8>), dim3(grid), dim3(256), smem, s, __VA_ARGS__); return;        \
        case 7: hipLaunchKernelGGL((KERNEL<7, 8>), dim3(grid), dim3(256), smem, s, __VA_ARGS__); return;        \
        default: hipLaunchKernelGGL((KERNEL<8, 8>), dim3(grid), dim3(256), smem, s, __VA_ARGS__); return;       \
      }                                                                                                         \
    }                                                                                                           \
    switch (nv) {                                                                                               \
      case 1: hipLaunchKernelGGL((KERNEL<1, 16>), dim3(grid), dim3(256), smem, s, __VA_ARGS__); return;         \
      case 2: hipLaunchKernelGGL((KERNEL<2, 16>), dim3(grid), dim3(256), smem, s, __VA_ARGS__); return;         \
      case 3: hipLaunchKernelGGL((KERNEL<3, 16>), dim3(grid), dim3(256), smem, s, __VA_ARGS__); return;         \
      case 4: hipLaunchKernelGGL((KERNEL<4, 16>), dim3(grid), dim3(256), smem, s, __VA_ARGS__); return;         \
      case 5: hipLaunchKernelGGL((KERNEL<5, 16>), dim3(grid), dim3(256), smem, s, __VA_ARGS__); return;         \
      case 6: hipLaunchKernelGGL((KERNEL<6, 16>), dim3(grid), dim3(256), smem, s, __VA_ARGS__); return;         \
      case 7: hipLaunchKernelGGL((KERNEL<7, 16>), dim3(grid), dim3(256), smem, s, __VA_ARGS__); return;         \
      default: hipLaunchKernelGGL((KERNEL<8, 16>), dim3(grid), dim3(256), smem, s, __VA_ARGS__); return;        \
    }                                                                                                           \
  } while (0)

// blocks of the grid-stride row kernels (each block first fills its LDS parameter image).  def_cap: 1024 measured best for the head
// kernel (16 lanes per row) and the LayerNorm kernel, 512 for the embed kernel (8 lanes per row: 33.5 us against 36.1 at C2,
// profiles/r05_seq_attn_notes.log); FDMI_ROW_GRID overrides all of them
static int row_grid(int max_rows, int lpt, int def_cap = 1024) {
  static const int env_cap = [] { const char* e = getenv("FDMI_ROW_GRID"); return e ? atoi(e) : 0; }();
  const int cap = env_cap > 0 ? env_cap : def_cap;
  const int per = 256 / lpt;
  int grid = (max_rows + per - 1) / per;
  return grid > cap ? cap : (grid < 1 ? 1 : grid);
}
static int row_lpt(int d, int F) {  // F > 0: the head kernel (one lane per output feature; measured faster with 16 lanes at d = 384)
  static const int force = [] { const char* e = getenv("FDMI_ROW_LPT"); return e ? atoi(e) : 0; }();
  if (force == 16 || d > 512 || F > 8) return 16;
  if (force == 8) return 8;
  return F > 0 ? 16 : 8;
}

void launch_embed_img(const EmbedImgArgs& a, int max_rows, hipStream_t s) {
  const int lpt = row_lpt(a.d, 0), nv = (a.d / 8 + lpt - 1) / lpt, grid = row_grid(max_rows, lpt, lpt == 8 ? 512 : 1024);
  const size_t smem = (size_t)(a.F + 4) * a.d * 4;
  FD_ROW_SWITCH(embed_img_kernel, a);
}

void launch_head_update_img(const UpdateArgs& a, const HeadImgArgs& ia, int max_rows, hipStream_t s) {
  const int lpt = row_lpt(a.d, a.F), nv = (a.d / 8 + lpt - 1) / lpt, grid = row_grid(max_rows, lpt);
  const size_t smem = (size_t)(a.F + 2) * a.d * 4;
  FD_ROW_SWITCH(head_update_img_kernel, a, ia);
}

void launch_ln_f32_img(const float* src, const float* gamma, const float* beta, float eps, const int* dims, void* out, int d,
                       float out_scale, int max_rows, hipStream_t s) {
  const int lpt = row_lpt(d, 0), nv = (d / 8 + lpt - 1) / lpt, grid = row_grid(max_rows, lpt);
  const size_t smem = 0;
  FD_ROW_SWITCH(ln_f32_img_kernel, src, gamma, beta, eps, dims, static_cast<unsigned char*>(out), d, out_scale);
}
#undef FD_ROW_SWITCH

void launch_f32_to_img(const float* src, void* dst, long long rows, int K, long long src_rows, float scale, hipStream_t s) {
  hipLaunchKernelGGL(f32_to_img_kernel, dim3(blocks_for(rows * (K / 4))), dim3(256), 0, s, src,
                     static_cast<unsigned char*>(dst), rows, K, src_rows, scale);
}

void launch_qkv_unpack(const void* src, float* dst, long long BH, int LTOT, int LP, int rowbytes, int is_vt, float scale,
                       hipStream_t s) {
  hipLaunchKernelGGL(qkv_unpack_kernel, dim3(blocks_for(BH * LTOT * 32)), dim3(256), 0, s,
                     static_cast<const unsigned char*>(src), dst, BH, LTOT, LP, rowbytes, is_vt, 1.0f / scale);
}

// N2: one block per (item, stored state); the item's len * F values of that state are contiguous in both buffers.
// Arithmetic of foldingdiff/sampling.py:218-222 on float32 arrays: s + offset (one rounded add), then for angular features
// utils.modulo_with_wrapped_range(., -pi, pi) = wrap_pi -- the same bits as numpy's float32 operations.
__global__ __launch_bounds__(256) void shift_trim_kernel(ShiftTrimArgs a) {
  const int i = blockIdx.x / a.rows, j = blockIdx.x - i * a.rows;
  const int n = a.lens[i] * a.F;
  const float* src = a.traj + ((size_t)j * a.B + i) * a.L * a.F;
  float* dst = a.out + a.item_off[i] + (size_t)j * n;
  for (int idx = threadIdx.x; idx < n; idx += 256) {
    float v = src[idx];
    if (a.has_offset) {
      const int f = idx % a.F;
      v = __fadd_rn(v, a.offset[f]);
      if ((a.angle_mask >> f) & 1u) v = wrap_pi(v);
    }
    dst[idx] = v;
  }
}
void launch_shift_trim(const ShiftTrimArgs& a, hipStream_t s) {
  if (a.B < 1 || a.rows < 1) return;
  hipLaunchKernelGGL(shift_trim_kernel, dim3((unsigned)a.B * a.rows), dim3(256), 0, s, a);
}

__global__ void wrap_test_img_kernel(const float* in, float* out, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = wrap_pi(in[i]);
}
void launch_wrap_test_img(const float* in, float* out, long long n, hipStream_t s) {
  hipLaunchKernelGGL(wrap_test_img_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n);
}

void launch_img_to_f32(const void* src, float* dst, long long rows, int K, float scale, hipStream_t s) {
  hipLaunchKernelGGL(img_to_f32_kernel, dim3(blocks_for(rows * (K / 4))), dim3(256), 0, s,
                     static_cast<const unsigned char*>(src), dst, rows, K, 1.0f / scale);
}

}  // namespace fdmi
