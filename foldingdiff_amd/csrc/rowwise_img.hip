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
// 16 lanes per token: lane k owns the 4-column groups k + 16 j (256-byte coalesced group accesses of fp32 data,
// 8-byte hi / lo pieces of an image block), row reductions are four DPP steps.
#include <cstdlib>

#include "fdmi_kernels.h"
#include "img_common.h"

namespace fdmi {

namespace {

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

// LayerNorm over the valid 4-column groups of a row spread over 16 lanes (biased variance, two passes)
template <int NV>
__device__ __forceinline__ void row16_layernorm(float4 (&v)[NV], const float4 (&gm)[NV], const float4 (&bt)[NV],
                                                const bool (&ok)[NV], int d, float eps) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) s += ok[j] ? (v[j].x + v[j].y) + (v[j].z + v[j].w) : 0.f;
  const float mean = row16_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    v[j].x -= mean; v[j].y -= mean; v[j].z -= mean; v[j].w -= mean;
    q += ok[j] ? (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w) : 0.f;
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
    const int n = packed ? lens[b] : L;
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
    const int n = packed ? lens[b] : L;
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
template <int NV>
__global__ __launch_bounds__(256) void embed_img_kernel(EmbedImgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wT[];  // [F][d]: w_in transposed
  const int d = a.d, F = a.F, ng = d >> 2;
  for (int i = threadIdx.x; i < F * d; i += 256) {
    const int f = i / d, c = i - f * d;
    wT[i] = a.w_in[c * F + f];
  }
  const int k = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int t = a.tslot[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) a.tslot[1] = t;  // the step's other kernels read slot 1 (see head_update_img)
  float4 bi[NV], gm[NV], bt[NV], tt[NV];
  bool ok[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    ok[j] = k + 16 * j < ng;
    const int c = ok[j] ? 4 * (k + 16 * j) : 0;
    bi[j] = *reinterpret_cast<const float4*>(a.b_in + c);
    gm[j] = *reinterpret_cast<const float4*>(a.gamma + c);
    bt[j] = *reinterpret_cast<const float4*>(a.beta + c);
    tt[j] = *reinterpret_cast<const float4*>(a.time_table + (size_t)t * d + c);
  }
  __syncthreads();
  const int rows = a.dims[1];  // every row of the padded range is written (pad rows: zeros)
  const int nb = d >> 5;
  for (int tg = blockIdx.x; tg * 16 < rows; tg += gridDim.x) {
    const int row = tg * 16 + g;
    if (row >= rows) continue;  // (no barriers below)
    const int2 ri = a.rowinfo[row];
    const bool real = ri.x >= 0 && ri.y < a.nrow[ri.x >= 0 ? ri.x : 0];
    const size_t xo = real ? ((size_t)ri.x * a.L + ri.y) * F : 0;
    float4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = bi[j];
    for (int f = 0; f < F; ++f) {
      const float xf = a.x[xo + f];
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const float4 w = *reinterpret_cast<const float4*>(wT + f * d + (ok[j] ? 4 * (k + 16 * j) : 0));
        v[j].x += xf * w.x; v[j].y += xf * w.y; v[j].z += xf * w.z; v[j].w += xf * w.w;
      }
    }
    if (a.pos_emb) {  // absolute positions only (modelling.py:164-166)
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const float4 pe = *reinterpret_cast<const float4*>(a.pos_emb + (size_t)(real ? ri.y : 0) * d + (ok[j] ? 4 * (k + 16 * j) : 0));
        v[j].x += pe.x; v[j].y += pe.y; v[j].z += pe.z; v[j].w += pe.w;
      }
    }
    row16_layernorm<NV>(v, gm, bt, ok, d, a.eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      if (!ok[j]) continue;
      const int grp = k + 16 * j;
      // time embedding added AFTER the LayerNorm (modelling.py:472); pad rows are zero rows
      const float4 o = real ? make_float4(v[j].x + tt[j].x, v[j].y + tt[j].y, v[j].z + tt[j].z, v[j].w + tt[j].w)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
      u32x2 hi, lo;
      split4(o, a.out_scale, hi, lo);
      unsigned char* blk = a.h + img_unit_offset(row, nb, grp >> 3, (grp & 7) >> 1) + 8 * (grp & 1);  // 4 columns = half a unit
      *reinterpret_cast<u32x2*>(blk) = hi;
      *reinterpret_cast<u32x2*>(blk + 4 * 512) = lo;
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
template <int NV>
__global__ __launch_bounds__(256) void head_update_img_kernel(UpdateArgs a, HeadImgArgs ia) {
  extern __shared__ __attribute__((aligned(16))) float w2s[];  // [F][d]
  const int d = a.d, F = a.F, ng = d >> 2, nb = d >> 5;
  for (int i = threadIdx.x; i < F * d; i += 256) w2s[i] = a.w2[i];
  const int k = threadIdx.x & 15, g = threadIdx.x >> 4;
  float4 gm[NV], bt[NV];
  bool ok[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    ok[j] = k + 16 * j < ng;
    const int c = ok[j] ? 4 * (k + 16 * j) : 0;
    gm[j] = a.do_ln ? *reinterpret_cast<const float4*>(a.gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    bt[j] = a.do_ln ? *reinterpret_cast<const float4*>(a.beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
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
  for (int tg = blockIdx.x; tg * 16 < rows; tg += gridDim.x) {
    const int row = tg * 16 + g;
    const int2 ri = row < rows ? ia.rowinfo[row] : make_int2(-1, -1);
    const bool real = ri.x >= 0 && ri.y < ia.nrow[ri.x >= 0 ? ri.x : 0];
    const int grow = row < rows ? row : 0;
    float4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int grp = ok[j] ? k + 16 * j : 0;
      const unsigned char* blk = ia.g + img_unit_offset(grow, nb, grp >> 3, (grp & 7) >> 1) + 8 * (grp & 1);
      v[j] = join4(*reinterpret_cast<const u32x2*>(blk), *reinterpret_cast<const u32x2*>(blk + 4 * 512), ia.g_inv);
    }
    if (a.do_ln) row16_layernorm<NV>(v, gm, bt, ok, d, a.ln_eps);
    float mine = 0.f;
    for (int f = 0; f < F; ++f) {
      float partial = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const float4 w = *reinterpret_cast<const float4*>(w2s + f * d + (ok[j] ? 4 * (k + 16 * j) : 0));
        partial += ok[j] ? (v[j].x * w.x + v[j].y * w.y) + (v[j].z * w.z + v[j].w * w.w) : 0.f;
      }
      partial = row16_sum(partial);
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
    if (is_vt == 1) {  // [bh][l / 32][d][128 B], 8-byte unit u at position u ^ ((d >> 1) & 15)
      const int kb = l >> 5, kl = l & 31, sz = (dd >> 1) & 15;
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

void launch_embed_img(const EmbedImgArgs& a, int max_rows, hipStream_t s) {
  int grid = (max_rows + 15) / 16;
  if (grid > 2048) grid = 2048;
  const size_t smem = (size_t)a.F * a.d * 4;
  const int nv = (a.d / 4 + 15) / 16;
#define FD_EI(NV) case NV: hipLaunchKernelGGL((embed_img_kernel<NV>), dim3(grid), dim3(256), smem, s, a); return;
  switch (nv) { FD_EI(1) FD_EI(2) FD_EI(3) FD_EI(4) FD_EI(5) FD_EI(6) }
#undef FD_EI
}

void launch_head_update_img(const UpdateArgs& a, const HeadImgArgs& ia, int max_rows, hipStream_t s) {
  int grid = (max_rows + 15) / 16;
  if (grid > 2048) grid = 2048;
  const size_t smem = (size_t)a.F * a.d * 4;
  const int nv = (a.d / 4 + 15) / 16;
#define FD_HI(NV) case NV: hipLaunchKernelGGL((head_update_img_kernel<NV>), dim3(grid), dim3(256), smem, s, a, ia); return;
  switch (nv) { FD_HI(1) FD_HI(2) FD_HI(3) FD_HI(4) FD_HI(5) FD_HI(6) }
#undef FD_HI
}

void launch_f32_to_img(const float* src, void* dst, long long rows, int K, long long src_rows, float scale, hipStream_t s) {
  hipLaunchKernelGGL(f32_to_img_kernel, dim3(blocks_for(rows * (K / 4))), dim3(256), 0, s, src,
                     static_cast<unsigned char*>(dst), rows, K, src_rows, scale);
}

void launch_qkv_unpack(const void* src, float* dst, long long BH, int LTOT, int LP, int rowbytes, int is_vt, float scale,
                       hipStream_t s) {
  hipLaunchKernelGGL(qkv_unpack_kernel, dim3(blocks_for(BH * LTOT * 32)), dim3(256), 0, s,
                     static_cast<const unsigned char*>(src), dst, BH, LTOT, LP, rowbytes, is_vt, 1.0f / scale);
}

void launch_img_to_f32(const void* src, float* dst, long long rows, int K, float scale, hipStream_t s) {
  hipLaunchKernelGGL(img_to_f32_kernel, dim3(blocks_for(rows * (K / 4))), dim3(256), 0, s,
                     static_cast<const unsigned char*>(src), dst, rows, K, 1.0f / scale);
}

}  // namespace fdmi
