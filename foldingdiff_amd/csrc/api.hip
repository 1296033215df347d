// C-ABI host side of libfdmi.so: model state, weight packing, workspaces, the
// hipGraph-captured reverse-diffusion loop and the measurement hooks.
// Boundary: include/fdmi.h (each entry point cites the reference function it replaces).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/fdmi.h"
#include "fdmi_kernels.h"

using namespace fdmi;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                               \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) return fail(FD_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                      __FILE__, __LINE__);                                          \
  } while (0)

struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
};

// split-precision image of one weight matrix (gemm_f16x3.hip)
struct SplitW {
  void* p = nullptr;
  float scale = 1.f;
};

struct LayerDev {
  SplitW demb_s;  // distance table as an fp16 hi|lo row image (row-image attention)
  // row-image path: weight images padded to 384 rows, and the (static, power-of-two) scales of this layer's
  // activation images -- derived from norm bounds of the weights at fd_finalize, so nothing can overflow fp16
  SplitW wqk_i, wv_i, wqkv_i, wo_i, wi_i, wd_i;  // wqkv_i: q | k | v rows in one image (n_heads % 6 == 0: one launch)
  SplitW wsa_i;  // the same q | k | v weights ordered per head for the 32-row fused projection + attention kernel (seq_attn.hip), or null
  SplitW wff_i;    // intermediate.dense + output.dense as ONE stream of ring stages (ffn16.hip), or null; scale = the first dense's
  float wff_scale_dn = 1.f;  // ... the second dense's
  SplitW wtail_i;  // attention.output.dense's stages in front of that stream (ffn16.hip TAIL), or null; scale = attention.output.dense's
  SplitW wsa16_i;  // ... for the 16-row kernel (seq_attn16.hip: rows of a head permuted into its operand tiles), or null
  float *bqk = nullptr, *bv = nullptr;  // bias slices of bqkv
  float s_h = 1.f, s_q = 1.f, s_k = 1.f, s_v = 1.f, s_a = 1.f, s_g = 1.f;
  float *wqkv = nullptr, *bqkv = nullptr, *demb = nullptr;
  float *wo = nullptr, *bo = nullptr, *ln1g = nullptr, *ln1b = nullptr;
  float *wi = nullptr, *bi = nullptr, *wd = nullptr, *bd = nullptr, *ln2g = nullptr, *ln2b = nullptr;
};

enum KClass {
  KC_EMBED = 0, KC_GEMM_QKV, KC_GEMM_V, KC_ATTN, KC_GEMM_OUT, KC_LN1, KC_GEMM_UP, KC_GEMM_DOWN, KC_LN2, KC_GEMM_HEAD,
  KC_HEAD_UPDATE, KC_ADVANCE, KC_SEQ_ATTN, KC_FFN, KC_TAIL, KC_COUNT
};
const char* const kClassName[KC_COUNT] = {
    "embed_ln_time", "gemm_qkv", "gemm_v", "attention", "gemm_attn_out", "layernorm_attn", "gemm_ffn_up",
    "gemm_ffn_down", "layernorm_ffn", "gemm_head_dense1", "head_update_wrap", "step_advance", "qkv_attention_fused", "ffn_fused", "attn_out_ffn_fused"};

struct Workspace {
  int B = 0, L = 0;
  float *x = nullptr, *eps = nullptr, *h = nullptr, *qkv = nullptr, *ctx = nullptr, *a = nullptr, *tmp = nullptr,
        *g = nullptr, *z = nullptr;
  int* lens = nullptr;
  int* t_dev = nullptr;  // [0] step index; the row-image kernels also use [1] (see rowwise_img.hip)
  UpdateDyn* dyn = nullptr;
  // row-image path (fdmi_kernels.h): images, per-(sequence, head) q / k / v^T, the token-row table
  bool img = false;
  int cap = 0, LPK = 0, LTOT = 0, NKT = 0;  // rows128 capacity = ceil128(B * ceil8(L)); key-tile geometry
  unsigned char *himg = nullptr, *aimg = nullptr, *cimg = nullptr, *gimg = nullptr, *qbuf = nullptr, *kbuf = nullptr,
                *vbuf = nullptr, *trash = nullptr;
  int2* rowinfo = nullptr;
  int *seq_row0 = nullptr, *nrow = nullptr, *dims = nullptr, *flag = nullptr;
  unsigned char* kmask = nullptr;  // [B][L] explicit key mask of fd_forward_ex (allocated on first use)
  int* pos_ids = nullptr;          // [B][L] explicit position ids of fd_forward_ex (allocated on first use)
  hipGraphExec_t graph = nullptr;
  int graph_fuse_ln = -2;  // option value the graph was captured with (-2: none)
  int graph_varlen = -1;   // ... and the row mode (packed rows launch the slice-capable GEMM instantiation)
  int graph_fuse_attn = -2;  // ... and the fused projection + attention choice
  int graph_fuse_ffn = -2;   // ... and the fused feed-forward choice
  int graph_rows_hint = -1;  // ... and the row count the launch-sequence choices were made for
  uint64_t last_use = 0;
  void release() {
    if (graph) (void)hipGraphExecDestroy(graph);
    graph = nullptr;
    for (void* p : {(void*)x, (void*)eps, (void*)h, (void*)qkv, (void*)ctx, (void*)a, (void*)tmp, (void*)g, (void*)z,
                    (void*)lens, (void*)t_dev, (void*)dyn, (void*)himg, (void*)aimg, (void*)cimg, (void*)gimg,
                    (void*)qbuf, (void*)kbuf, (void*)vbuf, (void*)trash, (void*)rowinfo, (void*)seq_row0, (void*)nrow,
                    (void*)dims, (void*)flag, (void*)kmask, (void*)pos_ids})
      if (p) (void)hipFree(p);
    *this = Workspace();
  }
};

constexpr long long kStampWords = 5 * 8 * 64 * 6 + 4 * 64 * 8 + 4 * 64 * 16 + 16384 + 8 * 16 * 16;  // (... + ffn16.hip's [8][16][16])  // (+ 32 K floats of register dumps, seq_attn.hip FDMI_SA_DUMP)

struct PendingEvent {
  int cls;
  hipEvent_t e0, e1;
};

}  // namespace

struct fd_model {
  fd_config cfg;
  int device = 0;
  hipStream_t stream = nullptr;
  std::map<std::string, HostTensor> host;
  bool finalized = false;
  int T = 0;
  int precision = 0;
  unsigned angle_mask = 0;
  // device weights
  std::vector<void*> allocs;
  float *w_in = nullptr, *b_in = nullptr, *pos_emb = nullptr, *emb_g = nullptr, *emb_b = nullptr;
  std::vector<LayerDev> layers;
  float *hd_w1 = nullptr, *hd_b1 = nullptr, *hd_g = nullptr, *hd_b = nullptr, *hd_w2 = nullptr, *hd_b2 = nullptr;
  SplitW hd_w1_i;
  float s_hfinal = 1.f, s_hg = 1.f;  // row-image scales: last hidden state, head activation
  bool img = false;                  // FD_PREC_F16X3 runs on the row-image kernels
  float *coef = nullptr, *time_table = nullptr;
  // options
  int fuse_ln = -1;  // -1 auto: LN-fused GEMMs with fp16x3 (measured 9.37 vs 10.15 ms/step), not with fp32 (slower there)
  int use_graph = 1;
  unsigned long long* stamps = nullptr;  // debug cycle stamps (FDMI_STAMPS=1): gemm [5][8][64][6], attention [4][64][8], fused attention [4][64][16]
  int debug_stop = 0;  // row-image path: stop a step after this many launches (debug dumps; 0 = off)
  int debug_layer = 0; // layer whose scales fd_debug_read uses
  int split_qkv = 0; // row-image path: 1 = q | k and v^T as two launches even when one would do (A/B, tests)
  int varlen = 0;    // row-image path: only the first lens[b] positions of a sequence are token rows
  int fuse_attn = -1;  // row-image path: q | k | v projection + attention as ONE kernel per sequence (seq_attn.hip): -1 auto (padded rows of
                     // 97 .. 128 positions), 0 never, 1 wherever the kernel applies (packed rows too)
  int fuse_ffn = -1;   // row-image path: BertIntermediate + BertOutput as ONE kernel (ffn16.hip): -1 auto (whole rounds of 128-row passes: 2),
                     // 0 never, 1 wherever the kernel applies, 2 with BertSelfOutput in front of it in the same launch
  int rows_hint = 0;   // packed rows: the caller's exact count of token rows of the next calls (sum of the lengths rounded up to 8), or 0 = unknown;
                     // the auto choices of the fused kernels then go by it instead of the bound B * ceil8(L) (the count itself lives in device memory)
  // workspaces (buffers + captured graph) are kept per (B, L): sample_length()-driven sampling and ragged chunks
  // alternate between a few shapes
  std::vector<Workspace> cache;
  uint64_t use_clock = 0;
  Workspace ws;      // the current one (moved in and out of `cache`)
  void* comm = nullptr;  // ncclComm_t (fd_comm_init)
  int comm_world = 0, comm_rank = 0;
  UpdateDyn dyn_host{};  // per-run values of the sampling loop in progress (fd_sample_begin_dev .. fd_sample_end_dev)
  int run_t = -1;        // next timestep of that run (-1: none left)
  bool run_open = false; // between fd_sample_begin_dev and fd_sample_end_dev, with no other call on the model in between
  int run_B = 0, run_L = 0;  // its shape: fd_sample_steps_dev / _end_dev refuse to run on another workspace
  // profiling
  int profile_every = 0;
  double prof_ms[KC_COUNT] = {0};
  int64_t prof_n[KC_COUNT] = {0};
  double prof_flops[KC_COUNT] = {0}, prof_bytes[KC_COUNT] = {0};
  std::vector<PendingEvent> pending;
  std::vector<hipEvent_t> event_pool;
};

namespace {

int dev_alloc(fd_model* m, float** out, size_t n_floats) {
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, n_floats * sizeof(float)));
  m->allocs.push_back(p);
  *out = static_cast<float*>(p);
  return FD_OK;
}

int upload(fd_model* m, float** out, const float* src, size_t n) {
  int rc = dev_alloc(m, out, n);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(*out, src, n * sizeof(float), hipMemcpyHostToDevice));
  return FD_OK;
}

// W [N][K] fp32 -> [Npad128][K/32][hi x32 | lo x32] fp16 with w*scale = hi + lo (gemm_f16x3.hip).
// scale = the power of two that puts max|w|*scale in [8192, 16384): every lo of a weight that
// matters is a normal fp16 and nothing overflows.
void pack_split_weight(const float* W, int N, int K, std::vector<uint16_t>* out, float* scale, int row_pad = 128) {
  float mx = 0.f;
  for (size_t i = 0; i < (size_t)N * K; ++i) mx = std::fmax(mx, std::fabs(W[i]));
  float s = 1.f;
  if (mx > 0.f && std::isfinite(mx)) s = std::exp2(std::floor(std::log2(16384.0f / mx)));
  *scale = s;
  const int npad = (N + row_pad - 1) / row_pad * row_pad, nk = K / 32;
  out->assign((size_t)npad * nk * 64, 0);
  for (int n = 0; n < N; ++n)
    for (int kt = 0; kt < nk; ++kt) {
      uint16_t* dst = out->data() + ((size_t)n * nk + kt) * 64;
      for (int j = 0; j < 32; ++j) {
        const float xs = W[(size_t)n * K + kt * 32 + j] * s;
        const _Float16 hi = (_Float16)xs;
        const _Float16 lo = (_Float16)(xs - (float)hi);
        memcpy(dst + j, &hi, 2);
        memcpy(dst + 32 + j, &lo, 2);
      }
    }
}

// Row-image GEMM weights, ordered [384-row tile][k-tile][48 KiB = the workgroup's LDS stage, byte for byte]: the loader copies a
// k-tile with lane-linear LDS-DMA from CONSECUTIVE cache lines.  The stage is 48 pieces of 8 rows, each piece unit-major:
// [piece j][position p][row % 8][16 B] with position p holding 16-byte unit p ^ (j & 1) of the row's block (the layout the
// compute waves' fragment reads are bank-conflict free on; gemm_img.hip uses the same for the activation pieces).
// Rows padded to whole tiles with zeros.
void pack_weight_tiles(const float* W, int N, int K, std::vector<uint16_t>* out, float* scale) {
  std::vector<uint16_t> rm;
  pack_split_weight(W, N, K, &rm, scale, 384);
  const int npad = (N + 383) / 384 * 384, nk = K / 32;
  out->assign(rm.size(), 0);
  for (int n = 0; n < npad; ++n)
    for (int kt = 0; kt < nk; ++kt) {
      const int R = n % 384, j = R >> 3, r = R & 7;
      uint16_t* stage = out->data() + ((size_t)(n / 384) * nk + kt) * 384 * 64;
      const uint16_t* blk = rm.data() + ((size_t)n * nk + kt) * 64;
      for (int u = 0; u < 8; ++u) memcpy(stage + (j * 1024 + (((u ^ (j & 1)) * 8 + r) << 4)) / 2, blk + u * 8, 16);
    }
}

int upload_split(fd_model* m, SplitW* dst, const float* W, int N, int K, int row_pad = 128) {
  std::vector<uint16_t> img;
  if (row_pad == 384) pack_weight_tiles(W, N, K, &img, &dst->scale);
  else pack_split_weight(W, N, K, &img, &dst->scale, row_pad);
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, img.size() * 2));
  m->allocs.push_back(p);
  HIP_TRY(hipMemcpy(p, img.data(), img.size() * 2, hipMemcpyHostToDevice));
  dst->p = p;
  return FD_OK;
}

// seq_attn.hip's weights: the q | k | v projection [3 d][d] ordered per head, [head][k-tile][unit 0-7][96 rows: q_h | k_h | v_h][16 B]
// (a k-tile of a head = one 12 KiB LDS ring stage, byte for byte, unit-major), split at the SAME scale as wqkv_i (one power of two for
// the whole matrix), so the fused kernel multiplies the very same fp16 hi / lo values as the tile GEMM.
int upload_seq_attn_weights(fd_model* m, SplitW* dst, const float* W, int d) {
  std::vector<uint16_t> rm, img;
  pack_split_weight(W, 3 * d, d, &rm, &dst->scale, 384);
  const int nk = d / 32, H = d / 32;
  img.assign((size_t)H * nk * 96 * 64, 0);
  for (int h = 0; h < H; ++h)
    for (int kt = 0; kt < nk; ++kt) {
      uint16_t* stage = img.data() + ((size_t)h * nk + kt) * 96 * 64;
      for (int r = 0; r < 96; ++r) {
        const int n = (r / 32) * d + h * 32 + (r % 32);
        const uint16_t* blk = rm.data() + ((size_t)n * nk + kt) * 64;
        for (int u = 0; u < 8; ++u) memcpy(stage + ((size_t)u * 96 + r) * 8, blk + u * 8, 16);
      }
    }
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, img.size() * 2));
  m->allocs.push_back(p);
  HIP_TRY(hipMemcpy(p, img.data(), img.size() * 2, hipMemcpyHostToDevice));
  dst->p = p;
  return FD_OK;
}

// seq_attn16.hip's weights: [head][k32 step][tile: q 0, q 1, k 0, k 1, v 0, v 1][unit 0-7][row 0-15][16 B] (a k32 step of a head = 12 KiB,
// two of them = one LDS ring stage byte for byte; a tile = one v_mfma_f32_16x16x32_f16 operand, unit-major).  Row i of tile j holds
// head feature 8 (i / 4) + 4 j + (i % 4): with that order the 16 x 16 C/D layout gives every lane eight consecutive features of its
// token (seq_attn16.hip).  Split at the SAME scale as wqkv_i.
int upload_seq_attn16_weights(fd_model* m, SplitW* dst, const float* W, int d) {
  std::vector<uint16_t> rm, img;
  pack_split_weight(W, 3 * d, d, &rm, &dst->scale, 384);
  const int nk = d / 32, H = d / 32;
  img.assign((size_t)H * nk * 96 * 64, 0);
  for (int h = 0; h < H; ++h)
    for (int kt = 0; kt < nk; ++kt) {
      uint16_t* step = img.data() + ((size_t)h * nk + kt) * 96 * 64;
      for (int t = 0; t < 6; ++t)
        for (int i = 0; i < 16; ++i) {
          const int f = 8 * (i >> 2) + 4 * (t & 1) + (i & 3);
          const int n = (t >> 1) * d + h * 32 + f;
          const uint16_t* blk = rm.data() + ((size_t)n * nk + kt) * 64;
          for (int u = 0; u < 8; ++u) memcpy(step + (size_t)t * 1024 + ((size_t)u * 16 + i) * 8, blk + u * 8, 16);
        }
    }
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, img.size() * 2));
  m->allocs.push_back(p);
  HIP_TRY(hipMemcpy(p, img.data(), img.size() * 2, hipMemcpyHostToDevice));
  dst->p = p;
  return FD_OK;
}

// ffn16.hip's weights: intermediate.dense [ff][d] and output.dense [d][ff] as ONE stream in consumption order.  Per group G of 64
// intermediate features (the first dense runs one group ahead of the second: up(0) | up(1) down(0) | ... | up(ng - 1) down(ng - 2) |
// down(ng - 1)): d / 32 steps of the first dense (k32 step ks: four tiles, tile t = 2 pair + j, row i = feature
// 64 G + 32 pair + 8 (i / 4) + 4 j + (i % 4)), then d / 32 steps of the second (pair 0, 1: its 32 features are one k32 step; output
// tiles T = 0 .. d / 16 - 1 four to a step, tile T = 2 kt + j, row i = output feature 32 kt + 8 (i / 4) + 4 j + (i % 4)).  A tile =
// [unit 0-7][row 0-15][16 B] (units 0-3: hi of k 8 u .. 8 u + 7, 4-7: lo), a step = 8 KiB, two steps = one LDS ring stage byte for
// byte.  Each matrix is split at its own power-of-two scale.
int upload_ffn16_weights(fd_model* m, LayerDev* lw, const float* Wi, const float* Wd, const float* Wo, int d, int ff) {
  std::vector<uint16_t> ri, rd, img;
  pack_split_weight(Wi, ff, d, &ri, &lw->wff_i.scale, 128);
  pack_split_weight(Wd, d, ff, &rd, &lw->wff_scale_dn, 128);
  const int nkt = d / 32, ng = ff / 64, spg = 2 * nkt, nkd = ff / 32;
  img.assign((size_t)ng * spg * 4 * 1024, 0);
  // stream position (in steps) of a group's blocks: up(0) | up(1) down(0) | up(2) down(1) | ... | up(ng - 1) down(ng - 2) | down(ng - 1)
  auto up_at = [&](int G) { return G == 0 ? 0 : nkt + (G - 1) * spg; };
  auto down_at = [&](int G) { return G == ng - 1 ? nkt + (ng - 1) * spg : nkt + G * spg + nkt; };
  auto put = [&](int step, int t, int i, const uint16_t* blk) {
    uint16_t* tile = img.data() + ((size_t)step * 4 + t) * 1024;
    for (int u = 0; u < 8; ++u) memcpy(tile + ((size_t)u * 16 + i) * 8, blk + u * 8, 16);
  };
  for (int G = 0; G < ng; ++G) {
    for (int ks = 0; ks < nkt; ++ks)
      for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) {
          const int f = 64 * G + 32 * (t >> 1) + 8 * (i >> 2) + 4 * (t & 1) + (i & 3);
          put(up_at(G) + ks, t, i, ri.data() + ((size_t)f * nkt + ks) * 64);
        }
    for (int pr = 0; pr < 2; ++pr)
      for (int T = 0; T < 2 * nkt; ++T)
        for (int i = 0; i < 16; ++i) {
          const int o = 32 * (T >> 1) + 8 * (i >> 2) + 4 * (T & 1) + (i & 3);
          put(down_at(G) + pr * (nkt / 2) + T / 4, T % 4, i, rd.data() + ((size_t)o * nkd + 2 * G + pr) * 64);
        }
  }
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, img.size() * 2));
  m->allocs.push_back(p);
  HIP_TRY(hipMemcpy(p, img.data(), img.size() * 2, hipMemcpyHostToDevice));
  lw->wff_i.p = p;
  if (Wo) {
    // ... and with attention.output.dense [d][d] in front (ffn16.hip TAIL): step kt * (d / 64) + q = k32 step kt of the context against
    // the output tiles 4 q .. 4 q + 3 (rows permuted like the second dense's)
    std::vector<uint16_t> ro, timg;
    pack_split_weight(Wo, d, d, &ro, &lw->wtail_i.scale, 128);
    const int nsa = nkt * nkt / 2;
    timg.assign((size_t)nsa * 4 * 1024 + img.size(), 0);
    for (int kt = 0; kt < nkt; ++kt)
      for (int T = 0; T < 2 * nkt; ++T)
        for (int i = 0; i < 16; ++i) {
          const int o = 32 * (T >> 1) + 8 * (i >> 2) + 4 * (T & 1) + (i & 3);
          uint16_t* tile = timg.data() + ((size_t)(kt * (nkt / 2) + T / 4) * 4 + T % 4) * 1024;
          const uint16_t* blk = ro.data() + ((size_t)o * nkt + kt) * 64;
          for (int u = 0; u < 8; ++u) memcpy(tile + ((size_t)u * 16 + i) * 8, blk + u * 8, 16);
        }
    memcpy(timg.data() + (size_t)nsa * 4 * 1024, img.data(), img.size() * 2);
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, timg.size() * 2));
    m->allocs.push_back(q);
    HIP_TRY(hipMemcpy(q, timg.data(), timg.size() * 2, hipMemcpyHostToDevice));
    lw->wtail_i.p = q;
  }
  return FD_OK;
}

void free_weights(fd_model* m) {
  for (void* p : m->allocs) (void)hipFree(p);
  m->allocs.clear();
  m->layers.clear();
  m->finalized = false;
}

// ---- static scales of the activation images (row-image path).
// Every image tensor gets the largest power of two s with  bound * s <= 30000 < fp16 max, where `bound` is a
// guaranteed bound of |x| derived from the weights: a LayerNorm output obeys |y_i| <= max|gamma| sqrt(d) + max|beta|
// and ||y||_2 <= max|gamma| sqrt(d) + ||beta||_2; a dense output |y W_j + b_j| <= ||y||_2 ||W_j||_2 + |b_j|; GELU and the
// softmax-weighted average of V do not grow their argument.  Nothing overflows for ANY weights (trained outliers
// included); elements far below the bound merely lose low bits of `lo` (absolute error bound * 2^-40).
struct Bound {
  float linf, l2;
};
float scale_for(float bound) {
  if (!(bound > 0.f) || !std::isfinite(bound)) return 1.f;
  float e = std::floor(std::log2(30000.0f / bound));
  e = e < -40.f ? -40.f : (e > 40.f ? 40.f : e);
  return std::exp2(e);
}
Bound ln_bound(const HostTensor* g, const HostTensor* b, float extra_each = 0.f) {
  const size_t d = g->data.size();
  float gm = 0.f, bm = 0.f;
  double b2 = 0.0;
  for (size_t i = 0; i < d; ++i) {
    gm = std::fmax(gm, std::fabs(g->data[i]));
    bm = std::fmax(bm, std::fabs(b->data[i]));
    b2 += (double)b->data[i] * b->data[i];
  }
  const float sd = std::sqrt((float)d);
  return Bound{gm * sd + bm + extra_each, gm * sd + (float)std::sqrt(b2) + extra_each * sd};
}
// bound of a dense layer's outputs (rows [r0, r1) of W [N][K]) for inputs with ||x||_2 <= in_l2
float dense_bound(const float* W, const float* bias, int r0, int r1, int K, float in_l2) {
  float worst = 0.f;
  for (int j = r0; j < r1; ++j) {
    double n2 = 0.0;
    for (int k = 0; k < K; ++k) n2 += (double)W[(size_t)j * K + k] * W[(size_t)j * K + k];
    worst = std::fmax(worst, (float)std::sqrt(n2) * in_l2 + std::fabs(bias[j]));
  }
  return worst;
}

// expected shape of a state_dict entry; returns false if the name is not a parameter we consume
bool expected_shape(const fd_config& c, const std::string& name, std::vector<int64_t>* shp, bool* ignored) {
  *ignored = false;
  const int64_t d = c.d_model, F = c.n_features, ff = c.d_ff;
  auto set = [&](std::initializer_list<int64_t> s) { *shp = s; return true; };
  if (name == "time_embed.W" || name == "embeddings.position_ids") { *ignored = true; return true; }
  if (name == "inputs_to_hidden_dim.weight") return set({d, F});
  if (name == "inputs_to_hidden_dim.bias") return set({d});
  if (name == "embeddings.LayerNorm.weight" || name == "embeddings.LayerNorm.bias") return set({d});
  if (name == "embeddings.position_embeddings.weight") return set({c.max_pos, d});
  if (c.decoder == FD_DEC_MLP) {
    if (name == "token_decoder.dense1.weight") return set({d, d});
    if (name == "token_decoder.dense1.bias") return set({d});
    if (name == "token_decoder.layer_norm.weight" || name == "token_decoder.layer_norm.bias") return set({d});
    if (name == "token_decoder.dense2.weight") return set({F, d});
    if (name == "token_decoder.dense2.bias") return set({F});
  } else {
    if (name == "token_decoder.weight") return set({F, d});
    if (name == "token_decoder.bias") return set({F});
  }
  const std::string pre = "encoder.layer.";
  if (name.compare(0, pre.size(), pre) == 0) {
    size_t dot = name.find('.', pre.size());
    if (dot == std::string::npos) return false;
    int li = atoi(name.substr(pre.size(), dot - pre.size()).c_str());
    if (li < 0 || li >= c.n_layers) return false;
    const std::string rest = name.substr(dot + 1);
    for (const char* p : {"attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"}) {
      if (rest == std::string(p) + ".weight") return set({d, d});
      if (rest == std::string(p) + ".bias") return set({d});
    }
    if (rest == "attention.self.distance_embedding.weight") return set({2 * (int64_t)c.max_pos - 1, (int64_t)(c.d_model / c.n_heads)});
    for (const char* p : {"attention.output.LayerNorm", "output.LayerNorm"}) {
      if (rest == std::string(p) + ".weight" || rest == std::string(p) + ".bias") return set({d});
    }
    if (rest == "intermediate.dense.weight") return set({ff, d});
    if (rest == "intermediate.dense.bias") return set({ff});
    if (rest == "output.dense.weight") return set({d, ff});
    if (rest == "output.dense.bias") return set({d});
  }
  return false;
}

const HostTensor* need(fd_model* m, const std::string& name) {
  auto it = m->host.find(name);
  if (it == m->host.end()) {
    fail(FD_E_MISSING, "weight '%s' was never provided (fd_set_weight)", name.c_str());
    return nullptr;
  }
  return &it->second;
}

void drop_workspaces(fd_model* m) {
  m->ws.release();
  for (Workspace& w : m->cache) w.release();
  m->cache.clear();
}

size_t kHeadBytes(const fd_config& c) { return (size_t)(c.d_model / c.n_heads) * 4; }  // q / k / v image bytes per position and head
// The q | k | v projections are stored per 32-column SUB-HEAD (d_model / 32 of them; a head of size 32 nb is nb consecutive
// sub-heads): head size 32 (every released configuration) runs attention_img.hip, 64 / 96 / 128 the general attention_gen.hip.
int head_dim(const fd_config& c) { return c.d_model / c.n_heads; }
int sub_heads(const fd_config& c) { return c.d_model / 32; }

int ensure_ws(fd_model* m, int B, int L) {
  Workspace& w = m->ws;
  // every entry point that takes a workspace comes through here: a sampling run begun earlier (fd_sample_begin_dev) is over --
  // its state (x, the step counter) is about to be overwritten or parked; fd_sample_begin_dev re-opens one afterwards
  m->run_t = -1;
  m->run_open = false;
  const fd_config& c = m->cfg;
  const size_t M = (size_t)B * L, d = c.d_model, F = c.n_features;
  // algorithmic work per launch: 2*M*N*K for GEMMs, 6*L*d per token for attention
  {
    const double Md = (double)M, dd = (double)d, ff = (double)c.d_ff, Ld = (double)L;
    double* fl = m->prof_flops;
    double* by = m->prof_bytes;
    const bool img = m->img;
    fl[KC_EMBED] = 2 * Md * F * dd;            by[KC_EMBED] = 4 * (Md * F + Md * dd);
    // the row-image path computes V (transposed) in its own launch unless n_heads % 6 == 0 (one q | k | v launch)
    const int qn = (img && !(sub_heads(c) % 6 == 0 && !m->split_qkv)) ? 2 : 3;
    fl[KC_GEMM_QKV] = 2 * Md * qn * dd * dd;
    by[KC_GEMM_QKV] = 4 * (Md * dd + qn * dd * dd + Md * qn * dd);
    fl[KC_GEMM_V] = 2 * Md * dd * dd;          by[KC_GEMM_V] = 4 * (2 * Md * dd + dd * dd);
    fl[KC_ATTN] = 6 * Ld * dd * Md;            by[KC_ATTN] = 4 * (Md * 3 * dd + Md * dd);
    fl[KC_GEMM_OUT] = 2 * Md * dd * dd;        by[KC_GEMM_OUT] = 4 * (3 * Md * dd + dd * dd);
    fl[KC_LN1] = 0;                            by[KC_LN1] = 4 * 2 * Md * dd;
    fl[KC_GEMM_UP] = 2 * Md * ff * dd;         by[KC_GEMM_UP] = 4 * (Md * dd + ff * dd + Md * ff);
    fl[KC_GEMM_DOWN] = 2 * Md * ff * dd;       by[KC_GEMM_DOWN] = 4 * (Md * ff + ff * dd + 2 * Md * dd);
    fl[KC_LN2] = 0;                            by[KC_LN2] = 4 * 2 * Md * dd;
    fl[KC_GEMM_HEAD] = 2 * Md * dd * dd;       by[KC_GEMM_HEAD] = 4 * (2 * Md * dd + dd * dd);
    fl[KC_HEAD_UPDATE] = 2 * Md * dd * F;      by[KC_HEAD_UPDATE] = 4 * (Md * dd + 3 * Md * F);
    fl[KC_ADVANCE] = 0;                        by[KC_ADVANCE] = 4;
    // fused q | k | v projection + attention: h in, ctx out, the weights once (q, k, v stay on chip)
    fl[KC_SEQ_ATTN] = 2 * Md * 3 * dd * dd + 6 * Ld * dd * Md;  by[KC_SEQ_ATTN] = 4 * (2 * Md * dd + 3 * dd * dd);
    // fused BertIntermediate + BertOutput: a in, h out, both weight matrices once (the intermediate stays on chip)
    fl[KC_FFN] = 4 * Md * ff * dd;             by[KC_FFN] = 4 * (2 * Md * dd + 2 * ff * dd);
    // ... with BertSelfOutput in front: ctx and h in, h out, three weight matrices once
    fl[KC_TAIL] = 4 * Md * ff * dd + 2 * Md * dd * dd;  by[KC_TAIL] = 4 * (3 * Md * dd + 2 * ff * dd + dd * dd);
  }
  w.last_use = ++m->use_clock;
  if (w.B == B && w.L == L) return FD_OK;
  // park the current workspace and look for a cached one of this shape
  if (w.B != 0) {
    m->cache.push_back(w);
    w = Workspace();
  }
  for (size_t i = 0; i < m->cache.size(); ++i)
    if (m->cache[i].B == B && m->cache[i].L == L) {
      w = m->cache[i];
      m->cache.erase(m->cache.begin() + i);
      w.last_use = m->use_clock;
      return FD_OK;
    }
  while (m->cache.size() >= 3) {  // keep at most 4 shapes alive (C2-sized workspaces are about 1 GB each)
    HIP_TRY(hipStreamSynchronize(m->stream));
    size_t old = 0;
    for (size_t i = 1; i < m->cache.size(); ++i)
      if (m->cache[i].last_use < m->cache[old].last_use) old = i;
    m->cache[old].release();
    m->cache.erase(m->cache.begin() + old);
  }
  if (m->img) {  // every shape check comes BEFORE the first allocation: an early return must not leave buffers behind
    const int Tt = L > 128 ? 4 : (L + 31) / 32;
    const size_t ltot = (size_t)(L > 128 ? (L + 127) / 128 : 1) * 32 * Tt;
    if ((size_t)B * c.n_heads * ltot * kHeadBytes(c) >= (1ull << 32) - 65536)  // (32-bit offsets, the top bytes mark dropped stores)
      return fail(FD_E_UNSUPPORTED, "B=%lld x heads x L=%d: the q / k / v images of one batch must stay below 4 GiB; use smaller batches",
                  (long long)B, (int)L);
  }
  const int rc_alloc = [&]() -> int {
  auto al = [&](void** p, size_t bytes) -> hipError_t { return hipMalloc(p, bytes); };
  auto alz = [&](void** p, size_t bytes) -> hipError_t {
    hipError_t e = hipMalloc(p, bytes);
    return e != hipSuccess ? e : hipMemsetAsync(*p, 0, bytes, m->stream);
  };
  HIP_TRY(al((void**)&w.x, M * F * 4));
  HIP_TRY(alz((void**)&w.eps, M * F * 4));
  HIP_TRY(al((void**)&w.z, M * F * 4));
  HIP_TRY(al((void**)&w.lens, (size_t)B * 4));
  HIP_TRY(alz((void**)&w.t_dev, 16));
  HIP_TRY(al((void**)&w.dyn, sizeof(UpdateDyn)));
  if (m->img) {
    // Row images: sequences start at multiples of 8 rows, the row count is rounded up to whole 128-row tiles.
    // Everything is zeroed once: padding rows / never-written key slots stay finite for ever.
    const size_t Lr = (size_t)(L + 7) / 8 * 8;
    const size_t cap = (B * Lr + 127) / 128 * 128;
    const int T = L > 128 ? 4 : (L + 31) / 32;
    w.LPK = 32 * T;
    w.NKT = L > 128 ? (L + 127) / 128 : 1;
    w.LTOT = w.NKT * w.LPK;
    w.cap = (int)cap;
    const size_t BH = (size_t)B * sub_heads(c), gmax = c.d_ff > c.d_model ? c.d_ff : c.d_model;
    HIP_TRY(alz((void**)&w.himg, cap * d * 4));
    HIP_TRY(alz((void**)&w.aimg, cap * d * 4));
    HIP_TRY(alz((void**)&w.cimg, cap * d * 4));
    HIP_TRY(alz((void**)&w.gimg, cap * gmax * 4));
    if (d > 384) HIP_TRY(alz((void**)&w.tmp, cap * d * 4));  // pre-LayerNorm fp32 rows (un-fused LayerNorm path)
    HIP_TRY(alz((void**)&w.qbuf, BH * w.LTOT * 128));
    HIP_TRY(alz((void**)&w.kbuf, BH * w.LTOT * 128));
    HIP_TRY(alz((void**)&w.vbuf, BH * w.LTOT * 128));
    HIP_TRY(alz((void**)&w.trash, 1024));
    HIP_TRY(alz((void**)&w.rowinfo, cap * sizeof(int2)));
    HIP_TRY(alz((void**)&w.seq_row0, ((size_t)B + 1) * 4));
    HIP_TRY(alz((void**)&w.nrow, (size_t)B * 4));
    HIP_TRY(alz((void**)&w.dims, 16));
    HIP_TRY(alz((void**)&w.flag, 16));
    w.img = true;
  } else {
    const size_t gmax = c.d_ff > c.d_model ? c.d_ff : c.d_model;
    // Token-row buffers that feed or leave the LN-fused GEMMs are padded to whole 128-row tiles and
    // zeroed once: the fused kernels then run on full tiles for any B*L (rows are independent; the
    // padding rows stay finite and are never read back).
    const size_t Mp = (M + 127) / 128 * 128;
    HIP_TRY(alz((void**)&w.h, Mp * d * 4));
    HIP_TRY(alz((void**)&w.qkv, Mp * 3 * d * 4));
    HIP_TRY(alz((void**)&w.ctx, Mp * d * 4));
    HIP_TRY(alz((void**)&w.a, Mp * d * 4));
    HIP_TRY(al((void**)&w.tmp, M * d * 4));
    HIP_TRY(alz((void**)&w.g, Mp * gmax * 4));
  }
    return FD_OK;
  }();
  if (rc_alloc) {  // (out of memory, ...): give back what was allocated; w.B stays 0
    (void)hipStreamSynchronize(m->stream);
    w.release();
    return rc_alloc;
  }
  w.B = B;
  w.L = L;
  w.last_use = m->use_clock;
  // the zero fills ran on the model's stream; the caller may continue on another one
  HIP_TRY(hipStreamSynchronize(m->stream));
  return FD_OK;
}

struct StepMode {
  bool forward_only;   // write eps only (fd_forward)
  bool use_dyn;        // read per-call values from ws.dyn (sampling loop)
  bool advance;        // decrement *t_dev at the end
  bool profile;        // bracket every kernel with events
  // by-value per-call values when !use_dyn (fd_p_sample_step)
  const float* noise = nullptr;
  int t_start = 0;
  bool no_wrap = false;  // p_sample alone, without the loop's wrap
  const unsigned char* kmask = nullptr;  // fd_forward_ex: [B][L] key mask of any pattern (device), else null: prefix masks from lens
  const int* pos_ids = nullptr;          // fd_forward_ex: [B][L] position ids of the absolute position embedding (device)
};

int prof_begin(fd_model* m, int cls, hipStream_t s, bool on) {
  if (!on) return FD_OK;
  hipEvent_t e0, e1;
  for (hipEvent_t* e : {&e0, &e1}) {
    if (!m->event_pool.empty()) {
      *e = m->event_pool.back();
      m->event_pool.pop_back();
    } else {
      HIP_TRY(hipEventCreate(e));
    }
  }
  m->pending.push_back({cls, e0, e1});
  HIP_TRY(hipEventRecord(e0, s));
  return FD_OK;
}
int prof_end(fd_model* m, hipStream_t s, bool on) {
  if (!on) return FD_OK;
  HIP_TRY(hipEventRecord(m->pending.back().e1, s));
  return FD_OK;
}

int harvest(fd_model* m) {
  for (PendingEvent& p : m->pending) {
    HIP_TRY(hipEventSynchronize(p.e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, p.e0, p.e1));
    m->prof_ms[p.cls] += ms;
    m->prof_n[p.cls] += 1;
    m->event_pool.push_back(p.e0);
    m->event_pool.push_back(p.e1);
  }
  m->pending.clear();
  return FD_OK;
}

#define PROF(cls, stmt)                                   \
  do {                                                    \
    int rc_ = prof_begin(m, cls, s, mode.profile);        \
    if (rc_) return rc_;                                  \
    stmt;                                                 \
    rc_ = prof_end(m, s, mode.profile);                   \
    if (rc_) return rc_;                                  \
  } while (0)

// One reverse-diffusion step = BertForDiffusionBase.forward (modelling.py:384-484) + the
// p_sample update and wrap (sampling.py:62-75, :119-130), as a fixed kernel sequence.
void gemm(fd_model* m, int epi, const float* A, const float* W, const float* bias, const float* resid, float* C, int M,
          int N, int K, hipStream_t s) {
  (void)m;
  launch_gemm_f32(epi, A, W, bias, resid, C, M, N, K, s);
}

int run_step_img(fd_model* m, hipStream_t s, const StepMode& mode);

int run_step(fd_model* m, hipStream_t s, const StepMode& mode) {
  if (m->ws.img) return run_step_img(m, s, mode);
  const fd_config& c = m->cfg;
  Workspace& w = m->ws;
  const int B = w.B, L = w.L, M = B * L, d = c.d_model, ff = c.d_ff, F = c.n_features;
  const bool fuse_ln = m->fuse_ln > 0;  // fp32 path: the LN-fused fp32 GEMM is slower than GEMM + LayerNorm, off unless asked for
  PROF(KC_EMBED, launch_embed(w.x, m->w_in, m->b_in, m->pos_emb, m->emb_g, m->emb_b, c.ln_eps, m->time_table, w.t_dev,
                              w.h, B, L, F, d, s));
  for (int li = 0; li < c.n_layers; ++li) {
    const LayerDev& lw = m->layers[li];
    PROF(KC_GEMM_QKV, gemm(m, EPI_BIAS, w.h, lw.wqkv, lw.bqkv, nullptr, w.qkv, M, 3 * d, d, s));
    bool ok = true;
    PROF(KC_ATTN, ok = launch_attention_f32(w.qkv, lw.demb, w.lens, w.ctx, B, L, c.n_heads, c.max_pos, s,
                                            c.pos_type == FD_POS_RELATIVE_KEY_QUERY));
    if (!ok) return fail(FD_E_UNSUPPORTED, "attention: sequence length %d not supported by the fp32 kernel (max 128)", L);
    bool fused = false;
    if (fuse_ln)
      PROF(KC_GEMM_OUT, fused = launch_gemm_f32_ln(w.ctx, lw.wo, lw.bo, w.h, lw.ln1g, lw.ln1b, c.ln_eps, w.a, M, d, d, s));
    if (!fused) {
      if (fuse_ln && mode.profile) {  // the attempted launch recorded an empty bracket; drop it
        PendingEvent p = m->pending.back();
        m->pending.pop_back();
        m->event_pool.push_back(p.e0);
        m->event_pool.push_back(p.e1);
      }
      PROF(KC_GEMM_OUT, gemm(m, EPI_BIAS_RESID, w.ctx, lw.wo, lw.bo, w.h, w.tmp, M, d, d, s));
      PROF(KC_LN1, launch_layernorm(w.tmp, lw.ln1g, lw.ln1b, c.ln_eps, w.a, M, d, s));
    }
    PROF(KC_GEMM_UP, gemm(m, EPI_BIAS_GELU, w.a, lw.wi, lw.bi, nullptr, w.g, M, ff, d, s));
    fused = false;
    if (fuse_ln)
      PROF(KC_GEMM_DOWN, fused = launch_gemm_f32_ln(w.g, lw.wd, lw.bd, w.a, lw.ln2g, lw.ln2b, c.ln_eps, w.h, M, d, ff, s));
    if (!fused) {
      if (fuse_ln && mode.profile) {
        PendingEvent p = m->pending.back();
        m->pending.pop_back();
        m->event_pool.push_back(p.e0);
        m->event_pool.push_back(p.e1);
      }
      PROF(KC_GEMM_DOWN, gemm(m, EPI_BIAS_RESID, w.g, lw.wd, lw.bd, w.a, w.tmp, M, d, ff, s));
      PROF(KC_LN2, launch_layernorm(w.tmp, lw.ln2g, lw.ln2b, c.ln_eps, w.h, M, d, s));
    }
  }
  UpdateArgs u;
  memset(&u, 0, sizeof u);
  if (c.decoder == FD_DEC_MLP) {
    PROF(KC_GEMM_HEAD, gemm(m, EPI_BIAS_GELU, w.h, m->hd_w1, m->hd_b1, nullptr, w.g, M, d, d, s));
    u.g = w.g;
    u.gamma = m->hd_g;
    u.beta = m->hd_b;
    u.do_ln = 1;
  } else {
    u.g = w.h;
    u.do_ln = 0;
  }
  u.w2 = m->hd_w2;
  u.b2 = m->hd_b2;
  u.x = w.x;
  u.coef = m->coef;
  u.t_dev = w.t_dev;
  u.T = m->T;
  u.M = M;
  u.L = L;
  u.F = F;
  u.d = d;
  u.ln_eps = 1e-12f;  // AnglesPredictor(eps=1e-12)  (modelling.py:187,199)
  u.angle_mask = mode.no_wrap ? 0u : m->angle_mask;
  if (mode.forward_only) {
    u.eps_out = w.eps;
    u.x_out = nullptr;
  } else {
    u.eps_out = w.eps;
    u.x_out = w.x;
    if (mode.use_dyn) {
      u.dyn = w.dyn;
    } else {
      u.noise = mode.noise;
      u.noise_stride = 0;
      u.t_start = mode.t_start;
    }
  }
  if (mode.use_dyn) u.noise_stride = (long long)M * F;
  PROF(KC_HEAD_UPDATE, launch_head_update(u, s));
  if (mode.advance) PROF(KC_ADVANCE, launch_step_advance(w.t_dev, s));
  HIP_TRY(hipGetLastError());
  return FD_OK;
}


// The same step on the row-image kernels (FD_PREC_F16X3): 6 launches per layer (QK, V^T, attention, attn-out + LN,
// FFN-up + GELU, FFN-down + LN), every activation an fp16 hi|lo image, no separate step-advance launch.
int run_step_img(fd_model* m, hipStream_t s, const StepMode& mode) {
  int launches = 0;
#define DBG_STOP() do { if (m->debug_stop > 0 && ++launches >= m->debug_stop) { HIP_TRY(hipGetLastError()); return FD_OK; } } while (0)
  const fd_config& c = m->cfg;
  Workspace& w = m->ws;
  const int B = w.B, L = w.L, d = c.d_model, ff = c.d_ff, F = c.n_features, H = sub_heads(c);  // H: 32-column sub-heads (= heads at head size 32)
  const int max_rows = w.cap;
  {
    EmbedImgArgs e;
    memset(&e, 0, sizeof e);
    e.x = w.x; e.w_in = m->w_in; e.b_in = m->b_in; e.pos_emb = m->pos_emb; e.gamma = m->emb_g; e.beta = m->emb_b;
    e.time_table = m->time_table; e.tslot = w.t_dev; e.rowinfo = w.rowinfo; e.nrow = w.nrow; e.dims = w.dims;
    e.h = w.himg; e.L = L; e.F = F; e.d = d; e.eps = c.ln_eps; e.out_scale = m->layers[0].s_h;
    e.pos_ids = mode.pos_ids;
    PROF(KC_EMBED, launch_embed_img(e, max_rows, s));
      DBG_STOP();
  }
  static const bool want_stamps = [] { const char* e = getenv("FDMI_STAMPS"); return e && atoi(e) != 0; }();
  if (want_stamps && !m->stamps) {
    HIP_TRY(hipMalloc((void**)&m->stamps, kStampWords * 8));
    HIP_TRY(hipMemset(m->stamps, 0, kStampWords * 8));
  }
  // do the tiles of an N-column GEMM over this workspace fill whole rounds of the launch's workgroups?  Padded rows: the row count
  // is the workspace's capacity, known here; packed rows (sampling.sample): data dependent -> the slice-capable instantiation
  auto tail_for = [&](int N) {
    if (m->varlen) return 1;
    const int ntiles = (max_rows / 128) * ((N + 383) / 384);
    return ntiles % gemm_img_grid(max_rows, N) != 0 ? 1 : 0;
  };
  auto base = [&]() {
    GemmImgArgs g;
    memset(&g, 0, sizeof g);
    g.stamps = m->stamps;
    g.trash = w.trash; g.rowinfo = w.rowinfo; g.dims = w.dims;
    g.qkv_bytes = (unsigned)((size_t)w.B * H * w.LTOT * 128);
    g.H = H; g.LPK = w.LPK; g.LTOT = w.LTOT; g.NKT = w.NKT;
    g.eps = c.ln_eps;
    return g;
  };
  for (int li = 0; li < c.n_layers; ++li) {
    const LayerDev& lw = m->layers[li];
    const float s_next = li + 1 < c.n_layers ? m->layers[li + 1].s_h : m->s_hfinal;
    // q | k | v projection + attention as ONE kernel per sequence: q, k and v never reach HBM.  fuse_attn: 1 = seq_attn16.hip (16-row
    // waves, two per SIMD; any L <= 128, padded or packed rows), 2 = seq_attn.hip (round 5: 32-row waves, 96 < L <= 128), 0 = never
    static const int fuse_attn_env = [] { const char* e = getenv("FDMI_FUSE_ATTN"); return e ? atoi(e) : -1; }();
    const int fuse_attn = m->fuse_attn >= 0 ? m->fuse_attn : fuse_attn_env;
    // auto: padded rows of 97..128 positions when the batch fills whole rounds of the CUs (one workgroup = one sequence at a time: 512
    // sequences on 256 CUs are two full rounds, 300 would leave the second round four-fifths empty and 8 sequences would run on 8 CUs).
    // Measured (profiles/r06_seq_attn16_notes.log): packed rows of BASELINE C3's first chunk (B 512, lengths 50..101) are a tie with
    // the two-kernel path (5.29 against 5.25 ms per step), its second chunk (B 268) and batches of a few sequences lose.
    bool fused_auto = !m->varlen && L > 96;
    if (fused_auto) {
      const int ncu = gemm_img_grid(1 << 30, 384);  // (= the CU count, rounded down to whole XCDs)
      const int rounds = (B + ncu - 1) / ncu;
      fused_auto = (double)B >= 0.94 * (double)rounds * ncu;
    }
    const bool fused_ok = fuse_attn != 0 && (fuse_attn > 0 || fused_auto) && !mode.kmask && !m->split_qkv &&
                          (size_t)w.cap * d * 4 < (1ull << 32) - 65536;
    const bool fused16 = fused_ok && fuse_attn != 2 && lw.wsa16_i.p && seq_attn16_supported(d, c.n_heads, L, c.max_pos);
    const bool fused32 = fused_ok && !fused16 && (fuse_attn == 2 || fuse_attn < 0) && lw.wsa_i.p && (fuse_attn == 2 || !m->varlen) &&
                         seq_attn_supported(d, c.n_heads, L, c.max_pos);
    const bool fused_attn = fused16 || fused32;
    if (fused_attn) {
      SeqAttnArgs a;
      memset(&a, 0, sizeof a);
      a.himg = w.himg; a.himg_bytes = (unsigned)((size_t)w.cap * d * 4);
      const SplitW& wsa = fused16 ? lw.wsa16_i : lw.wsa_i;
      a.wimg = static_cast<const unsigned char*>(wsa.p); a.bias = lw.bqkv;
      a.demb = static_cast<const u32x4_t*>(lw.demb_s.p);
      a.lens = w.lens; a.nrow = w.nrow; a.seq_row0 = w.seq_row0; a.ctx = w.cimg;
      a.B = B; a.H = H; a.maxpos = c.max_pos;
      a.acc_scale = 1.0f / (lw.s_h * wsa.scale);
      a.q_scale = lw.s_q; a.k_scale = lw.s_k; a.v_scale = lw.s_v; a.ctx_scale = lw.s_v;
      a.r_scale = lw.s_k / lw.demb_s.scale;
      a.stamps = m->stamps ? m->stamps + 5 * 8 * 64 * 6 + 4 * 64 * 8 : nullptr;
      if (fused16) {
        bool launched = false;
        PROF(KC_SEQ_ATTN, launched = launch_seq_attn16(a, s));
        if (!launched) return fail(FD_E_HIP, "the fused projection + attention kernel could not be launched (%d bytes of LDS per workgroup)", 160256);
      } else {
        bool launched = false;
        PROF(KC_SEQ_ATTN, launched = launch_seq_attn(a, s));
        if (!launched) return fail(FD_E_HIP, "the 32-row fused projection + attention kernel could not be launched (%d bytes of LDS per workgroup)", 160 * 1024);
      }
      DBG_STOP();
      DBG_STOP();  // (two launches of the other path: debug_stop counts stay comparable)
    } else if (lw.wqkv_i.p && !m->split_qkv) {
      // q | k | v in one launch: the three column tiles of a row panel run side by side on one XCD (h is read from HBM once)
      GemmImgArgs g = base();
      g.A = w.himg; g.W = static_cast<const unsigned char*>(lw.wqkv_i.p); g.bias = lw.bqkv;
      g.qbuf = w.qbuf; g.kbuf = w.kbuf; g.vbuf = w.vbuf; g.N = 3 * d; g.K = d;
      g.acc_scale = 1.0f / (lw.s_h * lw.wqkv_i.scale); g.q_scale = lw.s_q; g.k_scale = lw.s_k; g.v_scale = lw.s_v;
      g.tail = tail_for(g.N);
      PROF(KC_GEMM_QKV, launch_gemm_img(EPI_IMG_QKV, g, max_rows, s));
      DBG_STOP();
    } else {
      {
        GemmImgArgs g = base();
        g.A = w.himg; g.W = static_cast<const unsigned char*>(lw.wqk_i.p); g.bias = lw.bqk;
        g.qbuf = w.qbuf; g.kbuf = w.kbuf; g.N = 2 * d; g.K = d;
        g.acc_scale = 1.0f / (lw.s_h * lw.wqk_i.scale); g.q_scale = lw.s_q; g.k_scale = lw.s_k;
        g.tail = tail_for(g.N);
        PROF(KC_GEMM_QKV, launch_gemm_img(EPI_IMG_QK, g, max_rows, s));
        DBG_STOP();
      }
      {
        GemmImgArgs g = base();
        g.A = w.himg; g.W = static_cast<const unsigned char*>(lw.wv_i.p); g.bias = lw.bv;
        g.vbuf = w.vbuf; g.N = d; g.K = d;
        g.acc_scale = 1.0f / (lw.s_h * lw.wv_i.scale); g.v_scale = lw.s_v;
        g.tail = tail_for(g.N);
        PROF(KC_GEMM_V, launch_gemm_img(EPI_IMG_VT, g, max_rows, s));
        DBG_STOP();
      }
    }
    if (!fused_attn) {
      AttnImgArgs a;
      memset(&a, 0, sizeof a);
      a.qbuf = w.qbuf; a.kbuf = w.kbuf; a.vbuf = w.vbuf;
      a.demb = static_cast<const u32x4_t*>(lw.demb_s.p);
      a.lens = w.lens; a.nrow = w.nrow; a.seq_row0 = w.seq_row0; a.ctx = w.cimg; a.trash = w.trash;
      a.B = B; a.H = H; a.LTOT = w.LTOT; a.NKT = w.NKT; a.maxpos = c.max_pos;
      a.q_scale = lw.s_q; a.k_scale = lw.s_k; a.v_scale = lw.s_v; a.ctx_scale = lw.s_v;
      a.r_scale = lw.demb_s.p ? lw.s_k / lw.demb_s.scale : 1.f;
      a.r_scale_k = lw.demb_s.p ? lw.s_q / lw.demb_s.scale : 1.f;
      a.rkq = c.pos_type == FD_POS_RELATIVE_KEY_QUERY;
      a.stamps = m->stamps ? m->stamps + 5 * 8 * 64 * 6 : nullptr;
      bool ok = true;
      if (mode.kmask) {  // an arbitrary key mask: the general kernel (the tuned one builds its schedule on prefix masks)
        a.H = c.n_heads;
        a.kmask = mode.kmask;
        a.L = L;
        PROF(KC_ATTN, ok = launch_attention_gen(a, head_dim(c) / 32, s));
      } else if (head_dim(c) == 32) {
        PROF(KC_ATTN, ok = launch_attention_img(a, L, s));
      } else {  // head size 64 / 96 / 128: the general kernel, indexed by head; the images stay per sub-head
        a.H = c.n_heads;
        PROF(KC_ATTN, ok = launch_attention_gen(a, head_dim(c) / 32, s));
      }
      if (!ok) return fail(FD_E_UNSUPPORTED, "attention: L=%d, head size %d", L, head_dim(c));
      DBG_STOP();
    }
    // BertIntermediate + BertOutput as ONE kernel (ffn16.hip): the 2 d wide intermediate never reaches HBM.  fuse_ffn 2 (and auto):
    // BertSelfOutput (attention.output.dense + residual + LayerNorm) in front of it in the same launch, its output on chip too.  Passes of 128 rows, one
    // workgroup per CU: auto = the passes fill whole rounds of the CUs (512 passes on 256 CUs: two; 315 would leave the second round
    // a quarter full where the tile GEMMs deal 6 + 1 column tiles per pass)
    static const int fuse_ffn_env = [] { const char* e = getenv("FDMI_FUSE_FFN"); return e ? atoi(e) : -1; }();
    const int fuse_ffn = m->fuse_ffn >= 0 ? m->fuse_ffn : fuse_ffn_env;
    bool ffn_auto = false;
    {
      // (packed rows: by the caller's row count when it gave one -- the bound B * ceil8(L) overstates ragged batches)
      const int rows_known = m->varlen && m->rows_hint > 0 && m->rows_hint <= max_rows ? m->rows_hint : max_rows;
      const int ncu = gemm_img_grid(1 << 30, 384), passes = (rows_known + 127) / 128;
      const int rounds = (passes + ncu - 1) / ncu;
      ffn_auto = (double)passes >= 0.94 * (double)rounds * ncu;
    }
    const bool fused_ffn = fuse_ffn != 0 && (fuse_ffn > 0 || ffn_auto) && lw.wff_i.p && ffn16_supported(d, ff) &&
                           (size_t)w.cap * d * 4 < (1ull << 32) - 65536;
    const bool fused_tail = fused_ffn && fuse_ffn != 1 && lw.wtail_i.p && d <= 384;
    if (!fused_tail) {
      GemmImgArgs g = base();
      g.A = w.cimg; g.W = static_cast<const unsigned char*>(lw.wo_i.p); g.bias = lw.bo; g.gamma = lw.ln1g; g.beta = lw.ln1b;
      g.resid = w.himg; g.out = w.aimg; g.N = d; g.K = d;
      g.acc_scale = 1.0f / (lw.s_v * lw.wo_i.scale); g.resid_inv = 1.0f / lw.s_h; g.out_scale = lw.s_a;
      if (d <= 384) {
        g.tail = tail_for(g.N);
        PROF(KC_GEMM_OUT, launch_gemm_img(EPI_IMG_LN, g, max_rows, s));
      } else {  // a LayerNorm row does not fit one 384-column tile: fp32 rows, then the LayerNorm kernel
        g.out_f32 = w.tmp;
        g.tail = tail_for(g.N);
        PROF(KC_GEMM_OUT, launch_gemm_img(EPI_IMG_BIAS, g, max_rows, s));
        PROF(KC_LN1, launch_ln_f32_img(w.tmp, lw.ln1g, lw.ln1b, c.ln_eps, w.dims, w.aimg, d, lw.s_a, max_rows, s));
      }
      DBG_STOP();
    }
    if (fused_ffn) {
      FfnArgs a;
      memset(&a, 0, sizeof a);
      a.aimg = w.aimg; a.a_bytes = (unsigned)((size_t)w.cap * d * 4);
      a.wimg = static_cast<const unsigned char*>(fused_tail ? lw.wtail_i.p : lw.wff_i.p);
      if (fused_tail) {
        a.cimg = w.cimg; a.hres = w.himg; a.bo = lw.bo; a.g1 = lw.ln1g; a.b1 = lw.ln1b;
        a.ao_scale = 1.0f / (lw.s_v * lw.wtail_i.scale); a.hres_inv = 1.0f / lw.s_h; a.a_scale = lw.s_a; a.eps1 = c.ln_eps;
      }
      a.bi = lw.bi; a.bd = lw.bd; a.gamma = lw.ln2g; a.beta = lw.ln2b;
      a.out = w.himg; a.out_bytes = a.a_bytes;
      a.panels = max_rows / 128; a.dims = w.dims;
      a.up_scale = 1.0f / (lw.s_a * lw.wff_i.scale); a.g_scale = lw.s_g; a.down_scale = 1.0f / (lw.s_g * lw.wff_scale_dn);
      a.resid_inv = 1.0f / lw.s_a; a.out_scale = s_next; a.eps = c.ln_eps;
      a.stamps = m->stamps ? m->stamps + 5 * 8 * 64 * 6 + 4 * 64 * 8 + 4 * 64 * 16 + 16384 : nullptr;
      bool launched = false;
      if (fused_tail) PROF(KC_TAIL, launched = launch_ffn16(a, d, s));
      else PROF(KC_FFN, launched = launch_ffn16(a, d, s));
      if (!launched) return fail(FD_E_HIP, "the fused feed-forward kernel could not be launched");
      if (fused_tail) DBG_STOP();
      DBG_STOP();
      DBG_STOP();  // (two / three launches of the other path: debug_stop counts stay comparable)
      continue;
    }
    {
      GemmImgArgs g = base();
      g.A = w.aimg; g.W = static_cast<const unsigned char*>(lw.wi_i.p); g.bias = lw.bi; g.out = w.gimg; g.N = ff; g.K = d;
      g.acc_scale = 1.0f / (lw.s_a * lw.wi_i.scale); g.out_scale = lw.s_g;
      g.tail = tail_for(g.N);
      PROF(KC_GEMM_UP, launch_gemm_img(EPI_IMG_GELU, g, max_rows, s));
      DBG_STOP();
    }
    {
      GemmImgArgs g = base();
      g.A = w.gimg; g.W = static_cast<const unsigned char*>(lw.wd_i.p); g.bias = lw.bd; g.gamma = lw.ln2g; g.beta = lw.ln2b;
      g.resid = w.aimg; g.out = w.himg; g.N = d; g.K = ff;
      g.acc_scale = 1.0f / (lw.s_g * lw.wd_i.scale); g.resid_inv = 1.0f / lw.s_a; g.out_scale = s_next;
      if (d <= 384) {
        g.tail = tail_for(g.N);
        PROF(KC_GEMM_DOWN, launch_gemm_img(EPI_IMG_LN, g, max_rows, s));
      } else {
        g.out_f32 = w.tmp;
        g.tail = tail_for(g.N);
        PROF(KC_GEMM_DOWN, launch_gemm_img(EPI_IMG_BIAS, g, max_rows, s));
        PROF(KC_LN2, launch_ln_f32_img(w.tmp, lw.ln2g, lw.ln2b, c.ln_eps, w.dims, w.himg, d, s_next, max_rows, s));
      }
      DBG_STOP();
    }
  }
  UpdateArgs u;
  memset(&u, 0, sizeof u);
  HeadImgArgs hi;
  memset(&hi, 0, sizeof hi);
  if (c.decoder == FD_DEC_MLP) {
    GemmImgArgs g = base();
    g.A = w.himg; g.W = static_cast<const unsigned char*>(m->hd_w1_i.p); g.bias = m->hd_b1; g.out = w.gimg; g.N = d; g.K = d;
    g.acc_scale = 1.0f / (m->s_hfinal * m->hd_w1_i.scale); g.out_scale = m->s_hg;
    g.tail = tail_for(g.N);
    PROF(KC_GEMM_HEAD, launch_gemm_img(EPI_IMG_GELU, g, max_rows, s));
      DBG_STOP();
    hi.g = w.gimg; hi.g_inv = 1.0f / m->s_hg;
    u.gamma = m->hd_g; u.beta = m->hd_b; u.do_ln = 1;
  } else {
    hi.g = w.himg; hi.g_inv = 1.0f / m->s_hfinal;
    u.do_ln = 0;
  }
  hi.rowinfo = w.rowinfo; hi.nrow = w.nrow; hi.dims = w.dims; hi.tslot = w.t_dev; hi.flag = w.flag;
  hi.advance = mode.advance ? 1 : 0;
  u.w2 = m->hd_w2; u.b2 = m->hd_b2; u.x = w.x; u.coef = m->coef; u.t_dev = w.t_dev; u.T = m->T;
  u.M = B * L; u.L = L; u.F = F; u.d = d;
  u.ln_eps = 1e-12f;  // AnglesPredictor(eps=1e-12)  (modelling.py:187,199)
  u.angle_mask = mode.no_wrap ? 0u : m->angle_mask;
  u.eps_out = w.eps;
  if (!mode.forward_only) {
    u.x_out = w.x;
    if (mode.use_dyn) {
      u.dyn = w.dyn;
    } else {
      u.noise = mode.noise;
      u.noise_stride = 0;
      u.t_start = mode.t_start;
    }
  }
  if (mode.use_dyn) u.noise_stride = (long long)B * L * F;
  PROF(KC_HEAD_UPDATE, launch_head_update_img(u, hi, max_rows, s));
  HIP_TRY(hipGetLastError());
  return FD_OK;
#undef DBG_STOP
}

// token-row table of the batch (row-image path); `packed`: only the first lens[b] positions are rows
int prepare_rows(fd_model* m, hipStream_t s, int packed) {
  Workspace& w = m->ws;
  if (!w.img) return FD_OK;
  launch_build_rows(w.lens, w.B, w.L, packed, w.cap, w.seq_row0, w.nrow, w.rowinfo, w.dims, s);
  HIP_TRY(hipGetLastError());
  return FD_OK;
}

__global__ void set_int_kernel(int* p, int v) { *p = v; }
__global__ void set_dyn_kernel(UpdateDyn* p, UpdateDyn v) { *p = v; }

int set_t(fd_model* m, hipStream_t s, int t) {
  hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(1), 0, s, m->ws.t_dev, t);
  HIP_TRY(hipGetLastError());
  return FD_OK;
}

int check_shape(fd_model* m, int B, int L, int t) {
  if (!m) return fail(FD_E_INVALID, "null model");
  if (!m->finalized) return fail(FD_E_STATE, "fd_finalize has not been called");
  if (B < 1 || L < 1) return fail(FD_E_INVALID, "B=%d L=%d must be positive", B, L);
  if (t < 0 || t >= m->T) return fail(FD_E_INVALID, "timestep %d outside [0, %d)", t, m->T);
  if (m->cfg.pos_type != FD_POS_ABSOLUTE && L > m->cfg.max_pos)
    return fail(FD_E_INVALID, "L=%d exceeds max_position_embeddings=%d", L, m->cfg.max_pos);
  if (m->cfg.pos_type == FD_POS_ABSOLUTE && L > m->cfg.max_pos)
    return fail(FD_E_INVALID, "L=%d exceeds max_position_embeddings=%d", L, m->cfg.max_pos);
  if (L > 128 && !m->img)
    return fail(FD_E_UNSUPPORTED, "L=%d: the exact-fp32 attention kernel handles L <= 128; use FD_PREC_F16X3", L);
  return FD_OK;
}

int check_lens(const int32_t* lens, int B, int L) {
  for (int i = 0; i < B; ++i)
    if (lens[i] < 1 || lens[i] > L) return fail(FD_E_INVALID, "lens[%d]=%d outside [1, %d]", i, lens[i], L);
  return FD_OK;
}

// the workspace's captured graph still holds the launch sequence the model's options ask for
static bool graph_current(const fd_model* m, const Workspace& w) {
  return w.graph && w.graph_fuse_ln == m->fuse_ln && w.graph_varlen == m->varlen && w.graph_fuse_attn == m->fuse_attn && w.graph_fuse_ffn == m->fuse_ffn &&
         w.graph_rows_hint == m->rows_hint;
}

int ensure_graph(fd_model* m) {
  Workspace& w = m->ws;
  if (graph_current(m, w)) return FD_OK;  // (a precision change goes through fd_finalize, which drops the workspace)
  if (w.graph) {
    (void)hipGraphExecDestroy(w.graph);
    w.graph = nullptr;
  }
  // one eager step first: loads code objects and sets function attributes outside capture
  int rc = set_t(m, m->stream, 0);
  if (rc) return rc;
  StepMode warm{};
  warm.forward_only = true;
  rc = run_step(m, m->stream, warm);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(m->stream));
  StepMode mode{};
  mode.use_dyn = true;
  mode.advance = true;
  hipGraph_t graph = nullptr;
  HIP_TRY(hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal));
  rc = run_step(m, m->stream, mode);
  hipError_t e = hipStreamEndCapture(m->stream, &graph);
  if (rc) {
    if (graph) (void)hipGraphDestroy(graph);
    return rc;
  }
  if (e != hipSuccess) return fail(FD_E_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
  e = hipGraphInstantiate(&w.graph, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) return fail(FD_E_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
  w.graph_fuse_ln = m->fuse_ln;
  w.graph_varlen = m->varlen;
  w.graph_fuse_attn = m->fuse_attn;
  w.graph_fuse_ffn = m->fuse_ffn;
  w.graph_rows_hint = m->rows_hint;
  return FD_OK;
}

// SURVEY 5 (failure detection): the update kernel raises a device flag when the model output is not finite
// (it would silently poison every later step).  Call with the model's work complete.
int check_flag(fd_model* m) {
  Workspace& w = m->ws;
  if (!w.img || !w.flag) return FD_OK;
  int v = 0;
  HIP_TRY(hipMemcpy(&v, w.flag, 4, hipMemcpyDeviceToHost));
  if (!v) return FD_OK;
  HIP_TRY(hipMemset(w.flag, 0, 4));
  return fail(FD_E_NONFINITE, "the model produced a non-finite value (inf/NaN in the predicted noise)");
}

// Test hook plumbing of the row-image GEMM: fp32 host operands -> images (scales chosen from the data exactly as
// fd_finalize chooses them from weight bounds) -> production kernel -> fp32.
struct ImgHook {
  std::vector<void*> bufs;
  ~ImgHook() {
    for (void* p : bufs) (void)hipFree(p);
  }
  hipError_t up(const void* host, size_t bytes, void** dev) {
    hipError_t e = hipMalloc(dev, bytes);
    if (e != hipSuccess) return e;
    bufs.push_back(*dev);
    return host ? hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice) : hipMemset(*dev, 0, bytes);
  }
};

float max_abs(const float* p, size_t n) {
  float mx = 0.f;
  for (size_t i = 0; i < n; ++i) mx = std::fmax(mx, std::fabs(p[i]));
  return mx;
}

// epilogue: EPI_IMG_BIAS | EPI_IMG_GELU | EPI_IMG_LN
int img_gemm_hook(int epilogue, const float* A, const float* W, const float* bias, const float* resid, const float* gamma,
                  const float* beta, float eps, float* C, int M, int N, int K) {
  if (N % 32 || K % 32) return fail(FD_E_UNSUPPORTED, "row-image GEMM: N=%d and K=%d must be multiples of 32", N, K);
  if (epilogue == EPI_IMG_LN && N > 384) return fail(FD_E_UNSUPPORTED, "LN-fused GEMM: N=%d > 384", N);
  ImgHook hk;
#define I_TRY(expr)                                                                               \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return fail(FD_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));    \
  } while (0)
  const long long rows = ((long long)M + 127) / 128 * 128;
  float *dA, *db, *dg = nullptr, *dbt = nullptr, *dR = nullptr, *dC;
  void *dAi, *dWi, *dRi = nullptr, *dOi, *dtrash;
  int* ddims;
  I_TRY(hk.up(A, (size_t)M * K * 4, (void**)&dA));
  I_TRY(hk.up(bias, (size_t)N * 4, (void**)&db));
  I_TRY(hk.up(nullptr, (size_t)rows * K * 4, &dAi));
  I_TRY(hk.up(nullptr, (size_t)rows * N * 4, &dOi));
  I_TRY(hk.up(nullptr, (size_t)rows * N * 4, (void**)&dC));
  I_TRY(hk.up(nullptr, 1024, &dtrash));
  const int hd[2] = {M, (int)rows};
  I_TRY(hk.up(hd, sizeof hd, (void**)&ddims));
  std::vector<uint16_t> img;
  float wscale = 1.f;
  pack_weight_tiles(W, N, K, &img, &wscale);
  I_TRY(hk.up(img.data(), img.size() * 2, &dWi));
  const float a_scale = scale_for(max_abs(A, (size_t)M * K));
  launch_f32_to_img(dA, dAi, rows, K, M, a_scale, nullptr);
  // output bound as fd_finalize derives it: ||row of A||_2 ||row of W||_2 + |bias| (+ residual); LayerNorm: gamma/beta
  float in_l2 = 0.f;
  for (int r = 0; r < M; ++r) {
    double n2 = 0.0;
    for (int k = 0; k < K; ++k) n2 += (double)A[(size_t)r * K + k] * A[(size_t)r * K + k];
    in_l2 = std::fmax(in_l2, (float)std::sqrt(n2));
  }
  GemmImgArgs g;
  memset(&g, 0, sizeof g);
  g.A = static_cast<const unsigned char*>(dAi);
  g.W = static_cast<const unsigned char*>(dWi);
  g.bias = db;
  g.out = static_cast<unsigned char*>(dOi);
  g.trash = static_cast<unsigned char*>(dtrash);
  g.dims = ddims;
  g.N = N;
  g.K = K;
  g.acc_scale = 1.0f / (a_scale * wscale);
  g.eps = eps;
  float out_scale;
  if (epilogue == EPI_IMG_LN) {
    I_TRY(hk.up(gamma, (size_t)N * 4, (void**)&dg));
    I_TRY(hk.up(beta, (size_t)N * 4, (void**)&dbt));
    I_TRY(hk.up(resid, (size_t)M * N * 4, (void**)&dR));
    I_TRY(hk.up(nullptr, (size_t)rows * N * 4, &dRi));
    const float r_scale = scale_for(max_abs(resid, (size_t)M * N));
    launch_f32_to_img(dR, dRi, rows, N, M, r_scale, nullptr);
    g.resid = static_cast<const unsigned char*>(dRi);
    g.resid_inv = 1.0f / r_scale;
    g.gamma = dg;
    g.beta = dbt;
    out_scale = scale_for(max_abs(gamma, N) * std::sqrt((float)N) + max_abs(beta, N));
  } else {
    out_scale = scale_for(dense_bound(W, bias, 0, N, K, in_l2));
  }
  g.out_scale = out_scale;
  g.tail = 1;  // (the hook's ragged row counts exercise the tail slices)
  launch_gemm_img(epilogue, g, (int)rows, nullptr);
  launch_img_to_f32(dOi, dC, rows, N, out_scale, nullptr);
  I_TRY(hipGetLastError());
  I_TRY(hipDeviceSynchronize());
  I_TRY(hipMemcpy(C, dC, (size_t)M * N * 4, hipMemcpyDeviceToHost));
#undef I_TRY
  return FD_OK;
}

}  // namespace

// ---- RCCL, bound at run time (dlopen): the library has no link-time dependency on it, and a host process that already
// carries an RCCL (PyTorch does) shares that instance.  Only what the single gather of SURVEY 8(e) needs.
namespace {
struct RcclId {
  char internal[128];
};
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(RcclId*) = nullptr;
  int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (r.lib) {
      r.GetUniqueId = reinterpret_cast<int (*)(RcclId*)>(dlsym(r.lib, "ncclGetUniqueId"));
      r.CommInitRank = reinterpret_cast<int (*)(void**, int, RcclId, int)>(dlsym(r.lib, "ncclCommInitRank"));
      r.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(r.lib, "ncclCommDestroy"));
      r.AllGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(r.lib, "ncclAllGather"));
      r.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(r.lib, "ncclGetErrorString"));
      if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) r.lib = nullptr;
    }
  }
  return r.lib ? &r : nullptr;
}
int rccl_fail(const char* what, int rc) {
  Rccl* r = rccl();
  return fail(FD_E_HIP, "%s failed: %s", what, (r && r->GetErrorString) ? r->GetErrorString(rc) : "RCCL error");
}
}  // namespace

// ============================================================================ C ABI
extern "C" {

int fd_comm_unique_id(void* id_out) {
  if (!id_out) return fail(FD_E_INVALID, "null argument");
  Rccl* r = rccl();
  if (!r) return fail(FD_E_UNSUPPORTED, "librccl.so could not be loaded");
  RcclId id;
  if (int rc = r->GetUniqueId(&id)) return rccl_fail("ncclGetUniqueId", rc);
  memcpy(id_out, &id, sizeof id);
  return FD_OK;
}

int fd_comm_init(fd_model* m, int rank, int world, const void* unique_id) {
  if (!m || !unique_id) return fail(FD_E_INVALID, "null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(FD_E_INVALID, "rank %d of %d", rank, world);
  Rccl* r = rccl();
  if (!r) return fail(FD_E_UNSUPPORTED, "librccl.so could not be loaded");
  HIP_TRY(hipSetDevice(m->device));
  if (m->comm) {
    (void)r->CommDestroy(m->comm);
    m->comm = nullptr;
  }
  RcclId id;
  memcpy(&id, unique_id, sizeof id);
  if (int rc = r->CommInitRank(&m->comm, world, id, rank)) return rccl_fail("ncclCommInitRank", rc);
  m->comm_world = world;
  m->comm_rank = rank;
  return FD_OK;
}

int fd_comm_destroy(fd_model* m) {
  if (!m) return fail(FD_E_INVALID, "null argument");
  Rccl* r = rccl();
  if (m->comm && r) (void)r->CommDestroy(m->comm);
  m->comm = nullptr;
  m->comm_world = 0;
  return FD_OK;
}

int fd_gather_dev(fd_model* m, const void* local_dev, int64_t n_floats, void* out_dev, void* hip_stream) {
  if (!m || !local_dev || !out_dev || n_floats < 0) return fail(FD_E_INVALID, "bad argument");
  if (!m->comm) return fail(FD_E_STATE, "fd_comm_init has not been called");
  Rccl* r = rccl();
  HIP_TRY(hipSetDevice(m->device));
  hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : m->stream;
  if (int rc = r->AllGather(local_dev, out_dev, (size_t)n_floats, /*ncclFloat*/ 7, m->comm, s)) return rccl_fail("ncclAllGather", rc);
  return FD_OK;
}

int fd_abi_version(void) { return FDMI_ABI_VERSION; }

const char* fd_last_error(void) { return g_err.c_str(); }

int fd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int fd_create(const fd_config* cfg, int device_id, fd_model** out) {
  if (!cfg || !out) return fail(FD_E_INVALID, "null argument");
  *out = nullptr;
  const fd_config& c = *cfg;
  if (c.n_features < 1 || c.n_features > kMaxFeat) return fail(FD_E_INVALID, "n_features=%d outside [1,%d]", c.n_features, kMaxFeat);
  if (c.n_heads < 1 || c.d_model < 32 || c.d_model % c.n_heads != 0)
    return fail(FD_E_INVALID, "d_model=%d is not a multiple of n_heads=%d", c.d_model, c.n_heads);
  {
    const int hd = c.d_model / c.n_heads;
    if (hd != 32 && hd != 64 && hd != 96 && hd != 128)
      return fail(FD_E_UNSUPPORTED, "d_model=%d n_heads=%d: attention head size %d is not one of 32, 64, 96, 128", c.d_model, c.n_heads, hd);
  }
  if (c.d_model > 1024) return fail(FD_E_UNSUPPORTED, "d_model=%d > 1024", c.d_model);
  if (c.d_ff < 32 || c.d_ff % 32) return fail(FD_E_UNSUPPORTED, "d_ff=%d must be a positive multiple of 32", c.d_ff);
  if (c.n_layers < 1 || c.max_pos < 1) return fail(FD_E_INVALID, "n_layers=%d max_pos=%d", c.n_layers, c.max_pos);
  if (c.pos_type != FD_POS_ABSOLUTE && c.pos_type != FD_POS_RELATIVE_KEY && c.pos_type != FD_POS_RELATIVE_KEY_QUERY)
    return fail(FD_E_INVALID, "pos_type=%d", c.pos_type);
  if (c.decoder != FD_DEC_MLP && c.decoder != FD_DEC_LINEAR) return fail(FD_E_INVALID, "decoder=%d", c.decoder);
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev) return fail(FD_E_INVALID, "device %d of %d", device_id, ndev);
  HIP_TRY(hipSetDevice(device_id));
  fd_model* m = new fd_model();
  m->cfg = c;
  m->device = device_id;
  hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete m;
    return fail(FD_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
  }
  *out = m;
  return FD_OK;
}

int fd_set_weight(fd_model* m, const char* name, const float* host_data, const int64_t* shape, int ndim) {
  if (!m || !name || !host_data || (ndim > 0 && !shape)) return fail(FD_E_INVALID, "null argument");
  std::vector<int64_t> want;
  bool ignored = false;
  if (!expected_shape(m->cfg, name, &want, &ignored)) return fail(FD_E_INVALID, "unexpected state_dict entry '%s'", name);
  if (ignored) return FD_OK;
  std::vector<int64_t> got(shape, shape + ndim);
  if (got != want) {
    std::string a, b;
    for (auto v : got) a += std::to_string(v) + ",";
    for (auto v : want) b += std::to_string(v) + ",";
    return fail(FD_E_INVALID, "'%s': shape [%s] does not match config [%s]", name, a.c_str(), b.c_str());
  }
  size_t n = 1;
  for (auto v : got) n *= (size_t)v;
  HostTensor& t = m->host[name];
  t.shape = got;
  t.data.assign(host_data, host_data + n);
  m->finalized = false;
  return FD_OK;
}

int fd_finalize(fd_model* m, int T, const float* coef, const float* time_table, const uint8_t* is_angle, int precision) {
  if (!m || !coef || !time_table || !is_angle) return fail(FD_E_INVALID, "null argument");
  if (T < 1) return fail(FD_E_INVALID, "T=%d", T);
  if (precision != FD_PREC_F32 && precision != FD_PREC_F16X3) return fail(FD_E_UNSUPPORTED, "precision mode %d", precision);
  if (precision == FD_PREC_F32 && head_dim(m->cfg) != kHeadDim)
    return fail(FD_E_UNSUPPORTED, "head size %d: the exact-fp32 attention kernel is built for head size %d; use FD_PREC_F16X3", head_dim(m->cfg), kHeadDim);
  HIP_TRY(hipSetDevice(m->device));
  HIP_TRY(hipStreamSynchronize(m->stream));
  drop_workspaces(m);
  free_weights(m);
  const fd_config& c = m->cfg;
  const size_t d = c.d_model, F = c.n_features, ff = c.d_ff;
  // FD_PREC_F16X3 runs on the row-image kernels; the attention-output / FFN-down LayerNorm is fused into the GEMM when a row
  // fits one 384-column workgroup tile (d_model <= 384: every released configuration), else it is its own launch
  const bool img = precision == FD_PREC_F16X3;
  m->img = img;
#define NEED(var, nm)                         \
  const HostTensor* var = need(m, nm);        \
  if (!var) return FD_E_MISSING
#define UP(dst, t) \
  if (int rc_ = upload(m, &(dst), (t)->data.data(), (t)->data.size())) return rc_
  NEED(t_win, "inputs_to_hidden_dim.weight");
  NEED(t_bin, "inputs_to_hidden_dim.bias");
  NEED(t_eg, "embeddings.LayerNorm.weight");
  NEED(t_eb, "embeddings.LayerNorm.bias");
  UP(m->w_in, t_win);
  UP(m->b_in, t_bin);
  UP(m->emb_g, t_eg);
  UP(m->emb_b, t_eb);
  // bound of the hidden state entering layer 0: embeddings LayerNorm + time embedding (|sin|, |cos| <= 1)
  Bound hb = ln_bound(t_eg, t_eb, 1.0f);
  m->pos_emb = nullptr;
  if (c.pos_type == FD_POS_ABSOLUTE) {
    NEED(t_pe, "embeddings.position_embeddings.weight");
    UP(m->pos_emb, t_pe);
  }
  m->layers.resize(c.n_layers);
  for (int li = 0; li < c.n_layers; ++li) {
    const std::string p = "encoder.layer." + std::to_string(li) + ".";
    LayerDev& lw = m->layers[li];
    NEED(wq, p + "attention.self.query.weight");
    NEED(wk, p + "attention.self.key.weight");
    NEED(wv, p + "attention.self.value.weight");
    NEED(bq, p + "attention.self.query.bias");
    NEED(bk, p + "attention.self.key.bias");
    NEED(bv, p + "attention.self.value.bias");
    std::vector<float> wqkv, bqkv;  // rows: [query | key | value]  => one N = 3d GEMM
    wqkv.reserve(3 * d * d);
    for (const HostTensor* t : {wq, wk, wv}) wqkv.insert(wqkv.end(), t->data.begin(), t->data.end());
    for (const HostTensor* t : {bq, bk, bv}) bqkv.insert(bqkv.end(), t->data.begin(), t->data.end());
    if (int rc = upload(m, &lw.wqkv, wqkv.data(), wqkv.size())) return rc;
    if (int rc = upload(m, &lw.bqkv, bqkv.data(), bqkv.size())) return rc;
    if (img) {
      if (int rc = upload_split(m, &lw.wqk_i, wqkv.data(), 2 * (int)d, (int)d, 384)) return rc;
      if (int rc = upload_split(m, &lw.wv_i, wqkv.data() + 2 * d * d, (int)d, (int)d, 384)) return rc;
      if (sub_heads(c) % 6 == 0)
        if (int rc = upload_split(m, &lw.wqkv_i, wqkv.data(), 3 * (int)d, (int)d, 384)) return rc;
      lw.wsa_i = SplitW();
      if (c.pos_type == FD_POS_RELATIVE_KEY && seq_attn_supported((int)d, c.n_heads, c.max_pos < 128 ? c.max_pos : 128, c.max_pos))
        if (int rc = upload_seq_attn_weights(m, &lw.wsa_i, wqkv.data(), (int)d)) return rc;
      lw.wsa16_i = SplitW();
      if (c.pos_type == FD_POS_RELATIVE_KEY && seq_attn16_supported((int)d, c.n_heads, c.max_pos < 128 ? c.max_pos : 128, c.max_pos))
        if (int rc = upload_seq_attn16_weights(m, &lw.wsa16_i, wqkv.data(), (int)d)) return rc;
      lw.bqk = lw.bqkv;
      lw.bv = lw.bqkv + 2 * d;
      lw.s_h = scale_for(hb.linf);
      lw.s_q = scale_for(dense_bound(wqkv.data(), bqkv.data(), 0, (int)d, (int)d, hb.l2));
      lw.s_k = scale_for(dense_bound(wqkv.data(), bqkv.data(), (int)d, 2 * (int)d, (int)d, hb.l2));
      lw.s_v = scale_for(dense_bound(wqkv.data(), bqkv.data(), 2 * (int)d, 3 * (int)d, (int)d, hb.l2));  // also bounds ctx
    }
    lw.demb = nullptr;
    if (c.pos_type != FD_POS_ABSOLUTE) {  // relative_key and relative_key_query
      NEED(de, p + "attention.self.distance_embedding.weight");
      UP(lw.demb, de);
      if (precision == FD_PREC_F16X3)
        if (int rc = upload_split(m, &lw.demb_s, de->data.data(), 2 * c.max_pos - 1, head_dim(c))) return rc;
    }
    NEED(wo, p + "attention.output.dense.weight");
    NEED(bo, p + "attention.output.dense.bias");
    NEED(g1, p + "attention.output.LayerNorm.weight");
    NEED(b1, p + "attention.output.LayerNorm.bias");
    NEED(wi, p + "intermediate.dense.weight");
    NEED(bi, p + "intermediate.dense.bias");
    NEED(wd, p + "output.dense.weight");
    NEED(bd, p + "output.dense.bias");
    NEED(g2, p + "output.LayerNorm.weight");
    NEED(b2, p + "output.LayerNorm.bias");
    UP(lw.wo, wo); UP(lw.bo, bo); UP(lw.ln1g, g1); UP(lw.ln1b, b1);
    UP(lw.wi, wi); UP(lw.bi, bi); UP(lw.wd, wd); UP(lw.bd, bd); UP(lw.ln2g, g2); UP(lw.ln2b, b2);
    if (img) {
      if (int rc = upload_split(m, &lw.wo_i, wo->data.data(), (int)d, (int)d, 384)) return rc;
      if (int rc = upload_split(m, &lw.wi_i, wi->data.data(), (int)ff, (int)d, 384)) return rc;
      if (int rc = upload_split(m, &lw.wd_i, wd->data.data(), (int)d, (int)ff, 384)) return rc;
      lw.wff_i = SplitW();
      lw.wtail_i = SplitW();
      if (ffn16_supported((int)d, (int)ff))
        if (int rc = upload_ffn16_weights(m, &lw, wi->data.data(), wd->data.data(), wo->data.data(), (int)d, (int)ff)) return rc;
      const Bound ab = ln_bound(g1, b1);
      lw.s_a = scale_for(ab.linf);
      lw.s_g = scale_for(dense_bound(wi->data.data(), bi->data.data(), 0, (int)ff, (int)d, ab.l2));  // |gelu(u)| <= |u|
      hb = ln_bound(g2, b2);  // hidden state entering the next layer
    }
  }
  m->s_hfinal = scale_for(hb.linf);
  if (c.decoder == FD_DEC_MLP) {
    NEED(w1, "token_decoder.dense1.weight");
    NEED(b1, "token_decoder.dense1.bias");
    NEED(g, "token_decoder.layer_norm.weight");
    NEED(b, "token_decoder.layer_norm.bias");
    NEED(w2, "token_decoder.dense2.weight");
    NEED(b2, "token_decoder.dense2.bias");
    UP(m->hd_w1, w1); UP(m->hd_b1, b1); UP(m->hd_g, g); UP(m->hd_b, b); UP(m->hd_w2, w2); UP(m->hd_b2, b2);
    if (img) {
      if (int rc = upload_split(m, &m->hd_w1_i, w1->data.data(), (int)d, (int)d, 384)) return rc;
      m->s_hg = scale_for(dense_bound(w1->data.data(), b1->data.data(), 0, (int)d, (int)d, hb.l2));
    }
  } else {
    NEED(w2, "token_decoder.weight");
    NEED(b2, "token_decoder.bias");
    UP(m->hd_w2, w2); UP(m->hd_b2, b2);
  }
#undef NEED
#undef UP
  if (int rc = upload(m, &m->coef, coef, (size_t)4 * T)) return rc;
  if (int rc = upload(m, &m->time_table, time_table, (size_t)T * d)) return rc;
  m->angle_mask = 0;
  for (size_t f = 0; f < F; ++f)
    if (is_angle[f]) m->angle_mask |= 1u << f;
  m->T = T;
  m->precision = precision;
  m->finalized = true;
  return FD_OK;
}

void fd_destroy(fd_model* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  (void)fd_comm_destroy(m);
  if (m->stream) (void)hipStreamSynchronize(m->stream);
  drop_workspaces(m);
  free_weights(m);
  for (PendingEvent& p : m->pending) {
    (void)hipEventDestroy(p.e0);
    (void)hipEventDestroy(p.e1);
  }
  for (hipEvent_t e : m->event_pool) (void)hipEventDestroy(e);
  if (m->stream) (void)hipStreamDestroy(m->stream);
  delete m;
}

int fd_set_option(fd_model* m, const char* name, int value) {
  if (!m || !name) return fail(FD_E_INVALID, "null argument");
  const std::string n = name;
  if (n == "fuse_ln") m->fuse_ln = value < 0 ? -1 : (value ? 1 : 0);
  else if (n == "use_graph") m->use_graph = value ? 1 : 0;
  else if (n == "varlen") m->varlen = value ? 1 : 0;
  else if (n == "fuse_attn") m->fuse_attn = value < 0 ? -1 : (value > 2 ? 1 : value);
  else if (n == "fuse_ffn") m->fuse_ffn = value < 0 ? -1 : (value > 2 ? 2 : value);
  else if (n == "rows_hint") m->rows_hint = value < 0 ? 0 : value;
  else if (n == "split_qkv") {
    m->split_qkv = value ? 1 : 0;
    drop_workspaces(m);  // captured graphs hold the other launch sequence
  }
  else if (n == "debug_stop") {
    if (m->debug_stop != value) drop_workspaces(m);  // a graph captured with a truncated step must not be replayed afterwards
    m->debug_stop = value;
  }
  else if (n == "debug_layer") m->debug_layer = value;
  else return fail(FD_E_INVALID, "unknown option '%s'", name);
  return FD_OK;
}

int fd_fused_attn_supported(fd_model* m, int L) {
  if (!m) return fail(FD_E_INVALID, "null argument");
  if (!m->finalized) return fail(FD_E_STATE, "fd_finalize has not been called");
  if (m->precision != FD_PREC_F16X3 || m->layers.empty() || L < 1) return 0;
  const fd_config& c = m->cfg;
  const LayerDev& lw = m->layers[0];
  if (m->fuse_attn == 2) return lw.wsa_i.p && seq_attn_supported(c.d_model, c.n_heads, L, c.max_pos) ? 1 : 0;
  return lw.wsa16_i.p && seq_attn16_supported(c.d_model, c.n_heads, L, c.max_pos) ? 1 : 0;
}

int fd_forward(fd_model* m, const float* x, int t, const int32_t* lens, int B, int L, float* eps_out) {
  if (int rc = check_shape(m, B, L, t)) return rc;
  if (!x || !lens || !eps_out) return fail(FD_E_INVALID, "null argument");
  if (int rc = check_lens(lens, B, L)) return rc;
  HIP_TRY(hipSetDevice(m->device));
  if (int rc = ensure_ws(m, B, L)) return rc;
  Workspace& w = m->ws;
  const size_t n = (size_t)B * L * m->cfg.n_features;
  hipStream_t s = m->stream;
  HIP_TRY(hipMemcpyAsync(w.x, x, n * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(w.lens, lens, (size_t)B * 4, hipMemcpyHostToDevice, s));
  if (int rc = prepare_rows(m, s, 0)) return rc;  // the forward defines every position, masked ones included
  if (int rc = set_t(m, s, t)) return rc;
  StepMode mode{};
  mode.forward_only = true;
  if (int rc = run_step(m, s, mode)) return rc;
  HIP_TRY(hipMemcpyAsync(eps_out, w.eps, n * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return check_flag(m);
}

int fd_forward_ex(fd_model* m, const float* x, int t, const uint8_t* key_mask, const int32_t* position_ids, int B, int L,
                  float* eps_out) {
  if (int rc = check_shape(m, B, L, t)) return rc;
  if (!x || !eps_out) return fail(FD_E_INVALID, "null argument");
  if (!key_mask && !position_ids) return fail(FD_E_INVALID, "fd_forward_ex without a mask and without position ids: use fd_forward");
  if (!m->img) return fail(FD_E_UNSUPPORTED, "fd_forward_ex (arbitrary key masks / position ids) needs FD_PREC_F16X3");
  const fd_config& c = m->cfg;
  if (position_ids && c.pos_type == FD_POS_ABSOLUTE)
    for (size_t i = 0; i < (size_t)B * L; ++i)
      if (position_ids[i] < 0 || position_ids[i] >= c.max_pos)
        return fail(FD_E_INVALID, "position_ids[%zu]=%d outside [0, %d)", i, position_ids[i], c.max_pos);
  HIP_TRY(hipSetDevice(m->device));
  if (int rc = ensure_ws(m, B, L)) return rc;
  Workspace& w = m->ws;
  const size_t n = (size_t)B * L * c.n_features;
  hipStream_t s = m->stream;
  // the lazily allocated buffers first: an out-of-memory return must not leave copies from this frame in flight
  if (key_mask && !w.kmask) HIP_TRY(hipMalloc((void**)&w.kmask, (size_t)B * L));
  if (position_ids && c.pos_type == FD_POS_ABSOLUTE && !w.pos_ids) HIP_TRY(hipMalloc((void**)&w.pos_ids, (size_t)B * L * 4));
  std::vector<int32_t> lens(B, L);  // every position is a row and a key; what is attended to is the mask's business
  // `lens` is a pageable vector of this frame, and the early returns below would destroy it under an asynchronous copy: a blocking
  // one -- which goes through the NULL stream and is not ordered against work still queued on the model's non-blocking stream (an
  // earlier asynchronous entry point reading w.lens), so that stream is drained first
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipMemcpy(w.lens, lens.data(), (size_t)B * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpyAsync(w.x, x, n * 4, hipMemcpyHostToDevice, s));
  StepMode mode{};
  mode.forward_only = true;
  if (key_mask) {
    HIP_TRY(hipMemcpyAsync(w.kmask, key_mask, (size_t)B * L, hipMemcpyHostToDevice, s));
    mode.kmask = w.kmask;
  }
  // (relative position types: the reference's embeddings add no position embedding and HF's distance uses arange, so
  // position_ids change nothing there -- modelling.py:157-166, BertSelfAttention)
  if (position_ids && c.pos_type == FD_POS_ABSOLUTE) {
    HIP_TRY(hipMemcpyAsync(w.pos_ids, position_ids, (size_t)B * L * 4, hipMemcpyHostToDevice, s));
    mode.pos_ids = w.pos_ids;
  }
  if (int rc = prepare_rows(m, s, 0)) return rc;
  if (int rc = set_t(m, s, t)) return rc;
  if (int rc = run_step(m, s, mode)) return rc;
  HIP_TRY(hipMemcpyAsync(eps_out, w.eps, n * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return check_flag(m);
}

int fd_p_sample_step(fd_model* m, const float* x, int t, const int32_t* lens, int B, int L, const float* z, int wrap,
                     float* x_out) {
  if (int rc = check_shape(m, B, L, t)) return rc;
  if (!x || !lens || !x_out) return fail(FD_E_INVALID, "null argument");
  if (t > 0 && !z) return fail(FD_E_INVALID, "z is required for t > 0");
  if (int rc = check_lens(lens, B, L)) return rc;
  HIP_TRY(hipSetDevice(m->device));
  if (int rc = ensure_ws(m, B, L)) return rc;
  Workspace& w = m->ws;
  const size_t n = (size_t)B * L * m->cfg.n_features;
  hipStream_t s = m->stream;
  HIP_TRY(hipMemcpyAsync(w.x, x, n * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(w.lens, lens, (size_t)B * 4, hipMemcpyHostToDevice, s));
  if (z) HIP_TRY(hipMemcpyAsync(w.z, z, n * 4, hipMemcpyHostToDevice, s));
  if (int rc = prepare_rows(m, s, 0)) return rc;
  if (int rc = set_t(m, s, t)) return rc;
  StepMode mode{};
  mode.noise = w.z;
  mode.t_start = t;
  mode.no_wrap = !wrap;
  if (int rc = run_step(m, s, mode)) return rc;
  HIP_TRY(hipMemcpyAsync(x_out, w.x, n * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return check_flag(m);
}

int fd_sample_begin_dev(fd_model* m, const void* x_init_dev, const void* lens_dev, int B, int L, int t_start,
                        uint64_t seed, int64_t seq_offset, void* out_dev, int full_history, void* hip_stream) {
  if (int rc = check_shape(m, B, L, t_start)) return rc;
  if (!x_init_dev || !lens_dev || !out_dev) return fail(FD_E_INVALID, "null argument");
  if (full_history < 0) return fail(FD_E_INVALID, "full_history = %d", full_history);
  HIP_TRY(hipSetDevice(m->device));
  if (int rc = ensure_ws(m, B, L)) return rc;
  Workspace& w = m->ws;
  hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : m->stream;
  const size_t n = (size_t)B * L * m->cfg.n_features;
  HIP_TRY(hipMemcpyAsync(w.x, x_init_dev, n * 4, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(w.lens, lens_dev, (size_t)B * 4, hipMemcpyDeviceToDevice, s));
  if (m->varlen && full_history)  // packed rows: positions beyond a sequence's length are never written
    HIP_TRY(hipMemsetAsync(out_dev, 0, ((size_t)(t_start + full_history) / full_history) * n * 4, s));
  if (int rc = prepare_rows(m, s, m->varlen)) return rc;
  if (m->use_graph && !graph_current(m, w)) {
    // first use of this (B, L): the warm-up step and the capture run on the model's stream and need the
    // lengths / row table that were just queued on `s`
    if (s != m->stream) HIP_TRY(hipStreamSynchronize(s));
    if (int rc = ensure_graph(m)) return rc;
    HIP_TRY(hipStreamSynchronize(m->stream));
  }
  UpdateDyn& dyn = m->dyn_host;
  memset(&dyn, 0, sizeof dyn);
  dyn.hist = full_history ? static_cast<float*>(out_dev) : nullptr;
  dyn.seed = seed;
  dyn.seq_offset = seq_offset;
  dyn.t_start = t_start;
  dyn.hist_every = full_history;
  m->run_t = t_start;
  m->run_open = true;
  m->run_B = B;
  m->run_L = L;
  if (int rc = set_t(m, s, t_start)) return rc;
  return FD_OK;
}

int fd_sample_steps_dev(fd_model* m, int n_steps, const void* noise_dev, int noise_t0, void* hip_stream) {
  if (!m || !m->finalized || m->ws.B == 0) return fail(FD_E_STATE, "fd_sample_steps_dev without fd_sample_begin_dev");
  if (!m->run_open) return fail(FD_E_STATE, "fd_sample_steps_dev: no sampling run in progress (another call on this model has taken its workspace)");
  if (m->ws.B != m->run_B || m->ws.L != m->run_L)
    return fail(FD_E_STATE, "fd_sample_steps_dev: the workspace now holds B=%d L=%d, the run was begun with B=%d L=%d", m->ws.B, m->ws.L, m->run_B, m->run_L);
  if (n_steps < 0 || n_steps > m->run_t + 1) return fail(FD_E_INVALID, "n_steps = %d with %d steps left", n_steps, m->run_t + 1);
  if (noise_dev && (noise_t0 < 0 || noise_t0 > m->run_t - n_steps + 1))  // (the row of t = 0 is read too, and multiplied by sigma_0 = 0)
    return fail(FD_E_INVALID, "noise rows start at t = %d, the steps run down to t = %d", noise_t0, m->run_t - n_steps + 1);
  HIP_TRY(hipSetDevice(m->device));
  Workspace& w = m->ws;
  hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : m->stream;
  // the kernel reads the draw of step t at noise + t * (B L F): rebase the chunk (row 0 = step noise_t0)
  const size_t n = (size_t)w.B * w.L * m->cfg.n_features;
  UpdateDyn dyn = m->dyn_host;
  dyn.noise = noise_dev ? static_cast<const float*>(noise_dev) - (ptrdiff_t)noise_t0 * (ptrdiff_t)n : nullptr;
  hipLaunchKernelGGL(set_dyn_kernel, dim3(1), dim3(1), 0, s, w.dyn, dyn);
  for (int i = 0; i < n_steps; ++i) {
    const int done = m->dyn_host.t_start - m->run_t;  // steps already run
    const bool prof = m->profile_every > 0 && (done % m->profile_every) == m->profile_every / 2;
    if (m->use_graph && !prof) {
      HIP_TRY(hipGraphLaunch(w.graph, s));
    } else {
      StepMode mode{};
      mode.use_dyn = true;
      mode.advance = true;
      mode.profile = prof;
      if (int rc = run_step(m, s, mode)) return rc;
    }
    --m->run_t;
  }
  return FD_OK;
}

int fd_sample_end_dev(fd_model* m, void* out_dev, void* hip_stream) {
  if (!m || !m->finalized || m->ws.B == 0) return fail(FD_E_STATE, "fd_sample_end_dev without fd_sample_begin_dev");
  if (!m->run_open) return fail(FD_E_STATE, "fd_sample_end_dev: no sampling run in progress (another call on this model has taken its workspace)");
  if (m->run_t >= 0) return fail(FD_E_STATE, "%d reverse steps have not been run", m->run_t + 1);
  if (m->ws.B != m->run_B || m->ws.L != m->run_L)
    return fail(FD_E_STATE, "fd_sample_end_dev: the workspace now holds B=%d L=%d, the run was begun with B=%d L=%d", m->ws.B, m->ws.L, m->run_B, m->run_L);
  HIP_TRY(hipSetDevice(m->device));
  Workspace& w = m->ws;
  hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : m->stream;
  const size_t n = (size_t)w.B * w.L * m->cfg.n_features;
  if (!m->dyn_host.hist_every) HIP_TRY(hipMemcpyAsync(out_dev, w.x, n * 4, hipMemcpyDeviceToDevice, s));
  m->run_open = false;
  return FD_OK;
}

int fd_sample_dev(fd_model* m, const void* x_init_dev, const void* lens_dev, int B, int L, int t_start,
                  const void* noise_dev, uint64_t seed, int64_t seq_offset, void* out_dev, int full_history,
                  void* hip_stream) {
  if (int rc = fd_sample_begin_dev(m, x_init_dev, lens_dev, B, L, t_start, seed, seq_offset, out_dev, full_history, hip_stream))
    return rc;
  if (int rc = fd_sample_steps_dev(m, t_start + 1, noise_dev, 0, hip_stream)) return rc;
  return fd_sample_end_dev(m, out_dev, hip_stream);
}

int fd_sample(fd_model* m, const float* x_init, const int32_t* lens, int B, int L, int t_start, const float* noise,
              uint64_t seed, float* out, int full_history) {
  return fd_sample_ex(m, x_init, lens, B, L, t_start, noise, seed, 0, out, full_history);
}

int fd_sample_ex(fd_model* m, const float* x_init, const int32_t* lens, int B, int L, int t_start, const float* noise,
                 uint64_t seed, int64_t seq_offset, float* out, int full_history) {
  if (int rc = check_shape(m, B, L, t_start)) return rc;
  if (!x_init || !lens || !out) return fail(FD_E_INVALID, "null argument");
  if (full_history < 0) return fail(FD_E_INVALID, "full_history = %d", full_history);
  if (int rc = check_lens(lens, B, L)) return rc;
  HIP_TRY(hipSetDevice(m->device));
  const size_t n = (size_t)B * L * m->cfg.n_features;
  const size_t nsteps = (size_t)t_start + 1;
  // rows of `out`: 1 (final only), every state, or every full_history-th state plus the final one
  const size_t out_rows = full_history ? (nsteps + full_history - 1) / full_history : 1;
  float *d_x = nullptr, *d_noise = nullptr, *d_out = nullptr;
  int* d_lens = nullptr;
  int rc = FD_OK;
  auto cleanup = [&]() {
    for (void* p : {(void*)d_x, (void*)d_noise, (void*)d_out, (void*)d_lens})
      if (p) (void)hipFree(p);
  };
#define TRY_CLEAN(expr)                                                                    \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      cleanup();                                                                           \
      return fail(FD_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));                \
    }                                                                                      \
  } while (0)
  TRY_CLEAN(hipMalloc((void**)&d_x, n * 4));
  TRY_CLEAN(hipMalloc((void**)&d_lens, (size_t)B * 4));
  TRY_CLEAN(hipMalloc((void**)&d_out, out_rows * n * 4));
  TRY_CLEAN(hipMemcpy(d_x, x_init, n * 4, hipMemcpyHostToDevice));
  TRY_CLEAN(hipMemcpy(d_lens, lens, (size_t)B * 4, hipMemcpyHostToDevice));
  if (noise) {
    TRY_CLEAN(hipMalloc((void**)&d_noise, nsteps * n * 4));
    TRY_CLEAN(hipMemcpy(d_noise, noise, nsteps * n * 4, hipMemcpyHostToDevice));
  }
  rc = fd_sample_dev(m, d_x, d_lens, B, L, t_start, d_noise, seed, seq_offset, d_out, full_history, nullptr);
  if (rc) {
    cleanup();
    return rc;
  }
  TRY_CLEAN(hipStreamSynchronize(m->stream));
  TRY_CLEAN(hipMemcpy(out, d_out, out_rows * n * 4, hipMemcpyDeviceToHost));
#undef TRY_CLEAN
  cleanup();
  return check_flag(m);
}

int fd_philox_normal_dev(fd_model* m, uint64_t seed, int t, int64_t seq_offset, int B, int L, void* out_dev,
                         void* hip_stream) {
  if (!m || !out_dev) return fail(FD_E_INVALID, "null argument");
  HIP_TRY(hipSetDevice(m->device));
  hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : m->stream;
  launch_philox_fill(static_cast<float*>(out_dev), seed, t, seq_offset, B, L, m->cfg.n_features, s);
  HIP_TRY(hipGetLastError());
  return FD_OK;
}

int fd_nerf(int device_id, const float* feats, const int32_t* lens, int B, int L, int F, const int32_t* feat_idx,
            int center, double* coords_out) {
  if (!feats || !lens || !feat_idx || !coords_out || B < 1 || L < 1 || F < 3) return fail(FD_E_INVALID, "bad argument");
  for (int i = 0; i < 9; ++i)
    if (feat_idx[i] >= F || (i < 3 && feat_idx[i] < 0)) return fail(FD_E_INVALID, "feat_idx[%d]=%d (F=%d)", i, feat_idx[i], F);
  if (int rc = check_lens(lens, B, L)) return rc;
  HIP_TRY(hipSetDevice(device_id));
  float* d_f = nullptr;
  int* d_l = nullptr;
  double* d_o = nullptr;
  auto cleanup = [&]() {
    for (void* p : {(void*)d_f, (void*)d_l, (void*)d_o})
      if (p) (void)hipFree(p);
  };
#define N_TRY(expr)                                                          \
  do {                                                                       \
    hipError_t e_ = (expr);                                                  \
    if (e_ != hipSuccess) {                                                  \
      cleanup();                                                             \
      return fail(FD_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));  \
    }                                                                        \
  } while (0)
  const size_t nf = (size_t)B * L * F, no = (size_t)B * 3 * L * 3;
  N_TRY(hipMalloc((void**)&d_f, nf * 4));
  N_TRY(hipMalloc((void**)&d_l, (size_t)B * 4));
  N_TRY(hipMalloc((void**)&d_o, no * 8));
  N_TRY(hipMemcpy(d_f, feats, nf * 4, hipMemcpyHostToDevice));
  N_TRY(hipMemcpy(d_l, lens, (size_t)B * 4, hipMemcpyHostToDevice));
  NerfFeatures fx{feat_idx[0], feat_idx[1], feat_idx[2], feat_idx[3], feat_idx[4], feat_idx[5], feat_idx[6], feat_idx[7], feat_idx[8]};
  launch_nerf(d_f, d_l, B, L, F, fx, center, d_o, nullptr);
  N_TRY(hipGetLastError());
  N_TRY(hipDeviceSynchronize());
  N_TRY(hipMemcpy(coords_out, d_o, no * 8, hipMemcpyDeviceToHost));
#undef N_TRY
  cleanup();
  return FD_OK;
}

int fd_shift_trim_dev(fd_model* m, const void* traj_dev, int rows, int B, int L, const void* lens_dev, const void* item_off_dev,
                      const float* offset, void* out_dev, void* hip_stream) {
  if (!m || !traj_dev || !lens_dev || !item_off_dev || !out_dev) return fail(FD_E_INVALID, "null argument");
  if (!m->finalized) return fail(FD_E_STATE, "fd_finalize has not been called");
  if (rows < 1 || B < 1 || L < 1) return fail(FD_E_INVALID, "rows=%d B=%d L=%d must be positive", rows, B, L);
  if ((long long)rows * B > 0x7fffffffLL) return fail(FD_E_UNSUPPORTED, "rows x B = %lld blocks", (long long)rows * B);
  HIP_TRY(hipSetDevice(m->device));
  ShiftTrimArgs a;
  memset(&a, 0, sizeof a);
  a.traj = static_cast<const float*>(traj_dev); a.lens = static_cast<const int*>(lens_dev);
  a.item_off = static_cast<const long long*>(item_off_dev); a.out = static_cast<float*>(out_dev);
  a.rows = rows; a.B = B; a.L = L; a.F = m->cfg.n_features;
  a.angle_mask = m->angle_mask;
  a.has_offset = offset ? 1 : 0;
  if (offset)
    for (int f = 0; f < a.F; ++f) a.offset[f] = offset[f];
  launch_shift_trim(a, hip_stream ? static_cast<hipStream_t>(hip_stream) : m->stream);
  HIP_TRY(hipGetLastError());
  return FD_OK;
}

int fd_test_wrap(int device_id, int which, const float* in, int64_t n, float* out) {
  if (!in || !out || n < 1) return fail(FD_E_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(device_id));
  float *din = nullptr, *dout = nullptr;
  auto cleanup = [&]() {
    if (din) (void)hipFree(din);
    if (dout) (void)hipFree(dout);
  };
#define W_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(FD_E_HIP, "%s: %s", #x, hipGetErrorString(e_)); } } while (0)
  W_TRY(hipMalloc((void**)&din, (size_t)n * 4));
  W_TRY(hipMalloc((void**)&dout, (size_t)n * 4));
  W_TRY(hipMemcpy(din, in, (size_t)n * 4, hipMemcpyHostToDevice));
  if (which == 0) launch_wrap_test_f32(din, dout, n, nullptr);
  else launch_wrap_test_img(din, dout, n, nullptr);
  W_TRY(hipGetLastError());
  W_TRY(hipDeviceSynchronize());
  W_TRY(hipMemcpy(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost));
#undef W_TRY
  cleanup();
  return FD_OK;
}

int fd_test_gemm(int device_id, int precision, int epilogue, const float* A, const float* W, const float* bias,
                 const float* resid, float* C, int M, int N, int K) {
  if (!A || !W || !bias || !C || M < 1 || N < 1 || K < 16 || K % 32) return fail(FD_E_INVALID, "bad argument");
  if (epilogue < EPI_BIAS || epilogue > EPI_BIAS_RESID || (epilogue == EPI_BIAS_RESID && !resid))
    return fail(FD_E_INVALID, "bad epilogue");
  HIP_TRY(hipSetDevice(device_id));
  if (precision == FD_PREC_F16X3) {
    if (epilogue == EPI_BIAS_RESID)
      return fail(FD_E_UNSUPPORTED, "the row-image path has no unfused residual epilogue (LayerNorm is always fused): use fd_test_gemm_ln");
    return img_gemm_hook(epilogue == EPI_BIAS_GELU ? EPI_IMG_GELU : EPI_IMG_BIAS, A, W, bias, nullptr, nullptr, nullptr, 0.f, C, M, N, K);
  }
  float *dA = nullptr, *dW = nullptr, *db = nullptr, *dr = nullptr, *dC = nullptr;
  void* dWp = nullptr;
  int rc = FD_OK;
  auto cleanup = [&]() {
    for (void* p : {(void*)dA, (void*)dW, (void*)db, (void*)dr, (void*)dC, dWp})
      if (p) (void)hipFree(p);
  };
#define T_TRY(expr)                                                          \
  do {                                                                       \
    hipError_t e_ = (expr);                                                  \
    if (e_ != hipSuccess) {                                                  \
      cleanup();                                                             \
      return fail(FD_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));  \
    }                                                                        \
  } while (0)
  T_TRY(hipMalloc((void**)&dA, (size_t)M * K * 4));
  T_TRY(hipMalloc((void**)&dW, (size_t)N * K * 4));
  T_TRY(hipMalloc((void**)&db, (size_t)N * 4));
  T_TRY(hipMalloc((void**)&dC, (size_t)M * N * 4));
  T_TRY(hipMemcpy(dA, A, (size_t)M * K * 4, hipMemcpyHostToDevice));
  T_TRY(hipMemcpy(dW, W, (size_t)N * K * 4, hipMemcpyHostToDevice));
  T_TRY(hipMemcpy(db, bias, (size_t)N * 4, hipMemcpyHostToDevice));
  if (resid) {
    T_TRY(hipMalloc((void**)&dr, (size_t)M * N * 4));
    T_TRY(hipMemcpy(dr, resid, (size_t)M * N * 4, hipMemcpyHostToDevice));
  }
  if (precision == FD_PREC_F32) {
    launch_gemm_f32(epilogue, dA, dW, db, dr, dC, M, N, K, nullptr);
  } else {
    cleanup();
    return fail(FD_E_INVALID, "precision %d", precision);
  }
  T_TRY(hipGetLastError());
  T_TRY(hipDeviceSynchronize());
  T_TRY(hipMemcpy(C, dC, (size_t)M * N * 4, hipMemcpyDeviceToHost));
#undef T_TRY
  cleanup();
  return rc;
}

int fd_test_gemm_ln(int device_id, int precision, int use_fused, const float* A, const float* W, const float* bias,
                    const float* resid, const float* gamma, const float* beta, float eps, float* C, int M, int N,
                    int K) {
  if (!A || !W || !bias || !resid || !gamma || !beta || !C || M < 1 || N < 1 || K < 32 || K % 32)
    return fail(FD_E_INVALID, "bad argument");
  if (precision != FD_PREC_F32 && precision != FD_PREC_F16X3) return fail(FD_E_INVALID, "precision %d", precision);
  HIP_TRY(hipSetDevice(device_id));
  if (precision == FD_PREC_F16X3) {
    if (!use_fused) return fail(FD_E_UNSUPPORTED, "the row-image path always fuses the LayerNorm into the GEMM");
    return img_gemm_hook(EPI_IMG_LN, A, W, bias, resid, gamma, beta, eps, C, M, N, K);
  }
  std::vector<void*> bufs;
  auto cleanup = [&]() {
    for (void* p : bufs) (void)hipFree(p);
  };
#define T_TRY(expr)                                                          \
  do {                                                                       \
    hipError_t e_ = (expr);                                                  \
    if (e_ != hipSuccess) {                                                  \
      cleanup();                                                             \
      return fail(FD_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));  \
    }                                                                        \
  } while (0)
  auto up = [&](const void* host, size_t bytes, void** dev) -> hipError_t {
    hipError_t e = hipMalloc(dev, bytes);
    if (e != hipSuccess) return e;
    bufs.push_back(*dev);
    return host ? hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice) : hipSuccess;
  };
  float *dA, *dW, *db, *dr, *dg, *dbt, *dT, *dC;
  T_TRY(up(A, (size_t)M * K * 4, (void**)&dA));
  T_TRY(up(W, (size_t)N * K * 4, (void**)&dW));
  T_TRY(up(bias, (size_t)N * 4, (void**)&db));
  T_TRY(up(resid, (size_t)M * N * 4, (void**)&dr));
  T_TRY(up(gamma, (size_t)N * 4, (void**)&dg));
  T_TRY(up(beta, (size_t)N * 4, (void**)&dbt));
  T_TRY(up(nullptr, (size_t)M * N * 4, (void**)&dT));
  T_TRY(up(nullptr, (size_t)M * N * 4, (void**)&dC));
  if (use_fused) {
    const bool ok = launch_gemm_f32_ln(dA, dW, db, dr, dg, dbt, eps, dC, M, N, K, nullptr);
    if (!ok) {
      cleanup();
      return fail(FD_E_UNSUPPORTED, "no LN-fused GEMM for N=%d K=%d in this precision", N, K);
    }
  } else {
    launch_gemm_f32(EPI_BIAS_RESID, dA, dW, db, dr, dT, M, N, K, nullptr);
    launch_layernorm(dT, dg, dbt, eps, dC, M, N, nullptr);
  }
  T_TRY(hipGetLastError());
  T_TRY(hipDeviceSynchronize());
  T_TRY(hipMemcpy(C, dC, (size_t)M * N * 4, hipMemcpyDeviceToHost));
#undef T_TRY
  cleanup();
  return FD_OK;
}

int fd_test_gemm_time(int device_id, int precision, int M, int N, int K, int reps, double* ms_per_launch) {
  if (!ms_per_launch || M < 1 || N < 1 || K < 32 || K % 32 || reps < 1) return fail(FD_E_INVALID, "bad argument");
  if (precision == FD_PREC_F16X3 && N % 32) return fail(FD_E_UNSUPPORTED, "row-image GEMM: N=%d must be a multiple of 32", N);
  HIP_TRY(hipSetDevice(device_id));
  std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N, 0.1f);
  unsigned st = 12345u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
  for (auto& v : hA) v = rnd();
  for (auto& v : hW) v = 0.02f * rnd();
  ImgHook hk;
  hipEvent_t e0 = nullptr, e1 = nullptr;
#define T_TRY(expr)                                                          \
  do {                                                                       \
    hipError_t e_ = (expr);                                                  \
    if (e_ != hipSuccess) {                                                  \
      if (e0) (void)hipEventDestroy(e0);                                     \
      if (e1) (void)hipEventDestroy(e1);                                     \
      return fail(FD_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));  \
    }                                                                        \
  } while (0)
  const long long rows = ((long long)M + 127) / 128 * 128;
  float *dA, *dW, *db, *dC;
  T_TRY(hk.up(hA.data(), hA.size() * 4, (void**)&dA));
  T_TRY(hk.up(hW.data(), hW.size() * 4, (void**)&dW));
  T_TRY(hk.up(hb.data(), (size_t)N * 4, (void**)&db));
  T_TRY(hk.up(nullptr, (size_t)rows * N * 4, (void**)&dC));
  GemmImgArgs g;
  memset(&g, 0, sizeof g);
  if (precision == FD_PREC_F16X3) {
    void *dAi, *dWi, *dtrash;
    int* ddims;
    std::vector<uint16_t> img;
    float wscale = 1.f;
    pack_weight_tiles(hW.data(), N, K, &img, &wscale);
    T_TRY(hk.up(nullptr, (size_t)rows * K * 4, &dAi));
    T_TRY(hk.up(img.data(), img.size() * 2, &dWi));
    T_TRY(hk.up(nullptr, 1024, &dtrash));
    const int hd[2] = {M, (int)rows};
    T_TRY(hk.up(hd, sizeof hd, (void**)&ddims));
    launch_f32_to_img(dA, dAi, rows, K, M, 8192.0f, nullptr);
    g.A = static_cast<const unsigned char*>(dAi);
    g.W = static_cast<const unsigned char*>(dWi);
    g.bias = db;
    g.out = reinterpret_cast<unsigned char*>(dC);
    g.trash = static_cast<unsigned char*>(dtrash);
    g.dims = ddims;
    g.N = N;
    g.K = K;
    g.acc_scale = 1.0f / (8192.0f * wscale);
    g.out_scale = 1024.0f;
  }
  T_TRY(hipEventCreate(&e0));
  T_TRY(hipEventCreate(&e1));
  auto run = [&]() {
    if (precision == FD_PREC_F16X3) launch_gemm_img(EPI_IMG_BIAS, g, (int)rows, nullptr);
    else launch_gemm_f32(EPI_BIAS, dA, dW, db, nullptr, dC, M, N, K, nullptr);
  };
  for (int i = 0; i < 3; ++i) run();
  T_TRY(hipDeviceSynchronize());
  T_TRY(hipEventRecord(e0, nullptr));
  for (int i = 0; i < reps; ++i) run();
  T_TRY(hipEventRecord(e1, nullptr));
  T_TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  T_TRY(hipEventElapsedTime(&ms, e0, e1));
#undef T_TRY
  *ms_per_launch = ms / reps;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return FD_OK;
}

int fd_profile_every(fd_model* m, int n) {
  if (!m || n < 0) return fail(FD_E_INVALID, "bad argument");
  m->profile_every = n;
  return FD_OK;
}

int fd_profile_reset(fd_model* m) {
  if (!m) return fail(FD_E_INVALID, "null model");
  if (int rc = harvest(m)) return rc;
  for (int i = 0; i < KC_COUNT; ++i) {
    m->prof_ms[i] = 0;
    m->prof_n[i] = 0;
  }
  return FD_OK;
}

int fd_profile_count(fd_model* m) { return m ? (int)KC_COUNT : 0; }

int fd_profile_get(fd_model* m, int i, const char** name, double* total_ms, int64_t* launches, double* flops_per_launch,
                   double* bytes_per_launch) {
  if (!m || i < 0 || i >= KC_COUNT) return fail(FD_E_INVALID, "bad argument");
  if (int rc = harvest(m)) return rc;
  if (name) *name = kClassName[i];
  if (total_ms) *total_ms = m->prof_ms[i];
  if (launches) *launches = m->prof_n[i];
  if (flops_per_launch) *flops_per_launch = m->prof_flops[i];
  if (bytes_per_launch) *bytes_per_launch = m->prof_bytes[i];
  return FD_OK;
}

int fd_synchronize(fd_model* m) {
  if (!m) return fail(FD_E_INVALID, "null model");
  HIP_TRY(hipSetDevice(m->device));
  HIP_TRY(hipStreamSynchronize(m->stream));
  return FD_OK;
}

int fd_debug_read(fd_model* m, const char* name, float* out, int64_t n_floats) {
  if (!m || !name || !out) return fail(FD_E_INVALID, "null argument");
  Workspace& w = m->ws;
  if (!w.img) return fail(FD_E_STATE, "fd_debug_read: the current workspace is not on the row-image path");
  HIP_TRY(hipSetDevice(m->device));
  HIP_TRY(hipStreamSynchronize(m->stream));
  const fd_config& c = m->cfg;
  const std::string nm = name;
  const int li = m->debug_layer < c.n_layers ? m->debug_layer : c.n_layers - 1;
  const LayerDev& lw = m->layers[li];
  const float s_next = li + 1 < c.n_layers ? m->layers[li + 1].s_h : m->s_hfinal;
  const long long BH = (long long)w.B * sub_heads(c);
  long long need = 0;
  float* tmp = nullptr;
  auto img = [&](const unsigned char* src, int K, float scale) -> int {
    need = (long long)w.cap * K;
    if (n_floats < need) return fail(FD_E_INVALID, "fd_debug_read(%s): need %lld floats", name, need);
    HIP_TRY(hipMalloc((void**)&tmp, need * 4));
    launch_img_to_f32(src, tmp, w.cap, K, scale, m->stream);
    return FD_OK;
  };
  auto qkv = [&](const unsigned char* src, int rowbytes, int is_vt, float scale) -> int {
    need = BH * w.LTOT * 32;
    if (n_floats < need) return fail(FD_E_INVALID, "fd_debug_read(%s): need %lld floats", name, need);
    HIP_TRY(hipMalloc((void**)&tmp, need * 4));
    launch_qkv_unpack(src, tmp, BH, w.LTOT, w.LPK, rowbytes, is_vt, scale, m->stream);
    return FD_OK;
  };
  int rc;
  if (nm == "h") rc = img(w.himg, c.d_model, lw.s_h);            // input of layer debug_layer
  else if (nm == "h_out") rc = img(w.himg, c.d_model, s_next);   // output of layer debug_layer
  else if (nm == "a") rc = img(w.aimg, c.d_model, lw.s_a);
  else if (nm == "ctx") rc = img(w.cimg, c.d_model, lw.s_v);
  else if (nm == "g") rc = img(w.gimg, c.d_ff, lw.s_g);
  else if (nm == "g_head") rc = img(w.gimg, c.d_model, m->s_hg);
  else if (nm == "q") rc = qkv(w.qbuf, 128, 0, lw.s_q);
  else if (nm == "k") rc = qkv(w.kbuf, 128, 2, lw.s_k);
  else if (nm == "v") rc = qkv(w.vbuf, 0, 1, lw.s_v);
  else if (nm == "stamps") {
    need = n_floats >= 2 * kStampWords ? kStampWords : 5 * 8 * 64 * 6 + 4 * 64 * 8;  // (older scripts ask for the first two tables only)
    if (n_floats < 2 * need) return fail(FD_E_INVALID, "fd_debug_read(stamps): need %lld floats (uint64 view)", 2 * need);
    if (!m->stamps) return fail(FD_E_STATE, "no stamps were recorded (FDMI_STAMPS=1)");
    HIP_TRY(hipMemcpy(out, m->stamps, need * 8, hipMemcpyDeviceToHost));
    return FD_OK;
  } else if (nm == "rowinfo") {
    need = 2LL * w.cap;
    if (n_floats < need) return fail(FD_E_INVALID, "fd_debug_read(rowinfo): need %lld", need);
    std::vector<int> t(need);
    HIP_TRY(hipMemcpy(t.data(), w.rowinfo, need * 4, hipMemcpyDeviceToHost));
    for (long long i = 0; i < need; ++i) out[i] = (float)t[i];
    return FD_OK;
  } else return fail(FD_E_INVALID, "fd_debug_read: unknown buffer '%s'", name);
  if (rc) return rc;
  hipError_t e = hipStreamSynchronize(m->stream);
  if (e == hipSuccess) e = hipMemcpy(out, tmp, need * 4, hipMemcpyDeviceToHost);
  (void)hipFree(tmp);
  if (e != hipSuccess) return fail(FD_E_HIP, "fd_debug_read: %s", hipGetErrorString(e));
  return FD_OK;
}

int fd_check_finite(fd_model* m) {
  if (!m) return fail(FD_E_INVALID, "null model");
  HIP_TRY(hipSetDevice(m->device));
  return check_flag(m);
}

}  // extern "C"
