// Internal launch interface between the C-ABI host code (api.hip) and the
// gfx950 kernels.  Not installed; include/fdmi.h is the public boundary.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fdmi {

constexpr int kHeadDim = 32;   // head size of the tuned kernels (attention_img.hip, attention_f32.hip); 64 / 96 / 128: attention_gen.hip
constexpr int kMaxFeat = 16;   // F <= 16 (reference feature sets have 3..9)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// ---- epilogues of the token GEMM  C[M,N] = A[M,K] * W[N,K]^T + bias[N] ----
enum GemmEpilogue {
  EPI_BIAS = 0,        // QKV projection
  EPI_BIAS_GELU = 1,   // BertIntermediate / AnglesPredictor.dense1: exact-erf GELU
  EPI_BIAS_RESID = 2   // BertSelfOutput / BertOutput dense: + residual (LayerNorm follows)
};

// fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32).  K % 32 == 0; M, N arbitrary.
void launch_gemm_f32(int epilogue, const float* A, const float* W, const float* bias, const float* resid, float* C,
                     int M, int N, int K, hipStream_t s);

// Fused GEMM + bias + residual + LayerNorm over full rows (N == 384 or 192):
//   C = LN(A * W^T + bias + resid) * gamma + beta.   Returns false if (N) has no instantiation.
bool launch_gemm_f32_ln(const float* A, const float* W, const float* bias, const float* resid, const float* gamma,
                        const float* beta, float eps, float* C, int M, int N, int K, hipStream_t s);

// y[r,:] = LN(x[r,:]) * gamma + beta, rows of length d (d <= 1024), one wave per row.
void launch_layernorm(const float* x, const float* gamma, const float* beta, float eps, float* y, int rows, int d,
                      hipStream_t s);

// K1: h = LN(x W_in^T + b_in (+ pos_emb[l])) * g + b + time_table[*t_dev]
void launch_embed(const float* x, const float* w_in, const float* b_in, const float* pos_emb /*null unless absolute*/,
                  const float* gamma, const float* beta, float eps, const float* time_table, const int* t_dev,
                  float* h, int B, int L, int F, int d, hipStream_t s);

// K4: multi-head self-attention with additive key mask and (optionally) the
// relative_key score term.  qkv: [B*L, 3d] (q | k | v), ctx: [B*L, d].
// dist_emb: [2*maxpos-1, 32] of this layer, or null for absolute positions.
// Returns false when L is beyond what this build tiles (L > 128).
// rkq != 0: relative_key_query (the key term k_r . E[l - r + maxpos - 1] as well)
bool launch_attention_f32(const float* qkv, const float* dist_emb, const int* lens, float* ctx, int B, int L, int H,
                          int maxpos, hipStream_t s, int rkq = 0);

// K8 tail + K9: per token  y = do_ln ? LN(g)*gamma+beta : g ;  eps = y W2^T + b2 ;
//   x' = wrap_if_angle( c1[t] * (x - beta[t]*eps / c3[t]) + (t>0 ? sigma[t]*z : 0) )
// coef: [4][T] (c1, beta, c3, sigma).  t is read from *t_dev.
// z comes from noise[t][...] when noise != null, else Philox(seed, t, element).
// Writes eps_out (if non-null), x_out (may alias x), hist[T_hist_row] (if non-null).
// Per-call values the captured per-step graph must not bake in: they live in a
// small device struct that the host rewrites before each sampling run.
struct UpdateDyn {
  const float* noise;    // [t_start+1, M, F] or null
  float* hist;           // [t_start+1, M, F] or null
  unsigned long long seed;
  long long seq_offset;
  int t_start;
  int hist_every;        // keep every k-th state (and the final one) in `hist`; <= 1: every state
};
struct UpdateArgs {
  const UpdateDyn* dyn;  // device pointer; when non-null it overrides noise/hist/seed/seq_offset/t_start
  const float* g;        // [M, d]
  const float* gamma;    // [d] or null
  const float* beta;     // [d] or null
  const float* w2;       // [F, d]
  const float* b2;       // [F]
  const float* x;        // [M, F]
  const float* coef;     // [4, T]
  const float* noise;    // [T, M, F] or null
  long long noise_stride;  // elements between consecutive t rows of `noise` (0: a single [M,F] slab)
  const int* t_dev;      // current step index
  float* eps_out;        // [M, F] or null
  float* x_out;          // [M, F] or null (null => forward only)
  float* hist;           // [(t_start+1), M, F] or null
  unsigned long long seed;
  long long seq_offset;  // global index of sequence 0 (Philox key)
  int t_start;
  int T;
  int M, L, F, d;
  int do_ln;
  float ln_eps;
  unsigned angle_mask;   // bit f set => wrap feature f
};
void launch_head_update(const UpdateArgs& a, hipStream_t s);

// N1: NeRF internal -> Cartesian backbone coordinates, one lane per chain, fp64.
// Feature column of each quantity in the [B][L][F] float32 array; -1 => the reference's constant.
struct NerfFeatures {
  int phi, psi, omega;                      // required
  int ang_n_ca_c, ang_ca_c_n, ang_c_n_ca;   // "tau"/"N:CA:C", "CA:C:1N", "C:1N:1CA"
  int len_c_n, len_n_ca, len_ca_c;          // "0C:1N", "N:CA", "CA:C"
};
void launch_nerf(const float* feats, const int* lens, int B, int L, int F, const NerfFeatures& fx, int center,
                 double* out /* [B][3L][3] */, hipStream_t s);

// *t_dev -= 1  (last node of the per-step graph)
void launch_step_advance(int* t_dev, hipStream_t s);

// out[i] = Philox normal for (seed, t, element i of [B][L][F] with seq_offset)
void launch_philox_fill(float* out, unsigned long long seed, int t, long long seq_offset, int B, int L, int F,
                        hipStream_t s);

// ================================================================== row-image path (FD_PREC_F16X3)
// Activations live in HBM as fp16 hi|lo row images (img_common.h); see gemm_img.hip / attention_img.hip /
// rowwise_img.hip.  Token rows: sequences start at multiples of 8 rows; `rowinfo[row]` = (sequence, position)
// or (-1, -1) for padding rows; `dims` (device) = {rows, rows rounded up to 128}.  Every kernel is persistent /
// grid-stride and reads the row counts from `dims`, so a captured graph serves any lengths of one (B, L).
enum GemmImgEpilogue {
  EPI_IMG_GELU = 0, EPI_IMG_LN = 1, EPI_IMG_QK = 2, EPI_IMG_VT = 3, EPI_IMG_BIAS = 4 /* plain bias: test hook */,
  EPI_IMG_QKV = 5  // q | k | v in ONE launch (N = 3 d_model; needs n_heads % 6 == 0 so that v starts on a 384-column tile):
                   // column tiles below 2 d_model take the EPI_QK path, the others the EPI_VT path (stamps share EPI_QK's slot)
};

struct GemmImgArgs {
  const unsigned char* A;      // activation image [rows128][K/32][128 B]
  const unsigned char* W;      // weight image [N rounded up to 384][K/32][128 B] (zero padded)
  const float* bias;           // [N]
  const float* gamma;          // [N]   EPI_IMG_LN
  const float* beta;           // [N]   EPI_IMG_LN
  const unsigned char* resid;  // image [rows128][N/32]   EPI_IMG_LN
  unsigned char* out;          // image [rows128][N/32]   EPI_IMG_GELU / EPI_IMG_LN
  float* out_f32;              // EPI_IMG_BIAS only: when non-null the result (+ residual if `resid`) leaves as fp32 [rows128][N]
  unsigned char* qbuf;         // [B][H][LTOT][128 B]     EPI_IMG_QK
  unsigned char* kbuf;         // [B][H][LTOT][128 B]     EPI_IMG_QK (unit u of row l at u ^ ((l >> 1) & 7))
  unsigned char* vbuf;         // [B][H][LTOT / 32][32 d][128 B]  EPI_IMG_VT (V transposed, swizzled: gemm_img.hip)
  unsigned char* trash;        // >= 256 B scratch line for the stores of padding rows
  unsigned qkv_bytes;          // size of each of qbuf / kbuf / vbuf (< 4 GiB: the q | k | v epilogues address them with 32-bit offsets)
  const int2* rowinfo;
  const int* dims;
  int N, K;                    // valid output columns (multiple of 32), reduction length (multiple of 32)
  int H, LPK, LTOT, NKT;       // heads; keys per key tile; NKT * LPK; key tiles
  float acc_scale;             // 1 / (scale of A * scale of W)
  float out_scale;             // scale of the output image (GELU / LN)
  float resid_inv;             // 1 / scale of the residual image
  float eps;                   // LayerNorm eps
  float q_scale, k_scale, v_scale;
  unsigned long long* stamps;  // null, or [5 epilogues][8 waves][64 slots][6] cycle stamps of workgroup 0 (debug)
  int tail;                    // 1: the rows may not fill whole rounds of tiles -> the slice-capable instantiation (gemm_img.hip); 0: they do
};
// max_rows bounds the grid (B * ceil8(L) of the workspace); the kernel reads the actual count from p.dims.
void launch_gemm_img(int epilogue, const GemmImgArgs& p, int max_rows, hipStream_t s);
int gemm_img_grid(int max_rows, int N);
// weight-stationary variant (gemm_ws.hip): K = 384, GELU / bias epilogue
bool gemm_ws_supported(int epilogue, const GemmImgArgs& p);
void launch_gemm_ws(int epilogue, const GemmImgArgs& p, hipStream_t s);
// few-rows LayerNorm GEMM (gemm_ln_rows.hip): N = 384, K = 384 / 768, one workgroup per 32-row group
bool gemm_ln_rows_supported(const GemmImgArgs& p);
void launch_gemm_ln_rows(const GemmImgArgs& p, int max_rows, hipStream_t s);

struct AttnImgArgs {
  const unsigned char* qbuf;
  const unsigned char* kbuf;
  const unsigned char* vbuf;
  const u32x4_t* demb;         // distance table image [2 maxpos - 1][128 B], or null (absolute positions)
  const int* lens;             // [B] unmasked keys per sequence
  const int* nrow;             // [B] positions that are rows at all (L, or lens[b] when packed)
  const int* seq_row0;         // [B + 1] first token row of each sequence
  unsigned char* ctx;          // image [rows128][H][128 B]
  unsigned char* trash;
  int B, H, LTOT, NKT, maxpos;
  float q_scale, k_scale, v_scale, ctx_scale;
  float r_scale;               // k_scale / scale of the distance table
  float r_scale_k;             // q_scale / scale of the distance table (relative_key_query: the key term)
  int rkq;                     // position_embedding_type == relative_key_query
  const unsigned char* kmask;  // attention_gen.hip only: [B][L] 1 = attend, 0 = masked key (any pattern), or null: keys >= lens[b] are masked
  int L;                       // row stride of kmask
  unsigned long long* stamps;  // null, or [4 waves][64 slots][8] cycle stamps of workgroup 0 (debug)
};
bool launch_attention_img(const AttnImgArgs& p, int L, hipStream_t s);
// head sizes 32 * nb, nb = 1 .. 4 (attention_gen.hip; nb = 1 only for arbitrary key masks, which the tuned kernel does not take): p.H = heads; qbuf / kbuf / vbuf / ctx / demb are indexed by 32-column sub-head
bool launch_attention_gen(const AttnImgArgs& p, int nb, hipStream_t s);

// seq_attn.hip: BertSelfAttention's q | k | v projection AND the attention of a whole sequence in ONE kernel (L <= 128, head size 32,
// relative_key, maxpos <= 128): q, k and v never reach HBM.  One 4-wave workgroup per sequence; the hidden state stays in registers.
struct SeqAttnArgs {
  const unsigned char* himg;   // hidden-state image [rows128][d/32] (grouped, img_common.h)
  unsigned himg_bytes;         // its size (< 4 GiB: 32-bit lane offsets; rows beyond it read as zeros)
  const unsigned char* wimg;   // weight image [head][k-tile][unit 0-7][96 rows: q_h | k_h | v_h][16 B] at the scale of wqkv_i (api.hip: pack_seq_attn_weights)
  const float* bias;           // [3 d]: q | k | v
  const u32x4_t* demb;         // distance table image [2 maxpos - 1][128 B]
  const int* lens;             // [B] unmasked keys per sequence
  const int* nrow;             // [B] positions that are rows at all
  const int* seq_row0;         // [B + 1] first token row of each sequence
  unsigned char* ctx;          // image [rows128][H][128 B]
  int B, H, maxpos;
  float acc_scale;             // 1 / (scale of himg * scale of wimg)
  float q_scale, k_scale, v_scale, ctx_scale;
  float r_scale;               // k_scale / scale of the distance table
  unsigned long long* stamps;  // null, or [4 waves][64 slots][16] cycle stamps of workgroup 0 (debug)
};
bool seq_attn_supported(int d_model, int n_heads, int L, int maxpos);
bool launch_seq_attn(const SeqAttnArgs& p, hipStream_t s);   // false: the LDS opt-in or the launch was refused
// seq_attn16.hip (round 6): the same operation with 16-row waves, two per SIMD (v_mfma_f32_16x16x32_f16); wimg = [head][k32 step][tile]
// [unit][row][16 B] (api.hip: upload_seq_attn16_weights); any L <= 128, padded or packed rows.  false: the launch failed.
bool seq_attn16_supported(int d_model, int n_heads, int L, int maxpos);
bool launch_seq_attn16(const SeqAttnArgs& p, hipStream_t s);

// ffn16.hip (round 6): BertIntermediate + GELU + BertOutput (dense + residual + LayerNorm) of 128 token rows per pass in ONE kernel;
// the intermediate never reaches HBM.  wimg = ONE stream of 24 KiB stages in consumption order (api.hip: upload_ffn16_weights).
struct FfnArgs {
  const unsigned char* aimg;   // input image [rows128][d/32] (BertSelfOutput's LayerNorm output): operand AND residual
  unsigned a_bytes;            // its size (rows beyond it read as zeros)
  const unsigned char* wimg;   // weight stream
  const float* bi;             // [2 d]  intermediate.dense.bias
  const float* bd;             // [d]    output.dense.bias
  const float* gamma;          // [d]    output.LayerNorm
  const float* beta;
  unsigned char* out;          // output image [rows128][d/32] (must not alias aimg)
  unsigned out_bytes;          // its size (stores beyond it are dropped)
  int panels;                  // passes of 128 rows (upper bound)
  const int* dims;             // null, or device {rows, rows rounded up to 128}: passes beyond them are skipped
  float up_scale;              // 1 / (scale of aimg * scale of the first dense's weights)
  float g_scale;               // scale of the intermediate's hi / lo images (a power of two)
  float down_scale;            // 1 / (g_scale * scale of the second dense's weights)
  float resid_inv;             // 1 / scale of aimg
  float out_scale;             // scale of the output image
  float eps;
  // the layer's tail in one launch: cimg != null -> BertSelfOutput (attention.output.dense + residual + LayerNorm) runs in front, on the
  // attention context; aimg is then unused (its rows never reach HBM) and wimg starts with attention.output.dense's stages
  const unsigned char* cimg;   // attention context image [rows128][d/32], a_bytes long
  const unsigned char* hres;   // the layer's input rows (residual of BertSelfOutput), a_bytes long; may alias out (a row is read, then written, by one wave)
  const float* bo;             // [d]  attention.output.dense.bias
  const float* g1;             // [d]  attention.output.LayerNorm
  const float* b1;
  float ao_scale;              // 1 / (scale of cimg * scale of attention.output.dense's weights)
  float hres_inv;              // 1 / scale of hres
  float a_scale;               // scale of BertSelfOutput's output images (= 1 / resid_inv)
  float eps1;
  unsigned long long* stamps;  // null, or [8 waves][16 passes][16] cycle stamps of workgroup 0 (debug)
};
bool ffn16_supported(int d_model, int d_ff);
bool launch_ffn16(const FfnArgs& p, int d_model, hipStream_t s);   // false: the launch failed

struct EmbedImgArgs {
  const float* x;              // [B][L][F]
  const float* w_in; const float* b_in; const float* pos_emb; const float* gamma; const float* beta;
  const float* time_table;
  int* tslot;                  // [0]: step index of this step (host / head_update_img write it), [1]: copy for the step's other kernels
  const int2* rowinfo; const int* nrow; const int* dims;
  const int* pos_ids;          // [B][L] position ids of the absolute position embedding, or null = 0 .. L-1 (modelling.py:434-442)
  unsigned char* h;            // image [rows128][d/32]
  int L, F, d;
  float eps, out_scale;
};
void launch_embed_img(const EmbedImgArgs& a, int max_rows, hipStream_t s);

struct HeadImgArgs {
  const unsigned char* g;      // image [rows128][d/32]: head activation (mlp decoder) or final hidden state (linear)
  float g_inv;                 // 1 / its scale
  const int2* rowinfo; const int* nrow; const int* dims;
  int* tslot;
  int* flag;                   // bit 0 set when a non-finite prediction was seen
  int advance;                 // write tslot[0] = t - 1 at the end (the sampling loop)
};
// UpdateArgs: M = B * L (elements of one state / F), x / eps / noise / hist in the [B][L][F] layout.
void launch_head_update_img(const UpdateArgs& a, const HeadImgArgs& ia, int max_rows, hipStream_t s);

// y = LayerNorm(fp32 rows [rows128][d]) -> image (d_model > 384: the un-fused BertSelfOutput / BertOutput LayerNorm)
void launch_ln_f32_img(const float* src, const float* gamma, const float* beta, float eps, const int* dims, void* out, int d,
                       float out_scale, int max_rows, hipStream_t s);

// N2: the post-processing of sampling.sample (foldingdiff/sampling.py:200-222) on the device: per item the first lens[i]
// positions of every stored state, shifted by the training mean offset and re-wrapped where the feature is an angle, packed into
// one ragged buffer: item i's [rows][lens[i]][F] block starts at element item_off[i].
struct ShiftTrimArgs {
  const float* traj;           // [rows][B][L][F]
  const int* lens;             // [B]
  const long long* item_off;   // [B]
  float* out;                  // ragged
  int rows, B, L, F;
  unsigned angle_mask;         // bit f: feature f is wrapped to [-pi, pi) after the shift
  int has_offset;              // 0: neither shift nor wrap (the reference only wraps inside `if offset is not None`)
  float offset[kMaxFeat];
};
void launch_shift_trim(const ShiftTrimArgs& a, hipStream_t s);
// test hook: out[i] = wrap_pi(in[i]) through the update kernels' own device function (which: 0 rowwise.hip, 1 rowwise_img.hip)
void launch_wrap_test_f32(const float* in, float* out, long long n, hipStream_t s);
void launch_wrap_test_img(const float* in, float* out, long long n, hipStream_t s);

void launch_build_rows(const int* lens, int B, int L, int packed, int cap, int* seq_row0, int* nrow, int2* rowinfo,
                       int* dims, hipStream_t s);
// fp32 [src_rows][K] -> image [rows][K/32] (rows >= src_rows are zero rows), value * scale = hi + lo; and back
void launch_f32_to_img(const float* src, void* dst, long long rows, int K, long long src_rows, float scale, hipStream_t s);
void launch_img_to_f32(const void* src, float* dst, long long rows, int K, float scale, hipStream_t s);
// debug: q / k rows or v^T blocks -> fp32 [BH][LTOT][32]
void launch_qkv_unpack(const void* src, float* dst, long long BH, int LTOT, int LP, int rowbytes, int is_vt, float scale,
                       hipStream_t s);

}  // namespace fdmi
