/*
 * fdmi.h -- C ABI of libfdmi.so, the MI355X (gfx950) reverse-diffusion sampler
 * for foldingdiff's BertForDiffusion backbone-angle model.
 *
 * The reference (microsoft/foldingdiff) is 100% Python and has NO FFI / plugin
 * interface of its own; its boundary for this path is the Python API
 *     modelling.BertForDiffusionBase.from_dir / .forward   (foldingdiff/modelling.py:297-484)
 *     sampling.p_sample / p_sample_loop / sample           (foldingdiff/sampling.py:27-224)
 * Each entry point below names the reference function it replaces.  The Python
 * package `foldingdiff_amd` binds these with ctypes and re-exposes the
 * reference's names and signatures.
 *
 * Conventions
 *  - plain C types only; no torch / HIP types in signatures (streams and device
 *    buffers cross as void*).
 *  - every function returns FD_OK (0) or a negative FD_E_* code; the message is
 *    available from fd_last_error() (thread local).  Nothing throws across the ABI.
 *  - "host" pointers are ordinary host memory, copied synchronously.  "dev"
 *    pointers are device memory on the model's device (hipMalloc / torch CUDA
 *    tensors), used in place on the given stream.
 *  - one fd_model per device; calls on one model are serialised by the caller.
 *  - all tensors are contiguous row-major float32 unless stated otherwise:
 *    angles x[B][L][F], lengths int32[B], history [T][B][L][F].
 */
#ifndef FDMI_H
#define FDMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: fd_sample_steps_dev / fd_sample_end_dev return FD_E_STATE when another call has taken the run's workspace; history
 *    padding of packed rows is zeroed; head sizes other than 32 (multiples of 32) are accepted */
/* 3: fd_shift_trim_dev, fd_test_wrap, option "fuse_attn" (round 5) */
/* 4: fd_fused_attn_supported; option "fuse_attn" takes 2 (the round-5 kernel) and its default kernel no longer promises the bits of
 *    the two-kernel path; option "fuse_ffn" (round 6) */
#define FDMI_ABI_VERSION 4

enum {
  FD_OK = 0,
  FD_E_INVALID = -1,     /* bad argument / unsupported shape            */
  FD_E_STATE = -2,       /* call order (e.g. sample before finalize)    */
  FD_E_MISSING = -3,     /* a required weight was never set             */
  FD_E_HIP = -4,         /* HIP runtime error (message has hipGetErrorString) */
  FD_E_UNSUPPORTED = -5, /* valid in the reference, not implemented here */
  FD_E_NONFINITE = -6    /* the model produced inf / NaN (e.g. corrupt weights); results are not usable */
};

/* config.position_embedding_type (config.json; modelling.py:138-149, HF BertSelfAttention) */
enum { FD_POS_ABSOLUTE = 0, FD_POS_RELATIVE_KEY = 1, FD_POS_RELATIVE_KEY_QUERY = 2 };
/* training_args.json "decoder" (modelling.py:274-279) */
enum { FD_DEC_MLP = 0, FD_DEC_LINEAR = 1 };
/* arithmetic of the contraction kernels */
enum {
  FD_PREC_F32 = 0,  /* v_mfma_f32_32x32x2_f32: exact fp32 products + fp32 accumulate */
  FD_PREC_F16X3 = 1 /* GEMM operands split into fp16 hi + lo (22 significant bits), three
                       v_mfma_f32_32x32x16_f16 per product, fp32 accumulate: fp32-class error at
                       5.3x the fp32-MFMA rate (GEMMs and the attention contractions).  Softmax,
                       LayerNorm, GELU, residuals and all accumulation stay fp32. */
};

typedef struct fd_model fd_model;

/* Shapes of BertConfig + training_args.json that the hot path reads
 * (bin/train.py:425-435; modelling.py:239-295). */
typedef struct fd_config {
  int32_t n_features; /* F  = len(ft_is_angular)  (modelling.py:255-256) */
  int32_t d_model;    /* config.hidden_size */
  int32_t n_heads;    /* config.num_attention_heads; head size d_model / n_heads: 32 (tuned kernels), 64, 96 or 128 */
  int32_t d_ff;       /* config.intermediate_size */
  int32_t n_layers;   /* config.num_hidden_layers */
  int32_t max_pos;    /* config.max_position_embeddings */
  int32_t pos_type;   /* FD_POS_* */
  int32_t decoder;    /* FD_DEC_* */
  float ln_eps;       /* config.layer_norm_eps (embeddings + encoder LayerNorms);
                         the decoder head LayerNorm always uses 1e-12 (modelling.py:187) */
} fd_config;

/* ---- construction: replaces BertForDiffusionBase.__init__/from_dir (modelling.py:239-382) ---- */

int fd_abi_version(void);

/* Number of visible HIP devices (0 if none / no driver). */
int fd_device_count(void);

/* Create an empty model on HIP device `device_id`. */
int fd_create(const fd_config* cfg, int device_id, fd_model** out);

/* Provide one tensor of the reference state_dict, by its HuggingFace /
 * modelling.py name (e.g. "encoder.layer.3.attention.self.query.weight",
 * "token_decoder.dense2.bias"; full list in DESIGN.md).  Replaces
 * load_state_dict (modelling.py:362-363).  Shape is checked against cfg.
 * Non-parameter buffers ("time_embed.W", "embeddings.position_ids") are accepted
 * and ignored: the time embedding reaches the device as a table (fd_finalize). */
int fd_set_weight(fd_model* m, const char* name, const float* host_data, const int64_t* shape, int ndim);

/* Pack weights, upload the schedule + time-embedding tables, build kernels' state.
 *   T           number of diffusion timesteps (NoisedAnglesDataset.timesteps)
 *   coef        float[4][T] rows: sqrt(1/alpha_t), beta_t, sqrt(1-alphabar_t),
 *               sqrt(posterior_variance_t)  -- from compute_alphas
 *               (beta_schedules.py:45-62) as used by p_sample (sampling.py:41-72)
 *   time_table  float[T][d_model]: time_embed(t) for t = 0..T-1, computed on the
 *               host with the reference's fp32 op order (modelling.py:59-71)
 *   is_angle    uint8[F]: which features are wrapped to [-pi, pi) each step
 *               (sampling.py:119-130)
 *   precision   FD_PREC_*
 * May be called again to change T / tables. */
int fd_finalize(fd_model* m, int T, const float* coef, const float* time_table, const uint8_t* is_angle,
                int precision);

void fd_destroy(fd_model* m);

/* Runtime switches:
 *   "fuse_ln"    FD_PREC_F32 only (FD_PREC_F16X3 always fuses): 1: residual + LayerNorm run in the epilogue of
 *                the attention-output and FFN-down GEMMs (shapes without a fused instantiation fall back);
 *                0 / -1 (default): separate LayerNorm kernel (same arithmetic, faster in that mode on MI355X).
 *   "use_graph"  1 (default): the per-step kernel sequence is replayed from a hipGraph;
 *                0: eager launches.
 *   "varlen"     fd_sample / fd_sample_dev with FD_PREC_F16X3: 1 = positions >= lens[b] are not computed at all
 *                (the reference computes them and sampling.sample cuts them away, sampling.py:56-58, :201-203);
 *                positions < lens[b] are bit-identical either way.  Padded positions of the final `out` then keep x_init;
 *                padded positions of history rows are zero.
 *                0 (default): every position evolves as in the reference's p_sample_loop.
 *   "fuse_attn"  FD_PREC_F16X3, relative_key, head size 32, d_model 384 / 192, L <= 128: BertSelfAttention of a sequence (q | k | v
 *                projection + attention, modelling.py:473-480 -> HF BertSelfAttention.forward) as ONE kernel, q / k / v never
 *                reach HBM.  -1 (default): when the batch fills whole rounds of the device's CUs; 1: whenever the shape allows
 *                (fd_fused_attn_supported); 0: never (the q|k|v GEMM + attention kernels); 2: the round-5 kernel (32-row waves,
 *                96 < L <= 128; kept for A/B measurements).  Every choice meets the same tolerance against the reference
 *                (1e-5 on the forward, 1e-3 rad on a step); 0 and 2 give the same bits as each other, 1 does not (it sums in
 *                another order).
 *   "fuse_ffn"   FD_PREC_F16X3, d_model 384 / 192 with intermediate size 2 d_model: the tail of a BertLayer (modelling.py:473-480 -> HF
 *                BertLayer.forward behind the attention) as ONE kernel over passes of 128 token rows.  2: BertSelfOutput (dense +
 *                residual + LayerNorm), BertIntermediate (dense + GELU) and BertOutput (dense + residual + LayerNorm); neither the
 *                attention block's output nor the intermediate reaches HBM.  1: the feed-forward pair only.  0: never (three GEMM
 *                launches).  -1 (default): 2 when the passes fill whole rounds of the device's CUs.  Every choice meets the same
 *                tolerance against the reference; 1 and 2 sum in another order than 0 (not the same bits).
 *   "rows_hint"  with "varlen" 1: the exact number of token rows of the next sampling calls (sum over the batch of the lengths rounded up
 *                to 8; the library itself only has the lengths in device memory and the bound B * ceil8(L)), 0 (default) = unknown.
 *                It only steers the automatic kernel choices ("fuse_ffn" -1); results are within the same tolerances either way.
 *   "split_qkv"  FD_PREC_F16X3: 1 = project q | k and v^T in two launches even when n_heads % 6 == 0 would allow one
 *                (A/B measurements, tests); 0 (default).
 *   "debug_stop" n > 0: a step returns after its first n launches (FD_PREC_F16X3; stage-by-stage comparison with
 *                fd_debug_read, scripts/debug_img.py); "debug_layer": which encoder layer fd_debug_read sees. */
int fd_set_option(fd_model* m, const char* name, int value);

/* 1 if option "fuse_attn" = 1 (or 2, when that is the current value) runs the fused projection + attention kernel for batches of
 * padded length L on this (finalized) model, 0 if the q|k|v GEMM + attention kernels run instead; negative: FD_E_*.  No reference
 * counterpart (HF BertSelfAttention.forward is one code path); tests use it to know which kernels a gate exercised. */
int fd_fused_attn_supported(fd_model* m, int L);

/* ---- parity hooks ---- */

/* eps = model(x, t, attention_mask(lens)) -- BertForDiffusionBase.forward in eval
 * mode (modelling.py:384-484).  Host buffers.  t is constant over the batch, as
 * p_sample asserts (sampling.py:46-47).
 * Size limit of one call (every entry point that takes B and L): in the default precision the q / k / v images of the batch
 * are addressed with 32-bit offsets, so B * n_heads * ceil(L / 32) * 4096 bytes must stay below 4 GiB (B < 21,845 sequences of
 * L = 128 at 12 heads) -- FD_E_UNSUPPORTED beyond; sampling.sample chunks by batch_size long before that. */
int fd_forward(fd_model* m, const float* x, int t, const int32_t* lens, int B, int L, float* eps_out);

/* The same forward with what the reference's forward also honours (modelling.py:434-452, :464-467) and the sampler never produces:
 *   key_mask      uint8[B][L], 1 = attend, 0 = masked key, ANY pattern (NULL: every key is attended to); masked keys get the
 *                 additive -10000 of HF's extended attention mask, and every position is computed as a query
 *   position_ids  int32[B][L] rows of the absolute position embedding (NULL: 0 .. L-1); ignored for the relative position
 *                 types, as in the reference (no position embedding is added there and the distance uses arange)
 * FD_PREC_F16X3 only.  Arbitrary masks run on the general attention kernel (attention_gen.hip), not the tuned one. */
int fd_forward_ex(fd_model* m, const float* x, int t, const uint8_t* key_mask, const int32_t* position_ids, int B, int L,
                  float* eps_out);

/* One reverse step: x_out = p_sample(x, t) (sampling.py:27-75); with wrap != 0 the
 * loop's per-feature wrap to [-pi, pi) (sampling.py:119-130) is applied as well, i.e.
 * the result is one iteration of p_sample_loop.  z is the N(0,1) draw the reference
 * takes from torch.randn_like (required for t > 0, ignored at t == 0).  Host buffers. */
int fd_p_sample_step(fd_model* m, const float* x, int t, const int32_t* lens, int B, int L, const float* z,
                     int wrap, float* x_out);

/* ---- the sampler: replaces p_sample_loop (sampling.py:78-132) ----
 *   x_init   [B][L][F] start point (already wrapped noise, datasets.py:772-799)
 *   lens     int32[B], 1 <= lens[i] <= L; positions >= lens[i] are masked keys
 *   t_start  first timestep index to run (T-1 for a full run); the loop runs
 *            t = t_start, t_start-1, ..., 0  (t_start+1 steps)
 *   noise    [t_start+1][B][L][F] per-step N(0,1) draws, row i used at t = i
 *            (row 0 unused), or NULL => on-device Philox4x32-10 keyed by
 *            (seed, t, element) -- deterministic, but NOT torch's stream
 *   out      full_history == 0: [B][L][F], the final sample only;
 *            full_history == 1: [t_start+1][B][L][F] (row j = state after step t = t_start - j,
 *            i.e. the reference's stacked `imgs`);
 *            full_history == k > 1: [ceil((t_start+1)/k)][B][L][F] -- every k-th state (j = k-1,
 *            2k-1, ...) and the final one in the last row (strided history)
 * The per-step loop is a captured hipGraph replayed t_start+1 times; the step
 * index lives on the device. */
int fd_sample(fd_model* m, const float* x_init, const int32_t* lens, int B, int L, int t_start, const float* noise,
              uint64_t seed, float* out, int full_history);

/* fd_sample for a slice of a larger batch: `seq_offset` is the global index of sequence 0, added to the sequence
 * index in the Philox key (multi-GPU sharding of sampling.sample: every rank draws the noise of the unsharded batch). */
int fd_sample_ex(fd_model* m, const float* x_init, const int32_t* lens, int B, int L, int t_start, const float* noise,
                 uint64_t seed, int64_t seq_offset, float* out, int full_history);

/* Same, with every buffer already resident in device memory, asynchronous on
 * `hip_stream` (a hipStream_t, NULL = the model's own stream).  The caller
 * synchronises.  `seq_offset` is added to the sequence index in the Philox key
 * so a batch sharded over several GPUs draws the noise of the unsharded batch. */
int fd_sample_dev(fd_model* m, const void* x_init_dev, const void* lens_dev, int B, int L, int t_start,
                  const void* noise_dev, uint64_t seed, int64_t seq_offset, void* out_dev, int full_history,
                  void* hip_stream);

/* The same loop in pieces, for callers that stream the per-step noise instead of materialising [T][B][L][F]
 * (sampling.py draws the reference's torch.randn_like sequence chunk by chunk on a host thread and uploads it on a copy
 * stream while the previous chunk is consumed):
 *   fd_sample_begin_dev   everything fd_sample_dev does before its first step (uploads x_init / lens into the workspace,
 *                         row table, graph); `out_dev` receives the history rows as in fd_sample_dev
 *   fd_sample_steps_dev   the next n_steps reverse steps, enqueued on the stream; noise_dev = [rows][B][L][F] device
 *                         buffer whose row i is the draw of step t = noise_t0 + i (it must cover every step of the call,
 *                         t = 0 included -- that row is read and multiplied by sigma_0 = 0), or NULL (Philox)
 *   fd_sample_end_dev     after the last step: copies the final state to out_dev when full_history == 0
 * fd_sample_dev(...) == begin; steps(t_start + 1, noise, 0); end.  Stream-ordered like fd_sample_dev. */
int fd_sample_begin_dev(fd_model* m, const void* x_init_dev, const void* lens_dev, int B, int L, int t_start,
                        uint64_t seed, int64_t seq_offset, void* out_dev, int full_history, void* hip_stream);
int fd_sample_steps_dev(fd_model* m, int n_steps, const void* noise_dev, int noise_t0, void* hip_stream);
int fd_sample_end_dev(fd_model* m, void* out_dev, void* hip_stream);

/* ---- multi-GPU through the ABI: independent sequences are sharded across the GPUs of a node by the HOST (one
 * model per GPU, each sampling its slice with seq_offset = its first global sequence index; Philox noise is keyed by that
 * index, so results do not depend on the world size) and ONE collective returns the slices: an RCCL all-gather over xGMI.
 * The reference has no multi-GPU sampler (foldingdiff/sampling.py:91 is single-device); the Python package does the same
 * exchange through torch.distributed (foldingdiff_amd/distributed.py).  RCCL is bound at run time (dlopen librccl.so).
 *   fd_comm_unique_id  128 bytes, created by one process and distributed by the host (file, pipe, MPI ...)
 *   fd_comm_init       joins the communicator (collective over all ranks); one communicator per model
 *   fd_gather_dev      out_dev[r * n_floats ...] = rank r's local_dev[0 .. n_floats) for every r; equal counts on every rank
 *                      (pad ragged slices to the largest); stream-ordered on hip_stream (NULL: the model's stream) */
#define FD_COMM_ID_BYTES 128
int fd_comm_unique_id(void* id_out);
int fd_comm_init(fd_model* m, int rank, int world, const void* unique_id);
int fd_gather_dev(fd_model* m, const void* local_dev, int64_t n_floats, void* out_dev, void* hip_stream);
int fd_comm_destroy(fd_model* m);

/* Fill out_dev[n] (device, float32) with the Philox N(0,1) stream used for step
 * t of a [B][L][F] batch -- exposes the perf-mode generator for tests. */
int fd_philox_normal_dev(fd_model* m, uint64_t seed, int t, int64_t seq_offset, int B, int L, void* out_dev,
                         void* hip_stream);

/* ---- post-processing: angles -> backbone coordinates ----
 * Replaces NERFBuilder.cartesian_coords / .centered_cartesian_coords (foldingdiff/nerf.py:78-129,
 * place_dihedral :145-204) as create_new_chain_nerf calls them (foldingdiff/angles_and_coords.py:112-184).
 *   feats     float32 [B][L][F] sampled features (host), lens int32[B] residues per chain
 *   feat_idx  int32[9]: column of phi, psi, omega, N:CA:C ("tau"), CA:C:1N, C:1N:1CA, 0C:1N, N:CA, CA:C;
 *             -1 for an angle / length that is not a feature => the reference's constant
 *             (109, 115, 121 degrees; 1.34, 1.46, 1.54 Angstrom); the three dihedrals are required
 *   center    subtract the mean atom position of each chain (centered_cartesian_coords)
 *   coords_out float64 [B][3*L][3]: N, CA, C of each residue; rows of residues >= lens[b] are 0 */
int fd_nerf(int device_id, const float* feats, const int32_t* lens, int B, int L, int F, const int32_t* feat_idx,
            int center, double* coords_out);

/* ---- test hook ----
 * One token GEMM  C[M,N] = A[M,K] W[N,K]^T + bias (+GELU | +resid) through the production
 * kernels of the given precision (epilogue: 0 bias, 1 bias+GELU, 2 bias+residual).  Host buffers,
 * K % 32 == 0.  Used by tests/ to measure kernel error against fp64 in isolation. */
int fd_test_gemm(int device_id, int precision, int epilogue, const float* A, const float* W, const float* bias,
                 const float* resid, float* C, int M, int N, int K);

/* N2: the post-processing of sampling.sample -- foldingdiff/sampling.py:200-222: cut every item to its length, add the
 * training mean offset (datasets.py get_masked_means), re-wrap the angular features with utils.modulo_with_wrapped_range(., -pi, pi)
 * -- on the device, bit-identical to the reference's float32 numpy arithmetic.  traj_dev: [rows][B][L][F] float32 (the stored states
 * of fd_sample_dev, F = fd_config.n_features); lens_dev: int32 [B]; item_off_dev: int64 [B], element offset of item i's
 * [rows][lens[i]][F] block inside out_dev; offset: HOST float32 [F], or NULL = neither shift nor wrap (what the reference does
 * without an offset).  Angularity is what fd_finalize was given.  Asynchronous on hip_stream (NULL: the model's own stream). */
int fd_shift_trim_dev(fd_model* m, const void* traj_dev, int rows, int B, int L, const void* lens_dev, const void* item_off_dev,
                      const float* offset, void* out_dev, void* hip_stream);

/* Test hook for utils.modulo_with_wrapped_range (foldingdiff/utils.py:87-121): out[i] = the update kernels' own wrap of in[i]
 * (host arrays of n float32).  which: 0 = the FD_PREC_F32 kernels' copy (rowwise.hip), 1 = the default path's (rowwise_img.hip). */
int fd_test_wrap(int device_id, int which, const float* in, int64_t n, float* out);

/* C = LayerNorm(A W^T + bias + resid) * gamma + beta over full rows (HF BertSelfOutput / BertOutput,
 * reached from modelling.py:473-480).  use_fused != 0 asks for the LN-fused GEMM kernel of that precision
 * (FD_E_UNSUPPORTED if the shape has no fused instantiation); 0 runs GEMM(+residual) then the LayerNorm kernel. */
int fd_test_gemm_ln(int device_id, int precision, int use_fused, const float* A, const float* W, const float* bias,
                    const float* resid, const float* gamma, const float* beta, float eps, float* C, int M, int N,
                    int K);

/* Average launch time (ms) of the bias-epilogue token GEMM of the given precision on pseudo-random
 * operands, `reps` back-to-back launches bracketed by hipEvents (kernel micro-benchmark / ablations). */
int fd_test_gemm_time(int device_id, int precision, int M, int N, int K, int reps, double* ms_per_launch);

/* ---- measurement ---- */

/* When n > 0, every n-th reverse step of fd_sample* is launched eagerly with a
 * hipEvent pair around each kernel instead of replaying the graph, and the
 * durations are accumulated per kernel class.  0 disables (default). */
int fd_profile_every(fd_model* m, int n);
int fd_profile_reset(fd_model* m);
/* Number of kernel classes; name / accumulated ms / launch count / algorithmic
 * FLOPs and HBM bytes per launch for class i at the last profiled shape. */
int fd_profile_count(fd_model* m);
int fd_profile_get(fd_model* m, int i, const char** name, double* total_ms, int64_t* launches,
                   double* flops_per_launch, double* bytes_per_launch);

/* Block until all work queued on the model's own stream is done. */
int fd_synchronize(fd_model* m);

/* Failure detection for the asynchronous entry point: FD_E_NONFINITE if any reverse step since the last check
 * predicted a non-finite noise value (the host-buffer entry points check by themselves).  Call after the work
 * has completed (fd_synchronize / the caller's stream sync).  The reference has no such guard: a NaN there
 * silently propagates into the sampled angles (foldingdiff/sampling.py:62-75). */
int fd_check_finite(fd_model* m);

/* Debug / test aid (FD_PREC_F16X3 only): copy an intermediate of the last step back as float32.  `name`: "h", "a",
 * "ctx", "g", "g_head", "h_out" ([rows rounded up to 128][width]) or "q", "k", "v" ([B][H][padded L][32]); scales are
 * those of layer option "debug_layer"; option "debug_stop" = n ends a step after n kernel launches. */
int fd_debug_read(fd_model* m, const char* name, float* out, int64_t n_floats);

const char* fd_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* FDMI_H */
