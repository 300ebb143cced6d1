/* libbmhip -- C-ABI of the MI355X (gfx950) SimpleConv + ClipLoss training hot path.
 *
 * The reference (facebookresearch/brainmagick) is 100 % Python: its "FFI" for this path is the set
 * of PyTorch ops its modules call.  Each entry point below replaces the ATen kernels behind one such
 * call site (cited as bm/<file>:<line>, relative to the reference root).  The reference-side binding
 * is the ctypes table in brainmagick_amd/_lib.py.
 *
 * Conventions
 *   - every function returns int: 0 = ok, otherwise a hipError_t or a BM_ERR_* code; the message is
 *     available (thread-local) from bm_last_error().  Nothing aborts or throws across the ABI.
 *   - all tensors are fp32, contiguous, [B][channels][T] with T fastest, living in device memory
 *     owned by the caller.  Indices are int32 (widx, order, seg) or int64 (idx of group_by_index).
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); kernels are
 *     only enqueued, never synchronised.  The library allocates nothing; scratch is caller-provided.
 *   - deterministic: every reduction has a fixed order (split-K partials are folded in order).
 */
#ifndef BM_HIP_H
#define BM_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define BM_ACT_NONE 0
#define BM_ACT_GELU 1   /* exact erf GELU, nn.GELU()          bm/models/simpleconv.py:85-86 */
#define BM_ACT_RELU 2   /* nn.ReLU                             bm/models/simpleconv.py:89-90 */
#define BM_ACT_LEAKY 3  /* nn.LeakyReLU(relu_leakiness)        bm/models/simpleconv.py:87-88 */

/* ---- core ---- */
int bm_version(void);
const char* bm_last_error(void);
int bm_device_count(void);

/* ---- weight packing / grouping / split-K folding (pack.hip) ---- */
/* Tile geometry helpers shared with the host side. */
int bm_conv_mt_for(int M);
int bm_conv_mpad(int M);
int bm_conv_stats_tiles(int B, int T);
long bm_packed_weight_elems(int G, int M, int Cin, int KS);
/* dst[g][chunk][tap][16][Mpad] <- alpha * src[g*sg + m*sm + c*sc + tap'*sj]  (tap' flipped if flip).
 * Expresses nn.Conv1d weights (fwd and data-grad), SubjectLayers.weights (bm/models/common.py:49),
 * ConvTranspose1d(k=1) weights (simpleconv.py:189) and attention weights (common.py:357). */
int bm_pack_weights(const float* src, float* dst, int G, int M, int Cin, int KS, long sg, long sm,
                    long sc, long sj, int flip, const float* alpha_ptr, void* stream);
/* Stable grouping of segments by subject / layout index; replaces the gather at common.py:57. */
int bm_group_by_index(const long* idx, int B, int G, int* order, int* seg, int* err_flag, void* stream);
/* int64 -> int32 group indices with a range check: out-of-range entries set *err_flag (caller raises like
 * the reference's `self.weights.gather(0, subjects...)`, bm/models/common.py:57) and are clamped to 0. */
int bm_index_to_i32(const long* idx, int B, int G, int* out, int* err_flag, void* stream);
int bm_reduce_splits(const float* part, float* out, int G, int nsplit, int M, int Cn, int KS, long sg,
                     long sm, long sc, long sj, void* stream);
int bm_sum_over_batch(const float* x, float* out, int B, long n, void* stream);

/* ---- implicit-GEMM conv, fp32 MFMA (conv_nn.hip) ----
 * y[b][m][t] = ep(bias[m] + sum_{c,j} W[widx[b]][m][c][j] * x[b][c][t + (j-KS/2)*dil]).
 * Replaces F.conv1d (bm/models/common.py:113-114,133-138; simpleconv.py:113-120,185-189), the
 * einsums of SubjectLayers (common.py:58) and ChannelMerger (common.py:358), their data-gradients
 * and ClipLoss' dEstimate (losses.py:94 backward).  y_pre = value after bias; y_out = after
 * [affine ->] act [-> + res]; stats = per-tile (sum, sumsq) of y_pre for BatchNorm1d (common.py:119). */
int bm_conv1d_nn(const float* x, long x_bstride, const float* wpacked, const int* widx,
                 const float* bias, long bias_gstride, const float* ep_scale, const float* ep_shift, const float* res,
                 long res_bstride, float* y_pre, float* y_out, long y_bstride, float* stats, int B,
                 int Cin, int M, int T, int KS, int dil, int act, float leak, void* stream);

/* ---- fp32-ACCURATE contraction on the bf16 matrix cores: exact 3-way bf16 split of both operands,
 * six partial products per MFMA block, fp32 accumulate (conv_nn_x3.hip / gemm_nt_x3.hip).  Same
 * contracts as bm_conv1d_nn / bm_gemm_nt; compute mode "f32x3". */
long bm_packed_weight_elems_x3(int G, int M, int Cin, int KS);
int bm_pack_weights_x3(const float* src, void* dst, int G, int M, int Cin, int KS, long sg, long sm,
                       long sc, long sj, int flip, const float* alpha_ptr, void* stream);
int bm_conv1d_nn_x3(const float* x, long x_bstride, const void* wpacked, const int* widx,
                    const float* bias, long bias_gstride, const float* ep_scale, const float* ep_shift, const float* res,
                    long res_bstride, float* y_pre, float* y_out, long y_bstride, float* stats, int B,
                    int Cin, int M, int T, int KS, int dil, int act, float leak, void* stream);
int bm_gemm_nt_x3(const float* a, long a_sstride, long a_rstride, const float* x, long x_sstride,
                  long x_rstride, const int* order, const int* seg, float* part, int S, int G, int M,
                  int Cn, int T, int KS, int dil, int nsplit, void* stream);
/* tile height (MFMA row blocks: 3, 4, 5) and padded row count of the narrow f32x3 kernels */
int bm_conv_x3_mt_for(int M);
int bm_conv_x3_mpad(int M);

/* ---- fp32-ACCURATE contraction on the f16 matrix cores, compute mode "f16x2" (conv_nn_h2w.hip): every operand
 * is scaled by a power of two (per tensor for activations, per output row for weights) and split into two f16
 * planes; three partial products per MFMA block, fp32 accumulate, exact inverse scaling in the epilogue.  Half
 * the matrix-core work of "f32x3" at the same parity tolerances.  Same conv contract as bm_conv1d_nn plus
 * `x_amax` (device pointer to an amax slot of the input tensor: 8 floats whose maximum is max|x|, from bm_amax or
 * published by the producing kernel) and G (weight groups packed).
 * Shapes bm_conv_h2_covers() rejects go through bm_conv1d_nn_x3 (also fp32-accurate). */
int bm_conv_h2_mw_for(int M);
int bm_conv_h2_mpad(int M);
int bm_conv_h2_covers(int Cin, int M, int T, int KS, int dil);
long bm_packed_weight_bytes_h2(int G, int M, int Cin, int KS);
int bm_pack_weights_h2(const float* src, void* dst, int G, int M, int Cin, int KS, long sg, long sm, long sc,
                       long sj, int flip, const float* alpha_ptr, void* stream);
/* Batched packing: every weight tensor of a model re-packed by ONE launch per optimizer step.  The host fills a
 * table of bm_pack_h2_job_bytes()-byte jobs with bm_pack_h2_job_fill (same arguments as bm_pack_weights_h2 plus the
 * job's first workgroup `block0`; returns the job's workgroup count, < 0 on bad arguments), copies the table to
 * device memory, and calls bm_pack_weights_h2_batch(table, njobs, sum of the workgroup counts, largest Cin * KS of the jobs). */
int bm_pack_h2_job_bytes(void);
int bm_pack_h2_job_fill(void* job, const float* src, void* dst, int G, int M, int Cin, int KS, long sg, long sm,
                        long sc, long sj, int flip, const float* alpha_ptr, int block0);
int bm_pack_weights_h2_batch(const void* jobs_dev, int njobs, int total_blocks, int max_nk, void* stream);
/* amax slots and workspace: a slot is 8 floats whose maximum is max|x|; `amax_ws` is 16384 floats of scratch that
 * the producers of one stream may share.  bm_amax / bm_amax_checked overwrite the slot (per-workgroup partial maxima
 * in `amax_ws`, folded by a one-workgroup kernel), and so do the PRODUCERS (`amax_out` / `y_amax_out` arguments
 * below).  (bm_clip_cand_prep is the one producer that raises its slot with atomic max: it wants it ZEROED; hip_ops
 * carves every slot out of zeroed pools.) */
int bm_amax_ws_elems(void);
int bm_amax(const float* x, long n, float* out, float* amax_ws, void* stream);
/* Same pass, plus the reference's finiteness assert (bm/solver.py:258-260): *nonfinite_flag (device int, nullable)
 * is set to 1 when x holds an inf or a nan. */
int bm_amax_checked(const float* x, long n, float* out, float* amax_ws, int* nonfinite_flag, void* stream);
/* `stats` (nullable; only together with y_pre alone, no affine / activation / residual: the training-mode BatchNorm
 * layers): CHANNEL-MAJOR [M][bm_conv_h2_stats_tiles(B, T)][2] per-(segment, column tile, wavefront column) partial
 * sums and sums of squares of y_pre, every entry written; input of bm_bn_finalize_cm with ntiles =
 * bm_conv_h2_stats_tiles(B, T). */
int bm_conv_h2_stats_tiles(int B, int T);
int bm_conv1d_nn_h2(const float* x, long x_bstride, const float* x_amax, const void* wpacked, const int* widx,
                    const float* bias, long bias_gstride, const float* ep_scale, const float* ep_shift, const float* res,
                    long res_bstride, float* y_pre, float* y_out, long y_bstride, float* stats, int B, int Cin,
                    int M, int T, int KS, int dil, int act, float leak, int G, float* y_amax_out, float* amax_ws,
                    void* stream);

/* f16x2 weight gradients (gemm_nt_h2w.hip): part[split][m][c*KS + j] over S consecutive segments, one group,
 * KS in {1, 3}; a_amax / x_amax = device pointers to max|a| / max|x| (bm_amax).  Shapes bm_gemm_nt_h2_covers()
 * rejects (grouped / ordered calls, tiny shapes) go through bm_gemm_nt_x3. */
int bm_gemm_nt_h2_covers(int M, int Cn, int KS, int S, int T, int G, int dil, int ordered);
int bm_gemm_nt_h2_suggest_splits(int M, int Cn, int KS, int S, int T);
int bm_gemm_nt_h2(const float* a, long a_sstride, long a_rstride, const float* a_amax, const float* x,
                  long x_sstride, long x_rstride, const float* x_amax, float* part, int S, int M, int Cn, int T,
                  int KS, int dil, int nsplit, void* stream);
/* The same with per-row maxima of `a` ([M] floats, nullable): every row of A -- one gradient channel = one row of dW --
 * carries its own power-of-two scale (320-row tile family, T % 4 == 0; otherwise the per-tensor scale applies). */
int bm_gemm_nt_h2_rows(const float* a, long a_sstride, long a_rstride, const float* a_amax, const float* a_row_amax,
                       const float* x, long x_sstride, long x_rstride, const float* x_amax, float* part, int S,
                       int M, int Cn, int T, int KS, int dil, int nsplit, void* stream);

/* Grouped form (KS = 1): group g contracts the segments order[seg[g] .. seg[g + 1]) (order nullable = identity; seg [G + 1]),
 * part[g][split][m][c] -- the per-(layout, subject) weight gradient of the composed front end and the per-subject
 * gradient of SubjectLayers (bm/models/common.py:55-58 backward).  bm_gemm_nt_h2_covers(..., G, dil, ordered) tells
 * whether a shape is covered; bm_gemm_nt_h2_suggest_splits_grouped the splits per group. */
int bm_gemm_nt_h2_grouped(const float* a, long a_sstride, long a_rstride, const float* a_amax, const float* x,
                          long x_sstride, long x_rstride, const float* x_amax, const int* order, const int* seg,
                          float* part, int S, int G, int M, int Cn, int T, int nsplit, void* stream);
int bm_gemm_nt_h2_suggest_splits_grouped(int M, int Cn, int KS, int S, int T, int G);

/* ClipLoss score contraction in compute mode "f16x2" (bm/losses.py:94, torch.einsum("bct,oct,o->bo") before the
 * candidate norms): part[split][b][o] = sum over the split's share of k of est[b][k] * cand[o][k]; est [B][K] and
 * cand [Bc][K] dense fp32, K = F * T; est_amax / cand_amax as for bm_gemm_nt_h2.  256 candidates x 256 (long K) or 128
 * (short K) estimates per tile, the partial tiles written with 16-byte stores; folded by bm_clip_ce.  Only shapes bm_clip_scores_h2_covers() accepts (Bc % 4 == 0,
 * operands below 2 GB, at most half of a tile padded); otherwise bm_gemm_nt_h2 with M = B, Cn = Bc computes the same. */
int bm_clip_scores_h2_covers(int B, int Bc, long K);
int bm_clip_scores_h2_suggest_splits(int B, int Bc, long K);
int bm_clip_scores_h2(const float* est, const float* est_amax, const float* cand, const float* cand_amax, float* part,
                      int B, int Bc, long K, int nsplit, void* stream);

/* ---- time-contraction GEMM, fp32 MFMA, split-K (gemm_nt.hip) ----
 * part[g,split][m][c*KS+j] = sum_{s in group g} sum_t A[s][m][t] * X[s][c][t + (j-KS/2)*dil].
 * Replaces aten::convolution_backward (weight part), the weight-grad einsums of SubjectLayers /
 * ChannelMerger, torch.einsum("bct,oct,o->bo") (losses.py:94) and einsum("bcd,bod->boc") (common.py:355). */
int bm_gemm_nt_suggest_splits(int M, int Cn, int KS, int S, int T, int G);
int bm_clip_suggest_splits(int M, int Cn, int S, int T);
int bm_gemm_nt(const float* a, long a_sstride, long a_rstride, const float* x, long x_sstride,
               long x_rstride, const int* order, const int* seg, float* part, int S, int G, int M,
               int Cn, int T, int KS, int dil, int nsplit, void* stream);

/* ---- BatchNorm1d / activation / residual / GLU (norm_act.hip)  bm/models/common.py:113-151 ---- */
int bm_bn_finalize(const float* stats, int ntiles, int C, long count, const float* gamma,
                   const float* beta, float* running_mean, float* running_var, long* num_batches,
                   float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                   void* stream);
/* The same for channel-major partials stats[C][ntiles][2] (bm_conv1d_nn_h2's `stats`). */
int bm_bn_finalize_cm(const float* stats, int ntiles, int C, long count, const float* gamma,
                   const float* beta, float* running_mean, float* running_var, long* num_batches,
                   float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                   void* stream);
int bm_bn_eval_affine(int C, const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, float* mean, float* invstd, float* scale,
                      float* shift, void* stream);
/* `amax_out` (nullable amax slot) + `amax_ws` on the elementwise producers: the slot receives max|output|, the
 * f16x2 scale of the contraction that consumes the tensor -- saves a separate bm_amax pass over it.
 * `amax_rows_out` (bm_act_bn_bwd / bm_glu_bwd; nullable, [C] resp. [2 H] floats): max |output| per CHANNEL -- a
 * gradient channel is a row of the weight gradient, which bm_gemm_nt_h2_rows scales row by row. */
int bm_affine_act_res(const float* y, const float* scale, const float* shift, const float* res,
                      float* out, int B, int C, int T, int act, float leak, float* amax_out, float* amax_ws,
                      void* stream);
int bm_bwd_nsplit(int B);
long bm_act_bn_bwd_workspace_bytes(int B, int C);
int bm_act_bn_bwd_set_fused(int mode);                    /* 0: two passes; 1: one pass (-1: default = 1 unless BM_BN_BWD_FUSED=0); returns the previous setting */
int bm_act_bn_bwd_set_poll_limit(int polls);              /* test hook: polls before a workgroup computes its partners' sums itself (-1: default) */
long bm_act_bn_bwd_fused_fallbacks(void);                 /* debug counter (synchronises): workgroups of the one-pass kernel that gave up on a partner and computed its sums themselves */
int bm_act_bn_bwd_fused_covers(int B, int C, int T);      /* 1: train-mode BatchNorm backward runs as ONE pass (slab in registers) */
int bm_act_bn_bwd(const float* dout, const float* y, const float* scale, const float* shift,
                  const float* mean, const float* invstd, int bn_train, float* dy, float* dgamma,
                  float* dbeta, float* dbias, void* workspace, long workspace_bytes, int B, int C,
                  int T, int act, float leak, float* amax_out, float* amax_ws, float* amax_rows_out,
                  void* stream);
long bm_channel_sum_workspace_bytes(int B, int C);
int bm_channel_sum(const float* x, long bstride, float* out, void* workspace, long workspace_bytes,
                   int B, int C, int T, void* stream);
/* one-pass per-channel (sum, sumsq) partials [nsplit][C][2] of a conv output, folded by bm_bn_finalize */
int bm_channel_stats_splits(int B);
int bm_channel_stats(const float* x, float* stats, int B, int C, int T, void* stream);
int bm_glu_fwd(const float* u, float* out, int B, int H, int T, float* amax_out, float* amax_ws, void* stream);
long bm_glu_bwd_workspace_bytes(int B, int H);
int bm_glu_bwd(const float* dout, const float* u, float* du, float* dbias, void* workspace,
               long workspace_bytes, int B, int H, int T, float* amax_out, float* amax_ws,
               float* amax_rows_out, void* stream);
/* out[c][b] = sum_t x[b][c][t]  (per-segment time sums, transposed; bias gradient of the composed front end:
 * bm/models/simpleconv.py:113-120 + bm/models/common.py:55-58 chained) */
int bm_time_sums_t(const float* x, float* out, int B, int C, int T, void* stream);

/* ---- ScaleReject front end (scale.hip)  bm/norm.py:86-87,255-261,325-341; bm/solver.py:245-246 ----
 * out = clamp((x - center[group[b]][c]) / scale[group[b]][c]); maxabs[b] = max|out[b]| (zero-init). */
int bm_center_scale(const float* x, float* out, const long* group, const float* center,
                    const float* scale, int B, int C, int T, int clip, float limit, float* maxabs,
                    void* stream);

/* ---- ChannelMerger front end (merger.hip)  bm/models/common.py:239-271,334-357 ---- */
int bm_fourier_emb(const float* positions, float* emb, long rows, int D, float margin, void* stream);
int bm_masked_softmax(const float* scores, const float* positions, const float* ban_center,
                      float ban_radius, float* weights, int U, int O, int C, void* stream);
int bm_softmax_bwd(const float* w, const float* dw, float* ds, long rows, int C, void* stream);

/* ---- ClipLoss (clip.hip)  bm/losses.py:77-114 ---- */
int bm_clip_inv_norms(const float* cand, int Bc, long K, float* inv_norm, void* stream);
/* The same norms plus, in the same pass over the candidates: max |cand| raised into a ZEROED amax slot (nullable; the
 * f16x2 scale of the score contraction) and the finiteness assert of bm/solver.py:258-260 (nullable device int). */
int bm_clip_cand_prep(const float* cand, int Bc, long K, float* inv_norm, float* amax_slot, int* nonfinite_flag,
                      void* stream);
/* *flag |= 1 when any byte of the bool mask is 0: ClipLoss.forward's `assert mask.all()` (bm/losses.py:110) without
 * a host synchronisation in the middle of the step (the Solver reads the flag word at its one sync point). */
int bm_flag_unless_all_set(const unsigned char* mask, long n, int* flag, void* stream);
int bm_clip_ce(const float* part, int nsplit, const float* inv_norm, float* scores, float* probs,
               float* dscaled, float* loss_row, float* loss, int B, int Bc, int target_offset,
               void* stream);
/* The same with a column mask (col_valid [Bc] floats, nullable; 0 = the candidate is padding): a masked candidate gets
 * score -inf, probability 0 and gradient 0.  Whole-node negatives when ranks rejected different numbers of segments
 * (bm/solver.py:245-246 ScaleReject next to the candidate all-gather of brainmagick_amd.solver). */
int bm_clip_ce_masked(const float* part, int nsplit, const float* inv_norm, const float* col_valid, float* scores,
                      float* probs, float* dscaled, float* loss_row, float* loss, int B, int Bc, int target_offset,
                      void* stream);

/* Column term of the symmetric CLIP objective (opt-in extension named by the hot-path contract, "row/col softmax";
 * the reference computes the row term only, bm/losses.py:104-114): for every target candidate o = target_offset + j,
 * loss_col[j] = logsumexp_b scores[b][o] - scores[j][o].  dscaled (nullable; holds bm_clip_ce's row term) becomes
 * w_row * row term + w_col * (softmax over the column - onehot) / B * inv_norm[o] in place; loss (nullable; holds the
 * row loss) becomes w_row * loss + w_col * mean(loss_col). */
int bm_clip_ce_cols(const float* scores, const float* inv_norm, float* dscaled, float* loss_col, float* loss, int B,
                    int Bc, int target_offset, float w_row, float w_col, void* stream);

/* Gradient of ClipLoss w.r.t. the candidates (learnable feature model, bm/solver.py:304-320):
 * coef[o] = alpha * (sum_b dscaled[b,o]*scores[b,o]) / |cand_o| ;  y[r] -= coef[r] * x[r]. */
int bm_clip_cand_coef(const float* dscaled, const float* scores, const float* inv_norm,
                      const float* alpha, float* coef, int B, int Bc, void* stream);
int bm_row_axpy_sub(float* y, const float* x, const float* coef, int rows, long K, void* stream);
/* Retrieval evaluation: top-k columns per row + "own label among the top-k labels" hit flag.
 * Replaces probs.topk + label gather/compare of scripts/run_eval_probs.py:237-264 and bm/wer.py:104-111. */
int bm_topk_rows(const float* x, int rows, int cols, int k, int* idx_out, float* val_out,
                 const long* col_labels, const long* row_labels, int* hit_out, void* stream);

/* Word-level retrieval, batched (bm/wer.py:91-116): row softmax, per-row dot products (own target
 * replaces the last negative), per-vocabulary-word sums over columns pre-sorted by word. */
int bm_row_softmax(const float* x, float* y, int rows, int cols, void* stream);
int bm_rowwise_dot(const float* a, const float* b, const float* scale, float* out, int rows, long K,
                   void* stream);
int bm_segment_sum_cols(const float* p, const int* order, const int* seg, float* pv, int rows, int cols,
                        int V, void* stream);

/* ---- fused Adam on the flat bucket (adam.hip)  torch.optim.Adam @ bm/train.py:118-119 ---- */
int bm_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, int step,
                 double lr, double beta1, double beta2, double eps, double grad_scale, void* stream);

/* ---- data-parallel collectives: RCCL behind the C-ABI (comm.hip) ----
 * Replace flashy.distrib.init (bm/train.py:139), flashy.distrib.sync_model (bm/solver.py:386: gradient
 * mean = in-place reduce-scatter of the flat gradient bucket + Adam on the shard + in-place all-gather of
 * the parameters; BatchNorm buffers by all-reduce), flashy.distrib.average_metrics (bm/solver.py:395), and
 * add the whole-node candidate all-gather (README.md:139-143 keeps negatives per GPU).  One process per GPU.
 * librccl is dlopen'ed on first use (BM_RCCL_LIB overrides the path).  The opaque handle is the only state
 * the library owns; fp32 buffers; collectives are enqueued on `stream`, never synchronised; NCCL in-place
 * conventions (all-gather: send == recv + rank*count; reduce-scatter: recv == send + rank*count). */
int bm_comm_unique_id_bytes(void);
int bm_comm_available(void);                              /* 0 = librccl loads and has every symbol (local check) */
int bm_comm_unique_id(void* out_id);                       /* rank 0; the host hands the bytes to the others */
int bm_comm_init(const void* id, int world, int rank, int device, void** handle);
int bm_comm_destroy(void* handle);
int bm_comm_world(void* handle);
int bm_comm_rank(void* handle);
int bm_comm_reported_world(void* handle);                 /* ncclCommCount: the size RCCL itself reports (-1: unknown) */
int bm_comm_reported_rank(void* handle);                  /* ncclCommUserRank */
int bm_comm_allgather(void* handle, const float* send, float* recv, long count, void* stream);
int bm_comm_reduce_scatter(void* handle, const float* send, float* recv, long count, void* stream);
int bm_comm_allreduce(void* handle, const float* send, float* recv, long count, int op, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BM_HIP_H */
