/*
 * mm_native.h — C ABI of libmm_native.so: MI355X (gfx950) interaction-scoring kernels for
 * matchmaker's re-ranking forward pass.
 *
 * The reference (sebastian-hofstaetter/matchmaker) is pure Python: it has no FFI of its own.
 * Its "operator API" for this path is the nn.Module protocol of SURVEY.md §8(b).  Each entry
 * point below replaces the arithmetic of one reference method; the Python host side
 * (the .py modules of matchmaker_amd/) mirrors the method signatures and binds these symbols with ctypes
 * (INTEGRATION.md shows the stub a matchmaker maintainer would add).
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HBM) unless stated; tensors are dense, row-major,
 *     innermost dimension contiguous, 16-byte aligned base;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue work;
 *   - return 0 on success, a negative MM_E* code otherwise; mm_last_error() gives a
 *     thread-local message.  Nothing is allocated, retained or synchronised by the library.
 */
#ifndef MM_NATIVE_H
#define MM_NATIVE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM_ABI_VERSION 4 /* 4: mm_kernel_pool_ex_fwd2 / mm_kernel_pool_ex_bwd2 (the forward hands its pooled kernel sums to the backward),
                          *    mm_maxsim_fwd_batched (several eval.py-sized batches per launch).
                          * 3: 2: `flags` on the MaxSim forwards (the reference's 16-bit dtype flow); mm_tkl_fwd's ascending chunk_slot
                            contract and workspace layout.  3: + mm_tkl_fwd_peaks (the region search's three peak indices) */

/* element types of the embedding tensors */
#define MM_F32 0
#define MM_F16 1
#define MM_BF16 2

/* how a mask argument is encoded (the reference passes int64 HF attention masks to ColBERT —
 * colbert.py:69,73 — and float {0,1} masks to TK/TKL — neuralIR_encoder.py:35-37) */
#define MM_MASK_NONE 0    /* pointer ignored: every position is a real token              */
#define MM_MASK_LEN_I32 1 /* int32[rows]: positions >= len are padding (prefix masks)      */
#define MM_MASK_U8 2      /* uint8/bool[rows, L], nonzero = real token                     */
#define MM_MASK_I64 3     /* int64[rows, L]   (HF tokenizer attention_mask)                */
#define MM_MASK_F32 4     /* float[rows, L]   (matchmaker embedding-model masks)           */

/* error codes */
#define MM_OK 0
#define MM_EINVAL -1       /* bad argument (null pointer, non-positive size, bad enum)     */
#define MM_EUNSUPPORTED -2 /* shape/dtype outside what the kernels implement               */
#define MM_EWORKSPACE -3   /* workspace missing or too small                               */
#define MM_ELAUNCH -4      /* HIP reported a launch error                                  */

/* `flags` of the MaxSim forwards: the reference's dtype flow for 16-bit inputs.  In the reference the similarity matrix has
 * the dtype of the token vectors: under torch.cuda.amp.autocast(enabled=use_fp16) — config/train/defaults.yaml:21
 * `use_fp16: True`, colbert.py:60 — `bmm` returns fp16 (fp32 accumulation, ONE rounding per element), the -1000 fill and
 * `max` stay fp16 and only `sum` is promoted to fp32 (colbert.py:68-75).  Rounding is monotone, so rounding the per-token
 * maximum reproduces that arithmetic exactly.
 *   MM_SIM_ROUND  every per-query-token maximum is rounded (RNE) to the element type of q / d before the fp32 sum:
 *                 ColBERT.forward / forward_aggregation under autocast (colbert.py:60-75, indexing_heads.py:49-56)
 *   MM_SUM_ROUND  the pair's sum is rounded to that type as well: 16-bit tensors OUTSIDE autocast, where `sum` is a 16-bit op
 *                 too (fp32 accumulation, one rounding) — the dynamic teacher's all-pairs call, dynamic_teacher.py:245-246
 * Both are no-ops for MM_F32 inputs.  0 = fp32 accumulators through max and sum (the fp32 contract of `use_fp16: False`). */
#define MM_SIM_ROUND 1
#define MM_SUM_ROUND 2

int mm_abi_version(void);
const char* mm_last_error(void);

/* ------------------------------------------------------------------------------------------
 * ColBERT late-interaction MaxSim.
 *
 *   out[p] = sum_{i < Q, q_mask[i]}  max_{j < D} ( d_mask[p, j] ? <q[i,:], d[p,j,:]> : -1000 )
 *
 * Replaces: ColBERT.forward scoring block            matchmaker/models/colbert.py:68-75
 *           ColBERT.forward_aggregation (no masks)   matchmaker/models/colbert.py:100-112
 *
 *   q   [n_queries, Q, E]   n_queries = ceil(n_pairs / pairs_per_query)
 *   d   [n_pairs,   D, E]   pair p uses query p / pairs_per_query
 *   out [n_pairs] float32
 *
 * pairs_per_query = 1 is the reference's pair-per-row layout (query replicated per pair,
 * eval.py:108); pairs_per_query = C is the "1 query x C candidates" re-ranking layout in which
 * the query tile is read once per candidate list.
 * q_mask rows follow q (n_queries rows), d_mask rows follow d (n_pairs rows).
 * flags: MM_SIM_ROUND / MM_SUM_ROUND above (0 = fp32 through max and sum).
 * workspace: mm_maxsim_workspace_bytes() bytes of device scratch (may be 0 -> NULL allowed).  It holds the packed masks;
 *   calls whose int64 masks the kernel reads itself (the pair-per-row layout, and long queries when every wavefront scores
 *   one pair: eval.py's 512-pair batches) leave it untouched.
 */
size_t mm_maxsim_workspace_bytes(int64_t n_pairs, int64_t pairs_per_query, int Q, int D,
                                 int q_mask_kind, int d_mask_kind);

int mm_maxsim_fwd(const void* q, const void* d,
                  const void* q_mask, int q_mask_kind,
                  const void* d_mask, int d_mask_kind,
                  float* out,
                  int64_t n_pairs, int64_t pairs_per_query,
                  int Q, int D, int E, int dtype, int flags,
                  void* workspace, size_t workspace_bytes, void* stream);

/* Several pair-per-row batches of ONE shape scored by ONE launch: the scoring block of ColBERT.forward (colbert.py:68-75) as
 * eval.py:82-108 drives it — `batch_size_eval: 512` pairs per model.forward (config/train/defaults.yaml:115) — for a caller that
 * holds the token vectors of several batches (matchmaker_amd/rerank.py evaluate_batches(score_group=...)).  A 512-pair call at
 * dim 128 is 3.6 us of HBM time behind a ~9 us launch chain; the group pays the chain once.
 *   batches[i]: q [n_pairs, Q, E], d [n_pairs, D, E] (16-bit vectors, 16-byte aligned), q_mask [n_pairs, Q] / d_mask [n_pairs, D]
 *   (int64 tokenizer masks, or NULL for all of them), out [n_pairs] float32; the descriptors are HOST memory (read during the call).
 *   Shapes the pair-per-row kernel takes (Q <= 32, E in {128, 256, 384, 512, 768}, even Q and D <= 256 with masks); anything
 *   else returns MM_EUNSUPPORTED and the caller scores batch by batch with mm_maxsim_fwd.  n_batches <= MM_MAXSIM_MAX_BATCHES.
 * Scores: bit-equal to mm_maxsim_fwd on each batch alone (same kernel body; flags as there). */
#define MM_MAXSIM_MAX_BATCHES 16
typedef struct {
  const void* q;
  const void* d;
  const void* q_mask;
  const void* d_mask;
  float* out;
  int64_t n_pairs;
} mm_maxsim_batch_t;
int mm_maxsim_fwd_batched(const mm_maxsim_batch_t* batches, int n_batches, int q_mask_kind, int d_mask_kind,
                          int Q, int D, int E, int dtype, int flags, void* stream);

/* All-pairs MaxSim: out[i, j] over query i x document j.
 * Replaces ColBERT.forward_inbatch_aggregation       matchmaker/models/colbert.py:154-162
 * bug_compatible != 0 reproduces the reference's mask expansion (:158), which masks
 * score[i, j] with document i's mask and requires Bq == Bd (MM_EINVAL otherwise).
 *   q [Bq, Q, E], d [Bd, D, E], out [Bq, Bd] float32 */
size_t mm_maxsim_inbatch_workspace_bytes(int64_t Bq, int64_t Bd, int Q, int D,
                                         int q_mask_kind, int d_mask_kind);

int mm_maxsim_inbatch_fwd(const void* q, const void* d,
                          const void* q_mask, int q_mask_kind,
                          const void* d_mask, int d_mask_kind,
                          float* out,
                          int64_t Bq, int64_t Bd, int Q, int D, int E, int dtype,
                          int bug_compatible, int flags,
                          void* workspace, size_t workspace_bytes, void* stream);

/* Ragged (CSR) MaxSim over a resident token store: document p = rows [doc_begin[p], doc_end[p]) of
 * `tokens` [T, E]; no padding, no document mask (every stored row is a real token: zero rows were
 * stripped when the store was written, dense_retrieval.py:244).  One launch replaces the
 * per-candidate Python loop of the ColBERT retrieval aggregate
 *   matchmaker/dense_retrieval.py:398-412   (doc_infos[seq_id] = (file, start, end) -> storage[file][start:end])
 *   ColBERT.forward_aggregation             matchmaker/models/colbert.py:100-112
 * q [n_queries, Q, E] in the store's dtype, pair p uses query p / pairs_per_query; q_mask as for
 * mm_maxsim_fwd (MM_MASK_NONE reproduces forward_aggregation exactly).  An empty range scores like
 * a fully padded document (-1000 per query token).  out [n_pairs] float32. */
size_t mm_maxsim_ragged_workspace_bytes(int64_t n_pairs, int64_t pairs_per_query, int Q, int q_mask_kind);

int mm_maxsim_ragged_fwd(const void* q, const void* tokens, const int64_t* doc_begin, const int64_t* doc_end,
                         const void* q_mask, int q_mask_kind, float* out,
                         int64_t n_pairs, int64_t pairs_per_query, int Q, int E, int dtype, int flags,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Calibration, not scoring: mm_maxsim_fwd's HBM read stream with the arithmetic removed (same launch geometry, LDS-DMA
 * blocks, ring and counted waits; no MFMA, no maximum) over `bytes` of `src` (16-byte aligned, a multiple of 8192 bytes).
 * bench.py times it over the headline's document tensor and prints the headline kernel's rate as a fraction of it, so
 * that a slow box and a slow kernel can be told apart from the benchmark line alone.  nt != 0: non-temporal loads, as
 * the headline kernel issues them. */
int mm_hbm_stream_probe(const void* src, int64_t bytes, int nt, void* stream);

/* Backward of the paired MaxSim (pair-per-row layout, the one train.py uses: train.py:347-348,
 * loss.backward() :503-524).  Recomputes the similarities and routes grad_out[p] to the FIRST
 * arg-max document position of every real query token (torch.max's rule); nothing flows through
 * the -1000 sentinel (colbert.py:69) or padded query tokens (:73).
 *   grad_out [n_pairs] float32; grad_q [n_pairs, Q, E], grad_d [n_pairs, D, E] of element type grad_dtype: MM_F32, or the
 *   token vectors' own 16-bit type (what autograd hands back to an fp16 / bf16 encoder: summed in fp32, rounded once).
 *   Both are fully written by the call (rows without gradient are zeros: no memset needed in front of it).
 *   q/d/masks exactly as given to the forward. */
size_t mm_maxsim_bwd_workspace_bytes(int64_t n_pairs, int Q, int D, int q_mask_kind, int d_mask_kind);

int mm_maxsim_bwd(const void* q, const void* d,
                  const void* q_mask, int q_mask_kind,
                  const void* d_mask, int d_mask_kind,
                  const float* grad_out, void* grad_q, void* grad_d, int grad_dtype,
                  int64_t n_pairs, int Q, int D, int E, int dtype,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * TK kernel pooling (cosine match matrix + K RBF kernels + log-sum pooling + bin weights).
 *
 *   cos[i,j]  = <q_i, d_j> / ((|q_i| + 1e-13)(|d_j| + 1e-13))
 *   pkq[i,k]  = sum_j d_mask[j] * exp(-(cos[i,j] - mu[k])^2 / (2 sigma[k]^2))
 *   out[p]    = sum_k w[k] * sum_i q_mask[i] * log(max(pkq[i,k] * alpha[k], 1e-10))
 *
 * Replaces: ECAI20_TK.forward pooling block   matchmaker/models/published/ecai20_tk.py:105-124
 *           (cosine = allennlp CosineMatrixAttention, call site ecai20_tk.py:105)
 *
 *   q [n_queries, Q, E], d [n_pairs, D, E] float32 contextualised embeddings
 *   mu, sigma, alpha, w: float32[K] device pointers, K <= 32 (K = 11, the reference configs, takes the
 *          streaming kernels; any other count the generic kernel)
 *   per_kernel: optional float32 [n_pairs, K] (the reference's secondary output), may be NULL
 *   masks: float {0,1} as the reference passes them (MM_MASK_F32) or any other mm mask kind;
 *          nonzero = real token.  workspace as for mm_maxsim_fwd.
 */
size_t mm_kernel_pool_workspace_bytes(int64_t n_pairs, int64_t pairs_per_query, int Q, int D,
                                      int q_mask_kind, int d_mask_kind);

int mm_kernel_pool_fwd(const void* q, const void* d,
                       const void* q_mask, int q_mask_kind,
                       const void* d_mask, int d_mask_kind,
                       const float* mu, const float* sigma, const float* alpha, const float* w,
                       float* out, float* per_kernel,
                       int64_t n_pairs, int64_t pairs_per_query,
                       int Q, int D, int E, int K, int dtype,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Variants of the pooling block that share its arithmetic:
 *   d_gate     optional float32 [n_pairs, D], >= 0: every activation of document token j is multiplied by
 *              d_gate[p, j] on top of d_mask — TK-Sparse's learned stop-word vector
 *              (matchmaker/models/published/cikm20_tk_sparse.py:133-135; negative values count as 0, the
 *              reference's gate is a ReLU output).  NULL = no gate (= mm_kernel_pool_fwd).
 *   clamp_min  the floor inside the log: 1e-10 for TK / TK-Sparse (ecai20_tk.py:121, cikm20_tk_sparse.py:142),
 *              1e-4 for the IDCM passage sampler (matchmaker/models/published/sigir21_idcm.py:182-186, whose
 *              pre-normalised vectors make the cosine's own normalisation a no-op).  Must be > 0.
 *   pair_query optional int32 [n_pairs]: the query row of every pair, for ragged groups — IDCM scores a
 *              different number of passages per document against the document's query (the reference
 *              materialises one query copy per passage, sigir21_idcm.py:143-144).  With it q / q_mask have
 *              n_queries rows and pairs_per_query is ignored; NULL = the uniform pairs_per_query layout.
 *              Consecutive equal entries reuse the query tile already in registers.  The workspace then packs
 *              n_queries query-mask rows: size it with
 *              mm_kernel_pool_workspace_bytes(max(n_pairs, n_queries), 1, Q, D, kinds).
 * Everything else as mm_kernel_pool_fwd (which calls this with NULL, NULL, 0, ..., 1e-10). */
int mm_kernel_pool_ex_fwd(const void* q, const void* d,
                          const void* q_mask, int q_mask_kind,
                          const void* d_mask, int d_mask_kind,
                          const float* d_gate,
                          const int32_t* pair_query, int64_t n_queries,
                          const float* mu, const float* sigma, const float* alpha, const float* w,
                          float clamp_min,
                          float* out, float* per_kernel,
                          int64_t n_pairs, int64_t pairs_per_query,
                          int Q, int D, int E, int K, int dtype,
                          void* workspace, size_t workspace_bytes, void* stream);

/* mm_kernel_pool_ex_fwd with one more optional output, for training (train.py:347-348 forward, :503-524 backward):
 *   pooled [n_pairs, Q, K] float32 or NULL: the pooled kernel sums pkq[i][k] = sum_j mask_j gate_j exp(-(cos_ij - mu_k)^2 / (2 sigma_k^2))
 *   (ecai20_tk.py:120, before kernel_alpha_scaler / clamp / log), one row per query token.  Every position of the document
 *   enters each of them, so the backward cannot form a gradient before it has them; handed to mm_kernel_pool_ex_bwd2 they
 *   save its pooling pre-pass — the document's second trip through HBM.  Rows of padded / masked query tokens are unspecified. */
int mm_kernel_pool_ex_fwd2(const void* q, const void* d,
                           const void* q_mask, int q_mask_kind,
                           const void* d_mask, int d_mask_kind,
                           const float* d_gate,
                           const int32_t* pair_query, int64_t n_queries,
                           const float* mu, const float* sigma, const float* alpha, const float* w,
                           float clamp_min,
                           float* out, float* per_kernel, float* pooled,
                           int64_t n_pairs, int64_t pairs_per_query,
                           int Q, int D, int E, int K, int dtype,
                           void* workspace, size_t workspace_bytes, void* stream);

/* Several (query tensor, document tensor) combinations pooled in ONE launch and summed: Conv-KNRM scores every
 * n-gram width of the query against every n-gram width of the document (n_grams^2 match matrices,
 * matchmaker/models/conv_knrm.py:130-132) and its dense layer (:137) is a weighted sum over all of them:
 *   out[p] = sum_{i < n_q, t < n_d} kernel_pool(q_list[i], d_list[t]; bin weights w[(i * n_d + t) * K ...])[p]
 * (summed in (i, t) order: deterministic).  q_list[i] [n_queries, Q, E], d_list[t] [n_pairs, D, E] float32, one mask
 * pair for all of them (the n-gram tensors share the token masks).  1 <= n_q, n_d <= 4, K = 11.
 * Workspace: mm_kernel_pool_multi_workspace_bytes (mask packing + the n_q * n_d partial score rows). */
size_t mm_kernel_pool_multi_workspace_bytes(int64_t n_pairs, int64_t pairs_per_query, int n_q, int n_d, int Q, int D,
                                            int q_mask_kind, int d_mask_kind);
int mm_kernel_pool_multi_fwd(const void* const* q_list, int n_q, const void* const* d_list, int n_d,
                             const void* q_mask, int q_mask_kind, const void* d_mask, int d_mask_kind,
                             const float* mu, const float* sigma, const float* alpha, const float* w,
                             float clamp_min, float* out, int64_t n_pairs, int64_t pairs_per_query, int Q, int D,
                             int E, int K, int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of mm_kernel_pool_fwd in the pair-per-row layout (training: train.py:347-348, loss.backward()
 * :503-524; the embedding model is called from neuralIR_encoder.py:86-87).  Gradients of the score w.r.t.
 * the contextualised embeddings and the two trainable pooling parameters (kernel_alpha_scaler
 * ecai20_tk.py:85, kernel_bin_weights :81); mu / sigma are buffers.
 *   grad_out [n_pairs]; grad_q [n_pairs, Q, E], grad_d [n_pairs, D, E] float32;
 *   grad_alpha, grad_w [n_pairs, K]: per-pair contributions (sum over pairs on the host side:
 *   deterministic, no atomics). */
size_t mm_kernel_pool_bwd_workspace_bytes(int64_t n_pairs, int Q, int D, int q_mask_kind, int d_mask_kind);
/* ... plus the partial grad_q buffers of a small batch (<= 128 pairs: several workgroups share a pair's document blocks and a
 * combine kernel finishes grad_q).  Optional: with the smaller workspace above each pair gets one workgroup. */
size_t mm_kernel_pool_bwd_workspace_bytes2(int64_t n_pairs, int Q, int D, int E, int q_mask_kind, int d_mask_kind);

int mm_kernel_pool_bwd(const void* q, const void* d,
                       const void* q_mask, int q_mask_kind,
                       const void* d_mask, int d_mask_kind,
                       const float* mu, const float* sigma, const float* alpha, const float* w,
                       const float* grad_out, float* grad_q, float* grad_d, float* grad_alpha, float* grad_w,
                       int64_t n_pairs, int Q, int D, int E, int K,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Backward of mm_kernel_pool_ex_fwd: as mm_kernel_pool_bwd, plus grad_gate [n_pairs, D] (may be NULL;
 * needs d_gate), the gradient w.r.t. the gate values (training of TK-Sparse's stop-word MLP,
 * cikm20_tk_sparse.py:133-135). */
int mm_kernel_pool_ex_bwd(const void* q, const void* d,
                          const void* q_mask, int q_mask_kind,
                          const void* d_mask, int d_mask_kind,
                          const float* d_gate,
                          const float* mu, const float* sigma, const float* alpha, const float* w,
                          float clamp_min,
                          const float* grad_out, float* grad_q, float* grad_d, float* grad_gate,
                          float* grad_alpha, float* grad_w,
                          int64_t n_pairs, int Q, int D, int E, int K,
                          void* workspace, size_t workspace_bytes, void* stream);

/* mm_kernel_pool_ex_bwd with the forward's pooled kernel sums: pooled [n_pairs, Q, K] as written by mm_kernel_pool_ex_fwd2 on
 * the SAME inputs, or NULL (then the backward pools them itself first and needs the workspace
 * mm_kernel_pool_bwd_workspace_bytes reports).  Same gradients either way (the sums are the same arithmetic). */
int mm_kernel_pool_ex_bwd2(const void* q, const void* d,
                           const void* q_mask, int q_mask_kind,
                           const void* d_mask, int d_mask_kind,
                           const float* d_gate,
                           const float* mu, const float* sigma, const float* alpha, const float* w,
                           float clamp_min, const float* pooled,
                           const float* grad_out, float* grad_q, float* grad_d, float* grad_gate,
                           float* grad_alpha, float* grad_w,
                           int64_t n_pairs, int Q, int D, int E, int K,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * TKL: match + RBF kernels per document position, sliding-window (30, stride 2) pooling with
 * learned saturation, per-window score; then top-3 non-overlapping region scoring.
 *
 * Replaces: TKL_sigir20.forward   matchmaker/models/published/sigir20_tkl.py:180-252 (windows)
 *                                 matchmaker/models/published/sigir20_tkl.py:254-286 (regions)
 *
 *   q_ctx        [B, Q, E]   contextualised query, already multiplied by its mask (:306)
 *   chunks       [P, 50, E]  contextualised packed chunks (output of :172); the 40 centre
 *                            tokens of each are used (:174)
 *   chunk_mask   [P, 50]     float {0,1} (padding_packed, :163)
 *   chunk_slot   [P] int32   flat slot b*C + c of each packed chunk (packed_indices :159, as indices), ASCENDING — the order
 *                            boolean-mask packing (:160-162) and torch.nonzero produce; the kernels rely on a document's
 *                            chunks being adjacent and on its last kept chunk coming last
 *   q_mask       [B, Q]      float {0,1}
 *   params       float32[MM_TKL_NPARAMS(K, E)] device, packed as (matchmaker_amd/tkl.py pack_params()):
 *                  mu[K] sigma[K] dense.weight[K] kernel_mult[0][K]
 *                  saturation_linear{w[2], b}  saturation_linear2{w[2], b}  saturation_linear3{w[2], b}
 *                  sat_normer{weight[2], bias[2]}  chunk_scoring[15]  sat_emb_reduce1.weight[E]
 *   saturation   0 = "embedding" (:224-234), 1 = "log" (:245-246)
 *   win_scores   [B, W] float32 out (W = (max(C*40,30) - 30)/2 + 1), may be NULL if workspace given
 *   out          [B] float32
 *   workspace    mm_tkl_workspace_bytes() bytes of device scratch: the slot -> packed-chunk map, the hand-off between the
 *                match stage and the window stage (the scaled, masked cosines of the real query tokens, 4 B per document
 *                position and token; on the A/B paths the pair sums of round 2), packed masks, and the partial window
 *                scores of the query-token groups.  Nothing in it survives the call.
 */
#define MM_TKL_NPARAMS(K, E) (4 * (K) + 13 + 15 + (E))
#define MM_TKL_SAT_EMBEDDING 0
#define MM_TKL_SAT_LOG 1
size_t mm_tkl_workspace_bytes(int64_t B, int64_t P, int C, int Q, int K);

int mm_tkl_fwd(const void* q_ctx, const void* chunks, const float* chunk_mask,
               const int32_t* chunk_slot, const float* q_mask, const float* params,
               float* win_scores, float* out,
               int64_t B, int64_t P, int C, int Q, int E, int K, int saturation,
               void* workspace, size_t workspace_bytes, void* stream);

/* mm_tkl_fwd + the region search's own result (ABI 3): top_idx [B, 3] int32 out (may be NULL) receives, per document, the
 * three arg-max window indices in round order — the reference's `top_non_overlapping_idx` (sigir20_tkl.py:266-271, returned
 * under output_secondary_output :290).  With win_scores they give `top_k_non_overlapping` (:276-282) by a gather of 15
 * values per document (matchmaker_amd/tkl.py does that), and they are what a rank-parity check needs to tell "the device
 * picked another region of equal score" from "the device scored a region wrongly" (tie policy). */
int mm_tkl_fwd_peaks(const void* q_ctx, const void* chunks, const float* chunk_mask,
                     const int32_t* chunk_slot, const float* q_mask, const float* params,
                     float* win_scores, float* out, int32_t* top_idx,
                     int64_t B, int64_t P, int C, int Q, int E, int K, int saturation,
                     void* workspace, size_t workspace_bytes, void* stream);

/* Backward of mm_tkl_fwd (training: train.py:503-524 through sigir20_tkl.py:180-286).  The document score is a weighted
 * sum of at most 15 window scores whose indices are piecewise constant, so the exact gradient involves only those windows:
 * they are recomputed from the contextualised vectors and differentiated on the device.
 *   win_scores [B, W]  the forward's window scores (selects the windows; 0 = empty window = constant)
 *   grad_out [B];  grad_q [B, Q, E];  grad_chunks [P, 50, E] (zeroed by the call; only rows of selected windows are non-zero);
 *   grad_params [B, MM_TKL_NPARAMS(K, E)]: per-document gradients in the layout of `params` (mu / sigma columns are 0) —
 *   sum over the documents on the host side (deterministic, no atomics).  Q <= 32.
 * Workspace: mm_tkl_bwd_workspace_bytes(B, C). */
size_t mm_tkl_bwd_workspace_bytes(int64_t B, int C);
/* ... plus the per-region shares of grad_q and of the parameter rows of a small batch (3 B <= 256: three workgroups per document,
 * one per arg-max region).  Optional: with the smaller workspace above every document gets one workgroup. */
size_t mm_tkl_bwd_workspace_bytes2(int64_t B, int C, int Q, int E);
int mm_tkl_bwd(const void* q_ctx, const void* chunks, const float* chunk_mask, const int32_t* chunk_slot,
               const float* q_mask, const float* params, const float* win_scores, const float* grad_out,
               float* grad_q, float* grad_chunks, float* grad_params,
               int64_t B, int64_t P, int C, int Q, int E, int K, int saturation,
               void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Brute-force inner-product top-k over one GPU's shard of the collection (dense retrieval).
 *
 * Replaces: FaissIdIndexer / FaissBaseIndexer.search   matchmaker/retrieval/faiss_indices.py:22-36, 49-74
 *           (IndexIDMap(IndexFlatIP) sharded over the GPUs, useFloat16; faiss-gpu==1.7.0 is a
 *           third-party dependency, conda-requirements.txt:1), call site dense_retrieval.py:391;
 *           score = BERT_DOT dot product               matchmaker/models/bert_dot.py:62
 *
 *   queries [nq, E], corpus [n_docs, E]  float16 / bfloat16, E in {128, 256, 384, 512, 768}
 *   out_scores [nq, k] float32 descending; out_idx [nq, k] int64 = row of `corpus` (-1 and -inf pad
 *   a shard with fewer than k documents, as faiss does); ties: lower row first.
 *   status [nq] int32: 0 = exact top-k delivered; 1 / 2 = the sampled threshold of that query let
 *   too few / too many candidates through — call again for those queries with m_scale x4 / x0.25
 *   (matchmaker_amd/retrieval.py does).  m_scale = 1 on the first call.
 *   workspace: mm_dot_topk_workspace_bytes(n_docs, nq, k) bytes.  k <= 4096. */
size_t mm_dot_topk_workspace_bytes(int64_t n_docs, int nq, int k);

int mm_dot_topk_fwd(const void* queries, const void* corpus, int64_t n_docs, int nq, int E, int dtype, int k,
                    float m_scale, float* out_scores, int64_t* out_idx, int32_t* status,
                    void* workspace, size_t workspace_bytes, void* stream);

/* Final merge of a sharded index: in [nq, n_in] (score, id) rows (e.g. the all-gathered per-shard
 * top-k lists, ids < 0 = padding) -> the k best per row, score descending, input order on ties.
 * n_in <= 16384. */
int mm_topk_merge(const float* in_scores, const int64_t* in_ids, int nq, int n_in, int k,
                  float* out_scores, int64_t* out_ids, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MM_NATIVE_H */
