/*
 * libctrhip — C ABI of the MI355X (gfx950) CTR hot path.
 *
 * The reference (LiuJun523/recommendation) has no FFI of its own: its hot path is the
 * TensorFlow-1.x graph built in each model class's define_model() and executed by
 * session.run() (e.g. DeepFM/DeepFM.py:34-100, :114-123).  Each entry point below
 * replaces one cluster of graph ops; the cluster is cited at the declaration.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), allocates
 *     nothing, never synchronises the host and never throws;
 *   - scratch memory comes in through (workspace, workspace_bytes); the matching
 *     *_workspace_bytes() query is a pure host function;
 *   - return value: 0 = ok, <0 = bad argument (ctr_last_error() tells which),
 *     >0 = a hipError_t;
 *   - matrices are row-major fp32 with an explicit leading dimension (elements);
 *   - indices are int32 (the reference narrows int64 feeds to int32, DeepFM/DeepFM.py:37);
 *   - "keep threshold" = floor(keep_prob * 2^24); element i of dropout stream s is kept
 *     iff hash24(seed, s, i) < threshold (ctr_keep_hash24 is the host copy of the hash).
 */
#ifndef CTRHIP_H
#define CTRHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTR_OK 0
#define CTR_EINVAL (-1)
#define CTR_EWORKSPACE (-2)
#define CTR_EUNSUPPORTED (-3)

const char* ctr_last_error(void);
int ctr_version(void);
/* Host copy of the device dropout hash (24 random bits). */
uint32_t ctr_keep_hash24(uint64_t seed, uint64_t stream_id, uint64_t index);

/* Dropout descriptor: threshold == 0 or >= 2^24 with scale == 1 means "off". */
typedef struct {
  uint64_t seed;
  uint64_t stream_id;
  uint32_t keep_threshold; /* floor(keep_prob * 2^24) */
  float scale;             /* 1 / keep_prob */
  const uint64_t* seed_add; /* optional device word added to seed when the kernel runs (NULL: none): lets a
                               captured HIP graph be replayed with a new step count */
} ctr_dropout_t;

/* ------------------------------------------------------------------------------------
 * Embedding + FM front end.
 * Replaces: tf.one_hot + tf.nn.embedding_lookup + tf.multiply + tf.concat
 * (DeepFM/DeepFM.py:43-53 and its copies), the standard lookup (Wide_Deep/Wide_Deep.py:51,
 * DIN/DIN.py:26), the value-scaled continuous embedding (AFM/AFM.py:52-57, PNN/PNN.py:55-59),
 * FM first/second order (DeepFM/DeepFM.py:55-63), bi-interaction (NFM/NFM.py:53-57),
 * the sum over the embedding axis of ESMM (ESMM/ESMM.py:50) and the deep-input concat
 * (DeepFM/DeepFM.py:66).
 *
 *   Z[b,f,:]   = tables[row_offset[f] + idx[b,f], :]   (zero row if idx outside [0, V_f))
 *   Z[b,F+c,:] = cont_w[c,:] * x_cont[b,c]             (only when cont_w != NULL; N = F + C)
 *   first[b,n] = sum_e Z[b,n,e]                         (then dropout drop_first)
 *   s[b,e]     = sum_n Z[b,n,e]
 *   second[b,e]= 0.5 * (s^2 - sum_n Z^2)                (then dropout drop_second)
 * Outputs are optional (NULL = skip).  z_out row b starts at z_out + b*z_ld, holds N*E
 * values and, when cont_tail != 0, x_cont[b,0..C) right behind them.
 * E must be a power of two in [4, 256]; z_ld must be a multiple of 4.
 * oor_count (optional) is incremented once per out-of-range id.
 * ---------------------------------------------------------------------------------- */
int ctr_embed_fwd(const float* tables, const int64_t* row_offset, const int32_t* idx,
                  const float* x_cont, const float* cont_w,
                  int B, int F, int C, int E,
                  float* z_out, int64_t z_ld, int cont_tail,
                  float* first_out, int64_t first_ld,
                  float* s_out,
                  float* second_out, int64_t second_ld,
                  const ctr_dropout_t* drop_first, const ctr_dropout_t* drop_second,
                  int32_t* oor_count, void* stream);
/* ctr_embed_fwd that also raises the record amax_out (optional, CTR_AMAX_FLOATS floats zeroed by the caller) to the largest magnitude
 * it writes to z_out, continuous tail included: the a_amax word of the product that consumes z_out
 * (ctr_epilogue_t). */
int ctr_embed_fwd_amax(const float* tables, const int64_t* row_offset, const int32_t* idx,
                  const float* x_cont, const float* cont_w,
                  int B, int F, int C, int E,
                  float* z_out, int64_t z_ld, int cont_tail,
                  float* first_out, int64_t first_ld,
                  float* s_out,
                  float* second_out, int64_t second_ld,
                  const ctr_dropout_t* drop_first, const ctr_dropout_t* drop_second,
                  int32_t* oor_count, float* amax_out, void* stream);

/* Multi-hot fields ("gather and pool"; the sum combiner of tf.nn.embedding_lookup_sparse, which the reference's
 * one-id-per-field models are the one-member case of: DeepFM/DeepFM.py:43-53).  idx is a flat list of nnz ids and
 * bag_offsets a CSR array of B*F+1 int32: bag (b,f) holds idx[bag_offsets[b*F+f] .. bag_offsets[b*F+f+1]), all
 * ids of field f.
 *   Z[b,f,:] = sum over the bag's members j, in list order, of tables[row_offset[f] + idx[j], :]
 * (an empty bag gives a zero row; an id outside [0, V_f) adds nothing and bumps oor_count).  Continuous rows,
 * first / s / second, dropout and the x_cont tail are those of ctr_embed_fwd, computed over the pooled Z.
 * z_out is required here. */
int ctr_embed_bag_fwd(const float* tables, const int64_t* row_offset, const int32_t* idx,
                      const int32_t* bag_offsets, int64_t nnz,
                      const float* x_cont, const float* cont_w,
                      int B, int F, int C, int E,
                      float* z_out, int64_t z_ld, int cont_tail,
                      float* first_out, int64_t first_ld,
                      float* s_out,
                      float* second_out, int64_t second_ld,
                      const ctr_dropout_t* drop_first, const ctr_dropout_t* drop_second,
                      int32_t* oor_count, void* stream);

/* FM reductions over an already materialised Z (leading dimension z_ld, N rows of E per
 * sample): first / s / second exactly as in ctr_embed_fwd (DeepFM/DeepFM.py:55-63). */
int ctr_fm_fwd(const float* z, int64_t z_ld, int B, int N, int E,
               float* first_out, int64_t first_ld, float* s_out,
               float* second_out, int64_t second_ld,
               const ctr_dropout_t* drop_first, const ctr_dropout_t* drop_second, void* stream);

/* Reference-exact ("R-mode") embedding block, DeepFM/DeepFM.py:43-53 as written:
 *   Z[b, off_f + k, :] = tables[row_offset[f] + 1, :] if k == idx[b,f] else 0,  k in [0, V_f)
 * z_out row b has sum(V)*E values (+ the x_cont tail when cont_tail != 0).
 * pos_out (optional, [B,F] int32) receives off_f + idx[b,f], or -1 when out of range:
 * the row of Z that sample b activates for field f. */
int ctr_embed_onehot_fwd(const float* tables, const int64_t* row_offset, const int32_t* idx,
                         const float* x_cont, int B, int F, int C, int E,
                         float* z_out, int64_t z_ld, int cont_tail,
                         int32_t* pos_out, void* stream);

/* Backward of the front end with respect to Z (gradient of DeepFM/DeepFM.py:55-66):
 *   dZ[b,n,e] = gz[b,n,e] + g_first[b,n]*keep_first + g_second[b,e]*keep_second*(s[b,e] - Z[b,n,e])
 *               + g_sum[b,e]
 * Every g* term is optional.  z (the forward's Z, leading dimension z_ld) is needed only
 * with g_second; g_sum ([B,E] contiguous) is a gradient with respect to s = sum_n Z (PNN outer
 * product, PNN/PNN.py:83).  dz may alias gz (in-place).  N = rows of Z per sample. */
int ctr_embed_bwd_dz(const float* gz, int64_t gz_ld,
                     const float* g_first, int64_t g_first_ld,
                     const float* g_second, int64_t g_second_ld,
                     const float* s, const float* z, int64_t z_ld, const float* g_sum,
                     int B, int N, int E,
                     const ctr_dropout_t* drop_first, const ctr_dropout_t* drop_second,
                     float* dz, int64_t dz_ld, void* stream);
/* Sharded tables (no counterpart in the single-process reference; same arithmetic as ctr_embed_fwd /
 * ctr_embed_bwd_dz, DeepFM/DeepFM.py:42-63): the rows of field n were fetched by the field's owner rank and
 * arrive as one block of columns per owner.  Row n < F of sample b is the E floats at
 * blocks[fbase[n] + b * fstride[n]]; fbase / fstride are device arrays of F int64 (element offsets, multiples
 * of 4).  ctr_embed_assemble_fwd builds Z (+ continuous rows, FM sums, dropout) from them in one pass;
 * ctr_embed_bwd_dz_blocks writes dZ of those rows straight back into the same block layout (the buffer the
 * gradient exchange sends) and the other rows into dz. */
int ctr_embed_assemble_fwd(const float* blocks, const int64_t* fbase, const int64_t* fstride,
                           const float* x_cont, const float* cont_w, int B, int F, int C, int E,
                           float* z_out, int64_t z_ld, int cont_tail, float* first_out, int64_t first_ld,
                           float* s_out, float* second_out, int64_t second_ld,
                           const ctr_dropout_t* drop_first, const ctr_dropout_t* drop_second, void* stream);
int ctr_embed_bwd_dz_blocks(const float* gz, int64_t gz_ld, const float* g_first, int64_t g_first_ld,
                            const float* g_second, int64_t g_second_ld, const float* s, const float* z,
                            int64_t z_ld, const float* g_sum, int B, int N, int E,
                            const ctr_dropout_t* drop_first, const ctr_dropout_t* drop_second, float* dz,
                            int64_t dz_ld, float* blocks_out, const int64_t* fbase, const int64_t* fstride, int F,
                            void* stream);

/* ------------------------------------------------------------------------------------
 * Deterministic embedding backward: sort + segment-reduce + fused row update.
 * Replaces: the IndexedSlices gradient of tf.nn.embedding_lookup, its
 * unsorted_segment_sum de-duplication and the optimizer apply inside
 * optimizer.minimize (DeepFM/DeepFM.py:100; AFM/AFM.py:29,97; DIN/DIN.py:110).
 *
 * ctr_emb_sort builds one (key, src) pair per (b,f):
 *   S-mode (rmode == 0): key = row_offset[f] + idx[b,f],   src = b*N + f
 *   R-mode (rmode != 0): key = row_offset[f] + 1,          src = b*N + row_offset[f] + idx[b,f]
 * (out-of-range ids get key = total_rows and are ignored later), and sorts the pairs by
 * key with a stable LSD radix sort (pairs of equal valid key stay in sample order).  keys_out / src_out: [B*F] uint32.
 * ---------------------------------------------------------------------------------- */
size_t ctr_emb_sort_workspace_bytes(int64_t n_items);
/* Workspace with which ctr_emb_sort can take its field-segmented path (never less than
 * ctr_emb_sort_workspace_bytes(B*F)): the pairs are laid out field by field and every field is sorted on the
 * bits its own table needs (one pass for up to 511 rows, three for ten million), without scan launches; up to
 * B = 16384 a pass is one launch whose blocks exchange their digit counts through flagged words in the workspace
 * while it runs (calls that share a workspace must not overlap in time, as with any workspace).  Taken for 1024 <= B <= 131072
 * when the workspace is at least this large; otherwise the global sort runs.  Same result either way, except the
 * order of the out-of-range pairs among themselves (key = total_rows, behind everything: by sample with the
 * global sort, by field with this one; they are ignored later). */
size_t ctr_emb_sort_workspace_bytes_fields(int B, int F, int64_t total_rows, int rmode);
int ctr_emb_sort(const int32_t* idx, const int64_t* row_offset, int B, int F, int N,
                 int64_t total_rows, int rmode,
                 uint32_t* keys_out, uint32_t* src_out,
                 void* workspace, size_t workspace_bytes, void* stream);
/* ctr_emb_sort for multi-hot bags (same CSR arguments as ctr_embed_bag_fwd): one pair per member j of bag
 * (b,f), key = row_offset[f] + idx[j] (total_rows when out of range), src = b*N + f, sorted by key with the same
 * stable sort, so ctr_emb_sparse_apply(n_items = nnz) adds dZ[b,f,:] to every member row: the gradient of the
 * sum pooling.  keys_out / src_out: [nnz] uint32; workspace: ctr_emb_sort_workspace_bytes(nnz). */
int ctr_emb_bag_sort(const int32_t* idx, const int32_t* bag_offsets, int64_t nnz,
                     const int64_t* row_offset, int B, int F, int N, int64_t total_rows,
                     uint32_t* keys_out, uint32_t* src_out,
                     void* workspace, size_t workspace_bytes, void* stream);
/* ctr_emb_sort for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments and the same result,
 * bit for bit, as ctr_emb_sort, but the scratch is a 36 MiB array built into the library's device image (one copy
 * per GPU, held until the process ends) instead of a caller workspace; the call allocates and frees nothing.
 * Limit: B*F <= CTR_EMB_SORT_STATIC_MAX_ITEMS, larger inputs are rejected with CTR_EINVAL.  All calls on one GPU
 * share that GPU's array: calls that overlap in time on one GPU (different streams) race; calls on one stream,
 * or on different GPUs, do not.  The first call on a GPU resolves the array's address and must not come inside a
 * stream capture. */
#define CTR_EMB_SORT_STATIC_MAX_ITEMS 2097152
int ctr_emb_sort_static(const int32_t* idx, const int64_t* row_offset, int B, int F, int N,
                        int64_t total_rows, int rmode, uint32_t* keys_out, uint32_t* src_out, void* stream);
/* Plain stable radix sort of (key, value) uint32 pairs on keys < 2^key_bits. */
int ctr_sort_pairs(const uint32_t* keys_in, const uint32_t* vals_in, int64_t n, int key_bits,
                   uint32_t* keys_out, uint32_t* vals_out,
                   void* workspace, size_t workspace_bytes, void* stream);

#define CTR_OPT_GRAD_ONLY 0 /* grad_out[key,:] = summed gradient (dense-gradient path)  */
#define CTR_OPT_ADAM 1      /* row-wise Adam on rows that received a gradient           */
#define CTR_OPT_ADAGRAD 2   /* tf.train.AdagradOptimizer semantics (AFM/AFM.py:29)      */
#define CTR_OPT_SGD 3       /* tf.train.GradientDescentOptimizer (DIN/DIN.py:110)       */
#define CTR_OPT_FTRL 4      /* tf.train.FtrlOptimizer, lr_power -0.5 (Wide_Deep/Wide_Deep.py:107-108):
                             * slot0 = accumulator n (initial 0.1), slot1 = linear z (initial 0);
                             * n' = n + g^2; z += g - (sqrt(n') - sqrt(n))/lr * w;
                             * w = |z| <= l1 ? 0 : -(z - sign(z) l1) / (sqrt(n')/lr + 2 l2)        */

typedef struct {
  int kind;     /* CTR_OPT_*                                            */
  float lr;     /* Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t), computed by the host */
  float beta1;  /* Adam                                                 */
  float beta2;  /* Adam                                                 */
  float eps;    /* Adam (outside the sqrt, TF-1.x)                      */
  float l1;     /* FTRL l1_regularization_strength                      */
  float l2;     /* FTRL l2_regularization_strength                      */
  const float* lr_dev; /* optional device float read instead of lr when the kernel runs (NULL: use lr) */
} ctr_opt_t;

/* For every run of equal keys in the sorted pairs: g = sum (in sorted order, so bitwise
 * reproducible) of scale_j * dz[b_j*dz_ld + n_j*E : +E] with src_j = b_j*N + n_j and
 * scale_j = src_scale[src_j] (1 when src_scale == NULL), then apply `opt` to row `key`
 * of tables (slot0/slot1 = Adam m/v, slot0 = Adagrad accumulator) or, for
 * CTR_OPT_GRAD_ONLY, store g into slot0[key,:].  Keys >= total_rows are skipped.
 * Long runs (hot rows) are cut into 64-pair pieces summed by separate lane groups and
 * combined in piece order, so the time does not depend on the hottest row. */
size_t ctr_emb_sparse_apply_workspace_bytes(int64_t n_items, int E);
int ctr_emb_sparse_apply(const uint32_t* keys_sorted, const uint32_t* src_sorted, int64_t n_items,
                         int N, int E, int64_t total_rows,
                         const float* dz, int64_t dz_ld, const float* src_scale,
                         float* tables, float* slot0, float* slot1,
                         const ctr_opt_t* opt, void* workspace, size_t workspace_bytes, void* stream);
/* The same with the optimizer slots at a row pitch of their own: row r of slot0 / slot1 starts slot_ld floats after
 * row r - 1 (ctr_emb_sparse_apply: slot_ld = E).  For Adam's m and v kept side by side in one [rows, 2E] buffer
 * (slot1 = slot0 + E, slot_ld = 2E): at E = 16 a touched row's m and v are then ONE 128-byte line instead of two
 * half-used ones (tf.train.AdamOptimizer's slots, DeepFM/DeepFM.py:32,100, are separate variables; only their
 * values are part of the contract). */
int ctr_emb_sparse_apply_ld(const uint32_t* keys_sorted, const uint32_t* src_sorted, int64_t n_items,
                            int N, int E, int64_t total_rows,
                            const float* dz, int64_t dz_ld, const float* src_scale,
                            float* tables, float* slot0, float* slot1, int64_t slot_ld,
                            const ctr_opt_t* opt, void* workspace, size_t workspace_bytes, void* stream);

/* ctr_emb_sparse_apply_ld for a gradient whose rows lack the FM second-order term's product with the looked-up
 * row (ctr_fm_bwd_t with z == NULL): with g_second ([B, E] rows at pitch g_second_ld, 8-byte aligned, the
 * gradient with respect to the second-order term with its keep mask on) the gradient of table row k becomes
 *   sum_j dz[src_j] - (sum_j g_second[b_j, :]) * tables[k, :]        (j over the run of key k, b_j = src_j / N)
 * with tables[k] read before the update: gradient of DeepFM/DeepFM.py:59-63 with respect to the looked-up row,
 * d/dz 0.5 ((sum z)^2 - sum z^2) = s - z, whose -z part is the same row for every lookup of k.  Both sums keep
 * the fixed order of the plain form.  Rows of at most 32 floats; tables is required (read even for
 * CTR_OPT_GRAD_ONLY).  Workspace: ctr_emb_sparse_apply_fm_workspace_bytes. */
size_t ctr_emb_sparse_apply_fm_workspace_bytes(int64_t n_items, int E);
int ctr_emb_sparse_apply_fm(const uint32_t* keys_sorted, const uint32_t* src_sorted, int64_t n_items,
                            int N, int E, int64_t total_rows,
                            const float* dz, int64_t dz_ld, const float* g_second, int64_t g_second_ld,
                            float* tables, float* slot0, float* slot1, int64_t slot_ld,
                            const ctr_opt_t* opt, void* workspace, size_t workspace_bytes, void* stream);

/* tf.train.AdamOptimizer on an embedding variable (DeepFM/DeepFM.py:32,100) moves EVERY row each step: rows
 * without a gradient still decay m and v and take the step lr * m / (sqrt(v) + eps).  In two parts without a dense
 * gradient buffer: ctr_emb_sparse_apply (CTR_OPT_ADAM) on the rows that received a gradient, then
 * ctr_adam_untouched_rows on all others.  A byte per table row tells the second part what to do: bit 0 = updated by
 * the first part in this step (skip), bit 1 = the row's m or v may be nonzero.  ctr_mark_rows(value = 1) sets both bits
 * on the rows the sorted keys name before the pass; the pass clears bit 0 of every byte it reads, so the map is ready
 * for the next step (ctr_mark_rows(value = 0) clears bit 0 on the rows the keys name, for callers that mark without
 * running the pass).  A row
 * whose byte is 0 has no optimizer state: the formulas leave such a row exactly as it is (m = v = 0: no decay, no
 * step), so the pass does not even read it - on tables of tens of millions of rows, most of which never receive a
 * gradient, that is most of the work.  Start from a map of zeros together with zero m and v; a caller that fills m or
 * v by other means (a restored checkpoint) sets the bytes of those rows, or of all rows, to 2.  Values with m = v = +0
 * inside rows with state are skipped as well (same result).  tables, m, v: contiguous [rows, E], E % 4 == 0; the map
 * 16-byte aligned. */
int ctr_mark_rows(const uint32_t* keys_sorted, int64_t n_items, int64_t total_rows, uint8_t* touched, int value,
                  void* stream);
int ctr_adam_untouched_rows(float* tables, float* m, float* v, int64_t rows, int E, uint8_t* touched,
                            const ctr_opt_t* opt, void* stream);

/* ------------------------------------------------------------------------------------
 * Dense tower pieces.
 * Replaces: tf.matmul + bias + relu + tf.layers.dropout chains and their gradients
 * (DeepFM/DeepFM.py:65-88, DCN/DCN.py:67-89, NFM/NFM.py:59-81, PNN/PNN.py:98-117,
 * ESMM/ESMM.py:52-92, DIN/DIN.py:47-58).
 *
 * C[M,N] = epilogue( op(A)[M,K] * op(B)[K,N] )   on the fp32 MFMA (exact fp32 FMA chain)
 *   trans_a == 0: A is stored [M,K] (lda >= K);  != 0: stored [K,M] (lda >= M)
 *   trans_b == 0: B is stored [K,N] (ldb >= N);  != 0: stored [N,K] (ldb >= K)
 * epilogue, in this order: += bias[n] (optional); accumulate (C += result, when
 * accumulate != 0); relu (when relu != 0); dropout (optional); mask: where
 * mask_src[m*mask_ld+n] <= 0 the result is 0, elsewhere it is multiplied by mask_scale
 * (optional; this is the relu+dropout backward given the layer's stored output); fm_bwd (optional, last): the
 * product is the tower's gradient with respect to x0 = [Z | x_cont], and the FM terms of ctr_embed_bwd_dz are
 * added to its first n_rows*E columns on the way out (the same arithmetic in the same order, so the same bits
 * as the separate pass over gz):
 *   C[m, n*E + e] += g_first[m,n] + g_second[m,e]*(s[m,e] - z[m, n*E + e])
 * (gradient of DeepFM/DeepFM.py:55-66).  g_first / g_second are the gradients with respect to the first- and
 * second-order terms BEFORE their dropout (ctr_dropout_apply puts the keep masks on them); each is optional;
 * s ([M,E] contiguous) and z are needed with g_second and must be 16-byte aligned with z_ld % 4 == 0, E % 4 == 0.
 * z == NULL leaves the z term out (C += g_first + g_second * s): for the single-id lookup a row of Z IS the table
 * row, the same for every lookup of it, so the term is subtracted once per table row by the row update instead
 * (ctr_emb_sparse_apply_fm) and the product does not read Z.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const float* g_first;
  int64_t g_first_ld;
  const float* g_second;
  int64_t g_second_ld;
  const float* s;
  const float* z;
  int64_t z_ld;
  int n_rows; /* rows of Z per sample */
  int E;
} ctr_fm_bwd_t;

typedef struct {
  const float* bias;
  int relu; /* 0 none, 1 relu, 2 sigmoid (DIN attention MLP, DIN/DIN.py:175-176) */
  int accumulate;
  const ctr_dropout_t* dropout;
  const float* mask_src;
  int64_t mask_ld;
  float mask_scale;
  const ctr_fm_bwd_t* fm_bwd;
  /* Largest magnitudes (all optional).  Each is a device record of CTR_AMAX_FLOATS floats whose value is the
   * largest of its words 0, 32, 64, ... (concurrent raises are spread over 32 cache lines); zero the whole record to
   * reset it.
   *   a_amax / b_amax   max |A|, max |B| (or any upper bound within a factor of 2^8), written before this call by
   *                     whatever produced the operands (c_amax_out of an earlier product, ctr_embed_fwd's amax_out,
   *                     ctr_absmax).  With both present - and the split arithmetic selected - the product runs as
   *                     the two-term fp16 split (see ctr_set_gemm_f16_split): ~3e-7 relative instead of ~1e-7, at
   *                     half the matrix-core work.  *_floor: a lower bound the kernel applies to the word (an
   *                     operand view that includes a column of ones its producer did not write).
   *   c_amax_out        raised (integer atomic max on the floats' bits, never lowered: zero it first) to max |C|
   *                     over the elements this call stores. */
  const float* a_amax;
  const float* b_amax;
  float a_amax_floor;
  float b_amax_floor;
  float* c_amax_out;
} ctr_epilogue_t;

/* Arithmetic of the tiled GEMM (process-wide; default CTR_GEMM_SPLIT_BF16):
 *   CTR_GEMM_FP32_MFMA   v_mfma_f32_32x32x2_f32, an exact fp32 FMA chain.
 *   CTR_GEMM_SPLIT_BF16  every fp32 operand split exactly into three bf16 terms and the product
 *                        accumulated in fp32 from six bf16 MFMAs: fp32-level accuracy (~1e-7
 *                        relative, different rounding order) at 2.7x the matrix-core rate.
 *   CTR_GEMM_BF16        operands rounded to bf16 (nearest even), one bf16 MFMA per product, fp32
 *                        accumulate: a reduced-precision mode for the tile kernel (~4e-3 relative per
 *                        operand), opt-in only; the narrow-layer, chained and streamed kernels stay fp32. */
#define CTR_GEMM_FP32_MFMA 0
#define CTR_GEMM_SPLIT_BF16 1
#define CTR_GEMM_BF16 2
int ctr_set_gemm_mode(int mode);
int ctr_get_gemm_mode(void);
/* In CTR_GEMM_SPLIT_BF16 mode a product whose epilogue brings a_amax and b_amax runs as a two-term fp16 split
 * instead: each operand times a power of two that puts its largest magnitude just under 2^15 (exact), then
 * x = hi + lo in fp16 (22 significant bits) and three fp16 MFMAs (lo*hi + hi*lo + hi*hi) accumulated in fp32, the
 * powers of two taken out again in the epilogue.  on = 0 keeps every product on the bf16 split.  Returns the
 * previous setting (default 1). */
int ctr_set_gemm_f16_split(int on);
/* Raises the record `out` (CTR_AMAX_FLOATS floats, see ctr_epilogue_t) to max |X[r, c]| over a [rows, cols] view:
 * the a_amax / b_amax record of an operand whose producer does not raise it.  Zero the record first (it is only
 * ever raised). */
#define CTR_AMAX_FLOATS 1024
int ctr_absmax(const float* X, int64_t ld, int64_t rows, int cols, float* out, void* stream);
/* The same over n_tensors slices of one flat buffer: record i (at out + i * CTR_AMAX_FLOATS) is raised to
 * max |flat[offset[i] .. offset[i] + count[i])| (the weight matrices of a tower in the flat variable buffer).
 * offsets / counts: device int64[n_tensors]. */
int ctr_absmax_tensors(const float* flat, const int64_t* offsets, const int64_t* counts, int n_tensors,
                       float* out, void* stream);
/* Diagnostic: resident blocks per CU the runtime grants the 128x128 split tile kernel (designed for 2); < 0 on error. */
int ctr_gemm_split_blocks_per_cu(void);
size_t ctr_gemm_f32_workspace_bytes(int M, int N, int K);
int ctr_gemm_f32(int trans_a, int trans_b, int M, int N, int K,
                 const float* A, int64_t lda, const float* B, int64_t ldb,
                 float* C, int64_t ldc, const ctr_epilogue_t* epi,
                 void* workspace, size_t workspace_bytes, void* stream);

/* Weight and bias gradient of a dense layer y = x W + b (gradient of DeepFM/DeepFM.py:71,78):
 *   dW[M,N] = A^T B  (A = layer input stored [K,M], B = upstream gradient stored [K,N], K = batch),
 *   db[N]   = column sums of B (optional).  dW is written contiguously (leading dimension N).
 * Small outputs stream the operands straight into MFMA fragments (no LDS staging) with fixed-order
 * partial slabs; large ones go through ctr_gemm_f32's split-K path.  Bitwise reproducible. */
size_t ctr_dense_wgrad_workspace_bytes(int M, int N, int K);
int ctr_dense_wgrad(int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                    float* dW, float* db, void* workspace, size_t workspace_bytes, void* stream);
/* The same with the operands' largest magnitudes (ctr_epilogue_t.a_amax / b_amax; a_amax_floor as there): the
 * products that go through ctr_gemm_f32 then run as the two-term fp16 split; the streamed small-output kernel is
 * plain fp32 either way. */
int ctr_dense_wgrad_scaled(int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                           float* dW, float* db, const float* a_amax, float a_amax_floor, const float* b_amax,
                           void* workspace, size_t workspace_bytes, void* stream);

/* out[j] = sum_b X[b*ld + j] * (w ? w[b*w_ld + j / w_group] : 1),  j in [0, K).
 * Two fixed-order passes (bitwise reproducible).  Serves bias gradients (w == NULL),
 * output-unit weight gradients (w = dlogit, w_group = K) and the continuous-embedding
 * gradient (w = x_cont, w_group = E). */
size_t ctr_colsum_workspace_bytes(int B, int K);
int ctr_colsum(const float* X, int64_t ld, const float* w, int64_t w_ld, int w_group,
               int B, int K, float* out, int accumulate,
               void* workspace, size_t workspace_bytes, void* stream);

/* logit[b] = sum_k X[b*ld + k] * w[k] + bias[0]   (output unit, DeepFM/DeepFM.py:83-88) */
int ctr_rowdot_fwd(const float* X, int64_t ld, const float* w, const float* bias,
                   int B, int K, float* out, void* stream);
/* dX[b*ld + k] (+)= g[b] * w[k] */
int ctr_rowdot_bwd_x(const float* g, const float* w, int B, int K,
                     float* dX, int64_t ld, int accumulate, void* stream);
/* Whole backward of the output unit y = X w + b in one pass over X (DeepFM/DeepFM.py:82-88, same unit in
 * NFM/NFM.py:76-81, DCN/DCN.py:84-89):  dX[b,k] = dl[b] w[k], dw[k] = sum_b X[b,k] dl[b], db = sum_b dl[b].
 * Columns k >= mask_from of X are the output of the tower's last relu (+ dropout) layer: their dX is also
 * multiplied by mask_scale * [X[b,k] > 0] (mask_from = K or < 0: no masked range).  K <= 4096
 * (ctr_unit_bwd_supported); fixed summation order. */
int ctr_unit_bwd_supported(int K);
size_t ctr_unit_bwd_workspace_bytes(int B, int K);
int ctr_unit_bwd(int B, int K, const float* dl, const float* w, const float* X, int64_t ldx, float* dX, int64_t lddx,
                 int mask_from, float mask_scale, float* dw, float* db, void* workspace, size_t workspace_bytes,
                 void* stream);
/* The same, and dX's first keep_a_cols columns and the keep_b_cols after them leave multiplied by the keep masks
 * of two dropouts, element indices b*keep_a_cols + k and b*keep_b_cols + (k - keep_a_cols) (what ctr_dropout_apply2
 * would do to dX afterwards; same bits): DeepFM's unit sees [first | second | tower], and the gradients of the
 * two FM terms before their dropout (DeepFM/DeepFM.py:58,63) are what ctr_gemm_f32's fm_bwd epilogue takes.
 * masked_amax_out (optional, CTR_AMAX_FLOATS floats): the largest |dX| over the masked columns is raised there -
 * the tower's incoming gradient, whose maximum the two-term fp16 product that consumes it wants (ctr_epilogue_t). */
int ctr_unit_bwd_keep(int B, int K, const float* dl, const float* w, const float* X, int64_t ldx, float* dX, int64_t lddx,
                      int mask_from, float mask_scale, int keep_a_cols, const ctr_dropout_t* keep_a, int keep_b_cols,
                      const ctr_dropout_t* keep_b, float* masked_amax_out, float* dw, float* db, void* workspace,
                      size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Loss head: activation + loss + gradient with respect to the pre-activation.
 * Replaces: tf.sigmoid + tf.losses.log_loss (DeepFM/DeepFM.py:88-91) and
 * tf.nn.sigmoid_cross_entropy_with_logits (PNN/PNN.py:120, ESMM/ESMM.py:95-96, DIN/DIN.py:101).
 *   CTR_LOSS_SIGMOID_LOGLOSS   y_out = sigmoid(x);  loss = log_loss(y, y_out)  (eps 1e-7)
 *   CTR_LOSS_IDENTITY_LOGLOSS  y_out = x;           loss = log_loss(y, y_out)  (AFM/AFM.py:85-88)
 *   CTR_LOSS_SIGMOID_SCE_PROB  y_out = sigmoid(x);  loss = sce(labels=y, logits=y_out)
 *   CTR_LOSS_SCE_LOGIT         y_out = sigmoid(x);  loss = sce(labels=y, logits=x)
 *   CTR_LOSS_PROB_SCE          y_out = x (already a probability); loss = sce(labels=y, logits=x)
 * loss_out[0] = mean over B (fixed-order reduction); dx[b] = grad_scale * d loss / d x[b]
 * (grad_scale = 1, or 1/world_size when B is one rank's share of a data-parallel batch).
 * y_out, dx, loss_out optional; y may be NULL when only y_out is wanted (inference).
 * ---------------------------------------------------------------------------------- */
#define CTR_LOSS_SIGMOID_LOGLOSS 0
#define CTR_LOSS_IDENTITY_LOGLOSS 1
#define CTR_LOSS_SIGMOID_SCE_PROB 2
#define CTR_LOSS_SCE_LOGIT 3
#define CTR_LOSS_PROB_SCE 4
size_t ctr_loss_head_workspace_bytes(int B);
int ctr_loss_head(int kind, const float* x, const float* y, int B, float grad_scale,
                  float* y_out, float* loss_out, float* dx,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ESMM head (ESMM/ESMM.py:95-98): y_out = sigmoid(ctr) * sigmoid(cvr);
 * loss = mean sigmoid_cross_entropy_with_logits(labels=y, logits=y_out) (y_out is already a
 * probability; reproduced as written).  dctr/dcvr = d loss / d logits.  Workspace as ctr_loss_head. */
int ctr_esmm_head(const float* ctr_logit, const float* cvr_logit, const float* y, int B, float grad_scale,
                  float* y_out, float* loss_out, float* dctr, float* dcvr,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Dense optimizers over one flat parameter buffer (multi-tensor in one launch).
 * Replaces: tf.train.AdamOptimizer / AdagradOptimizer / GradientDescentOptimizer apply
 * for dense variables (DeepFM/DeepFM.py:93-100).
 * ---------------------------------------------------------------------------------- */
int ctr_dense_update(float* param, const float* grad, float* slot0, float* slot1,
                     int64_t n, const ctr_opt_t* opt, void* stream);

/* Writes the two device words a captured training step reads instead of by-value scalars
 * (ctr_dropout_t.seed_add, ctr_opt_t.lr_dev), in stream order; either pointer may be NULL.
 * Replaces: the feed of global_step into tf.train.exponential_decay and Adam's beta powers
 * (DeepFM/DeepFM.py:94-100), which TF re-evaluates inside every session.run. */
int ctr_set_step_words(uint64_t* seed_add, uint64_t seed_add_value, float* lr_dev, float lr_value,
                       void* stream);

/* ------------------------------------------------------------------------------------
 * DCN cross network, all layers fused (DCN/DCN.py:56-65 without the [B,D,D] product):
 *   x_{l+1} = x0 * (x_l . w_l) + b_l + x_l,   l = 0..L-1,   1 <= L <= 8,  D <= 8192
 * w, b, dw, db: L rows of D values, row stride wld (>= D).  Every x_l lies in span{x0, prefix sums of b}, so the kernels
 * carry only per-sample scalars: aux_out[b, 0..L) = x0 . w_l and aux_out[b, L..2L) = c_{l+1}
 * with x_l = x0*c_l + sum_{l'<l} b_l'.  aux_out ([B, 2L]) is what the backward needs.
 * Backward: given g = dLoss/dx_L, produces dx0 (+= when accumulate), dw[L,D], db[L,D].
 * ---------------------------------------------------------------------------------- */
size_t ctr_cross_fwd_workspace_bytes(int D, int L);
int ctr_cross_fwd(const float* x0, int64_t x0_ld, const float* w, const float* b, int64_t wld,
                  int B, int D, int L, float* xL, int64_t xL_ld, float* aux_out,
                  void* workspace, size_t workspace_bytes, void* stream);
size_t ctr_cross_bwd_workspace_bytes(int B, int D, int L);
int ctr_cross_bwd(const float* x0, int64_t x0_ld, const float* w, const float* b, int64_t wld,
                  const float* aux, const float* g, int64_t g_ld,
                  int B, int D, int L,
                  float* dx0, int64_t dx0_ld, int accumulate, float* dw, float* db,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * PNN product layer pieces (PNN/PNN.py:61-96).  lz and the outer-product contraction are plain
 * GEMMs (ctr_gemm_f32); these are the parts that are not.
 *   inner: u[b,p,e] = sum_n Z[b,n,e]*w[p,n];  usq_out[b,p] = sum_e u^2.  The reference's
 *          lp[p] = ||u_p||_F over the WHOLE batch is sqrt(column sum of usq) (ctr_colsum + ctr_vec_op).
 *   inner backward: du[b,p,e] = coef * u[b,p,e] with coef = coef[p] (coef_ld == 0) or
 *          coef[b*coef_ld + p];  dz += du . w;  dw_out[p,n] = sum_{b,e} du * Z.
 *   outer: ss_out[b, e*E+f] = s[b,e]*s[b,f];  ds_out[b,e] = sum_f (dss[b,e,f]+dss[b,f,e])*s[b,f].
 * E must be a power of two in [4,64]; the backward needs P <= 64 and N <= 128.
 * ---------------------------------------------------------------------------------- */
int ctr_pnn_inner_fwd(const float* z, int64_t z_ld, const float* w, int B, int N, int E, int P,
                      float* usq_out, void* stream);
size_t ctr_pnn_inner_bwd_workspace_bytes(int B, int N, int P);
int ctr_pnn_inner_bwd(const float* z, int64_t z_ld, const float* w, const float* coef, int64_t coef_ld,
                      int B, int N, int E, int P, float* dz, int64_t dz_ld, float* dw_out,
                      void* workspace, size_t workspace_bytes, void* stream);
int ctr_outer_fwd(const float* s, int B, int E, float* ss_out, void* stream);
int ctr_outer_bwd(const float* dss, const float* s, int B, int E, float* ds_out, void* stream);
/* x[r,c] = dropout(relu?(x[r,c] + bias1[c] + bias2[c])) in place (PNN/PNN.py:95-96). */
int ctr_bias_act(float* x, int64_t ld, int rows, int cols, const float* bias1, const float* bias2,
                 int relu, const ctr_dropout_t* drop, void* stream);
/* kind 0: out = sqrt(a);  kind 1: out = b != 0 ? a/b : 0   (n-element vectors) */
int ctr_vec_op(int kind, const float* a, const float* b, float* out, int n, void* stream);

/* ------------------------------------------------------------------------------------
 * AFM pair layer with the reference's attention (AFM/AFM.py:60-85): the softmax is taken over
 * a size-1 axis, so all attention weights are exactly 1 and the attention variables receive
 * zero gradient;  x[b,e] = sum_{i<j} m_ij * Z[b,i,e]*Z[b,j,e]  with m_ij the dropout multiplier
 * of pair k = lexicographic index of (i,j) (dropout element b*P + k, P = N(N-1)/2).
 * Backward: dz[b,i,e] = dx[b,e] * sum_{j != i} m_ij * Z[b,j,e].
 * ---------------------------------------------------------------------------------- */
int ctr_afm_pair_fwd(const float* z, int64_t z_ld, int B, int N, int E,
                     const ctr_dropout_t* pair_drop, float* x_out, void* stream);
int ctr_afm_pair_bwd(const float* z, int64_t z_ld, const float* dx, int B, int N, int E,
                     const ctr_dropout_t* pair_drop, float* dz, int64_t dz_ld, void* stream);

/* ------------------------------------------------------------------------------------
 * AFM pair layer with the attention softmax taken ACROSS the P = N(N-1)/2 pairs (the network
 * AFM/AFM.py:60-85 builds, with the softmax axis of AFM/AFM.py:79 moved from the size-1 axis to
 * the pair axis):
 *   p_k = Z_i * Z_j;  e_k = h . relu(p_k W + b);  a = softmax_k(e);  x = sum_k a_k m_k p_k
 * W[E,A] row-major, b[A], h[A], A <= 32; m_k = dropout multiplier of element b*P + k.
 * Backward writes dz and the three parameter gradients (overwritten, fixed summation order).
 * ---------------------------------------------------------------------------------- */
int ctr_afm_attn_fwd(const float* z, int64_t z_ld, int B, int N, int E, int A, const float* W,
                     const float* b, const float* h, const ctr_dropout_t* pair_drop, float* x_out,
                     void* stream);
size_t ctr_afm_attn_bwd_workspace_bytes(int B, int E, int A);
int ctr_afm_attn_bwd(const float* z, int64_t z_ld, const float* dx, int B, int N, int E, int A,
                     const float* W, const float* b, const float* h, const ctr_dropout_t* pair_drop,
                     float* dz, int64_t dz_ld, float* dW, float* db, float* dh, void* workspace,
                     size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Two or three consecutive narrow tower layers in one launch (DeepFM/DeepFM.py:65-80 for the layers
 * after the first):  h_i = dropout_i(relu(h_{i-1} W_i + b_i)),  h_0 = x [M, K0].
 * Every h_i is written once (the backward reads them) and never read back: a layer's output stays
 * in the registers that are the next layer's matrix-core operands.  Dropout element index of h_i is
 * m * N_i + n, as in ctr_gemm_f32.  Covered widths: ctr_mlp_chain_supported (K0 % 16 == 0, K0 <= 256,
 * (N1, N2, N3) = (64, 32, 16 | 0)); N3 = 0 chains two layers.
 * Rows of x and of every output but the last must be 16-byte aligned (the last may be any column
 * range of a wider buffer).
 * ---------------------------------------------------------------------------------- */
int ctr_mlp_chain_supported(int K0, int N1, int N2, int N3);
int ctr_mlp_chain_fwd(int M, int K0, const float* X, int64_t ldx,
                      int N1, const float* W1, const float* b1, const ctr_dropout_t* drop1, float* H1, int64_t ld1,
                      int N2, const float* W2, const float* b2, const ctr_dropout_t* drop2, float* H2, int64_t ld2,
                      int N3, const float* W3, const float* b3, const ctr_dropout_t* drop3, float* H3, int64_t ld3,
                      void* stream);

/* Backward of that tail with respect to the activations, two layers per launch (the gradient walks
 * in -> mid -> out; widths (32, 64, 128) or (16, 32, 64), ctr_mlp_chain_bwd_supported):
 *   Gmid = (Gin Wb^T) * [Hmid > 0] * scale_mid,  Wb [mid, in]  = weights of the layer mid -> in
 *   Gout = (Gmid Wa^T) * [Hout > 0] * scale_out, Wa [out, mid] = weights of the layer out -> mid
 * Gin is the gradient of the last layer's output with its own relu/dropout mask already applied.
 * Rows of Hmid, Hout, Gmid, Gout must be 16-byte aligned; Gin may be any column range of a wider buffer. */
/* ctr_mlp_chain_bwd and the two layers' weight / bias gradients in one pass over the activations (same widths):
 * Gout as there (the mid gradient is not written); dWa [out,mid] / dba [mid] = gradients of the layer out -> mid
 * (Hout^T g_mid, column sums of g_mid), dWb [mid,in] / dbb [in] those of the layer mid -> in (Hmid^T Gin, column sums
 * of Gin).  Contiguous outputs, overwritten; fixed summation order. */
int ctr_tail_bwd_fused_supported(int in, int mid, int out);
size_t ctr_tail_bwd_fused_workspace_bytes(int M, int in, int mid, int out);
int ctr_tail_bwd_fused(int M, int in, const float* Gin, int64_t ldgi, int mid, const float* Wb,
                       const float* Hmid, int64_t ldhm, float scale_mid, int out, const float* Wa,
                       const float* Hout, int64_t ldho, float scale_out, float* Gout, int64_t ldgo,
                       float* dWa, float* dba, float* dWb, float* dbb,
                       void* workspace, size_t workspace_bytes, void* stream);
/* Weight and bias gradients of the last two narrow tower layers in one launch (gradient of
 * DeepFM/DeepFM.py:71,78 for those layers):  dWa = Xa^T Ga [ka,na], dba = column sums of Ga, dWb = Xb^T Gb [kb,nb],
 * dbb = column sums of Gb; all operands have M rows; dW* / db* are contiguous and overwritten.  Covered widths:
 * ctr_tail_wgrad2_supported.  Xa, Ga, Xb need 16-byte aligned rows; Gb may sit at any column of a wider buffer. */
int ctr_tail_wgrad2_supported(int ka, int na, int kb, int nb);
size_t ctr_tail_wgrad2_workspace_bytes(int M, int ka, int na, int kb, int nb);
int ctr_tail_wgrad2(int M, int ka, const float* Xa, int64_t ldxa, int na, const float* Ga, int64_t ldga,
                    float* dWa, float* dba, int kb, const float* Xb, int64_t ldxb, int nb, const float* Gb, int64_t ldgb,
                    float* dWb, float* dbb, void* workspace, size_t workspace_bytes, void* stream);
int ctr_mlp_chain_bwd_supported(int in, int mid, int out);
int ctr_mlp_chain_bwd(int M, int in, const float* Gin, int64_t ldgi, int mid, const float* Wb,
                      const float* Hmid, int64_t ldhm, float scale_mid, float* Gmid, int64_t ldgm, int out,
                      const float* Wa, const float* Hout, int64_t ldho, float scale_out, float* Gout,
                      int64_t ldgo, float* gout_amax /* optional largest-magnitude record of Gout (ctr_epilogue_t) */,
                      void* stream);

/* ------------------------------------------------------------------------------------
 * Evaluation metrics on the device (the reference round-trips predictions to the host:
 * DeepFM/main.py:105-126, DIN/main.py:12-61).  auc_out[0] = area under the ROC curve of
 * (score, label > 0.5) with average ranks for tied scores (NaN when one class is absent);
 * logloss_out[0] = mean of -y log(p + 1e-7) - (1-y) log(1 - p + 1e-7) (tf.losses.log_loss).
 * Either output may be NULL.  Sums are double precision in a fixed order.
 * ---------------------------------------------------------------------------------- */
size_t ctr_auc_workspace_bytes(int64_t n);
int ctr_auc(const float* score, const float* label, int64_t n, float* auc_out, float* logloss_out,
            void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Row-wise sharding of the embedding tables over W ranks (new design: the reference is
 * single-process; the single-device numerics it must reproduce are DeepFM/DeepFM.py:43-53 in
 * the standard-lookup form of Wide_Deep/Wide_Deep.py:51).  Row r of field f lives on rank
 * (r + f) mod W at local row r / W of that rank's shard of field f (the rotation by f keeps the
 * hottest row of every field, often the same small id, from landing on one rank).
 *   ctr_shard_route   per lookup p = b*F + f: dest[p] = owner rank, lkey[p] = row inside the owner's
 *                     concatenated shard (local_offset[W,F] int32 = first local row of each field on
 *                     each rank); an id outside [0, V_f) gets key -1 (zero row) on rank f mod W and is
 *                     added to oor_count (optional device int32) on the rank whose batch holds it.
 *   ctr_shard_bucket  fixed-capacity send buckets from the lookups stably sorted by destination
 *                     (dest_sorted, perm): lookup number r of destination d fills slot d*cap + r with
 *                     slot_key = lkey[perm] and slot_perm = perm; unfilled slots hold -1 / 0xFFFFFFFF;
 *                     lookups beyond a bucket's capacity are dropped and counted in overflow_count
 *                     (a device int32 the caller zeroes and checks); with z_zero (optional, the [B, z_ld]
 *                     activation buffer the rows will be scattered into, F rows of E floats per sample)
 *                     the row of a dropped lookup is zeroed there.  Every split of the exchange is
 *                     then `cap` entries: no count exchange and no host synchronisation.  W <= 64.
 *   ctr_rows_gather   out[s, 0:E] = src[b, f*E:(f+1)*E], (b, f) = divmod(perm[s], F): packs rows of the
 *                     strided [B, ld] layout in send order (32-bit words of any type; E = 1 packs ids);
 *                     perm[s] = 0xFFFFFFFF (an empty slot) gives a zero row.
 *   ctr_rows_scatter  the inverse: dst[b, f*E:(f+1)*E] = rows[s, 0:E]; empty slots are skipped.
 * ---------------------------------------------------------------------------------- */
int ctr_shard_route(const int32_t* idx, const int64_t* row_offset, const int32_t* local_offset, int B, int F,
                    int W, uint32_t* dest_out, int32_t* lkey_out, int32_t* oor_count, void* stream);
int ctr_shard_bucket(const uint32_t* dest_sorted, const uint32_t* perm, const int32_t* lkey, int64_t n, int W,
                     int64_t cap, int32_t* slot_key_out, uint32_t* slot_perm_out, int32_t* overflow_count,
                     float* z_zero, int64_t z_ld, int F, int E, void* stream);
int ctr_rows_gather(const void* src, int64_t ld, const uint32_t* perm, int64_t n, int F, int E, void* out,
                    void* stream);
int ctr_rows_scatter(const void* rows, const uint32_t* perm, int64_t n, int F, int E, void* dst, int64_t ld,
                     void* stream);

/* ------------------------------------------------------------------------------------
 * DIN pieces (DIN/DIN.py:17-71,149-196).  The dense layers of the attention MLP and of the
 * FCN are ctr_gemm_f32 calls over all B*T (resp. B) rows; these are the rest.
 *   ctr_din_build_idx   out[k] = (ids[k], cate_list[ids[k]]) as int32 pairs: with the item and
 *                       category tables concatenated as two "fields" of width H/2, ctr_embed_fwd
 *                       on these pairs yields [item_emb_w[id], cate_emb_w[cate_list[id]]]
 *                       (DIN/DIN.py:24-42); an id outside [0, item_count) gives (-1,-1) = zero row.
 *   ctr_din_concat_*    row (b,t) = [q_b, k_bt, q_b - k_bt, q_b * k_bt]  (DIN/DIN.py:171-174)
 *   ctr_din_attn_pool_* att = softmax(where(t < sl[b], score, -2^32+1) / sqrt(H)) over t;
 *                       out[b,:] = sum_t att[b,t] K[b,t,:]                (DIN/DIN.py:179-194)
 *   ctr_bn_infer        tf.layers.batch_normalization with training=False (DIN/DIN.py:45,53)
 *   ctr_dice_*          dice() with inference-mode statistics             (DIN/DIN.py:149-166)
 * ---------------------------------------------------------------------------------- */
int ctr_din_build_idx(const int32_t* ids, const int32_t* cate_list, int64_t n, int item_count,
                      int32_t* out, void* stream);
int ctr_din_concat_fwd(const float* q, const float* k, int B, int T, int H, float* out, void* stream);
int ctr_din_concat_bwd(const float* d_all, const float* q, const float* k, int B, int T, int H,
                       float* dq, int accumulate_q, float* dk, int accumulate_k, void* stream);
/* The attention scoring network of DIN/DIN.py:169-185 in one launch, [B*T, 4H] never formed:
 *   din_all = [q, k, q - k, q*k];  d1 = sigmoid(din_all W1 + b1);  d2 = sigmoid(d1 W2 + b2);  score = d2 w3 + b3
 * q: [B,H] (the candidate item), k: [B,T,H] (the behaviour sequence), W1: [4H,80], W2: [80,40], w3: [40].
 * d1_out [B*T,80] / d2_out [B*T,40] (optional, kept for the backward), score_out [B*T] (unmasked, unscaled: feed
 * ctr_din_attn_pool_fwd).  H in {16, 32, 64}; ctr_din_attn_mlp_supported says whether all three calls below take
 * a shape (the weight-gradient call holds H <= 32).
 * _bwd: from dscore (gradient of score) and the stored d1, d2: g1_out [B*T,80] and g2_out [B*T,40] (gradients at
 * the two pre-activations), dq_out [B,H] (overwritten) and dk [B,T,H] (added to its content when accumulate_dk).
 * _wgrad: every parameter gradient of the network (the tf.layers.dense kernels and biases f1_att, f2_att, f3_att),
 * summed over rows in a fixed order; workspace from ctr_din_attn_mlp_wgrad_workspace_bytes(B, H). */
int ctr_din_attn_mlp_supported(int H, int n1, int n2);
int ctr_din_attn_mlp_fwd(const float* q, const float* k, int B, int T, int H, const float* W1, const float* b1,
                         const float* W2, const float* b2, const float* w3, const float* b3,
                         float* d1_out, float* d2_out, float* score_out, void* stream);
int ctr_din_attn_mlp_bwd(const float* q, const float* k, int B, int T, int H, const float* W1, const float* W2,
                         const float* w3, const float* d1, const float* d2, const float* dscore,
                         float* g1_out, float* g2_out, float* dq_out, float* dk, int accumulate_dk, void* stream);
size_t ctr_din_attn_mlp_wgrad_workspace_bytes(int B, int H);
int ctr_din_attn_mlp_wgrad(const float* q, const float* k, int B, int T, int H, const float* d1, const float* d2,
                           const float* dscore, const float* g1, const float* g2,
                           float* dW1, float* db1, float* dW2, float* db2, float* dw3, float* db3,
                           void* workspace, size_t workspace_bytes, void* stream);
/* The same three calls with the sequence lengths sl [B] (NULL: as above).  The softmax that follows masks the positions
 * from a sample's length on (DIN/DIN.py:186-190): their scores are never read and their dscore is 0, so every
 * contribution of such a position to dq, dk and the parameter gradients is an exact zero.  With sl the kernels skip
 * the 16-row strips that hold no valid position: score_out, d1_out, d2_out, g1_out and g2_out are left unwritten there
 * (pass the same sl to all three calls), dk keeps its content there; everything else has the same value.  At lengths
 * uniform in [1, T] this halves the work of the three launches. */
int ctr_din_attn_mlp_fwd_sl(const float* q, const float* k, const int32_t* sl, int B, int T, int H, const float* W1,
                            const float* b1, const float* W2, const float* b2, const float* w3, const float* b3,
                            float* d1_out, float* d2_out, float* score_out, void* stream);
int ctr_din_attn_mlp_bwd_sl(const float* q, const float* k, const int32_t* sl, int B, int T, int H, const float* W1,
                            const float* W2, const float* w3, const float* d1, const float* d2, const float* dscore,
                            float* g1_out, float* g2_out, float* dq_out, float* dk, int accumulate_dk, void* stream);
int ctr_din_attn_mlp_wgrad_sl(const float* q, const float* k, const int32_t* sl, int B, int T, int H, const float* d1,
                              const float* d2, const float* dscore, const float* g1, const float* g2,
                              float* dW1, float* db1, float* dW2, float* db2, float* dw3, float* db3,
                              void* workspace, size_t workspace_bytes, void* stream);
int ctr_din_attn_pool_fwd(const float* score, const int32_t* sl, const float* K, int B, int T, int H,
                          float* att_out, float* out, void* stream);
int ctr_din_attn_pool_bwd(const float* dout, const float* att, const int32_t* sl, const float* K,
                          int B, int T, int H, float* dscore, float* dK, int accumulate_k, void* stream);
int ctr_bn_infer(const float* x, int64_t x_ld, int rows, int cols,
                 const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                 float* y, int64_t y_ld, float* xhat, int64_t xhat_ld, void* stream);
int ctr_dice_fwd(const float* x, int64_t x_ld, int rows, int cols, const float* alpha, float eps,
                 float* y, int64_t y_ld, void* stream);
int ctr_dice_bwd(const float* x, int64_t x_ld, int rows, int cols, const float* alpha, float eps,
                 const float* dy, int64_t dy_ld, float* dx, int64_t dx_ld,
                 float* dalpha_src, int64_t dalpha_ld, void* stream);
/* Backward of ctr_bn_infer and of ctr_dice_fwd together with the column sums their parameters need, one launch
 * each (a block per 16 columns walks all rows; fixed summation order): for the [batch, <= a few hundred] tensors
 * after DIN's attention (DIN/DIN.py:45,53-57), where an elementwise pass plus two-stage column sums are five
 * launches.  ctr_bn_infer_bwd: dx = gamma * dy * rsqrt(var + eps), dgamma = sum_r dy * xhat, dbeta = sum_r dy
 * (dbeta, var optional).  ctr_dice_bwd_sum: dx as ctr_dice_bwd, dalpha = sum_r dy * (1 - p) * x. */
int ctr_bn_infer_bwd(const float* dy, int64_t dy_ld, const float* xhat, int64_t xhat_ld, int rows, int cols,
                     const float* gamma, const float* var, float eps, float* dx, int64_t dx_ld,
                     float* dgamma, float* dbeta, void* stream);
int ctr_dice_bwd_sum(const float* x, int64_t x_ld, int rows, int cols, const float* alpha, float eps,
                     const float* dy, int64_t dy_ld, float* dx, int64_t dx_ld, float* dalpha, void* stream);
/* g[r,c] *= y[r,c]*(1-y[r,c]): backward of a sigmoid layer given its output (DIN/DIN.py:175-176). */
int ctr_dsigmoid_apply(float* g, int64_t g_ld, const float* y, int64_t y_ld, int rows, int cols, void* stream);
/* out_mean[0] = mean(a[i] - b[i] > 0)   (mf_auc, DIN/DIN.py:92) */
int ctr_mean_positive_diff(const float* a, const float* b, int n, float* out_mean, void* stream);

/* g[r,c] = act[r,c] > 0 ? g[r,c]*scale : 0 — gradient of relu followed by dropout through a
 * layer whose stored output is `act` (DeepFM/DeepFM.py:71-72). */
int ctr_relu_mask_apply(float* g, int64_t g_ld, const float* act, int64_t act_ld,
                        int rows, int cols, float scale, void* stream);

/* Small strided 2-D helpers (concat/slice plumbing of the graphs). */
int ctr_copy2d(const float* src, int64_t src_ld, float* dst, int64_t dst_ld,
               int rows, int cols, int accumulate, void* stream);
/* out[r,c] = a[r,c] * b[r,c] on strided 2-D views: the GMF product of NCF (NCF/NCF.py:66) and its gradient */
int ctr_mul2d(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo, int rows, int cols,
              void* stream);
int ctr_dropout_apply(float* x, int64_t ld, int rows, int cols, const ctr_dropout_t* drop,
                      void* stream);
/* Two neighbouring column blocks of one buffer, each under its own dropout, in one launch: the keep masks of the
 * first- and second-order FM terms (DeepFM/DeepFM.py:58,63) put onto their gradients [g_first | g_second] before
 * ctr_gemm_f32's fm_bwd epilogue consumes them:
 *   x[r, c] *= keep_a(r*cols_a + c) for c < cols_a;  x[r, cols_a + c] *= keep_b(r*cols_b + c) for c < cols_b
 * (the element indices ctr_embed_fwd / ctr_embed_bwd_dz use).  Either descriptor may be NULL (block untouched). */
int ctr_dropout_apply2(float* x, int64_t ld, int rows, int cols_a, const ctr_dropout_t* drop_a,
                       int cols_b, const ctr_dropout_t* drop_b, void* stream);

/* ------------------------------------------------------------------------------------
 * Collectives of the sharded tables and the data-parallel towers, one communicator per process and GPU, over
 * RCCL (xGMI inside a node).  The reference is one process (DeepFM/main.py:69 opens a single tf.Session) and has
 * no counterpart; these are what a host that binds only this header needs for the multi-GPU path:
 *   - ids out / rows back / gradient rows out of the table- and row-sharded lookups: ctr_all2all (equal pieces,
 *     sizes known on the host: no count exchange) or ctr_all2allv;
 *   - the towers' flat gradient buffer: ctr_allreduce_sum_f32 (the loss gradient is pre-scaled by 1 / world).
 * RCCL is loaded at the first call; every call enqueues on `stream` and returns.  Bootstrap: rank 0 calls
 * ctr_comm_unique_id and hands the CTR_COMM_ID_BYTES bytes to every rank by any means (file, socket, MPI); every
 * rank then calls ctr_comm_init with its GPU current.  Offsets and sizes are bytes; piece p goes to / comes from
 * rank p.
 * ---------------------------------------------------------------------------------- */
#define CTR_COMM_ID_BYTES 128
typedef struct ctr_comm ctr_comm_t;
int ctr_comm_unique_id(void* id_out);
int ctr_comm_init(const void* id, int world, int rank, ctr_comm_t** comm_out);
int ctr_comm_destroy(ctr_comm_t* comm);
int ctr_comm_world(const ctr_comm_t* comm);
int ctr_comm_rank(const ctr_comm_t* comm);
int ctr_allreduce_sum_f32(ctr_comm_t* comm, float* buf, int64_t n, void* stream);
int ctr_all2all(ctr_comm_t* comm, const void* send, void* recv, int64_t bytes_per_peer, void* stream);
int ctr_all2allv(ctr_comm_t* comm, const void* send, const int64_t* send_offset, const int64_t* send_bytes,
                 void* recv, const int64_t* recv_offset, const int64_t* recv_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Input text of the Criteo display-advertising sets -> the arrays a step is fed with (host code, no GPU).
 * Stands before the path where the reference's loaders stand (pandas.read_csv + the feed conversion,
 * DeepFM/main.py:13-43, DeepFM/DeepFM.py:37-39); the reference reads its own two-column CSV only.
 * A line is  label \t I1..I13 \t C1..C26 : 0/1, thirteen integers, twenty-six hexadecimal tokens of up to eight
 * digits; any field may be empty (dense 0, token value 0).  ids[r][f] = value(C_f) mod cards[f] (the token's own
 * 32 bits as its hash), dense[r][j] = I_j, or log(1 + max(I_j, 0)) with log_dense != 0.  The text is cut into
 * n_threads pieces at line boundaries and parsed in parallel.  At most max_rows samples are written (label
 * [max_rows], dense [max_rows, 13], ids [max_rows, 26], host memory); *rows_out = samples written.
 * ctr_criteo_tsv_count: the number of non-empty lines.
 * ---------------------------------------------------------------------------------- */
int64_t ctr_criteo_tsv_count(const char* text, int64_t len);
int ctr_criteo_tsv_parse(const char* text, int64_t len, const int64_t* cards, int log_dense, int n_threads,
                         int64_t max_rows, float* label, float* dense, int32_t* ids, int64_t* rows_out);

#ifdef __cplusplus
}
#endif
#endif /* CTRHIP_H */
