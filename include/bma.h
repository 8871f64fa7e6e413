/*
 * bma.h -- C ABI of libbma_hip.so: the per-step kernels of the joint GCG+PGD
 * attack loop, hand-written for gfx950 (MI355X).
 *
 * Drop-in boundary.  The reference is pure Python and has no FFI;
 * each entry point below replaces the PyTorch op sequence cited next to it
 * (file:line into /root/reference/bimodalattack/bimodal_attack.py).  A host
 * binds these with ctypes (INTEGRATION.md shows the stub) and passes raw device
 * pointers plus the HIP stream to launch on -- no torch types cross this line.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - `stream` is a hipStream_t (NULL = the default stream); every call only
 *     enqueues work on it and returns -- no allocation, no synchronisation, so
 *     calls can be captured into a hipGraph;
 *   - the caller owns every buffer, including scratch (`ws`), whose size the
 *     matching *_ws_bytes() function returns;
 *   - return value: 0 on success, a negative BMA_E* code otherwise; nothing is
 *     launched when a negative code is returned; no exceptions cross the line.
 */
#ifndef BMA_H
#define BMA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BMA_VERSION 112 /* 0.1.12: bma_prefix_attention_set_plan (128-wide heads on the 32x32x16 kernel: 64-key chunks by LDS-DMA); 0.1.11: bma_causal_attention_gqa(+_bwd_gqa) (grouped-query heads, 256-wide heads: Gemma-3's decoder at batch 1); 0.1.10: bma_add_layernorm(+_bwd) (CLIP's residual add + LayerNorm pairs in one launch each way); 0.1.9: bma_gemm_nt_next (cross-product weight prefetch), bma_gemm_nt_set_plan flags bit 3 (fenced split-K hand-off); 0.1.8: bma_causal_attention(+_bwd) (one long sequence at batch 1); 0.1.7: bma_gemm_mid (the 599-644-row products of the pass with the image in the prompt); 0.1.6: bma_b1_attention(+_bwd); bma_gemm_nt_plan / _set_plan (slab height chosen with the split count); 0.1.5: bma_quick_gelu(+_bwd); 0.1.4: bma_qknorm_rope2; 0.1.3: bma_allgather_f32; 0.1.2: bma_add_rmsnorm(+_bwd), bma_rope2, bma_splice_rows, bma_gemm_nt; 0.1.1: bma_mask_topk takes a workspace; bma_ragged_attention takes blocks of any length, 256-wide heads */

/* element types of model-dtype tensors */
enum { BMA_F32 = 0, BMA_BF16 = 1, BMA_F16 = 2 };

/* error codes */
enum {
  BMA_OK = 0,
  BMA_EINVAL = -1,   /* bad argument (null pointer, negative size, k > V, ...) */
  BMA_EDTYPE = -2,   /* unsupported dtype code */
  BMA_EALIGN = -3,   /* pointer / stride not aligned as the kernel requires */
  BMA_ELAUNCH = -4,  /* hipLaunchKernel reported an error */
  BMA_ELIMIT = -5,   /* size beyond what the kernel was built for */
  BMA_ECOLL = -6     /* bma_allgather_f32: no RCCL in the process, or RCCL reported an error */
};

int bma_version(void);
const char* bma_strerror(int code);

/* ---------------------------------------------------------------------------
 * a5  perform_pgd_step  (:1030-1037)
 *   out[i] = clamp(clamp(x[i] - step*sign(g[i]), x0[i]-eps, x0[i]+eps), 0, 1)
 * fp32, exact arithmetic (bit-comparable with the reference's three torch ops).
 * `step` is the reference's alpha*eps already multiplied by the caller in double
 * and rounded to fp32.  out may alias x.  Pointers 4-byte aligned; the 16-byte
 * vector path is used when all four are 16-byte aligned.
 * ------------------------------------------------------------------------- */
int bma_linf_step(const float* x, const float* g, const float* x0, int64_t n,
                  float eps, float step, float* out, void* stream);

/* ---------------------------------------------------------------------------
 * a2  cross-entropy over the target slice
 *     (:1006-1012 gradient pass, :1289-1306 candidate scoring)
 * logits: B candidates x T rows x V columns of `dtype`; element (b,t,v) lives at
 *   logits + b*ld_cand + t*ld_row + v   (strides in ELEMENTS; rows contiguous).
 * labels: T int64 target token ids, shared by all candidates (the reference
 *   `.repeat`s them, :1291).
 * Outputs (fp32 accumulation throughout):
 *   loss      [B]    mean over the T rows of a candidate, summed in row order
 *   match     [B]    1 if argmax(row) == label for all T rows (first maximum
 *                    wins, as torch.argmax) -- the early-stop test (:1300-1306);
 *                    may be NULL
 *   dlogits   [B*T*V] of `dtype`, contiguous: (softmax(row) - onehot(label)) *
 *                    grad_scale / T -- the backward of the mean CE; may be NULL
 *   ws        scratch of bma_ce_target_ws_bytes(B,T) bytes = 3*B*T 4-byte words,
 *                    left holding: [0,BT) per-row loss f32 (logsumexp - x[label]),
 *                    [BT,2BT) per-row argmax==label i32, [2BT,3BT) logsumexp f32
 * ------------------------------------------------------------------------- */
size_t bma_ce_target_ws_bytes(int B, int T);
int bma_ce_target(const void* logits, int64_t ld_cand, int64_t ld_row,
                  const int64_t* labels, int B, int T, int V, int dtype,
                  float* ws, float* loss, int32_t* match,
                  void* dlogits, float grad_scale, void* stream);

/* ---------------------------------------------------------------------------
 * a3  sample_ids_from_grad, first half  (:144-147)
 *   grad[:, not_allowed] = +inf ; topk(-grad, k).indices
 * grad: rows x V of `dtype`, row r at grad + r*ld_row (elements).
 * mask_bits: ceil(V/32) uint32 words, bit (v & 31) of word (v >> 5) set = token
 *   v is not allowed (counts as +inf); NULL = nothing masked.
 * idx_out [rows*k] int64: per row the k allowed tokens with the most negative
 *   gradient, ordered by gradient ascending, ties by token id ascending; NaN
 *   gradients rank first (torch.topk ranks NaN highest in -grad); -0.0 == +0.0.
 *   The gradient itself is not modified.
 * Limits: 1 <= k <= 2048, k <= V, V < 2^31.
 * Workspace: rows longer than 4096 tokens are cut across workgroups (slice-local
 *   select, then a merge per row); the two stages hand over through `ws`,
 *   bma_mask_topk_ws_bytes(rows, V, k) bytes, 8-byte aligned (0 = not needed).
 *   ws == NULL is allowed: one workgroup per row then does the whole select.
 * ------------------------------------------------------------------------- */
size_t bma_mask_topk_ws_bytes(int rows, int V, int k);
int bma_mask_topk(const void* grad, int64_t ld_row, int rows, int V, int dtype,
                  const uint32_t* mask_bits, int k, int64_t* idx_out, void* ws, void* stream);

/* ---------------------------------------------------------------------------
 * a3  sample_ids_from_grad, second half  (:150-162)
 * bma_rand_positions: pos[b, 0:n_rep] = the n_rep positions of row b of `rnd`
 *   (B x n_opt uniform floats) with the smallest values, in ascending order of
 *   value (ties by position) == argsort(rnd)[..., :n_rep].  Any n_opt >= n_rep.
 * bma_sample_scatter: out[b,:] = ids; out[b, pos[b,j]] = topk_idx[pos[b,j]*k +
 *   rank[b,j]] for j < n_rep.   ids [n_opt], topk_idx [n_opt*k], pos/rank
 *   [B*n_rep], out [B*n_opt], all int64.
 * ------------------------------------------------------------------------- */
int bma_rand_positions(const float* rnd, int B, int n_opt, int n_rep,
                       int64_t* pos_out, void* stream);
int bma_sample_scatter(const int64_t* ids, const int64_t* topk_idx,
                       const int64_t* pos, const int64_t* rank, int B, int n_opt,
                       int n_rep, int k, int64_t* out, void* stream);

/* ---------------------------------------------------------------------------
 * a7  _build_input_embeds  (:1112-1225): emb(sampled_ids) gather + repeat + cat
 * Builds out[B][S][D] (contiguous, `dtype`) from up to BMA_MAX_SEGS segments laid
 * end to end along the sequence axis.  Segment kinds:
 *   BMA_SEG_SHARED  ptr -> [len][D], the same rows for every candidate
 *   BMA_SEG_PERCAND ptr -> [B][len][D], candidate-major
 *   BMA_SEG_GATHER  rows emb[ids[b][j]] * emb_scale for j < len (len == n_opt);
 *                   ptr unused.  emb_scale == 1.0f copies bits; otherwise the
 *                   product is formed in fp32 and rounded to `dtype` (Gemma's
 *                   scaled embedding, :1142).
 * S is the sum of the segment lengths.  D*sizeof(elem) must be a multiple of 16
 * and every pointer 16-byte aligned.  ids are int64 in [0, V).
 * ------------------------------------------------------------------------- */
#define BMA_MAX_SEGS 8
enum { BMA_SEG_SHARED = 0, BMA_SEG_PERCAND = 1, BMA_SEG_GATHER = 2 };
typedef struct bma_segment {
  const void* ptr;
  int32_t len;
  int32_t kind;
} bma_segment;

int bma_splice(const bma_segment* segs_host, int n_segs, const void* emb, int V,
               const int64_t* ids, int B, int n_opt, int D, int dtype,
               float emb_scale, void* out, void* stream);
/* bma_splice_rows: the ROW-LIST form for ragged scoring (a candidate is computed from its first replaced suffix
 *   position on): out[n][D] = row slot[n] of the [B][S][D] block bma_splice would build from the same arguments
 *   (slot[n] = b*S + s, int32, clamped into the block), for n < n_rows; the block itself is never materialised.
 *   Replaces bma_splice + bma_gather_rows (one pass over the rows instead of a write, a read and a write). */
int bma_splice_rows(const bma_segment* segs_host, int n_segs, const void* emb, int V,
                    const int64_t* ids, int B, int n_opt, int D, int dtype, float emb_scale,
                    const int* slot, int64_t n_rows, void* out, void* stream);

/* ---------------------------------------------------------------------------
 * a6  the elementwise tail of a Llama-family decoder layer during candidate scoring
 *     (the model forward of :1287).  Each replaces a chain of eager HuggingFace ops with
 *     one pass over HBM and keeps the chain's rounding points (see fused_elementwise.hip).
 * bma_rmsnorm:      out[r,:] = weight * dt(x[r,:] * rsqrt(mean(x[r,:]^2) + eps))   (gemma_style=0)
 *                   out[r,:] = dt(x[r,:] * rsqrt(mean(..)+eps) * (1 + weight))     (gemma_style=1)
 *                   x, out: rows x D contiguous; D*es a multiple of 16 and <= 16 KiB.
 * bma_swiglu:       out = dt(dt(silu(gate)) * up), n contiguous elements each.
 * bma_rope_inplace: q <- dt(dt(q*cos) + dt(rotate_half(q)*sin)) in place; element (b,h,l,d) of q
 *                   at q + b*stride_b + h*stride_h + l*stride_l + d (elements, d contiguous);
 *                   cos/sin: [cos_batch][L][Dh] contiguous, cos_batch = 1 or B.
 * ------------------------------------------------------------------------- */
int bma_rmsnorm(const void* x, const void* weight, float eps, int64_t rows, int D,
                int dtype, int gemma_style, void* out, void* stream);
/* bma_add_rmsnorm: the residual add of a decoder layer and the norm that follows it, in one pass (2 reads + 2 writes
 *   per row instead of 3 + 2):   sum_out = dt(residual + a),   out = rmsnorm(sum_out; weight, eps)   as bma_rmsnorm,
 *   where a = h, or, with pre_weight != NULL, a = rmsnorm(h; pre_weight, pre_eps) rounded to `dtype` first (Gemma-3's
 *   sandwich norm on the branch output: HF Gemma3DecoderLayer.forward).  Bit-identical to the eager add followed by
 *   bma_rmsnorm.  All of residual, h, sum_out, out: rows x D contiguous; limits as bma_rmsnorm.
 * bma_add_rmsnorm_bwd: dx = dsum + rmsnorm_bwd(x, weight, dy) (dsum may be NULL: then == bma_rmsnorm_bwd): the
 *   gradient of both inputs of the fused add (x is the forward's sum_out). */
int bma_add_rmsnorm(const void* residual, const void* h, const void* pre_weight, float pre_eps,
                    const void* weight, float eps, int64_t rows, int D, int dtype, int gemma_style,
                    void* sum_out, void* out, void* stream);
int bma_add_rmsnorm_bwd(const void* x, const void* weight, const void* dy, const void* dsum, float eps,
                        int64_t rows, int D, int dtype, int gemma_style, void* dx, void* stream);

/* bma_add_layernorm / bma_add_layernorm_bwd: the residual add and the LayerNorm behind it of a pre-LN vision-tower block in ONE
 *   launch each way (a1 with PGD on, :970-979: `get_image_features` runs CLIP's 24 encoder layers at batch 1, 577 x 1024, where
 *   HuggingFace's modeling_clip.CLIPEncoderLayer.forward issues an add and a LayerNorm twice per layer and autograd an add and a
 *   LayerNorm backward twice more -- all launch-bound).  Forward: sum_out = dt(residual + h), out = dt(weight * (rstd * (sum -
 *   mean)) + bias) with mean / rstd of the row in fp32 (aten's expression and rounding points); residual == NULL: a plain
 *   LayerNorm of h (sum_out unused).  stats [rows][2] fp32 receives (mean, rstd) for the backward (may be NULL).  Backward:
 *   dx = dt(rstd * (g - mean(g) - xhat * mean(g * xhat))) with g = dy * weight, xhat = (x - mean) * rstd, x the SUM the forward
 *   normalised; + dsum when given (the gradient arriving through the residual stream; rounding as the eager chain: the norm's
 *   gradient rounded to dtype, then the add).  Weight and bias are constants: no gradient for them.  Rows of D elements,
 *   D * es a multiple of 16 and at most 16 KiB; f32 / bf16 / f16. */
int bma_add_layernorm(const void* residual, const void* h, const void* weight, const void* bias, float eps, int64_t rows, int D,
                      int dtype, void* sum_out, void* out, float* stats, void* stream);
int bma_add_layernorm_bwd(const void* x, const void* weight, const void* dy, const void* dsum, const float* stats, int64_t rows,
                          int D, int dtype, void* dx, void* stream);
int bma_swiglu(const void* gate, const void* up, int64_t n, int dtype, void* out, void* stream);
/* bma_gated_act: out = dt(dt(act(gate)) * up); act 0 = SiLU (== bma_swiglu), 1 = GELU-tanh as
 *   aten evaluates gelu(x, approximate="tanh") (Gemma's gated MLP). */
int bma_gated_act(const void* gate, const void* up, int64_t n, int dtype, int act, void* out, void* stream);
/* bma_gated_act_il: the same with gate and up as ALTERNATING 16-byte chunks of one array of 2n elements (chunk
 *   2i = gate chunk i, chunk 2i+1 = up chunk i): the output of ONE product against the chunk-interleaved
 *   gate_proj/up_proj weights instead of two products.  out: n elements, contiguous. */
int bma_gated_act_il(const void* gate_up, int64_t n, int dtype, int act, void* out, void* stream);
int bma_rope_inplace(void* q, int64_t stride_b, int64_t stride_h, int64_t stride_l,
                     int B, int H, int L, int Dh, const void* cos, const void* sin,
                     int cos_batch, int dtype, void* stream);
/* bma_rope: the same rotation read from q and written to dst (own strides; dst == q is the in-place
 *   form); sin_sign = -1 applies the inverse rotation, which is the backward of the forward one. */
int bma_rope(const void* q, int64_t stride_b, int64_t stride_h, int64_t stride_l, void* dst, int64_t dst_b,
             int64_t dst_h, int64_t dst_l, int B, int H, int L, int Dh, const void* cos, const void* sin,
             int cos_batch, float sin_sign, int dtype, void* stream);
/* bma_rope2: q AND k of one attention block in one launch (same cos/sin, B, L, Dh; own base pointers, strides, head
 *   counts and destinations, each as in bma_rope; qd == q / kd == k is the in-place form). */
int bma_rope2(const void* q, int64_t q_b, int64_t q_h, int64_t q_l, void* qd, int64_t qd_b, int64_t qd_h,
              int64_t qd_l, int Hq, const void* k, int64_t k_b, int64_t k_h, int64_t k_l, void* kd,
              int64_t kd_b, int64_t kd_h, int64_t kd_l, int Hk, int B, int L, int Dh, const void* cos,
              const void* sin, int cos_batch, float sin_sign, int dtype, void* stream);

/* bma_quick_gelu / bma_quick_gelu_bwd: CLIP's MLP activation, y = x * sigmoid(1.702 x), on n contiguous elements as
 *   HuggingFace's QuickGELUActivation evaluates it -- three aten kernels forward, five in its autograd backward, each
 *   rounding to `dtype` -- in ONE launch each with the roundings where aten has them (bit-identical).  The backward
 *   recomputes the sigmoid from x: dx = dy*s + 1.702 * (dy*x) * (1 - s) * s.  (bf16 and fp32 are bit-identical to
 *   aten; fp16 is computed the same way, but aten's own fp16 sigmoid is not correctly rounded, so the host keeps fp16 eager.)  n * element size a multiple of 16,
 *   16-byte aligned pointers; out may alias x (forward) or dy (backward). */
int bma_quick_gelu(const void* x, int64_t n, int dtype, void* out, void* stream);
int bma_quick_gelu_bwd(const void* x, const void* dy, int64_t n, int dtype, void* dx, void* stream);

/* bma_qknorm_rope2: bma_rope2 with the per-head RMSNorm of q and k in front of the rotation (Gemma-3's q_norm / k_norm:
 *   weights wq / wk [Dh] of `dtype`, `eps`, `gemma` != 0 for the (1 + w) form) in the same pass: bit for bit bma_rmsnorm
 *   on the head rows followed by bma_rope2, one read and one write of q and k instead of two.  Forward rotation only
 *   (the no-grad scoring forward); same layout rules as bma_rope2, Dh * es / 16 a power of two <= 64. */
int bma_qknorm_rope2(const void* q, int64_t q_b, int64_t q_h, int64_t q_l, void* qd, int64_t qd_b, int64_t qd_h,
                     int64_t qd_l, int Hq, const void* k, int64_t k_b, int64_t k_h, int64_t k_l, void* kd,
                     int64_t kd_b, int64_t kd_h, int64_t kd_l, int Hk, int B, int L, int Dh, const void* wq,
                     const void* wk, float eps, int gemma, const void* cos, const void* sin, int cos_batch,
                     int dtype, void* stream);
/* Backward halves, used by the gradient pass (autograd at batch 1; weights are constants, so no
 * weight gradients): bma_rmsnorm_bwd: dx from x, weight, dy (D*es <= 16 KiB);
 * bma_swiglu_bwd: dgate, dup from gate, up, dy.  RoPE's backward is bma_rope_inplace with -sin. */
int bma_rmsnorm_bwd(const void* x, const void* weight, const void* dy, float eps, int64_t rows,
                    int D, int dtype, int gemma_style, void* dx, void* stream);
int bma_swiglu_bwd(const void* gate, const void* up, const void* dy, int64_t n, int dtype,
                   void* dgate, void* dup, void* stream);
int bma_gated_act_bwd(const void* gate, const void* up, const void* dy, int64_t n, int dtype, int act,
                      void* dgate, void* dup, void* stream);
/* ... and for the chunk-interleaved layout of bma_gated_act_il: dgate_up has gate_up's layout (2n elements). */
int bma_gated_act_il_bwd(const void* gate_up, const void* dy, int64_t n, int dtype, int act, void* dgate_up,
                         void* stream);
/* bma_attn_merge: merges the two partial attentions of the shared-prefix scheme (new tokens
 *   vs the prompt prefix shared by all candidates; new tokens vs themselves, causal):
 *   out = w*o1 + (1-w)*o2 with w = 1/(1+exp(lse2-lse1)).  o1, o2, out: [B][L][H][Dh] contiguous
 *   of `dtype`; lse1: [H][B*L] fp32 (prefix launch: batch 1, B*L queries); lse2: [B][H][L] fp32. */
int bma_attn_merge(const void* o1, const void* o2, const float* lse1, const float* lse2,
                   int B, int L, int H, int Dh, int dtype, void* out, void* stream);
/* Ragged scoring (a candidate differs from its parent suffix from position p on, so only its
 *   tokens >= p are computed; the N computed tokens of all candidates form one row list):
 * bma_attn_merge_rows: as bma_attn_merge, with o1/out [N][H][Dh], lse1 [H][N], and row n paired
 *   with row map[n] (= b*L + l) of the padded o2 [B2][L][H][Dh] / lse2 [B2][H][L];
 * bma_gather_rows: out[r] = src[idx[r]] for rows of row_bytes (multiple of 16) bytes -- builds the
 *   padded (B2,L) query/key/value blocks from the row list (a candidate's rows < p come from
 *   its parent's rows).  Indices are clamped into the source. */
int bma_attn_merge_rows(const void* o1, const void* o2, const float* lse1, const float* lse2,
                        const int* map, int64_t N, int B2, int L, int H, int Dh, int dtype, void* out,
                        void* stream);
int bma_gather_rows(const void* src, const int* idx, int64_t n_out, int64_t n_src, int64_t row_bytes,
                    void* out, void* stream);

/* bma_allgather_f32 (the reference, bimodal_attack.py:1282-1299, scores every candidate on one GPU): the
 *   collective of the sharded chunk loop -- every rank's n_local fp32 values (its candidates' losses, padded with
 *   +inf to the common count) gathered into out[world][n_local] on every rank, in rank order, on `stream`.
 *   `comm` is the host's ncclComm_t (RCCL), created by the host with ncclCommInitRank; this library calls
 *   ncclAllGather of the RCCL instance already loaded into the process (it does not link one).  comm == NULL is
 *   accepted for world == 1 only (out = local).  local may be out + rank*n_local (in place).  BMA_ECOLL when no RCCL
 *   can be found or RCCL reports an error.  Device pointers, 4-byte aligned. */
int bma_allgather_f32(const float* local, int64_t n_local, float* out, int rank, int world, void* comm, void* stream);

/* bma_gemm_nt: y[M][N] = x[M][K] . w[N][K]^T for the SKINNY products of the batch-1 gradient pass (a1, :953-1028: every
 *   linear layer of the language model applied to a handful of rows): bf16 / f16 operands with K contiguous, fp32
 *   accumulation on the matrix cores, one rounding to `dtype`.  Leading dimensions ldx/ldw/ldy in elements (multiples
 *   of 8 / 8 / 4); K a multiple of 64; any M (tiles of 64 or 96 rows), any N.  The weight is streamed once (non-temporal
 *   loads); the grid is (slabs of w rows) x (splits of K), slab height (<= 128 or 192 rows, a multiple of 4) and split
 *   count chosen together so that the workgroups fill the 256 CUs in whole rounds (bma_gemm_nt_plan reports them).
 *   With more than one split the partial sums pass through `ws` (bma_gemm_nt_ws_bytes(M,N,K) bytes, 16-byte aligned)
 *   and a ticket per tile in `counters` (bma_gemm_nt_tiles(M,N,K) ints, ZERO before the first launch; every launch
 *   leaves them zero); the last workgroup of a tile adds the partials in split order, so the result does not depend on
 *   arrival order.  ws/counters may be NULL when bma_gemm_nt_ws_bytes returns 0.  Launches that share ws/counters
 *   must be ordered on one stream.
 * bma_gemm_nt_plan: the decomposition bma_gemm_nt will use, out8 = {row tiles of 16 per workgroup, row-tile count,
 *   16-row w tiles per wave, w rows per slab, slabs, K splits, splits of a tile kept on one XCD (0/1), non-temporal w
 *   loads (0/1)}.  bma_gemm_nt_set_plan: measurement only (tools/gemm_bench.py --sweep) -- pins w tiles per wave /
 *   rows per slab / splits (0 = the planner's choice) and the two flags (bit 0 XCD grouping, bit 1 non-temporal; -1 =
 *   default; bit 3: the split-K hand-off with an agent-scope release / acquire fence pair on top of the write-through
 *   stores -- the form the HIP memory model asks for, ~2 us per split launch slower; the default relies on gfx942 /
 *   gfx950 cache behaviour) for every later call in the process; results never depend on it.  Process-global and not
 *   synchronised: not to be called while another thread sizes or launches a product -- nor once hipGraphs that hold
 *   bma_gemm_nt_next launches exist (they baked the next launch's plan in at capture time).
 * bma_gemm_nt_next: bma_gemm_nt that also knows the NEXT product of the caller's chain -- its weight next_w [next_N][next_ldw]
 *   (next_K columns used), applied to the same M rows (the gradient pass walks qkv -> gate/up -> down -> the next layer's
 *   qkv, and the transposed copies in reverse, bimodal_attack.py:1003, :1016-1025; weights do not depend on activations).
 *   Workgroups of a split launch that leave early -- every split of a tile but the last arriver -- load the first four
 *   64-column stages of the weight rows the next launch's workgroups on the same XCD will start with, so that launch finds
 *   them in the L2 / Infinity Cache instead of ramping HBM up from idle.  A hint: results are those of bma_gemm_nt;
 *   next_w == NULL is bma_gemm_nt.  Only addresses inside next_w [0, next_N) x [0, next_K) are touched. */
size_t bma_gemm_nt_ws_bytes(int M, int N, int K);
int bma_gemm_nt_tiles(int M, int N, int K);
int bma_gemm_nt_plan(int M, int N, int K, int* out8);
void bma_gemm_nt_set_plan(int w_tiles_per_wave, int rows_per_slab, int splits, int flags);
int bma_gemm_nt(const void* x, int64_t ldx, const void* w, int64_t ldw, void* y, int64_t ldy, int M, int N, int K,
                int dtype, void* ws, size_t ws_bytes, int* counters, int n_counters, void* stream);
int bma_gemm_nt_next(const void* x, int64_t ldx, const void* w, int64_t ldw, void* y, int64_t ldy, int M, int N, int K,
                     int dtype, void* ws, size_t ws_bytes, int* counters, int n_counters, const void* next_w,
                     int64_t next_ldw, int next_N, int next_K, void* stream);

/* bma_causal_attention / bma_causal_attention_bwd: causal self-attention of ONE sequence at batch 1 and its backward, for the
 *   gradient pass with the image in the prompt (a1, :953-1028 with PGD on: 599-644 tokens per layer) and for the rows
 *   behind a reused prefix (joint mode: 44 new tokens against 643 keys).  The Lq queries are the LAST Lq positions of the
 *   Lk keys: query i attends to keys 0 .. (Lk - Lq) + i.  q [Lq][H][Dh], k / v [Lk][H][Dh] through (row, head) strides in
 *   elements (multiples of 8; views of a fused projection are fine), H query heads = H key/value heads, Dh = 64, 72, 128 or 256, bf16 /
 *   f16, rotary already applied.  causal = 0: every query sees every key (a vision tower's attention: CLIP's 577 tokens x 16
 *   heads of 64).  Forward: out [Lq][H][Dh] contiguous and lse2 [H][Lq] fp32 = log2 of the softmax
 *   denominator in units of the scaled scores (an opaque token for the backward).  Backward: dq [Lq][H][Dh], dk / dv
 *   [Lk][H][Dh] with rows d_row_stride elements apart (H*128 when contiguous; 3*H*128 writes the three straight into
 *   the gradient of a fused q/k/v projection) from d_out [Lq][H][Dh] contiguous; `delta` [H][Lq] fp32 is scratch.  Two launches (dq, then
 *   dk and dv), every output element with one owner: no atomics, bitwise reproducible.  Probabilities and score
 *   gradients are rounded to `dtype` before their products, as in a flash kernel. */
int bma_causal_attention(const void* q, int64_t q_row_stride, int64_t q_head_stride, const void* k, int64_t k_row_stride,
                         int64_t k_head_stride, const void* v, int64_t v_row_stride, int64_t v_head_stride, int64_t Lq,
                         int64_t Lk, int H, int Dh, int dtype, int causal, float scale, void* out, float* lse2, void* stream);
int bma_causal_attention_bwd(const void* q, int64_t q_row_stride, int64_t q_head_stride, const void* k, int64_t k_row_stride,
                             int64_t k_head_stride, const void* v, int64_t v_row_stride, int64_t v_head_stride, const void* out,
                             const float* lse2, const void* d_out, int64_t Lq, int64_t Lk, int H, int Dh, int dtype,
                             int causal, float scale, void* dq, void* dk, void* dv, int64_t d_row_stride, float* delta,
                             void* stream);
/* ... with grouped-query heads: H query heads read Hkv key/value heads (H a multiple of Hkv; query heads h*rep .. h*rep+rep-1
 *   share key/value head h, HuggingFace's repeat_kv order) -- k / v [Lk][Hkv][Dh], dk / dv [Lk][Hkv][Dh] with their own row
 *   stride, summed over the group inside the launch (no repeated copies of k / v, no reduction afterwards) -- and head widths
 *   64, 72, 128 or 256 (Gemma-3's decoder in the gradient pass, reference :953-1028 on a Gemma-3 model: ~320 tokens x 8 heads
 *   over 4 of 256).  Hkv == H is bma_causal_attention / _bwd. */
/* experiment knob: 72-wide heads (SigLIP) with at least this many queries run the forward with two 16-query tiles per wave
 *   (128 rows per workgroup: every K / V^T fragment read from LDS feeds two MFMAs); 0 = never; default 1024.  Results are
 *   the same attention either way (the accumulation order over keys does not change). */
void bma_causal_attention_set_plan(int64_t fwd_two_tiles_min_rows);
int bma_causal_attention_gqa(const void* q, int64_t q_row_stride, int64_t q_head_stride, const void* k, int64_t k_row_stride,
                             int64_t k_head_stride, const void* v, int64_t v_row_stride, int64_t v_head_stride, int64_t Lq,
                             int64_t Lk, int H, int Hkv, int Dh, int dtype, int causal, float scale, void* out, float* lse2,
                             void* stream);
int bma_causal_attention_bwd_gqa(const void* q, int64_t q_row_stride, int64_t q_head_stride, const void* k, int64_t k_row_stride,
                                 int64_t k_head_stride, const void* v, int64_t v_row_stride, int64_t v_head_stride,
                                 const void* out, const float* lse2, const void* d_out, int64_t Lq, int64_t Lk, int H, int Hkv,
                                 int Dh, int dtype, int causal, float scale, void* dq, void* dk, void* dv,
                                 int64_t dq_row_stride, int64_t dkv_row_stride, float* delta, void* stream);

/* bma_gemm_mid: y[M][N] = x[M][K] . w[N][K]^T for the products of the batch-1 gradient pass when the image is part of the
 *   prompt (a1, :953-1028 with PGD on: 576 image rows + the text = 599-644 rows; the same shapes occur in the prefix
 *   pass of joint candidate scoring, :605-612).  Operands, accumulation, rounding and leading dimensions as for
 *   bma_gemm_nt; K a multiple of 64; any M (row tiles of 224), any N; a tile's operands within 2 GiB (256 rows x the larger
 *   leading dimension + K, in bytes: BMA_ELIMIT beyond).  Tiles of 224 x {192, 256} on one workgroup per CU;
 *   the tile width and a split of K -- of every tile, or of the last columns of tiles only -- are chosen so that the
 *   grid fits the 256 CUs (bma_gemm_mid_plan reports them).  Split tiles pass their fp32 partials through `ws`
 *   (bma_gemm_mid_ws_bytes(M,N,K) bytes, 16-byte aligned; may be NULL when that is 0) and a second launch on the same
 *   stream adds them in split order: results do not depend on scheduling.
 * bma_gemm_mid_plan: out8 = {16-row x fragments per wave, row-tile count, 16-row w fragments per wave, column-tile
 *   count, K splits, XCD-contiguous tile order (0/1), workgroups, tiles that run unsplit}.  bma_gemm_mid_set_plan:
 *   measurement only -- pins w fragments per wave / splits / columns of tiles that are split (0, 0, -1 = the planner's
 *   choice; 0 columns = every tile) and the flags (bit 0 XCD order; -1 = default; other bits ignored) for every later call in the process;
 *   results never depend on it. */
size_t bma_gemm_mid_ws_bytes(int M, int N, int K);
int bma_gemm_mid_plan(int M, int N, int K, int* out8);
void bma_gemm_mid_set_plan(int w_frags_per_wave, int splits, int tail_columns, int flags);
int bma_gemm_mid(const void* x, int64_t ldx, const void* w, int64_t ldw, void* y, int64_t ldy, int M, int N, int K,
                 int dtype, void* ws, size_t ws_bytes, void* stream);

/* bma_b1_attention / bma_b1_attention_bwd: rotary embedding + causal self-attention of ONE short sequence, for the batch-1
 *   gradient pass over a text-only prompt (a1, :953-1028: what HuggingFace's attention block does between the q/k/v
 *   projections and o_proj, and autograd's backward of it).  qkv [S][ld_qkv] holds, per token, H query heads, H key heads
 *   and H value heads of 128 (the fused projection's own output); cos / sin [S][128] contiguous, HuggingFace's duplicated-
 *   halves tables (rotate_half convention); out [S][ld_out] = H heads of 128 in the layout o_proj reads; lse [H][S] fp32,
 *   natural log.  The backward takes dout [S][ld_dout] and writes dqkv [S][ld_dqkv] in qkv's layout (the rotation's
 *   backward applied).  S <= 80 (BMA_ELIMIT beyond), bf16 / f16, one workgroup per head; leading dimensions in elements,
 *   multiples of 8; pointers 16-byte aligned.  Rounding points are those of bma_rope2 + a flash-attention kernel: the
 *   rotated q / k, the probabilities and dS are rounded to `dtype` before their products, sums are fp32. */
int bma_b1_attention(const void* qkv, int64_t ld_qkv, const void* cos, const void* sin, int S, int H, int dtype, float scale,
                     void* out, int64_t ld_out, float* lse, void* stream);
int bma_b1_attention_bwd(const void* qkv, int64_t ld_qkv, const void* cos, const void* sin, const void* out, int64_t ld_out,
                         const float* lse, const void* dout, int64_t ld_dout, int S, int H, int dtype, float scale, void* dqkv,
                         int64_t ld_dqkv, void* stream);

/* bma_prefix_attention: N rows (q [N][H][Dh] through row/head strides) against the P keys/values of the SHARED
 *   prefix (pk/pv [P][Hk][Dh] through strides; grouped heads in place), no mask: out [N][H][Dh] contiguous of
 *   `dtype` and lse [H][N] fp32 (natural log) -- the partial that bma_ragged_attention merges (o1, lse1).  A
 *   flash-attention forward on the matrix cores (bf16 / f16, Dh 64 or 128): 4*N*P*Dh*H flops per launch. */
int bma_prefix_attention(const void* q, int64_t q_row_stride, int64_t q_head_stride, const void* pk,
                         int64_t pk_row_stride, int64_t pk_head_stride, const void* pv, int64_t pv_row_stride,
                         int64_t pv_head_stride, int P, int64_t N, int H, int Hk, int Dh, int dtype, float scale,
                         void* out, float* lse, void* stream);
/* Which kernel takes a launch -- 0: by shape (128-wide heads and a prefix of more than 64 keys: v_mfma_f32_32x32x16, 64-key
 *   chunks L2 -> LDS by LDS-DMA, four waves of 32 rows; anything else: the 16x16x32 kernel), 1: the 16x16x32 kernel always,
 *   4 / 8: the 32x32x16 kernel on workgroups of that many waves.  Measurement and tests; process-wide, not thread-safe
 *   against concurrent launches.  Replaces nothing in the reference (bimodal_attack.py:1150-1163 recomputes the prefix per
 *   candidate). */
void bma_prefix_attention_set_plan(int kernel);

/* bma_ragged_attention: the attention of a ragged scoring forward in one launch (bf16 / f16, MFMA).
 *   Candidate i (0 <= i < B2) owns rows start[i] .. start[i]+len[i]-1 of the row list: its tokens at
 *   positions first[i] .. first[i]+len[i]-1 behind the shared prefix (len[i] <= max_len <= 4096).  A query
 *   at position j attends to the P prefix keys (pk/pv), to rows t of the row list for positions
 *   t < first[i] (its parent's keys/values) and to its own rows for first[i] <= t <= j.
 *   q/k/v: row-list tensors addressed as base + row*rs + head*hs (elements; multiples of 8);
 *   pk/pv likewise with P rows; H query heads, Hk key/value heads (H % Hk == 0), Dh in {32,64,128,256}.
 *   out: [N][H][Dh] contiguous.  If o1/lse1 are given (o1 [N][H][Dh], lse1 [H][N] fp32: a prefix
 *   partial computed elsewhere, then pass P = 0) the result is merged with it as bma_attn_merge does.
 *   Two kernels behind the one entry: blocks of max_len >= 96 tokens at Dh 128 / 256 (Gemma-3's padded candidates)
 *   take a persistent flash-attention-style kernel -- one workgroup per CU, LDS-DMA ring, two query heads of a
 *   key/value head per workgroup; shorter blocks and the other head widths one workgroup per (candidate, head,
 *   64 queries).  Same arithmetic (32-key online-softmax steps, fp32 accumulation, one rounding), not bit-identical.
 *   bma_ragged_attention_set_long(mode, min_len) -- measurement only, process-wide: mode 0 keeps every shape on the
 *   second kernel, 1 (default) routes by length; min_len > 0 moves the 96-token threshold (0 = default).  The library
 *   reads no environment variables. */
void bma_ragged_attention_set_long(int mode, int min_len);
int bma_ragged_attention(const void* q, int64_t q_rs, int64_t q_hs, const void* k, int64_t k_rs, int64_t k_hs,
                         const void* v, int64_t v_rs, int64_t v_hs, const void* pk, int64_t pk_rs, int64_t pk_hs,
                         const void* pv, int64_t pv_rs, int64_t pv_hs, int P, const int* start, const int* first,
                         const int* len, int B2, int max_len, int64_t N, int H, int Hk, int Dh, int dtype,
                         float scale, const void* o1, const float* lse1, void* out, void* stream);

/* ---------------------------------------------------------------------------
 * Measurement aid (bench.py).  When enabled, the dominant kernel of
 * each entry point is bracketed by HIP events on the launch stream and the launch's
 * ALGORITHMIC bytes (the figures DESIGN.md states per unit) are tallied.  Disabled
 * by default -- nothing is recorded and calls stay graph-capturable.
 * bma_profile_enable(on) resets all tallies; bma_profile_read() waits for the
 * recorded events and returns launches, summed device milliseconds, summed bytes.
 * ------------------------------------------------------------------------- */
enum {
  BMA_K_LINF = 0, BMA_K_CE_ROWS = 1 /* B > 1: candidate scoring */, BMA_K_CE_DLOGITS = 2,
  BMA_K_TOPK = 3, BMA_K_SCATTER = 4, BMA_K_SPLICE = 5,
  BMA_K_CE_ROWS_B1 = 6 /* B == 1: the gradient pass */, BMA_K_RMSNORM = 7, BMA_K_SWIGLU = 8,
  BMA_K_ROPE = 9, BMA_K_ATTN_MERGE = 10, BMA_K_GATHER_ROWS = 11,
  BMA_K_RAGGED_ATTN = 12, BMA_K_PREFIX_ATTN = 13, BMA_K_ADD_RMSNORM = 14, BMA_K_GEMM_NT = 15, BMA_K_B1_ATTN = 16, BMA_K_GEMM_MID = 17, BMA_K_CAUSAL_ATTN = 18, BMA_K_COUNT = 19
};
int bma_profile_enable(int on);
int bma_profile_read(int kernel, int64_t* launches, double* total_ms, double* total_bytes);
const char* bma_profile_kernel_name(int kernel);

#ifdef __cplusplus
}
#endif
#endif /* BMA_H */
