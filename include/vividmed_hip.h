/*
 * vividmed_hip.h — C ABI of libvividmed_hip.so (gfx950 / MI355X).
 *
 * This is the "B-inner" boundary of SURVEY.md §8(b): the reference
 * (function2-llx/MMMM) has no FFI of its own — its hot path is Python calling
 * torch/xformers/peft ops — so every entry point below replaces one *op site*
 * of the reference training step, cited as `file:line` into /root/reference.
 *
 * Conventions
 *   - extern "C", plain pointers + sizes; no torch / C++ types.
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`.
 *   - matrices are row-major with an explicit leading dimension (elements).
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it,
 *     nothing synchronises, nothing allocates (graph-capture safe).
 *   - return value: 0 = ok, negative = VM_ERR_* (never throws).
 *   - `nrows_dev`: optional device pointer to an int32 row count that is
 *     produced on the device (e.g. number of valid / vision-expert tokens).
 *     When non-NULL the host-side `rows` argument is only an upper bound used
 *     to size the grid; kernels read the true count from the device, so no
 *     host synchronisation is needed (SURVEY.md §7 "index lists must be built
 *     on-device without a host sync").
 */
#ifndef VIVIDMED_HIP_H
#define VIVIDMED_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VM_OK 0
#define VM_ERR_BAD_ARG (-1)
#define VM_ERR_UNSUPPORTED (-2)
#define VM_ERR_LAUNCH (-3)

/* element types */
#define VM_BF16 0
#define VM_F32 1

/* GEMM epilogue activations */
#define VM_ACT_NONE 0
#define VM_ACT_GELU 1 /* exact erf GELU (torch nn.GELU / ACT2FN['gelu']) */
#define VM_ACT_RELU 2

/* library / device info --------------------------------------------------- */
int vm_version(void);   /* 600 = round 6 (vm_attn_f32_args grew by causal / row_of_pos; head width 112); 510 = VM_TN_GROUP_MAX 24 -> 32; 500 = round 5 (vm_gemm_args grew by workspace / workspace_bytes; vm_gemm_workspace_bytes added; vm_lora_down_fused removed), 400 = round 4
                          * (vm_attn_args grew by workspace / workspace_bytes). The argument structs below only ever GROW at their end; a caller built against an older
                          * header must be rebuilt when this number changes (there is no struct_size field). */
/* fills name[0..len) with the gcnArchName of the current device */
int vm_device_arch(char* name_host, int len);

/* per-kernel profiling with HIP events (used by bench.py for `roofline`).
 * When enabled for a kind, every launch of that kind is bracketed by two events recorded on
 * the launch stream (two hipEventRecord per launch: enable only the kinds you report); vm_prof_collect synchronises those events and returns the
 * summed duration and the summed algorithmic FLOPs. */
int vm_prof_enable(int kind_mask);   /* bit k set: bracket launches of kind VM_PROF_* == k; 0 disables */
int vm_prof_reset(void);
int vm_prof_stride(int every);        /* bracket every `every`-th launch of an enabled kind (uniform sample; default 1) */
int vm_prof_collect(int kind, double* total_ms_host, double* total_flops_host, int64_t* launches_host);
/* algorithmic bytes (operands read once + result written once) summed by the last vm_prof_collect (GEMM kinds) */
int vm_prof_last_bytes(double* bytes_host);
#define VM_PROF_GEMM_BF16 0
#define VM_PROF_GEMM_F32 1
#define VM_PROF_ATTN 2
#define VM_PROF_LORA 3 /* vm_lora_down + vm_gemm_tn_bf16 */

/* ------------------------------------------------------------------------
 * Token routing metadata.
 * Replaces: get_expert_mask            mmmm/models/cogvlm/modeling_cogvlm.py:58-70
 *           boolean gather/scatter     modeling_cogvlm.py:243-245,277-279,95-97
 *           _to_tensor_list / padding  modeling_cogvlm.py:100-128
 *
 * Input  token_type_ids[B,L] int64, attention_mask[B,L] (int64, 0/1).
 * Output (all int32, device):
 *   counts[4]      = { n_vision_rows, n_total_rows, max_seqlen, 0 }
 *   row_of_tok[B*L]  packed row of token (b,l), -1 for padding.
 *                    rows [0,n_vision) are vision-expert tokens, rows
 *                    [n_vision,n_total) language-expert tokens; inside each
 *                    segment tokens keep (b,l) order.
 *   tok_of_row[B*L]  inverse map (b*L+l), -1 for rows >= n_total
 *   cu_seqlens[B+1]  prefix sums of valid tokens per sample
 *   row_of_pos[B*L]  row of the i-th valid token of sample b at index cu_seqlens[b]+i
 *                    (what the var-len attention kernel gathers through)
 *   expert_mask[B*L] uint8: bit0 = vision expert, bit1 = language expert
 *                    (bit-exact restatement of get_expert_mask)
 * One workgroup, no host sync.
 */
int vm_expert_index_build(const int64_t* token_type_ids, const int64_t* attention_mask,
                          int B, int L,
                          int32_t* counts, int32_t* row_of_tok, int32_t* tok_of_row,
                          int32_t* cu_seqlens, int32_t* row_of_pos, uint8_t* expert_mask, void* stream);

/* ------------------------------------------------------------------------
 * bf16 MFMA GEMM, "NT" form:  C[M,N] = act( A[M,K]·B[N,K]^T + ext + bias ) + residual
 * Replaces every nn.Linear on the path: visual.py:93,100,120-122,174-177;
 * modeling_cogvlm.py:55,244-245,278-279,701; peft lora.Linear (conf/lora.yaml).
 *
 * Two-segment grouped form (token-type gated experts, modeling_cogvlm.py:87-98,
 * 243-245, 277-279): rows [0,split) use B/bias/B2, rows [split,M) use
 * B_1/bias_1/B2_1. `split`/`M` may come from the device (counts_dev[0],
 * counts_dev[1]) — then M is the grid upper bound.
 *
 * K-extension (LoRA, K14): ext = alpha2 * A2[M,K2]·B2[N,K2]^T accumulated in the
 * same fp32 accumulator ( y = W x + s·B(A x) with A2 = A x precomputed ).
 * If drop_p > 0 the extension part of the accumulator is multiplied by the
 * inverted-dropout mask of element (row, col) (used by the LoRA dgrad).
 *
 * Requirements: K % 64 == 0, K2 % 64 == 0, lda/ldb/lda2/ldb2 % 8 == 0, ldc % 4 == 0,
 * all base pointers 16-byte aligned. M, N arbitrary. bias/residual have the dtype of C.
 */
typedef struct vm_gemm_args {
  const void* A; int64_t lda;
  const void* B; const void* B_1; int64_t ldb;
  const void* A2; int64_t lda2;
  const void* B2; const void* B2_1; int64_t ldb2;
  int32_t K2; float alpha2;
  const void* bias; const void* bias_1;      /* [N], same dtype as C, or NULL */
  const void* residual; int64_t ldr;         /* [M,N] dtype of C, or NULL */
  void* C; int64_t ldc;
  int32_t M, N, K;
  const int32_t* counts_dev;                 /* NULL, or {split, M_true} on device */
  int32_t split;                             /* host-side split when counts_dev == NULL; <0 = single segment */
  int32_t act;                               /* VM_ACT_* */
  int32_t out_dtype;                         /* VM_BF16 or VM_F32 */
  float drop_p; uint64_t drop_seed;          /* dropout on the extension accumulator */
  float alpha;                               /* scale on the main product (1.0 default) */
  int32_t ksplit;                            /* > 1: split K over that many workgroups per tile and ACCUMULATE into a
                                                pre-zeroed fp32 C with atomics (tiny M x N, long K: the weight gradients of
                                                the mask-decoder hyper-network products). Needs out_dtype VM_F32, no
                                                activation, K2 == 0. 0 / 1 = off. */
  int32_t b_nn;                              /* vm_gemm_bf16 only. != 0: B (and B_1) are given as [K, N] — the contraction index is the ROW,
                                                ldb the row pitch — i.e. a weight W [N_w, K_w] as it sits in HBM is the operand of the input
                                                gradient dx [M, K_w] = dy [M, N_w] . W with K = N_w, N = K_w: no transposed copy of the frozen
                                                weights (35 GB for the 7B decoder + ViT-E) is kept. bf16 output, N % 8 == 0, no split-K. */
  int32_t f32_split;                         /* vm_gemm_f32 only: arithmetic of THIS call. 0 = the process default (vm_gemm_f32_mode),
                                                1 = exact f32 MFMA, 2 = split-bf16 with 3 products, 3 = split-bf16 with 6 products. */
  void* workspace; int64_t workspace_bytes;  /* vm_gemm_bf16 / vm_gemm_fp8: device scratch of >= vm_gemm_workspace_bytes() bytes, ZERO-FILLED once when it is
                                                allocated and owned by ONE stream (launches that may overlap need separate ones), or NULL. With it the library
                                                may run the shape as a stream-K launch — one persistent workgroup per CU, the tiles that do not fill a whole
                                                round over the CUs cut along K, fp32 partial tiles handed over through this scratch — when its cost model
                                                says a partial round would otherwise be paid (e.g. 288 tiles over 256 CUs); results then differ from the
                                                one-tile-per-workgroup kernel by the fp32 rounding of ONE extra addition per element of a split tile, and are
                                                bit-identical from launch to launch (the split points and the summation order are functions of the shape and
                                                of the device-side row counts only). NULL: every tile is computed by one workgroup, as before round 5.
                                                AS SHIPPED the cost model never picks that form (it measured slower on every shape of the six workloads: DESIGN.md
                                                section 3), so passing a workspace changes nothing unless the internal switch vm_gemm_sched_mode_(1 | 2) was set —
                                                which only the tests and tools/ubench/gemm_bench do. Under that switch an owner workgroup that gives up waiting for a
                                                contributor's slab (a bounded spin: it cannot happen while all `multiProcessorCount` workgroups are co-resident, i.e.
                                                while nothing else occupies CUs of the device) marks word `workers` of the flag area behind the slabs; the tests read
                                                it, the library does not: the form is a measured experiment, not a supported mode. */
} vm_gemm_args;

int vm_gemm_bf16(const vm_gemm_args* args_host, void* stream);
/* bytes of vm_gemm_args.workspace that let the library choose the stream-K form on this device (64 MiB + flags on MI355X) */
int vm_gemm_workspace_bytes(int64_t* bytes_host);

/* LoRA down-projection (peft lora.Linear, conf/lora.yaml r = 64): t[M,64] = drop(x)[M,K] · A[64,K]^T with the
 * inverted dropout of lora_dropout fused on the activation fragment ((seed, row*K+col) hash, same mask as
 * vm_dropout). Also computes u = dy · B in the backward (A = B^T). Optional two row segments (gated experts):
 * rows [0,split) use A0, rows [split,M) use A1; split / M from counts_dev when non-NULL. R must be 64, K % 8 == 0. */
int vm_lora_down(const void* x, int64_t ldx, const void* A0, const void* A1, int64_t lda, void* t, int64_t ldt,
                 int M, int K, int R, const int32_t* counts_dev, int split, float drop_p, uint64_t drop_seed,
                 void* workspace, int64_t workspace_bytes, void* stream);
/* fp32 scratch the K-split form of vm_lora_down wants for (M, K) (0: none). Without it the kernel runs one pass per
 * 64-row block, which is correct but leaves most CUs idle for M << 16k. The split sum order is fixed (deterministic). */
int vm_lora_down_workspace(int M, int K, int segmented, int64_t* bytes_host);

/* Row-contraction ("TN") bf16 GEMM for weight gradients: C[P,Q] = alpha * X[M,P]^T · drop(Y)[M,Q] contracted over
 * token rows — dB = dy^T t, dA = u^T drop(x), dW = dy^T x of peft lora.Linear / trainable nn.Linear. Operands stay
 * row-major as stored (transposed LDS reads feed the MFMA); no transposed activation copies.
 * Row range: all rows (bounded by nrows_dev when non-NULL), or with counts_dev: segment 0 = [0,counts[0]),
 * segment 1 = [counts[0],counts[1]), segment -1 = [0,counts[1]).
 * splits > 1 distributes the row range over `splits` workgroups per tile that accumulate with fp32 atomics into C
 * (must be fp32 and zero-filled) — used when P*Q alone gives too few tiles to fill 256 CUs.
 * drop_p > 0 applies the inverted dropout mask of element (row, col) of a [*, drop_cols] tensor to Y. */
int vm_gemm_tn_bf16(const void* X, int64_t ldx, int P, const void* Y, int64_t ldy, int Q, void* C, int64_t ldc,
                    int out_dtype, int M, const int32_t* counts_dev, int segment, const int32_t* nrows_dev, int splits,
                    float alpha, float drop_p, uint64_t drop_seed, int drop_cols, void* stream);

/* fp8 path of BASELINE configs[4] ("model-hr ... fp8 MFMA path"; shapes conf/model-hr.yaml:5-18). The reference has no fp8 arithmetic:
 * this is the build's own reduced-precision mode for the FROZEN base-weight linears of the two bf16 towers (visual.py:93,100,120-122,
 * modeling_cogvlm.py:54-56,243-245,277-279), forward and input gradient; LoRA factors, norms, attention, the heads stay as they are.
 * vm_quant_rows_fp8: x8[r][c] = e4m3(x[r][c] 448 / amax_r) (OCP e4m3fn), scale[r] = amax_r / 448, inv_scale[r] = 1 / scale[r] (optional).
 *   cols % 16 == 0, ld8 % 16 == 0; rows >= nrows_dev[0] are written as zeros with unit scale.
 * vm_gemm_fp8: the NT form of vm_gemm_bf16 with A [M, K], B / B_1 [N, K] holding e4m3 bytes (lda / ldb in bytes, K % 128 == 0) on
 *   v_mfma_f32_16x16x128_f8f6f4 (twice the bf16 matrix rate), fp32 accumulation:
 *     C = act(row_scale[m] col_scale[n] sum_k A8 B8 + alpha2 (A2 B2^T, dropout-masked) + bias) + residual.
 *   The bf16 extension operands must be pre-divided: A2[m][:] / row_scale[m], B2[n][:] / col_scale[n] (B2_1 by col_scale_1). */
int vm_quant_rows_fp8(const void* x, int64_t ldx, void* x8, int64_t ld8, float* scale, float* inv_scale, int rows, int cols, int dtype,
                      const int32_t* nrows_dev, void* stream);
int vm_gemm_fp8(const vm_gemm_args* args_host, const float* row_scale, const float* col_scale, const float* col_scale_1, void* stream);
/* out[r][c] = bf16(x[r][c] * s[r]): the pre-division of the extension operands (s = inv_scale of vm_quant_rows_fp8). cols % 8 == 0. */
int vm_scale_rows_bf16(const void* x, int64_t ldx, const float* s, void* out, int64_t ldo, int rows, int cols, void* stream);

/* fp32 GEMM for the fp32 islands `sam`, `isam_model`, `vg_proj` (mmmm/models/mmmm.py:137-138): every nn.Linear of
 * segvol/modeling/{image_encoder,transformer,mask_decoder}.py and their weight gradients. Same NT form and argument struct (all
 * dtypes f32; K % 32 == 0, K2 % 32 == 0, ld % 4 == 0). Arithmetic (vm_gemm_f32_mode; default 3, or VM_F32_SPLIT in the environment
 * of the first call):
 *   0  v_mfma_f32_16x16x4_f32, the exact f32 fma chain (1/16 of the bf16 matrix rate);
 *   2  split-bf16: a = a0 + a1, b = b0 + b1 with bf16 terms split in registers, a.b ~ a0 b0 + a0 b1 + a1 b0 on
 *      v_mfma_f32_16x16x32_bf16 with fp32 accumulation — products carry 16 mantissa bits. Measured (tools/bench_gemm_f32.py,
 *      tools/f32_mode_accuracy.py): 1.6-2.3x faster than mode 0 on the heads' shapes, 3e-6 relative error per product matrix.
 *      As the PROCESS-WIDE mode (every linear of the islands, the mask decoder's included) it gives 1.2e-5 end to end on SAM-B's
 *      masks and up to 9e-4 on the prompt gradients through iSAM — outside the 1e-4 bar of the fp32 islands, so it is not the
 *      process default. PER LAYER (vm_gemm_args.f32_split = 2) the product uses it for the 12 blocks of the two SAM-B image
 *      encoders only (models/segvol/modeling/image_encoder.py); the mask decoder, through which the prompt gradients flow, keeps
 *      mode 3. Measured in that configuration at TRUE width against the fp32 oracle (tests/test_config0_gpu.py,
 *      profiles/r3_parity_report.json): masks 1.3e-6, boxes 2.8e-7, discriminator 7.3e-7, encoder parameter gradients
 *      8e-7 .. 3.3e-6, mask-decoder / box-head parameter gradients 5e-7 .. 1.2e-6 — all two orders inside the 1e-4 bar;
 *   3  (default) three terms, six products: 24 mantissa bits, i.e. fp32 products — error as mode 0 (4e-7 per product matrix,
 *      1.3e-6 end to end), 1.2-1.4x faster than mode 0. */
int vm_gemm_f32(const vm_gemm_args* args_host, void* stream);
int vm_gemm_f32_mode(int mode);


/* ------------------------------------------------------------------------
 * Row-wise kernels (HBM-bound). `dtype` is the storage type of x/y/dx/dy;
 * statistics and accumulation are fp32.
 */

/* RMSNorm — modeling_cogvlm.py:30-41 (fp32 maths, weight in `dtype`).
 * rstd[rows] (fp32) is saved for the backward. */
int vm_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd,
                   int rows, int cols, float eps, int dtype,
                   const int32_t* nrows_dev, void* stream);
/* dx = d/dx ; dw_accum is an fp32 [cols] accumulator that is atomically ADDED to (a zeroed scratch or an fp32 gradient
 * slot). dx == NULL skips the input gradient, dw_accum == NULL the weight gradient: the two halves may run on different
 * streams (the parameter gradient is off the critical path of backward). Same for vm_layernorm_bwd. */
int vm_rmsnorm_bwd(const void* x, const void* w, const void* dy, const float* rstd,
                   void* dx, float* dw_accum,
                   int rows, int cols, int dtype,
                   const int32_t* nrows_dev, void* stream);

/* LayerNorm (+ optional residual): y = residual + LN(x)*w + b
 * visual.py:129-141 (post-LN on the branch output), image_encoder.py:121-124,
 * transformer.py:743-753, mask_decoder.py:15-26. */
int vm_layernorm_fwd(const void* x, const void* w, const void* b, const void* residual,
                     void* y, float* mean, float* rstd,
                     int rows, int cols, float eps, int dtype, void* stream);
int vm_layernorm_bwd(const void* x, const void* w, const void* dy,
                     const float* mean, const float* rstd,
                     void* dx, float* dw_accum, float* db_accum,
                     int rows, int cols, int dtype, void* stream);
/* The same with the gradient of the residual branch that forked off x summed in: dx = d norm / dx + dx_add (fp32 sum, ONE rounding to
 * `dtype`). In a pre-norm block (`h + f(norm(h))`: modeling_cogvlm.py:323-343, segvol image_encoder.py blocks) h feeds the norm and the
 * residual; autograd would add the two gradients of h with a separate element-wise kernel (three passes over [rows, cols]) per norm. */
int vm_rmsnorm_bwd_res(const void* x, const void* w, const void* dy, const float* rstd, const void* dx_add,
                       void* dx, float* dw_accum, int rows, int cols, int dtype, const int32_t* nrows_dev, void* stream);
int vm_layernorm_bwd_res(const void* x, const void* w, const void* dy, const float* mean, const float* rstd, const void* dx_add,
                         void* dx, float* dw_accum, float* db_accum, int rows, int cols, int dtype, void* stream);

/* RoPE, rotate_half form, looked up by explicit position ids —
 * modeling_cogvlm.py:183-193. In place on q and k inside a packed qkv buffer
 * [rows, 3*H*hd] (q at col 0, k at col H*hd). cos/sin: fp32 tables [n_pos, hd]
 * built on the host exactly as modeling_cogvlm.py:162-170 does (so the
 * bf16 `inv_freq` quirk of SURVEY.md §7 is reproduced by construction).
 * inverse != 0 applies the transposed rotation (the backward). */
int vm_rope_inplace(void* qkv, int64_t ld, const int32_t* row_pos,
                    const float* cos_tab, const float* sin_tab, int n_pos,
                    int rows, int n_heads, int head_dim, int dtype, int inverse,
                    const int32_t* nrows_dev, void* stream);

/* SwiGLU gate: out = silu(gate) * up — modeling_cogvlm.py:55, visual.py:176 */
int vm_silu_mul_fwd(const void* gate, const void* up, void* out, int64_t n, int dtype, void* stream);
int vm_silu_mul_bwd(const void* gate, const void* up, const void* dout,
                    void* dgate, void* dup, int64_t n, int dtype, void* stream);

/* exact GELU — visual.py:121,175; image_encoder.py MLP; mask_decoder.py:287-289 */
int vm_gelu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream);
int vm_gelu_bwd(const void* x, const void* dy, void* dx, int64_t n, int dtype, void* stream);
int vm_relu_bwd(const void* y, const void* dy, void* dx, int64_t n, int dtype, void* stream);

/* inverted dropout with a counter-based hash RNG: y = x * keep(seed, i) / (1-p)
 * (peft lora.Linear lora_dropout, conf/lora.yaml:3). Same (seed, index) ->
 * same mask, so gradient-checkpoint recompute and backward regenerate it. */
int vm_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, int dtype, void* stream);

/* y = a + b (residual adds kept on our own stream/kernel set) */
int vm_add(const void* a, const void* b, void* y, int64_t n, int dtype, void* stream);
/* dtype conversion */
int vm_cast(const void* x, int src_dtype, void* y, int dst_dtype, int64_t n, void* stream);

/* out[r, :] = src[idx[r], :] (idx < 0 -> zeros). Embedding lookup
 * (modeling_cogvlm.py:449) and packed<->padded layout changes. */
int vm_gather_rows(const void* src, int64_t ld_src, const int32_t* idx, void* out, int64_t ld_out,
                   int rows, int cols, int dtype, const int32_t* nrows_dev, void* stream);
/* out[idx[r], :] = src[r, :] for idx[r] >= 0 (image-feature scatter,
 * modeling_cogvlm.py:451-453; attention output scatter :126) */
int vm_scatter_rows(const void* src, int64_t ld_src, const int32_t* idx, void* out, int64_t ld_out,
                    int rows, int cols, int dtype, const int32_t* nrows_dev, void* stream);
/* Embedding weight gradient: rows sorted by id; one workgroup per segment head
 * sums its rows in fp32 (deterministic, no atomics). */
int vm_embedding_bwd(const void* dout, int64_t ld, const int32_t* sorted_ids, const int32_t* sorted_rows,
                     int n, void* dweight, int64_t ld_w, int cols, int dtype, void* stream);

/* out[c, r] = in[r, c]; columns >= rows_true (device count, optional) are zero-filled
 * up to `rows` — and on to min(ld_out, rows rounded up to 64): the pad columns of a K-padded output row need no separate
 * fill — so that the result can be the K-contiguous operand of an NT GEMM (weight-gradient GEMMs contract over tokens).
 * NOTE for callers that transpose INTO A COLUMN SLICE of a wider matrix (ld_out > rows with live data to the right of the slice):
 * columns [rows, min(ld_out, roundup64(rows))) of every output row are OVERWRITTEN with zeros (vm_transpose, vm_transpose_segment and
 * vm_transpose_colsum alike; since round 2). Give such an output a row pitch of exactly `rows`' slice only if rows % 64 == 0, or
 * transpose into a buffer of its own. */
int vm_transpose(const void* in, int64_t ld_in, void* out, int64_t ld_out,
                 int rows, int cols, int dtype, const int32_t* nrows_dev, void* stream);

/* vm_transpose that also accumulates the column sums of `in` atomically into colsum_accum[cols] (fp32): the bias gradient of a trainable
 * linear is the column sum of dy, which its weight gradient transposes anyway (every nn.Linear of the unfrozen heads under segvol/modeling). */
int vm_transpose_colsum(const void* in, int64_t ld_in, void* out, int64_t ld_out, int rows, int cols, int dtype, float* colsum_accum,
                        void* stream);

/* fp32 weight gradient in TN form: C[P, Q] += X[M, P]^T Y[M, Q] with X (= dy) and Y (= x) row-major as autograd holds them — the
 * weight gradients of the unfrozen fp32 islands (every nn.Linear under segvol/modeling, mmmm.py:137-138; README Stage 1 / 3:
 * --model.freeze_sam false) without transposed copies of the activations. Split-bf16 arithmetic as vm_gemm_f32 (f32_split 2: three
 * products, 3: six, 0: the process default; the exact mode 1 is not available here: VM_ERR_UNSUPPORTED). C is ACCUMULATED into
 * (a gradient-bucket slot): plain read-modify-write when one workgroup per tile walks all rows, fp32 atomics when the rows are split.
 * colsum (optional, fp32 [P]): += the column sums of X, i.e. the bias gradient. P % 8 == 0, Q % 8 == 0, ldx % 4 == 0, ldy % 4 == 0. */
int vm_gemm_tn_f32(const float* X, int64_t ldx, int P, const float* Y, int64_t ldy, int Q, float* C, int64_t ldc, int M, float* colsum,
                   int f32_split, void* stream);

/* LoRA factor gradients: row contraction of a wide streamed operand W [M, C] with a rank-64 operand S [M, 64]
 * (peft lora.Linear backward: dB = s * dy^T t, dA = s * u^T drop(x); functional._Linear.backward).
 *   transpose_out == 0: out[c][n] (C rows, 64 columns, ldo)   = [accumulate ? out : 0] + alpha * sum_m W[m][c] S[m][n]
 *   transpose_out == 1: out[n][c] (64 rows, C columns, ldo)   = the same, stored transposed
 * drop_p > 0 applies the inverted-dropout mask of element (m, c) of an [*, C] tensor to W (same hash as vm_dropout).
 * Rows: all M, a routed segment (counts_dev + segment as in vm_gemm_tn_bf16) or nrows_dev; segment == 2 computes BOTH
 * routed segments in one launch: segment 0 -> out, segment 1 -> out1 (the two experts of a gated linear). Partial sums over row
 * ranges go through `workspace` (size from vm_tn_skinny_workspace) and are added in a fixed order: deterministic.
 * C % 8 == 0, ldw % 8 == 0, lds % 8 == 0. out_dtype VM_BF16 or VM_F32. */
int vm_tn_skinny_workspace(int M, int C, int64_t* bytes_host);
int vm_tn_skinny_bf16(const void* W, int64_t ldw, int C, const void* S, int64_t lds, void* out, void* out1, int64_t ldo, int out_dtype,
                      int transpose_out, int accumulate, int M, const int32_t* counts_dev, int segment,
                      const int32_t* nrows_dev, float alpha, float drop_p, uint64_t drop_seed, void* workspace,
                      int64_t workspace_bytes, void* stream);

/* A batch of up to VM_TN_GROUP_MAX independent LoRA factor gradients in ONE launch (the weight gradients a transformer layer's
 * backward produces: peft lora.Linear backward of modeling_cogvlm.py:44-56,87-98,243-245 / visual.py:93-123). Item i:
 *   out (+)= alpha * sum over rows m in its range of drop(W)[m][c] * S[m][n]        (W [M, C] wide, S [M, 64]; bf16)
 * stored as out[c][n] (transpose_out == 0: dB [C, 64]) or out[n][c] (transpose_out == 1: dA [64, C]); `out` is ALWAYS accumulated
 * into (a zeroed or partially filled gradient slot), bf16 (one rounding per call) or fp32. Row range: all M rows, or with
 * counts_dev the routed segment `segment` (0: [0, counts[0]), 1: [counts[0], counts[1]), other: [0, counts[1])).
 * One workgroup per 64 columns of one item walks all of the item's rows: no partial sums, no atomics — deterministic.
 * `block0` is filled in by the call. C % 8 == 0, ldw % 8 == 0, lds % 8 == 0. */
#define VM_TN_GROUP_MAX 32
typedef struct {
  const void* W; int64_t ldw; int32_t C; int32_t M;
  const void* S; int64_t lds;
  void* out; int64_t ldo; int32_t out_f32; int32_t transpose_out;
  const int32_t* counts_dev; int32_t segment; int32_t block0;
  float alpha; float drop_p; uint64_t seed;
} vm_tn_group_item;
int vm_tn_skinny_group_bf16(const vm_tn_group_item* items_host, int n, void* stream);

/* A table of independent small transposes in ONE launch: desc_dev holds n records of six int64
 * {src ptr, dst ptr, rows, cols, ld_src, ld_dst}; dst[c, r] = src[r, c]. Used to refresh the K-contiguous copies of
 * every LoRA factor (peft lora_A/lora_B of scripts/cli.py:82-85) once per optimizer step instead of once per use.
 * tiles_per_entry = grid.x (entries with more 64x64 tiles loop). n <= 65535. */
int vm_transpose_batched(const int64_t* desc_dev, int n, int tiles_per_entry, int dtype, void* stream);

/* Table of fp32 side accumulators folded into bf16 gradient slots in ONE launch: desc_dev holds n records of three int64
 * {dst bf16*, src float*, count}; dst[j] = bf16(dst[j] + bf16(src[j])) (the rounding of AccumulateGrad's `grad += g.to(bf16)`), then
 * src[j] = 0. The column-sum gradients of the bf16 norm layers (RMSNorm modeling_cogvlm.py:30-41, LayerNorm visual.py:129-141) are
 * accumulated atomically in fp32 by vm_rmsnorm_bwd / vm_layernorm_bwd; this moves all of a gradient bucket's at once. n <= 65535. */
int vm_accum_f32_table(const int64_t* desc_dev, int n, int blocks_per_entry, void* stream);

/* same, restricted to one row segment of the token-routed layout:
 * segment 0 = rows [0, counts[0]), segment 1 = rows [counts[0], counts[1]) (device counts);
 * out[c, i] = in[begin + i, c], zero beyond the segment. Feeds the per-expert LoRA weight gradients. */
int vm_transpose_segment(const void* in, int64_t ld_in, void* out, int64_t ld_out,
                         int rows, int cols, int dtype, const int32_t* counts_dev, int segment, void* stream);

/* Fused gradient clip + AdamW over one flat bucket (the optimizer side of the step: conf/phase-vg/fit.yaml gradient_clip_val
 * 1.0 + torch.optim.AdamW). p, g, m, v: n elements of `dtype` (n % 8 == 0 for bf16, % 4 for fp32, 16-byte
 * aligned); maths in fp32. g is scaled by clip_coef_dev[0] (device scalar; NULL = 1) before it enters the moments.
 * step >= 1 is the 1-based step count for the bias corrections. */
int vm_adamw(void* p, const void* g, void* m, void* v, int64_t n, float lr, float beta1, float beta2, float eps,
             float weight_decay, int step, const float* clip_coef_dev, int dtype, void* stream);

/* partials_out[b] = sum of x[i]^2 over workgroup b's share of the n elements (fp32 accumulation; n_partials workgroups, a fixed
 * element -> workgroup map: deterministic). The gradient-norm side of gradient_clip_val (conf/phase-vg/fit.yaml:8-9; the reference's
 * torch.nn.utils.clip_grad_norm_): the caller sums the partials of all buckets and takes the root. n % 8 == 0 (bf16) / % 4 (fp32). */
int vm_sumsq_partials(const void* x, int64_t n, int dtype, float* partials_out, int n_partials, void* stream);

/* out_accum[c] += sum_r x[r, c] (fp32, atomically accumulated: zero it first). Bias gradients. */
int vm_colsum(const void* x, int64_t ld, float* out_accum, int rows, int cols, int dtype,
              const int32_t* nrows_dev, void* stream);

/* ------------------------------------------------------------------------
 * Per-token weighted cross-entropy over the vocabulary —
 * _sample_weighted_ce, modeling_cogvlm.py:610-627, fed by lm_head :701.
 * logits [rows, vocab] (ld elements) in `dtype`; maths fp32.
 * fwd: row_loss[r] = lse_r - logit[r,label_r] (0 if label == -100); lse saved.
 * bwd: dlogits[r,j] = (softmax_rj - [j==label_r]) * row_scale[r]; row_scale is
 *      weight_r / n_valid * dloss, prepared by the host wrapper on the device.
 */
int vm_ce_fwd(const void* logits, int64_t ld, const int64_t* labels, float* row_loss, float* lse,
              int rows, int vocab, int dtype, const int32_t* nrows_dev, void* stream);
int vm_ce_bwd(const void* logits, int64_t ld, const int64_t* labels, const float* lse,
              const float* row_scale, void* dlogits, int64_t ld_d,
              int rows, int vocab, int dtype, const int32_t* nrows_dev, void* stream);

/* ------------------------------------------------------------------------
 * Variable-length (block-diagonal) flash attention, bf16 MFMA, fp32 softmax.
 * Replaces the four xformers memory_efficient_attention sites:
 *   modeling_cogvlm.py:113-128 (BlockDiagonalCausalMask, hd 128)
 *   visual.py:91-99            (BlockDiagonalMask, hd 112, scale 112^-0.5)
 * q,k,v: [rows, H, hd] views with row strides ldq/ldk/ldv (elements); they may
 * alias one packed qkv buffer. Sequences are given by cu_seqlens[n_seq+1] over
 * *sequence positions*; `row_of_pos` (optional) maps a sequence position to
 * its physical row (expert-sorted layout), NULL = identity.
 * out: [rows, H, hd] (ldo), lse: [H, total_pos] fp32.
 * With `row_of_pos` the forward keeps the sequence's slice of it in LDS: max_seqlen <= 16 384 (VM_ERR_UNSUPPORTED beyond).
 */
typedef struct vm_attn_args {
  const void* q; const void* k; const void* v; void* out;
  int64_t ldq, ldk, ldv, ldo;
  float* lse;
  const int32_t* cu_seqlens; int32_t n_seq;
  const int32_t* row_of_pos;
  int32_t total_pos_max;   /* upper bound of cu_seqlens[n_seq] (grid sizing) */
  int32_t max_seqlen;      /* upper bound of any sequence length */
  int32_t n_heads, head_dim;
  float scale;
  int32_t causal;
  /* backward only */
  const void* dout; int64_t lddo;
  void* dq; void* dk; void* dv; int64_t lddq, lddk, lddv;
  float* delta;            /* [H, total_pos] scratch: rowsum(dO*O) */
  /* backward, optional: device scratch of vm_attn_bwd_workspace_bytes() bytes. With it the dK/dV kernel leaves dS^T (bf16, the operand
   * of its own dK product) behind and dQ = dS K is a plain product over it; without it (NULL / too small) dQ recomputes S and dP. */
  void* workspace; int64_t workspace_bytes;
} vm_attn_args;
int vm_attn_fwd_bf16(const vm_attn_args* args_host, void* stream);
int vm_attn_bwd_bf16(const vm_attn_args* args_host, void* stream);
int vm_attn_bwd_workspace_bytes(const vm_attn_args* args_host, int64_t* bytes_out);

/* Trilinear up-sampling of fp32 volumes, F.interpolate(x, size, mode='trilinear', align_corners=False) as used on the mask
 * logits in Sam._predict_masks (segvol/modeling/sam.py:57-87). x [n, di, hi, wi] -> y [n, dout, ho, wo] (n = prompts x channels).
 * bwd: gx [n, di, hi, wi] from gy [n, dout, ho, wo] by GATHERING (one thread per input voxel, fixed order): deterministic,
 * unlike the atomic scatter of ATen's backward. */
int vm_upsample_trilinear3d_fwd(const float* x, float* y, int n, int di, int hi, int wi, int dout, int ho, int wo, void* stream);
int vm_upsample_trilinear3d_bwd(const float* gy, float* gx, int n, int di, int hi, int wi, int dout, int ho, int wo, void* stream);

/* Fused Dice + sigmoid-focal loss of full-resolution mask logits: DiceFocalLoss.dice / .focal, mmmm/models/loss.py:32-56
 * (focal = luolib.losses.sigmoid_focal_loss [external]: the torchvision formula, reduction none).
 * x fp32 [rows, n] logits (rows = prompts x channels, n = D*H*W), target [rows, n] bytes (non-zero = foreground) or NULL
 * (no target: dice = 1, focal against zeros). alpha < 0 = no alpha weighting.
 * fwd: sums[row] = (sum t*p, sum p, sum t, sum focal), out[row] = (dice = 1 - 2 sum(t p) / max(sum t + sum p, 1e-8), sum focal);
 *      one streaming pass + a fixed-order second stage (deterministic); workspace from vm_dice_focal_workspace.
 * bwd: dx[row, i] = g_dice[row] * d dice_row / dx_i + g_focal[row] * d focal_i / dx_i (g_* may be NULL = 0). */
int vm_dice_focal_workspace(int rows, int64_t n, int64_t* bytes_host);
int vm_dice_focal_fwd(const float* x, const unsigned char* target, int rows, int64_t n, float gamma, float alpha, float* sums,
                      float* out, void* workspace, int64_t workspace_bytes, void* stream);
int vm_dice_focal_bwd(const float* x, const unsigned char* target, int rows, int64_t n, float gamma, float alpha, const float* sums,
                      const float* g_dice, const float* g_focal, float* dx, void* stream);

/* Rectangular linear sum assignment (Hungarian matching) of many small cost matrices on the device:
 * InstanceSamLoss._match_instances, segvol/modeling/sam.py:243 (`scipy.optimize.linear_sum_assignment` on the host in the
 * reference: one device->host synchronisation per target). cost[p * ld_prob + r * ld_row + c], fp32, solved in double by the
 * algorithm SciPy uses (Crouse 2016) with its scan order and tie rule: the result equals SciPy's. dims_dev[2p], [2p+1] =
 * rows, cols of problem p (rows <= cols <= 64; rows == 0 marks a padding entry); col4row[p * ld_out + r] = assigned
 * column of row r. max_cols: host upper bound of the column counts (> 64 -> VM_ERR_UNSUPPORTED, solve on the host). */
int vm_lsap_f32(const float* cost, int64_t ld_prob, int64_t ld_row, const int32_t* dims_dev, int32_t* col4row,
                int64_t ld_out, int n_prob, int max_cols, void* stream);

/* Cost matrices of the box-only Hungarian matching of every target of every sample in ONE launch:
 * InstanceSamLoss._match_instances, segvol/modeling/sam.py:178-250 (per target: box_loss(reduce_batch=False) :148-160 of every
 * query against every label box + the discriminator cost of a positive; the cost of a negative on the dummy columns that make the
 * matrix square). desc_dev[p] = {reg, logit, label, n_pos, n_col, nq} (int64; reg -> fp32 [nq, 6] centre-size rows of the
 * target's instance queries, logit -> fp32 [nq], label -> fp32 [n_pos, 6]); cost[p] is [rows x width] fp32, zero outside
 * [nq x n_col] — the layout vm_lsap_f32 reads. match_ce != 0: discriminator cost = disc_weight * (1 - p | p); else
 * disc_weight * focal(logit, 1 | 0) with gamma / alpha (alpha < 0: none). */
int vm_box_match_cost(const int64_t* desc_dev, int n_problems, float* cost, int rows, int width, float l1_weight, float giou_weight,
                      float disc_weight, int match_ce, float gamma, float alpha, void* stream);

/* Instance losses of one sample (InstanceSamLoss.compute_loss, sam.py:252-361, branch without instance masks) in one launch
 * each way. logit fp32 [n_targets, n_queries]; reg fp32 [n_targets, 1 + n_queries, 6] (row 0 of a target = its semantic box:
 * not part of the loss); label fp32 [n_boxes, 6]; match int64 [n_targets, n_queries] = index of the matched label box or < 0.
 * out6 = { focal mean over all entries (label = matched; alpha), focal mean of matched entries vs 1 (no alpha; log only),
 *          focal mean of unmatched entries vs 0 (no alpha; log only), l1 mean over matched pairs (F.l1_loss),
 *          1 - mean box_pair_giou over matched pairs [monai, external; eps = FLT_EPSILON], number of matched entries }.
 * bwd: grad_out[0], [3], [4] are the gradients of out6[0], [3], [4]; d_logit like logit, d_reg like reg (every row written).
 * The sub-gradients follow torch's: min / max split on ties, clamp(min=0) passes at 0, sign(0) = 0. */
int vm_instance_loss_fwd(const float* logit, const float* reg, const float* label, const int64_t* match, int n_targets, int n_queries,
                         float gamma, float alpha, float* out6, void* stream);
int vm_instance_loss_bwd(const float* logit, const float* reg, const float* label, const int64_t* match, int n_targets, int n_queries,
                         float gamma, float alpha, const float* out6, const float* grad_out, float* d_logit, float* d_reg, void* stream);

/* Skinny-M linear of the decode step (one row per sample): out[M,N] = x[M,K] W[N,K]^T + alpha2 * x2[M,K2] W2[N,K2]^T
 * + bias[N], then (rounded to bf16) + residual[M,N] — the language-expert nn.Linear calls of
 * modeling_cogvlm.py:243-245, 277-279, 54-56 and lm_head (:706) when L == 1. bf16 operands, fp32 accumulation.
 * M <= 16, K % 64 == 0, K2 % 64 == 0 (else VM_ERR_UNSUPPORTED: use vm_gemm_bf16). x2 / W2 / bias / residual may be
 * NULL. HBM-bound: W is read once, N/16 workgroups of 16 waves, no workspace, deterministic. */
int vm_gemv_bf16(const void* x, int64_t ldx, const void* W, int64_t ldw, const void* x2, int64_t ldx2, const void* W2, int64_t ldw2,
                 float alpha2, const void* bias, const void* residual, int64_t ldr, void* out, int64_t ldo, int M, int N, int K,
                 int K2, void* stream);

/* Single-query attention against a KV cache: the generation branch of attention_fn
 * (modeling_cogvlm.py:129-141) with the cache handling of VisionExpertAttention.forward (:253-262).
 * The cache of one layer is two bf16 arrays of token-major rows, k_cache / v_cache[b * ld_seq + t * ld_row + h * head_dim + d]
 * for t < kv_lens_dev[b] (valid tokens only: no padding rows, no mask); q[b * ldq + h * head_dim + d] is the rotated
 * query of the one new token of sample b, whose own K / V row has already been appended (kv_lens counts it).
 * out[b * ldo + h * head_dim + d] = softmax_t(bf16(bf16(q * scale) . k_t)) . v_t, fp32 softmax. head_dim 32 / 64 / 128.
 * max_len: host-side upper bound of kv_lens (sizes the launch; no device->host sync). Two launches: per-chunk partials
 * (one wave per sample x head x 32 keys) and their merge; workspace from vm_attn_decode_workspace. Deterministic. */
int vm_attn_decode_workspace(int batch, int n_heads, int head_dim, int max_len, int64_t* bytes_host);
int vm_attn_decode_bf16(const void* q, int64_t ldq, const void* k_cache, const void* v_cache, int64_t ld_row, int64_t ld_seq,
                        const int32_t* kv_lens_dev, void* out, int64_t ldo, int batch, int n_heads, int head_dim, int max_len,
                        float scale, void* workspace, int64_t workspace_bytes, void* stream);

/* fp32 attention (SAM ViT-B encoder image_encoder.py:126-136 hd 64; two-way
 * transformer transformer.py:224-239 hd 96/48, tiny Lq or tiny Lk).
 * Dense batched form: q [Bn, Lq, H, hd], k/v [Bn, Lk, H, hd] given by strides.
 * Also the attention of the towers' fp32 ("32-true") mode — the reference's own precision for BASELINE configs[0]
 * (mmmm.py:468-492 MyPrecision is optional): EVA-ViT-E visual.py:91-99 (hd 112, block-diagonal) and the decoder
 * modeling_cogvlm.py:106-128 (hd 128, `causal`, positions mapped to the packed expert-sorted rows through `row_of_pos`
 * exactly as vm_attn_*_bf16 does). head_dim in {8, 16, 32, 48, 64, 96, 112, 128}. */
typedef struct vm_attn_f32_args {
  const float* q; const float* k; const float* v; float* out;
  int64_t q_bs, q_ls, k_bs, k_ls, v_bs, v_ls, o_bs, o_ls;   /* batch / position strides (elements); head stride = hd */
  float* lse;              /* [Bn, H, Lq] */
  int32_t Bn, Lq, Lk, n_heads, head_dim;
  float scale;
  const int32_t* cu_seqlens; int32_t n_seq;  /* optional varlen (self-attention, Bn == 1) */
  /* backward */
  const float* dout; int64_t do_bs, do_ls;
  float* dq; float* dk; float* dv;           /* same strides as q/k/v */
  float* delta;
  int32_t f32_split;       /* arithmetic of the four products (head_dim 64 only; other head dims always take the exact form):
                              0 / 1 = exact f32 MFMA (v_mfma_f32_32x32x2_f32), 2 = split-bf16 with 3 products (~2^-17 per product),
                              3 = split-bf16 with 6 products (fp32 products) on v_mfma_f32_32x32x16_bf16 — cf. vm_gemm_args.f32_split */
  /* packed self-attention (cu_seqlens given) in the exact arithmetic only; VM_ERR_BAD_ARG otherwise: */
  int32_t causal;          /* key position <= query position within a sequence */
  const int32_t* row_of_pos;   /* optional [cu_seqlens[n_seq]]: physical row (of q, k, v, out, dout, dq, dk, dv) of sequence position
                              cu_seqlens[b] + i — vm_expert_index_build's table; lse / delta stay indexed by position */
} vm_attn_f32_args;
int vm_attn_fwd_f32(const vm_attn_f32_args* args_host, void* stream);
int vm_attn_bwd_f32(const vm_attn_f32_args* args_host, void* stream);

/* ------------------------------------------------------------------------
 * Patch embedding im2col — visual.py:64 / resample.py:55-62 /
 * image_encoder.py:72. image [C, D, H, W] -> cols [n_patch, C*pz*py*px]
 * (K order c,z,y,x = conv3d weight.flatten(1)), patch order d,h,w. */
int vm_im2col3d(const void* image, int C, int D, int H, int W, int pz, int py, int px,
                void* cols, int64_t ld, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * Measurement helper (bench.py's `roofline.peak_measured`; not on the training path): the sustained rate of bare
 * v_mfma_f32_16x16x32_bf16 on THIS device with uniform random operands in registers, two waves per SIMD, every CU busy, for about
 * `seconds` (first half load only, second half timed with HIP events on `stream`). SYNCHRONOUS: returns after the measurement with
 * TFLOP/s in *tflops_host (host memory). SURVEY.md §8d: "re-measure with an MFMA micro-bench on the box and use the measured peak
 * alongside" the nominal 2.5 PFLOP/s. */
int vm_ubench_mfma_bf16(float seconds, float* tflops_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VIVIDMED_HIP_H */
