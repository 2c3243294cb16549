/*
 * comat_hip.h — C ABI of libcomat_hip.so: the MI355X (gfx950) kernels behind the CoMat training step.
 *
 * The reference (CaraJ7/CoMat) has no FFI: its "operator API" for the hot path is the set of Python call
 * conventions listed in SURVEY.md §8(b).  Every entry point below replaces the third-party kernel class that the
 * cited reference line launches (through diffusers / transformers / torch).  Conventions:
 *   - plain C types only; device pointers are raw `void*`; the caller (PyTorch-ROCm) owns every buffer,
 *     including workspaces; the library never allocates, frees or keeps state between calls;
 *   - all work is enqueued on the caller's `hipStream_t` (passed as void*), no internal synchronisation;
 *   - return 0 on success, a negative COMAT_E* code otherwise; `comat_last_error()` gives a thread-local message;
 *   - activations are channels-last token matrices: an image tensor (B,H,W,C) is the row-major matrix
 *     [B*H*W, C]; attention maps are [B, heads, N, L] row-major (== the reference's (B*h, N, 77) layout,
 *     attn_utils/tc_attn_utils.py:198-217);
 *   - dtype codes: COMAT_F32 (fp32 storage, exact-f32 MFMA 32x32x2) and COMAT_BF16 (bf16 storage,
 *     MFMA 32x32x16, fp32 accumulate).  Biases, norm statistics, losses and optimizer state are always fp32.
 */
#ifndef COMAT_HIP_H
#define COMAT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COMAT_ABI_VERSION 8

enum { COMAT_F32 = 0, COMAT_BF16 = 1,
       COMAT_FP8_E4M3 = 2 /* OCP e4m3fn bytes; operand dtype of comat_gemm / comat_conv2d only (comat_fp8_quantize) */ };
enum { COMAT_OK = 0, COMAT_EINVAL = -1, COMAT_ELAUNCH = -2, COMAT_EUNSUPPORTED = -3 };
enum { COMAT_ACT_NONE = 0, COMAT_ACT_SILU = 1, COMAT_ACT_GELU = 2 };

int comat_abi_version(void);
const char* comat_last_error(void);
/* 16 hex digits: a hash of the sources (kernels, headers, plan table, build flags) this library was built from.  Profiles of
 * the library (rocprofv3 counter passes under profiles/) record it; bench.py quotes them only for the library they describe. */
const char* comat_build_id(void);
/* Kernel-selection options for A/B runs, microbenchmarks and the parity tests of every kernel variant: "flash_trim",
 * "flash_tr", "flash_kt", "flash_merge", "gemm2", "gemm2_tt", "g2_cfg", "g2_splits", "force_splits", "norm_fused".  Each defaults to the environment variable
 * COMAT_<NAME> (read once) or its built-in default.  They select among kernels that compute the same function; results
 * differ at most in floating-point summation order. */
int comat_set_option(const char* name, int32_t value);
/* Which kernel served the calling thread's last comat_gemm / comat_gemm_segments / comat_conv2d call: 0 the general
 * 64x64 kernel, 1 the LDS-DMA pipelined kernel, 2 its k-major (transA && transB) variant, 3 its fp8 (e4m3, 32x32x64
 * MFMA) variant, 4 the grouped k-major kernel (comat_gemm_tt_grouped); -1 before the first call.
 * bench.py and the tests use it to attribute time and to assert that a problem runs on the kernel the docs say. */
int comat_last_gemm_kernel(void);

/* ------------------------------------------------------------------------------------------------------------
 * GEMM:  C = act(alpha * op(A) op(B)^T + bias + bias2) + beta * R         (fp32 accumulate on MFMA)
 *   transA == 0: A is [M,K] row-major (lda = row stride);  transA == 1: A is stored [K,M] (lda = stride of k).
 *   transB == 0: B is [N,K] row-major (a torch Linear weight);  transB == 1: B is stored [K,N].
 *   Two-level batch (batch1 x batch2) with independent element strides lets attention heads be addressed in
 *   place inside [tokens, heads*dim] matrices.
 * Replaces: every nn.Linear / 1x1 conv / torch.bmm on the path — UNet to_q/to_k/to_v/to_out + LoRA
 * (training_utils/pipeline.py:94-115), proj_in/out, GEGLU FFN, time embedding; the patched attention's
 * QK^T and `torch.bmm(attention_probs, value)` (attn_utils/tc_attn_utils.py:126-146); BLIP ViT/decoder linears
 * (concept_mat_utils/caption_blip.py:57); the VAE attention (TrainableSDPipeline.py:220).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    const void* A; const void* B; void* C;
    const float* bias;   /* [N] fp32 or NULL */
    const float* bias2;  /* [M / rows_per_bias2, N] fp32 or NULL (per-row-group bias, e.g. time embedding) */
    const void* R;       /* residual, same shape as C, or NULL.  May alias C (accumulate). */
    int64_t M, N, K;
    int64_t lda, ldb, ldc, ldr;
    int64_t batch1, batch2;
    int64_t sA1, sA2, sB1, sB2, sC1, sC2, sR1, sR2;
    int64_t rows_per_bias2;
    float alpha, beta;
    int32_t transA, transB;
    int32_t act;
    int32_t in_dtype;   /* dtype of A and B */
    int32_t out_dtype;  /* dtype of C */
    int32_t r_dtype;    /* dtype of R */
    void* ws;           /* optional caller-owned split-K workspace (NULL: never split), see below */
    int64_t ws_bytes;
    /* in_dtype == COMAT_FP8_E4M3: device pointers to the per-tensor dequantisation scales of A and B (value = scale *
     * fp8), NULL = 1.  The product is scaled by *scale_a * *scale_b before bias / activation / residual.  fp8 operands
     * need transA == transB == 0, K % 64 == 0, lda/ldb % 16 == 0; anything else is COMAT_EINVAL (no silent fallback). */
    const float* scale_a;
    const float* scale_b;
    /* Second epilogue (ABI 5; comat_gemm only; ONE launch for the problems the pipelined kernel takes, else the plain product
     * followed by comat_geglu_il_fwd - epi2 = 2 then needs M * N * 2 bytes of `ws` behind the counters - the same function):
     *   epi2 = 1 / 2: GEGLU.  The N = 2 D output columns hold value and gate channels INTERLEAVED in sixteens (columns
     *   32 t .. 32 t + 15 = value channels 16 t .., columns 32 t + 16 .. 32 t + 31 = their gate channels: the caller permutes
     *   the rows of B and the bias once); C2 [M, D] (leading dimension ldc2) receives value * gelu(gate), both rounded to bf16
     *   first.  epi2 = 1 also stores the pre-activations to C (what the backward pass reads), epi2 = 2 does not (C may be
     *   NULL).  bf16 output, N % 32 == 0, 16-byte aligned rows; no residual, bias2, activation or batch.
     *   epi2 = 3 (ABI 6): the GEGLU BACKWARD.  The product is dF [M, N] (the data-gradient of `ff.net.2`: the gradient of the hidden
     *   activations f = value * gelu(gate)); C2 [M, 2 N] holds the saved pre-activations in the interleaved layout (input), C
     *   [M, 2 N] receives their gradient in the same layout: d value = dF * gelu(gate), d gate = dF * value * gelu'(gate), dF
     *   rounded to bf16 first (the bits of the two-launch form).  bf16, N % 16 == 0, ldc, ldc2 >= 2 N, 16-byte aligned rows,
     *   no bias / residual / activation / batch.
     * (epi2 = 1 .. 3) replace the separate GEGLU kernel behind `ff.net.0.proj` of every BasicTransformerBlock (3P diffusers GEGLU, reached
     * from TrainableSDPipeline.py:144-150): one launch and one [M, 2 D] read less per block, and no [M, 2 D] write at all in
     * the no-grad denoise steps.
     *   epi2 = 4 (ABI 7): TAIL COLUMNS.  The product has N = N1 + n2 columns; its last n2 rows of B come from a second matrix B2
     *   (row-major [n2, K], leading dimension ldb like B, batch stride sB2_tail) and its last n2 COLUMNS go to a second output
     *   C2 [M, n2] (leading dimension ldc2, batch stride sC2_tail, out_dtype) as alpha2 * A B2^T - no bias, activation or residual;
     *   the first N1 = N - n2 columns get the whole first epilogue (bias [N1], bias2 [.., N1], R, act) into C.  transA = transB = 0,
     *   batch2 = 1; n2 and N1 multiples of 8 and 16-byte aligned rows for the one-launch form, else (and outside the pipelined
     *   kernel) two launches with the same results.
     *   One launch replaces a frozen-plus-merged projection and the rank-r product that shares its A operand: the forward
     *   `y = x W_eff^T` with `h = s x D^T` (the LoRA down projection, training_utils/pipeline.py:94-115, needed again by the factor
     *   gradient dU += g^T h), and the data-gradient `dx = g W_eff` with `u = s g U` (needed by dD += u^T x). */
    void* C2;
    int64_t ldc2;
    int32_t epi2;
    /* epi2 = 4 only (ABI 7) */
    const void* B2;
    int64_t n2, sB2_tail, sC2_tail;
    float alpha2;
    /* ABI 8, fp8 operands with batch1 > 1: batch z is scaled by scale_b[z * s_scale_b] (0: one scale for every batch) - the
     * q / k / v projections of one attention share their input (one scale_a) but carry a scale per frozen weight */
    int64_t s_scale_b;
    /* ABI 8, fp8 operands only: a MIXED-PRECISION K TAIL in the same launch,
     *     C = act(alpha * (scale_a * scale_b * A B^T + A2k B2k^T) + bias + bias2) + beta * R
     * with A2k [M, K2] and B2k [N, K2] in bf16 (row strides lda2k / ldb2k, batch1 strides sA2k / sB2k, in elements; K2 a multiple
     * of 16, 16-byte aligned rows; NULL: none).  This is the LoRA branch of a frozen projection under the fp8 forward:
     * y = x8 W8^T (e4m3 MFMA) + h U^T (bf16 MFMA, h = s x D^T) - `training_utils/pipeline.py:94-115` - without a second launch and
     * without the read-modify-write of y. */
    const void* A2k;
    const void* B2k;
    int64_t K2, lda2k, ldb2k, sA2k, sB2k;
    /* ABI 8, epi2 = 1 / 2 on the pipelined kernel only (else COMAT_EINVAL): the GEGLU epilogue ALSO emits q8 [M, N / 2] (leading
     * dimension ldq8) = the e4m3 bytes of value * gelu(gate) under *q_scale, and folds its abs-max into *q_amax - what
     * comat_fp8_quantize_scaled would make of C2, for the `ff.net.2` product that consumes it (delayed scaling, see the fp8 section).
     * With q8 given C2 may be NULL: the bf16 copy is then never written. */
    void* q8;
    const float* q_scale;
    uint32_t* q_amax;
    int64_t ldq8;
} comat_gemm_params;
int comat_gemm(const comat_gemm_params* p, void* stream);

/* Split-K workspace (comat_gemm, comat_gemm_segments, comat_conv2d).  Problems with few output tiles and a long
 * contraction are cut along k; the partial tiles are combined INSIDE the launch by the last-arriving workgroup of
 * each tile, in slice order (bit-reproducible; no reduce launch).  Layout of `ws`:
 *   [0, COMAT_WS_COUNTER_BYTES)  uint32 ticket counters, one per output tile.  The caller zeroes this region ONCE
 *                                 (before the first call that receives the buffer); every launch restores the zeros.
 *   [COMAT_WS_COUNTER_BYTES, ws_bytes)  fp32 partial tiles.
 * One workspace serves any number of calls on ONE stream; calls that may overlap (different streams) need a
 * workspace each.  comat_gemm_workspace_bytes() returns the size that lets a problem use its full planned split (a
 * smaller buffer only lowers the split count).  It does NOT include the scratch of the two-launch forms of the second epilogue
 * (epi2 = 2 or 3 on a problem the pipelined kernel declines: M * N * 2 bytes behind the counters, see comat_gemm_params::epi2):
 * a caller that relies on those forms adds it. */
#define COMAT_WS_COUNTER_BYTES (256 * 1024)
int64_t comat_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t batch, int32_t in_dtype);

/* K-segmented GEMM:  C = act(alpha * sum_s A_s[M, K_s] B_s[N, K_s]^T + bias + bias2) + beta * R   (1 <= nseg <= 8).
 * Every A_s / B_s is k-contiguous (row-major [rows, K_s] with leading dimension lda / ldb); M, N, the dtypes, the
 * epilogue and the workspace come from `p` (its A, B, K, lda, ldb, trans*, batch2 fields are ignored / must be 0).
 * p->batch1 > 1 runs that many independent problems in one launch (q/k/v projections of one activation): C and R
 * advance by p->sC1 / p->sR1 elements, bias (if any) is [batch1, N], each segment's operands by sA / sB.
 * One launch replaces the chains  y = x W^T + b;  y += (x D^T) U^T  (LoRALinearLayer added to a frozen nn.Linear,
 * training_utils/pipeline.py:95-115) and  dx = sum_i g_i W_i + u D  (the data-gradient of q/k/v projections that
 * share one input plus their low-rank branches). */
typedef struct {
    const void* A; const void* B;
    int64_t K, lda, ldb;
    int64_t sA, sB;   /* element stride of A_s / B_s between batch items (p->batch1 items; 0 = shared operand) */
} comat_gemm_segment;
int comat_gemm_segments(const comat_gemm_params* p, const comat_gemm_segment* segs, int32_t nseg, void* stream);

/* Grouped k-major products:  C_p[M_p, N_p] += A_p^T B_p  for nprob INDEPENDENT problems, fp32 accumulation in place.
 *   A_p is stored [K_p, M_p] (lda = element stride between k-rows), B_p is stored [K_p, N_p], both bf16; C_p is fp32
 *   row-major with leading dimension ldc.  M_p, N_p multiples of 8 (>= 8); any K_p >= 1; lda, ldb multiples of 8, ldc of 4;
 *   operands 16-byte aligned.  The C_p of one call must not overlap each other (the problems run concurrently); two calls
 *   on one stream are ordered.  A violated requirement is COMAT_EINVAL before anything is launched.
 * These are the LoRA weight gradients of a backward pass - dU = g^T h and dD = u^T x contract over the token axis of two
 * row-major token matrices - which the reference leaves to autograd's addmm, one small launch per factor
 * (training_utils/pipeline.py:84-115: LoRALinearLayer on to_q / to_k / to_v / to_out.0 of every Attention; ~720 factors
 * gradients per SD1.5 step).  Nothing reads them before the optimizer, so the caller queues them and hands them over in
 * groups: one launch per <= 48 problems fills the chip, and the split-K combines (in-launch, fixed slice order:
 * bit-reproducible for a given grouping) overlap across problems.  `ws` as for comat_gemm (zeroed ticket counters at its
 * head); without a workspace no problem is split. */
typedef struct {
    const void* A; const void* B; void* C;
    int64_t M, N, K;
    int64_t lda, ldb, ldc;
} comat_tt_problem;
int comat_gemm_tt_grouped(const comat_tt_problem* probs, int32_t nprob, int32_t in_dtype, void* ws, int64_t ws_bytes,
                          void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * conv2d as implicit GEMM on channels-last tensors.
 *   X: [B, Hin, Win, Cin], W: [Cout, KH, KW, Cin] (K-contiguous), Y: [B, Hout, Wout, Cout].
 *   mode 0 (forward gather):    src = dst*stride + k - pad   (in the `ups`-times nearest-upsampled input)
 *   mode 1 (transposed gather): src = (dst + k - pad)/stride, taken only when divisible  (dgrad of a strided conv)
 *   The data-gradient of a stride-1 conv is mode 0 with the tap-flipped, channel-transposed weight.
 *   Epilogue as in comat_gemm (bias [Cout], bias2 [B, Cout] = per-sample time-embedding add, residual R).
 * Replaces: cuDNN conv2d of ResnetBlock2D / Downsample2D / Upsample2D / conv_in / conv_out in the UNet call
 * (TrainableSDPipeline.py:144-150), the discriminator UNet (training_utils/gan_sdxl.py:72-88,112-131) and the
 * VAE decoder (TrainableSDPipeline.py:220).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    const void* X; const void* W; void* Y;
    const float* bias; const float* bias2; const void* R;
    int32_t B, Hin, Win, Cin, Hout, Wout, Cout;
    int32_t KH, KW, stride, pad, mode, ups;
    float alpha, beta;
    int32_t act;
    int32_t in_dtype, out_dtype, r_dtype;
    void* ws;           /* optional caller-owned split-K workspace (layout: see comat_gemm) */
    int64_t ws_bytes;
    const float* scale_a;  /* in_dtype == COMAT_FP8_E4M3 (mode 0, Cin % 64 == 0): per-tensor scales of X and W, as in */
    const float* scale_b;  /* comat_gemm_params */
} comat_conv_params;
int comat_conv2d(const comat_conv_params* p, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * GroupNorm (+ optional fused SiLU) on [B, HW, C] channels-last, G groups.  stats: [B, G, 2] fp32 (mean, rstd),
 * written by fwd and read by bwd.  ws: caller workspace of COMAT_GN_WS_DOUBLES(B, G) doubles: 4 KiB of uint32 ticket
 * counters (the caller zeroes them ONCE; the kernels re-arm them) followed by the final sums and up to 1024 per-block
 * partial slabs.  The statistics pass and their fixed-order combination (by the last-arriving block of each sample)
 * are ONE launch, the normalisation a second one; sums are bit-reproducible.  One workspace per stream.
 * gamma/beta fp32 [C].  bwd returns dx only (norm affine parameters are frozen: training_utils/pipeline.py:68-70);
 * `add` (same layout and dtype as dx, or NULL) is added to dx: the gradient of the branch that bypasses the norm (a
 * ResnetBlock's shortcut, a transformer block's residual), which autograd would otherwise sum in a kernel of its own.
 * Replaces: torch GroupNorm + SiLU in ResnetBlock2D / Transformer2DModel / VAE decoder.
 * ---------------------------------------------------------------------------------------------------------- */
#define COMAT_GN_WS_DOUBLES(B, G) (512 + (int64_t)(B) * (G) * 2 * 1025)
int comat_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, double* ws,
                        int32_t B, int64_t HW, int32_t C, int32_t G, float eps, int32_t silu, int32_t dtype,
                        void* stream);
int comat_groupnorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* stats,
                        void* dx, double* ws, int32_t B, int64_t HW, int32_t C, int32_t G, int32_t silu,
                        const void* add, int32_t dtype, void* stream);

/* LayerNorm over the last dim of [M, C]; stats [M, 2] fp32 (mean, rstd).  bwd: `add` as in comat_groupnorm_bwd. */
int comat_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, int64_t M,
                        int32_t C, float eps, int32_t dtype, void* stream);
int comat_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* stats, void* dx, int64_t M,
                        int32_t C, const void* add, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Row softmax of attention scores.  S: [rows, cols] (s_dtype), P: [rows, cols] (p_dtype).
 *   causal != 0: row r attends keys j <= (r % q_len) + causal_offset.   key_mask: int8 [rows / rows_per_mask, cols]
 *   (1 = keep) or NULL.  P is the materialised probability map that the attention-store clones
 *   (attn_utils/tc_attn_utils.py:60-68): written once, coalesced, never copied.
 * bwd: dS = scale * P * (dP - sum_j dP*P).
 * ---------------------------------------------------------------------------------------------------------- */
int comat_softmax_fwd(const void* S, void* P, int64_t rows, int32_t cols, int32_t q_len, int32_t causal,
                      int32_t causal_offset, const int8_t* key_mask, int64_t rows_per_mask, int32_t s_dtype,
                      int32_t p_dtype, void* stream);
int comat_softmax_bwd(const void* P, const void* dP, void* dS, int64_t rows, int32_t cols, float scale,
                      int32_t p_dtype, int32_t dp_dtype, int32_t ds_dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fused (flash-style) attention for layers whose probability map is not captured: O = softmax(scale Q K^T) V per
 * (batch, head), scores never written to HBM.  Q: [B*Nq, ldq], K/V: [B*Nk, ldk/ldv], O: [B*Nq, ldo] with head h in
 * columns h*d .. (h+1)*d; lse: [B, H, Nq] fp32 log-sum-exp of the scaled scores (saved for backward).
 * Head dim d <= 160, multiple of 8 (bf16) / 4 (fp32); leading dims and base pointers 16-byte aligned.
 * bwd: Dbuf [B, H, Nq] fp32 caller workspace; dO has the layout of O; dQ/dK/dV the layouts of Q/K/V.  ws (optional,
 * fp32, ws_bytes): when there are few keys (cross-attention to 77 text tokens) the dK/dV pass cuts its query loop
 * into ranges that write fp32 partials here and are summed in fixed order (bit-reproducible); NULL: never split.
 * Replaces the materialised softmax(QK^T)V of the reference's patched Attention.forward
 * (attn_utils/tc_attn_utils.py:126-146) for self-attention (268 MB of fp16 scores per layer per sample at 64x64).
 * ---------------------------------------------------------------------------------------------------------- */
int comat_flash_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* lse, int32_t B, int32_t H,
                         int32_t Nq, int32_t Nk, int32_t d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                         float scale, int32_t dtype, void* stream);
/* ABI 8 (fp8 forward, delayed scaling - see the fp8 section below): the same forward that ALSO stores q8 [B*Nq, ldq8] = the e4m3 bytes of
 * its rounded output under *q_scale (layout of O) and folds max |O| into *q_amax: what comat_fp8_quantize_scaled would make of O,
 * for the `to_out` projection that consumes it (attn_utils/tc_attn_utils.py:140-146 feeds `attn.to_out[0]`). */
int comat_flash_attn_fwd_q(const void* Q, const void* K, const void* V, void* O, float* lse, int32_t B, int32_t H,
                           int32_t Nq, int32_t Nk, int32_t d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                           float scale, int32_t dtype, void* q8, int64_t ldq8, const float* q_scale, uint32_t* q_amax,
                           void* stream);
int comat_flash_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                         const float* lse, float* Dbuf, void* dQ, void* dK, void* dV, int32_t B, int32_t H,
                         int32_t Nq, int32_t Nk, int32_t d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                         float scale, int32_t dtype, float* ws, int64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Elementwise family (HBM-bound, 16-byte vectorised).
 * ---------------------------------------------------------------------------------------------------------- */
enum { COMAT_UN_COPY = 0, COMAT_UN_SILU = 1, COMAT_UN_GELU = 2, COMAT_UN_AFFINE = 3 };
/* y = f(x) (AFFINE: p0*x + p1); dtype conversion allowed (COPY == cast). */
int comat_unary(int32_t op, const void* x, void* y, int64_t n, float p0, float p1, int32_t x_dtype, int32_t y_dtype,
                void* stream);
/* dx = dy * f'(x) */
int comat_unary_bwd(int32_t op, const void* dy, const void* x, void* dx, int64_t n, int32_t dtype, void* stream);
/* out = a*x + b*y (y may be NULL) */
int comat_axpby(float a, const void* x, float b, const void* y, void* out, int64_t n, int32_t x_dtype,
                int32_t y_dtype, int32_t out_dtype, void* stream);
/* GEGLU: x [M, 2D] -> y[m, d] = x[m, d] * gelu(x[m, D + d]) */
int comat_geglu_fwd(const void* x, void* y, int64_t M, int32_t D, int32_t dtype, void* stream);
int comat_geglu_bwd(const void* dy, const void* x, void* dx, int64_t M, int32_t D, int32_t dtype, void* stream);
/* The same on the INTERLEAVED layout of the fused projection (comat_gemm_params::epi2): x, dx [M, 2 D] with value / gate
 * channels interleaved in sixteens, y, dy [M, D]; D % 16 == 0, 16-byte aligned operands.  16-byte accesses. */
int comat_geglu_il_fwd(const void* x, void* y, int64_t M, int32_t D, int32_t dtype, void* stream);
int comat_geglu_il_bwd(const void* dy, const void* x, void* dx, int64_t M, int32_t D, int32_t dtype, void* stream);
/* strided 2-D copy (channel concat / split): dst[r, c] = src[r, c] for r < rows, c < cols */
int comat_copy2d(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int64_t cols,
                 int32_t src_dtype, int32_t dst_dtype, void* stream);
/* Two strided 2-D copies of the same row count in ONE launch (ABI 6): dst_i[r, c] = src_i[r, c] for c < cols_i.  The channel
 * concat of a UNet skip connection (`torch.cat([hidden_states, res_hidden_states], dim=1)` in every up-path ResnetBlock of the 3P
 * UNet, reached from TrainableSDPipeline.py:144-150) and its backward split.  16-byte accesses: cols_i and every leading dimension
 * in whole 16-byte vectors, 16-byte aligned pointers (else COMAT_EINVAL: use comat_copy2d twice). */
int comat_copy2d_pair(const void* src0, int64_t ld_src0, void* dst0, int64_t ld_dst0, int64_t cols0, const void* src1,
                      int64_t ld_src1, void* dst1, int64_t ld_dst1, int64_t cols1, int64_t rows, int32_t dtype, void* stream);
/* out[r, c] = x[r, c] + v[c] broadcast over rows (positional embeddings) */
int comat_add_rowvec(const void* x, const void* v, void* out, int64_t rows, int64_t cols, int32_t dtype,
                     void* stream);
/* 2x2 sum pooling [B, 2H, 2W, C] -> [B, H, W, C]: adjoint of the nearest-2x upsample fused in comat_conv2d. */
int comat_sumpool2x2(const void* x, void* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream);
/* Batched transpose + cast of many small fp32 matrices in ONE launch (the per-optimizer-step refresh of the
 * transposed compute-dtype copies of the LoRA down factors, which lets every LoRA data-gradient run k-contiguous).
 * tiles: device int64 [n_tiles, 6] = (src_off, dst_off, rows, cols, r0, c0) per 32x32 tile;
 * dst[dst_off + c*rows + r] = cast(src[src_off + r*cols + c]). */
int comat_transpose_cast_tiles(const float* src, void* dst, const int64_t* tiles, int64_t n_tiles, int32_t out_dtype,
                               void* stream);
/* Merged LoRA weights of MANY projections in ONE launch (ABI 6; once per optimizer step):
 *     Wm_p[N, K] = bf16(W_p + scale * U_p[N, r] D_p[r, K]),     WmT_p[K, N] = Wm_p^T
 * i.e. the weight of the function the reference's LoRA-wrapped Linear computes, y = W x + s * up(down(x))
 * (training_utils/pipeline.py:94-115).  The trained and the no-grad calls of a step then run ONE plain GEMM per projection
 * (forward on Wm, data-gradient on WmT) instead of a rank-r product in front of a K-segmented one.
 *   problems: device int64 [n_problems, 10] = (W, U, Dt, Wm, WmT: device addresses; WmT may be 0), N, K, r, ldu, lddt:
 *             W, Wm bf16 [N, K] contiguous; WmT bf16 [K, N] contiguous; U bf16 [N, r] with leading dimension ldu;
 *             Dt = D^T bf16 [K, r] with leading dimension lddt.  N % 8 == K % 8 == 0, r % 16 == 0, ldu % 8 == lddt % 8 == 0,
 *             every address 16-byte aligned.
 *   tiles:    device int32 [n_tiles, 3] = (problem, n0, k0), one 64 x 64 output tile each (a launch may cover any subset). */
int comat_lora_merge(const int64_t* problems, const int32_t* tiles, int64_t n_tiles, float scale, void* stream);
/* NCHW <-> NHWC permutation of small boundary tensors (latents, images). to_nhwc != 0: [B,C,H,W] -> [B,H,W,C]. */
int comat_permute_nchw_nhwc(const void* x, void* y, int32_t B, int32_t C, int32_t H, int32_t W, int32_t to_nhwc,
                            int32_t x_dtype, int32_t y_dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Classifier-free guidance + DDPM ancestral step, fused (fp32, tiny):
 *   eps = e_u + s (e_c - e_u);   x_prev = cx * x + ce * eps + sigma * z
 * with cx, ce the closed-form coefficients of DDPMScheduler.step.  eps2 holds [uncond; cond].
 * Replaces TrainableSDPipeline.py:155-157,166.  bwd: dx = cx*g, de_u = ce*(1-s)*g, de_c = ce*s*g.
 * ---------------------------------------------------------------------------------------------------------- */
int comat_cfg_ddpm_fwd(const float* x, const void* eps2, const float* z, float* x_prev, int64_t n, float s,
                       float cx, float ce, float sigma, int32_t eps_dtype, void* stream);
int comat_cfg_ddpm_bwd(const float* g, float* dx, void* deps2, int64_t n, float s, float cx, float ce,
                       int32_t eps_dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Image path between VAE and BLIP (training_script.py:606-611, concept_mat_utils/caption_blip.py:33-36,45):
 * separable resampling with precomputed sparse taps (crop + antialiased bicubic + per-channel affine in one
 * pass); the same kernel with transposed tap tables is its adjoint.
 *   in: [B, Hin, Win, C], out: [B, Hout, Wout, C];  ystart[Hout], ywt[Hout, KT], xstart[Wout], xwt[Wout, KT];
 *   out = scale[c] * sum_ty sum_tx ywt*xwt*in[ystart+ty, xstart+tx, c] + shift[c]
 * ---------------------------------------------------------------------------------------------------------- */
int comat_resample2d(const void* in, void* out, int32_t B, int32_t Hin, int32_t Win, int32_t Hout, int32_t Wout,
                     int32_t C, const int32_t* ystart, const float* ywt, const int32_t* xstart, const float* xwt,
                     int32_t KT, const float* scale, const float* shift, int32_t in_dtype, int32_t out_dtype,
                     void* stream);
/* ViT patch embedding gather: img [B, H, W, C] -> patches [B*(H/P)*(W/P), P*P*C] (fwd) and its adjoint (bwd). */
int comat_patchify(const void* img, void* patches, int32_t B, int32_t H, int32_t W, int32_t C, int32_t P,
                   int32_t inverse, int32_t dtype, void* stream);
/* rows of a table: out[i, :] = table[ids[i], :] */
int comat_embedding(const int64_t* ids, const void* table, void* out, int64_t n, int32_t dim, int64_t vocab,
                    int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Losses.
 * ---------------------------------------------------------------------------------------------------------- */
/* Token cross-entropy with ignore_index and label smoothing (BLIP decoder LM loss, caption_blip.py:51-58).
 * logits [T, V]; labels int64 [T] (already shifted by the caller); logp[T] = log-prob of the label ("token-level
 * concept score", 0 where ignored); loss_sum_cnt[2] = {sum of per-token losses, number of valid tokens}. */
int comat_cross_entropy_fwd(const void* logits, const int64_t* labels, float* logp, float* row_lse,
                            float* row_loss /* [T] workspace */, float* loss_sum_cnt, int64_t T, int32_t V, int64_t ld,
                            int32_t ignore_index, float label_smoothing, int32_t dtype, void* stream);
/* dlogits = (g_up[0] / loss_sum_cnt[1]) * (softmax - smoothed one-hot) on valid rows, 0 elsewhere.  The upstream
 * gradient and the valid-token count are DEVICE scalars: no host synchronisation in the middle of backward. */
int comat_cross_entropy_bwd(const void* logits, const int64_t* labels, const float* row_lse, void* dlogits,
                            int64_t T, int32_t V, int64_t ld, int32_t ignore_index, float label_smoothing,
                            const float* g_up, const float* loss_sum_cnt, int32_t dtype, void* stream);
/* Discriminator head (training_utils/gan_sdxl.py:32-35,83-88,124-131): pred = x[p, 0:4] . w + b per pixel,
 * BCE-with-logits against target[p / pix_per_sample], mean over pixels.  x: [P, 4] channels-last UNet output.
 * fwd writes loss[0]; bwd writes dx [P,4] and accumulates dw[4], db[1] (fp32, caller zeroes). */
int comat_disc_head_fwd(const void* x, const float* w, const float* b, const float* target, float* loss,
                        float* ws /* >= 512 floats */, int64_t P, int64_t pix_per_sample, int32_t dtype, void* stream);
/* g_up: upstream gradient, a DEVICE scalar.  dwb: [5] = {dw[0..3], db}, accumulated (caller zeroes) or NULL;
 * ws: >= 512*5 floats.  All reductions of this family are two-stage with a fixed order (bit-reproducible). */
int comat_disc_head_bwd(const void* x, const float* w, const float* b, const float* target, const float* g_up,
                        void* dx, float* dwb, float* ws, int64_t P, int64_t pix_per_sample, int32_t dtype,
                        void* stream);

/* Attribute-concentration losses on one captured cross-attention map (attn_utils/tc_loss_utils.py:104-167).
 *   amap: [heads, res*res, L] probabilities of ONE sample and ONE layer; mask: [n_obj, res*res] fp32 {0,1};
 *   tok_idx: int32 [n_tok] token positions, tok_obj: int32 [n_tok] owning object.
 * Stage 1 (HBM-bound strided gather, one pass over the map):
 *   num[h, t] = sum_px A[h,px,tok_t] * mask[obj_t, px];  den[h, t] = sum_px A[h,px,tok_t];
 *   avg[t, px] += (1/heads) * A[h,px,tok_t]   (head-mean map per token, accumulated across layers by the caller)
 * The (tiny) remaining reductions are host-side torch on [heads, n_tok] and [n_tok, res*res] tensors.
 * bwd: dA[h,px,tok_t] = sum over the listed t of g_num[h,t]*mask + g_den[h,t] + g_avg[t,px]/heads, zero elsewhere: the
 * kernel writes EVERY element of dA (no zero-fill by the caller).
 * Both passes move the map as one coalesced stream staged through LDS (16-byte accesses) and pick / place the token
 * columns there. */
int comat_attnmap_gather_fwd(const void* amap, const float* mask, const int32_t* tok_idx, const int32_t* tok_obj,
                             float* num, float* den, float* avg,
                             float* ws /* (ceil(npix/128)*2 + npix) * heads * n_tok floats */,
                             int32_t heads, int32_t npix, int32_t L, int32_t n_tok, int32_t dtype, void* stream);
int comat_attnmap_gather_bwd(const float* g_num, const float* g_den, const float* g_avg, const float* mask,
                             const int32_t* tok_idx, const int32_t* tok_obj, void* damap, int32_t heads,
                             int32_t npix, int32_t L, int32_t n_tok, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Optimizer tail on the flat fp32 LoRA buffers (training_script.py:661-664,692-694).
 * ---------------------------------------------------------------------------------------------------------- */
/* out[0] += sum(x^2)  (caller zeroes out); ws: >= 1024 floats.  Fixed summation order: every data-parallel rank gets
 * the bit-identical norm (and clip factor) from the all-reduced gradient. */
int comat_sumsq(const float* x, int64_t n, float* out, float* ws, void* stream);
/* AdamW with the global-norm clip folded in: g' = s g * min(1, max_norm / (s sqrt(*gnorm_sq) + 1e-6)), s = grad_scale:
 * the gradient buffer (and *gnorm_sq, the sum of ITS squares) may hold the SUM over data-parallel ranks, grad_scale =
 * 1 / world makes it the mean (DDP semantics, training_script.py:659) without another pass over the buffer.  A non-finite
 * *gnorm_sq skips the update entirely (p, m, v untouched): the inf/NaN check of a mixed-precision optimizer step.
 * Step count t of the bias correction: `step` (host value, >= 1) when step_dev is NULL; otherwise *step_dev + 1 with
 * the count of APPLIED updates kept in device memory — advanced by comat_adamw_tick after the adamw launches of one
 * optimizer step, and only when the update was applied (so a skipped step does not run the bias correction ahead of
 * the moments, and a captured hipGraph of the whole step replays with the right count). */
int comat_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                float eps, float weight_decay, int32_t step, const int32_t* step_dev, const float* gnorm_sq,
                float max_norm, float grad_scale, void* stream);
/* counters[0] += 1 if *gnorm_sq is finite (update applied), else counters[1] += 1 (update skipped). */
int comat_adamw_tick(int32_t* counters, const float* gnorm_sq, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Per-tensor fp8 (OCP e4m3fn) quantisation: the operand format of the fp8-forward configuration (BASELINE.json
 * configs[4]: "fp8 MFMA UNet forward with bf16 backward"; the reference itself trains in fp16 autocast,
 * training_script.py:449-456 - fp8 is this project's MI355X target, restated on the CPU by oracle/fp8.py).
 *   comat_fp8_scale:    *scale = max(max_i |x_i|, 2^-100) / 448      (one launch; ws: 8 bytes the caller zeroed once).  ABI 8:
 *                       amax_bits (may be NULL) = the site's running maximum below, which receives the tensor's abs-max too
 *   comat_fp8_quantize: y_i = e4m3fn(x_i * (1 / *scale)), round to nearest even, saturating; y: n bytes
 * x: n elements of `dtype` (fp32 or bf16), 16-byte aligned.  Feed y and scale to comat_gemm / comat_conv2d with
 * in_dtype = COMAT_FP8_E4M3.
 *
 * DELAYED SCALING (ABI 8).  A quantisation SITE (the input of one frozen layer) owns two device words: `scale` (fp32) and
 * `amax_bits` (the float bits of a running abs-max, zeroed once).  During an optimizer step every tensor that passes the site
 * is quantised under the scale that is already there and folds its own abs-max into amax_bits; once per optimizer step
 * comat_fp8_scales_update turns every running maximum that saw a tensor into next step's scale and re-arms it.  One launch per
 * tensor, no reduction anybody waits for - and none at all where the producer of the tensor emits the bytes itself:
 *   comat_fp8_quantize_scaled: y_i = e4m3fn(x_i * (1 / *scale)); *amax_bits = max(*amax_bits, bits(max_i |x_i|))
 *   comat_fp8_scales_update:   for i < n with amax_bits[i] != 0: scale[i] = max(float(amax_bits[i]), 2^-100) / 448, amax_bits[i] = 0
 *   comat_layernorm_fwd_q / comat_groupnorm_fwd_q: the normalisation of comat_layernorm_fwd / comat_groupnorm_fwd that ALSO
 *       stores q8 = the e4m3 bytes of its (rounded) output y under *scale and folds max |y| into *amax_bits: the bits of
 *       comat_fp8_quantize_scaled(y), one launch and one read of y less.  Only the vectorised forms (comat_*_fwd_q_ok -> 1);
 *       other shapes return COMAT_EUNSUPPORTED and the caller runs the two calls.
 * (The reference has no fp8 path: nothing to cite; arithmetic restated by oracle/fp8.py.)
 * ---------------------------------------------------------------------------------------------------------- */
int comat_fp8_scale(const void* x, int64_t n, int32_t dtype, float* scale, void* ws, uint32_t* amax_bits, void* stream);
int comat_fp8_quantize(const void* x, int64_t n, int32_t dtype, const float* scale, void* y, void* stream);
int comat_fp8_quantize_scaled(const void* x, int64_t n, int32_t dtype, const float* scale, void* y, uint32_t* amax_bits,
                              void* stream);
int comat_fp8_scales_update(uint32_t* amax_bits, float* scale, int32_t n, void* stream);
int comat_layernorm_fwd_q_ok(int32_t C, int32_t dtype);
int comat_layernorm_fwd_q(const void* x, const float* gamma, const float* beta, void* y, float* stats, int64_t M, int32_t C,
                          float eps, int32_t dtype, void* q8, const float* scale, uint32_t* amax_bits, void* stream);
int comat_groupnorm_fwd_q_ok(int32_t B, int64_t HW, int32_t C, int32_t G, int32_t dtype);
int comat_groupnorm_fwd_q(const void* x, const float* gamma, const float* beta, void* y, float* stats, double* ws, int32_t B,
                          int64_t HW, int32_t C, int32_t G, float eps, int32_t silu, int32_t dtype, void* q8, const float* scale,
                          uint32_t* amax_bits, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* COMAT_HIP_H */
