/* afft_hip.h -- C-ABI of the MI355X (gfx950) kernels behind the AFFT hot path.
 *
 * The reference (zeyun-zhong/AFFT) is pure Python/PyTorch and has NO FFI of its own
 * ("Native components: none"), so this ABI is what a maintainer would bind
 * from the reference's Python modules with ctypes.  Every entry point
 * names the reference code whose arithmetic it replaces (paths relative to the reference).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch allocator); the library
 *     never allocates, frees or retains device memory (scratch space is passed in: afft_gemm_t.workspace,
 *     the `partial` argument of afft_layernorm_bwd, the workspaces of the composite entry points);
 *     `stream` is a hipStream_t passed as void*;
 *   - no entry point synchronises the host; all work is enqueued on `stream`;
 *   - entry points may be called from several host threads and for several devices at once (the device is
 *     the calling thread's current HIP device); a workspace must not be shared by two streams;
 *   - return value 0 = success; non-zero = error, text via afft_last_error() (thread-local);
 *   - matrices are row-major with explicit element strides; dtype codes below.
 */
#ifndef AFFT_HIP_H
#define AFFT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { AFFT_F32 = 0, AFFT_BF16 = 1,
       AFFT_F16 = 2 };   /* fp16 planes of the "fp16x2" precision: outputs / copies / images only (a GEMM's operand dtype stays AFFT_BF16 =
                          * "16-bit planes", afft_gemm_t.split3 = 2 says they are fp16) */
enum { AFFT_ACT_NONE = 0, AFFT_ACT_GELU_ERF = 1, AFFT_ACT_GELU_TANH = 2,
       AFFT_ACT_DGELU_ERF = 3,   /* v *= d/du gelu_erf(aux[m,n])  (backward of nn.GELU)  */
       AFFT_ACT_DGELU_TANH = 4,  /* v *= d/du gelu_new(aux[m,n])  (backward of HF gelu_new) */
       AFFT_ACT_RELU = 5,        /* nn.ReLU: MATT (models/fusion.py:40-46), NonLinear mapping (feature_mapping.py:95) */
       AFFT_ACT_SIGMOID_GATE = 6 };/* v = aux[m,n] * sigmoid(v): ContextGating = glu(cat(x, fc(x))) (feature_mapping.py:22-31) */
enum { AFFT_MASK_NONE = 0, AFFT_MASK_DIAG = 1, AFFT_MASK_CAUSAL = 2, AFFT_MASK_BLOCKCAUSAL = 3 };

/* Dropout description shared by the kernels that apply or replay a dropout mask.  keep(idx) is a pure
 * function of (key, element index), so backward passes regenerate the forward mask instead of storing it.
 * Element index of [m, n] in an [M, N] tensor is m*N + n.  p = 0 disables.  DropPath (stochastic depth,
 * models/transformerblock.py:96-104): rows m with equal m / path_group form one sample. */
typedef struct {
  float p;            uint32_t key;        /* nn.Dropout(p) on elements */
  float path_p;       uint32_t path_key;   /* DropPath(p) on row groups */
  int32_t path_group;
} afft_dropout_t;

const char* afft_last_error(void);
int afft_version(void);

/* ------------------------------------------------------------------ GEMM + fused epilogue
 * C[m,n] = epilogue( alpha * sum_k A(m,k) * B(k,n) ),  A(m,k) = A[m*a_rs + k*a_cs],
 * B(k,n) = B[k*b_rs + n*b_cs].   Replaces every nn.Linear / HF Conv1D on the path:
 *   models/transformerblock.py:14,16,21,34,48-50,84-89 ; models/feature_mapping.py:59-61 ;
 *   models/future_prediction.py:108-121,246-255 ; HF modeling_gpt2.py Conv1D (c_attn, c_proj, c_fc)
 * and their backward (dgrad: dX = dY*W ; wgrad: dW = dY^T*X).
 * epilogue order: v = alpha*acc ; v += bias[n] ; pre[m,n] = v ; v = act(v | aux[m,n]) ; v = dropout(v) ;
 *                 v *= rowscale[m]*droppath(m) ; v += residual[m,n] ; v += out[m,n] if accumulate ; store out, out2
 *                 (with `sgd` set: v = alpha*acc is consumed by the optimizer update of p[m,n] and nothing is stored).
 * dtype = operand dtype of A and B (both the same).  With AFFT_BF16 operands, k-contiguous
 * ("NT": a_cs==1,b_rs==1) and k-strided ("TN": a_rs==1,b_cs==1) layouts with K%64==0 and 16-byte
 * aligned rows take the MFMA bf16 fast path (fp32 accumulate); anything else, and every AFFT_F32 call,
 * runs the exact-fp32 MFMA path (v_mfma_f32_32x32x2_f32).                                            */
typedef struct {
  int32_t M, N, K;
  int32_t dtype;
  const void* A; int64_t a_rs, a_cs;
  const void* B; int64_t b_rs, b_cs;
  float alpha;
  const float* bias;                 /* [N] or NULL */
  int32_t act;                       /* AFFT_ACT_*.  The bf16-operand kernels evaluate erf (Abramowitz & Stegun 7.1.26) and tanh
                                      * (1 - 2 / (1 + exp 2u)) on the hardware exp / reciprocal: absolute error 2e-7 (GELU) to 2e-6
                                      * (GELU' of gelu_new), far inside their operands' rounding; the exact-fp32 kernel (dtype
                                      * AFFT_F32, or any shape off the fast path) uses the library's erff / tanhf, and
                                      * AFFT_EXACT_ACT=1 in the environment makes every kernel use them.                       */
  const void* aux; int64_t ldaux; int32_t aux_dtype;   /* read by DGELU_* */
  void* pre; int64_t ldpre; int32_t pre_dtype;         /* optional store of the pre-activation */
  const float* rowscale;             /* [M] or NULL (DropPath / per-row loss weights) */
  const float* residual; int64_t ldres;                /* fp32 or NULL */
  int32_t accumulate;                /* out (fp32 only) += */
  void* out; int64_t ldo; int32_t out_dtype;
  void* out2; int64_t ldo2; int32_t out2_dtype;        /* optional second copy */
  afft_dropout_t drop;               /* dropout / DropPath on the output (train mode), p = 0 -> off */
  /* Split-K scratch, owned by the caller, private to `stream`: AFFT_GEMM_WS_HEADER bytes of counters (zeroed ONCE by the
   * caller before the first use; every launch leaves them zero) followed by fp32 partial tiles.  NULL or smaller than
   * afft_gemm_workspace_bytes() for the problem: the launch simply runs without split-K. */
  void* workspace; int64_t workspace_bytes;
  /* "bf16x3": fp32-accurate products from bf16 MFMAs.  A and B each point at the HI plane of a two-plane split
   * x = hi + lo (afft_split_bf16: hi = bf16(x), lo = bf16(x - hi)); the lo plane sits a_lo / b_lo ELEMENTS behind it.
   * The kernel accumulates A_hi*B_hi + A_lo*B_hi + A_hi*B_lo in one pass over a 3x longer K (fp32 accumulate):
   * products exact to ~2^-17 relative, the precision mode whose logits meet the 1e-3 tolerance at MFMA speed.
   * split3 = 2, "fp16x2" (forward layouts only: A k-contiguous): the planes are FP16 (afft_split_f16) and only the first two
   * segments run, A_hi*B_hi + A_lo*B_hi on v_mfma_f32_16x16x32_f16 -- A exact to ~2^-22, B rounded once to fp16 (2^-11): the
   * evaluation forward at 2/3 of the bf16x3 MFMA work (operands must stay inside fp16's range: weights and activations do,
   * gradients do not, hence no backward).
   * split3 = 4: ONE fp16 pass, A_hi*B_hi alone (same operands and layouts as split3 = 2; a_lo is not read): a GEMM whose activation
   * operand is rounded once to fp16 like its weight -- the sub-layers' AFFT_F16X2_ONE_PASS_* sites. */
  int32_t split3; int64_t a_lo, b_lo;
  /* Optional: apply the optimizer IN the epilogue instead of storing the result (see afft_sgd_fused_t below): the GEMM is a
   * weight gradient whose value is consumed once, by the update of that weight.  Needs accumulate = 0, no bias / act. */
  const struct afft_sgd_fused* sgd;
  /* Optional (NT layout, B = a weight [N, K] with N % 16 == 0, K % 64 == 0, b_cs == K): the FRAGMENT-PACKED copy of B made by
   * afft_pack_weight / kept fresh by afft_sgd_fused_t.p_pk16.  With it the launch may take the "B direct" kernels (csrc/gemm_bd.hip:
   * 160 x 256 tiles for M = 5120-type row counts, A through LDS, B global -> register in MFMA operand layout, no LDS traffic for
   * B); without it, or when the cost model prefers the LDS-staged kernels, B is read row-major as before.  Same result up to
   * the summation order inside a tile (both accumulate K in order, fp32). */
  const void* b_packed;
  /* out_dtype AFFT_F16 and out_lo != 0: `out` is the HI plane of a two-plane fp16 split of the result, hi = fp16(v), and
   * lo = fp16(v - hi) is stored out_lo ELEMENTS behind it (same leading dimension): the producing GEMM writes the A operand of the
   * next fp16 two-pass GEMM (split3 = 2, a_lo = out_lo) directly -- no fp32 round trip, no split kernel.  out2 (any dtype, e.g. the
   * bf16 copy the backward pass reads) and pre are stored as before. */
  int64_t out_lo;
  /* split3 = 3, "fp16x2 with an fp8 lo pass" (NT layout on the 256x256 kernel only: afft_gemm_lo8_ok): A points at the fp16 HI plane, B at
   * the weight's FP16 image (first pass, as split3 = 2); the second pass A_lo W runs on the block-scaled fp8 MFMA
   * (v_mfma_scale_f32_16x16x128_f8f6f4, twice the bf16 rate) over two BYTE planes: a8[m, k] = e4m3(2^11 (a - hi)) (row pitch a8_ld bytes)
   * and b8[n, k] = e4m3(2^8 w) (row pitch b8_ld bytes), with the constant block scales 2^-11 and 2^-8.  The correction term it adds
   * is ~2^-12 of the product, so its own 2^-4 operand rounding stays ~2^-16 of the result -- far below the weight's fp16 rounding.
   * out_lo8 (with out_dtype AFFT_F16): the result is stored as the same pair for the NEXT such GEMM: hi at out, the e4m3 lo byte plane
   * at out_lo8 (row pitch ldo bytes, i.e. the element pitch of out). */
  const void* a8; int64_t a8_ld; const void* b8; int64_t b8_ld;
  void* out_lo8;
} afft_gemm_t;
/* 1 when a split3 = 3 problem of this size runs (NT, the 256x256 kernel's territory, K a multiple of 128) */
int afft_gemm_lo8_ok(int M, int N, int K);
/* Nesterov-SGD update fused into the epilogue of the weight-gradient GEMM that produces the gradient (single-GPU training, no
 * gradient clipping, a weight that receives exactly one gradient contribution per step): element [m, n] of the result is the
 * gradient g of p[m * ldo + n] and is never stored;  g' = gscale*g + wd*p ; buf = mom*buf + g' (g' on the first step) ;
 * p -= lr*(g' + mom*buf) ; p_bf16 = bf16(p)  -- bit for bit what afft_sgd_nesterov computes from a stored gradient (one shared
 * device function).  p / buf / p_bf16 have the layout of the GEMM output (row stride ldo).  Saves the gradient's round trip
 * through HBM (8 of 26 bytes per parameter and step) and the separate update kernels.  The caller orders the launch behind the
 * kernels of the same step that still read p_bf16 (the data-gradient GEMM of the same layer). */
typedef struct afft_sgd_fused {
  float* p; float* buf; void* p_bf16;
  float lr, mom, wd, gscale; int32_t first_step;   /* first_step: AFFT_SGD_* flags (below) */
  void* p_pk16;                                    /* optional: the fragment-packed bf16 image (afft_pack_weight) of the same weight, */
                                                   /* refreshed by the same epilogue (ldo % 32 == 0 and M % 16 == 0)               */
  void* p_f16;                                     /* optional: the row-major FP16 image (layout of p_bf16): the B operand of the   */
                                                   /* fp16 two-pass forward GEMMs ("fp16x2" precision)                              */
  void* p_f8;                                      /* optional: the e4m3 byte image e4m3(2^8 p) (afft_gemm_t.b8), same element offsets */
  const float* ok;                                 /* optional device flag: the update is applied iff ok[0] != 0 (afft_loss_reduce_bwd_ok */
                                                   /* writes isfinite(total loss) there: a non-finite step leaves p / buf / images    */
                                                   /* untouched, like the reference's 'The loss is NaN!' raised before backward)      */
} afft_sgd_fused_t;
/* `first_step` of every optimizer entry point is a flag word: AFFT_SGD_FIRST_STEP = the momentum buffer does not exist yet (it
 * starts as the gradient: torch.optim.SGD's first step); AFFT_SGD_PLAIN_MOMENTUM = torch.optim.SGD(nesterov=False):
 * p -= lr * buf instead of the Nesterov form p -= lr * (g' + mom * buf) (conf/opt/optimizer/sgd.yaml ships nesterov: false,
 * every training recipe under expts/ sets opt.optimizer.nesterov=true). */
enum { AFFT_SGD_FIRST_STEP = 1, AFFT_SGD_PLAIN_MOMENTUM = 2 };
enum { AFFT_GEMM_WS_HEADER = 4096 };
/* 1 when afft_gemm would run an NT problem of this size on the B-direct kernel if it were handed a fragment-packed B
 * (afft_gemm_t.b_packed), 0 when it would ignore the packed copy: callers keep packed images only of the weights that use them
 * (every image costs 2 bytes per parameter of optimizer traffic per step). */
int afft_gemm_packed_wanted(int M, int N, int K);

/* bytes of afft_gemm_t.workspace that let this problem use split-K under the current mode (0 = it would not split) */
int64_t afft_gemm_workspace_bytes(int M, int N, int K, int a_kstrided, int b_kstrided);
int afft_gemm(const afft_gemm_t* g, void* stream);
/* Tuning / test hook: force the bf16 kernel (0 = auto, 1 = 128x128x64 2-stage, 3 = 256x256x64 ping-pong, 4 = 128x128x64 4-stage,
 * 7 / 8 = "B direct" 256x256 / 160x256 tiles on NT layouts with a row-major B (other layouts: as auto), 9 / 10 = the same with B
 * pointing at a fragment-packed image -- what the dispatcher takes by itself when afft_gemm_t.b_packed is given). */
int afft_set_gemm_variant(int variant);
/* Split-K for small grids of the 128x128 kernel (<= 128 tiles with K >= 2048, <= 256 tiles with K >= 4096): K is cut into 2 or 4 slices, every slice
 * parks its fp32 partial tile in the caller's workspace (afft_gemm_t.workspace) and the slice that arrives last adds them up in slice order
 * and runs the epilogue (no waiting, any epilogue, bitwise repeatable).  mode: 0 off, 1 auto (default), 2 / 4 force. */
int afft_set_gemm_splitk(int mode);
/* Which bf16 tile shape afft_gemm picks for a fast-path problem (1 / 3 as above); used by bench.py to attribute
 * launches to kernel symbols. */
int afft_gemm_variant_for(int M, int N, int K, int a_kstrided, int b_kstrided);
/* K-slices afft_gemm uses for that problem under the current split-K mode (1 = none). */
int afft_gemm_splitk_for(int M, int N, int K, int a_kstrided, int b_kstrided);
/* Measurement hook (bench.py's roofline object): while a trace is open, every afft_gemm launch of the bf16 fast path --
 * from any entry point, the composite ones included -- is bracketed by a HIP event pair ON THE STREAM IT IS LAUNCHED ON.
 * afft_gemm_trace_end synchronises those events and returns the records (at most `capacity`; return value = count, < 0 on
 * error).  One trace at a time, process-wide.  variant: 1 = 128x128 tile, 3 = 256x256 ping-pong (general kernel), 13 = 256x256 ping-pong,
 * steady-state kernel (whole tiles, even K-tile count, plain bf16: gemm_bf16_pp2_kernel), 10 = B-direct 160x256 (4, 7-9: tests). */
typedef struct {
  int32_t M, N, K, a_kstrided, b_kstrided, variant, splitk, split3, fused_update;
  float ms;
} afft_gemm_trace_rec_t;
int afft_gemm_trace_begin(int32_t capacity);
int afft_gemm_trace_end(afft_gemm_trace_rec_t* out, int32_t capacity);

/* The same hook for the HBM-bound kernels of the path (bench.py's `hbm_kernels` / `roofline.sublayers`): while a kernel trace is
 * open every afft_attention_fwd / _bwd and afft_layernorm_fwd / _bwd call -- from any entry point, the composite ones included --
 * is bracketed by a HIP event pair on its stream.  bytes = the algorithmic HBM bytes of the call (every operand read once, every
 * result written once), flops = its MFMA-shaped work (attention only).  One trace at a time, process-wide. */
enum { AFFT_K_ATTN_FWD = 1, AFFT_K_ATTN_BWD = 2, AFFT_K_LN_FWD = 3, AFFT_K_LN_BWD = 4 };
typedef struct {
  int32_t kind, rows, width, reserved;
  int64_t bytes, flops;
  float ms;
} afft_kernel_trace_rec_t;
int afft_kernel_trace_begin(int32_t capacity);
int afft_kernel_trace_end(afft_kernel_trace_rec_t* out, int32_t capacity);

/* ------------------------------------------------------------------ LayerNorm
 * nn.LayerNorm(eps) fwd/bwd: models/fusion.py:281,362 ; transformerblock.py:122,127,150-152 ;
 * HF GPT2 ln_1/ln_2/ln_f (eps 1e-5).   x is fp32 [rows, d] (row stride ldx); y dtype selectable.
 * A row stride ldx = S*d selects token 0 of every frame (fuser tail, models/fusion.py:362-364: only
 * token 0 of the final LayerNorm is used).  w / b may be NULL (elementwise_affine=False). */
int afft_layernorm_fwd(const float* x, int64_t ldx, const float* w, const float* b,
                       float eps, int32_t rows, int32_t d, void* y, int64_t ldy, int32_t y_dtype,
                       float* mean, float* rstd, void* stream);
/* The same LayerNorm with the result written as the operand planes of the "fp16x2" precision: y_hi [rows, d] fp16 = fp16(y),
 * y_hi + y_lo (ELEMENTS) = fp16(y - hi) (the A operand of an fp16 two-pass GEMM, afft_gemm_t.split3 = 2), and -- optional --
 * y_bf16 = bf16(y), the copy the single-pass bf16 backward reads (weight-gradient operand).  y_lo = 0: hi plane only. */
int afft_layernorm_fwd_split(const float* x, int64_t ldx, const float* w, const float* b, float eps, int32_t rows, int32_t d,
                             void* y_hi, int64_t ldy, int64_t y_lo, void* y_bf16, int64_t ldyb, float* mean, float* rstd,
                             void* y_lo8, void* stream);      /* y_lo8 (optional, then y_lo = 0): the lo part as the e4m3 byte plane of afft_gemm_t.a8, row pitch ldy bytes */
/* dx_out[r] = (dx_in ? dx_in[r] : 0) + LN'(dy)[r]; dw/db are written (accumulate = 0) or added to (+=).
 * dy dtype selectable.
 * dx_bf16 (optional) receives a bf16 copy of dx_out with the dropout / DropPath mask `copy_drop` (optional) replayed
 * on it: the operand of the dgrad/wgrad GEMMs of the sub-layer that produced this LayerNorm's input, whose output
 * dropout it is.  dcol (optional, fp32 [d]) receives the column sums of that masked copy (= that sub-layer's output
 * bias gradient), written or added to per dcol_accumulate.
 * partial: fp32 workspace of at least 3*d*afft_layernorm_bwd_nparts(rows) floats. */
int afft_layernorm_bwd_nparts(int32_t rows);
int afft_layernorm_bwd(const void* dy, int64_t lddy, int32_t dy_dtype, const float* x, int64_t ldx,
                       const float* w, const float* mean, const float* rstd,
                       int32_t rows, int32_t d, const float* dx_in, float* dx_out, int64_t lddx,
                       void* dx_bf16, const afft_dropout_t* copy_drop, float* dw, float* db, int32_t accumulate,
                       float* dcol, int32_t dcol_accumulate, float* partial, void* stream);

/* ------------------------------------------------------------------ small-sequence attention
 * softmax(q k^T * scale + mask) v per (sequence, head); L <= 128 tokens per sequence (bf16 MFMA kernel for
 * L <= 32 and the first three masks, a generic kernel otherwise).
 *   SA-Fuser: L = M+1 modality tokens of one frame  (models/transformerblock.py:24-33, fusion.py:338-349)
 *   GPT-2:    L = T frames, causal                   (HF modeling_gpt2.py eager_attention_forward)
 *   CA-Fuser: causal self and cross attention        (models/transformerblock.py:56-76)
 *   T-SA-Fuser: L = M*T tokens, AFFT_MASK_BLOCKCAUSAL with mask_period = T: key j is hidden from query i when
 *               (j mod T) > (i mod T) = generate_square_subsequent_mask(T).repeat(M, M)  (models/fusion.py:170-171)
 * q/k/v: [nseq*L, *] with row strides ldq/ldk/ldv, head h at columns h*hd..; out [nseq*L, H*hd].
 * probs: fp32 [nseq, H, L, L] (the attention weights the reference returns; also saved for backward).
 * drop_p/drop_key: attention-probability dropout (attn_drop / attn_pdrop); probs always holds the
 * PRE-dropout probabilities (backward regenerates the mask). */
int afft_attention_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                       int32_t dtype, int32_t nseq, int32_t L, int32_t H, int32_t hd, float scale, int32_t mask,
                       int32_t mask_period, float drop_p, uint32_t drop_key, void* out, int64_t ldo, float* probs,
                       void* stream);
/* The same with an ARBITRARY additive mask: mask_table fp32 [L, L] (device), added to the scaled scores before the softmax -- what
 * `attn = attn + attn_mask` does for any (N, N) tensor the caller passes (models/transformerblock.py:26-28 Attention, :66-68
 * CrossAttention); entries may be -inf.  Generic kernel (the in-register masks above stay on the MFMA kernels).  The backward pass
 * needs no table: afft_attention_bwd works from the saved probabilities. */
int afft_attention_fwd_table(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                             int32_t dtype, int32_t nseq, int32_t L, int32_t H, int32_t hd, float scale,
                             const float* mask_table, float drop_p, uint32_t drop_key, void* out, int64_t ldo, float* probs,
                             void* stream);
/* "fp16x2" forward: q / k / v are the HI planes of two-plane fp16 splits (lo planes in_lo ELEMENTS behind, same strides); scores and
 * P V are accumulated from three fp16 MFMA products each (hi*hi + lo*hi + hi*lo: fp32-grade, ~2^-21) so that the attention core
 * adds no 2^-11 operand rounding to the forward pass; the result is written as planes again (out_hi, lo out_lo elements behind)
 * and, optional, as the bf16 copy the backward pass reads.  MFMA path only: L <= 64, hd % 64 == 0. */
int afft_attention_fwd_split(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, int64_t in_lo,
                             int32_t nseq, int32_t L, int32_t H, int32_t hd, float scale, int32_t mask, int32_t mask_period,
                             float drop_p, uint32_t drop_key, void* out_hi, int64_t ldo, int64_t out_lo, void* out_bf16,
                             int64_t ldob, float* probs, void* out_lo8, void* stream);      /* out_lo8: as y_lo8 above (row pitch ldo bytes) */
int afft_attention_bwd(const void* dout, int64_t lddo, const void* q, int64_t ldq, const void* k, int64_t ldk,
                       const void* v, int64_t ldv, int32_t dtype, const float* probs, int32_t nseq, int32_t L,
                       int32_t H, int32_t hd, float scale, float drop_p, uint32_t drop_key, void* dq, int64_t lddq,
                       void* dk, int64_t lddk, void* dv, int64_t lddv, void* stream);

/* afft_layernorm_bwd whose incoming residual gradient dx_in is [rows / in_take, d] (row pitch lddx_in) and belongs to rows 0, in_take,
 * 2 in_take, ..: the other rows take no residual gradient (afft_attn_sublayer_t.take; in_take = 1: every row). */
int afft_layernorm_bwd_take(const void* dy, int64_t lddy, int32_t dy_dtype, const float* x, int64_t ldx, const float* w, const float* mean,
                            const float* rstd, int32_t rows, int32_t d, const float* dx_in, int64_t lddx_in, int32_t in_take,
                            float* dx_out, int64_t lddx, void* dx_bf16, const afft_dropout_t* copy_drop, float* dw, float* db,
                            int32_t accumulate, float* dcol, int32_t dcol_accumulate, float* partial, void* stream);

/* ------------------------------------------------------------------ losses (common/runner.py:13-37,112-168)
 * Softmax cross-entropy over C classes, fwd + bwd in one pass.
 *   hard: labels int64[rows], label -1 = ignored row (loss 0, grad 0); any other label outside [0, C) makes the row's
 *   loss and gradient NaN (torch raises a device-side assert there; nothing out of bounds is read);
 *   soft: targets fp32 [rows, C] and
 *   optional keep uint8[rows] (0 = row removed).  row_loss[r] = the row's loss; loss_sum (optional, needs row_loss)
 *   += the sum of row_loss added up in row order by a second one-workgroup kernel (no float atomics);
 *   dlogits[r,:] = gscale * row_g[r] * (softmax*sum(target) - target) for kept rows, 0 otherwise (row_g NULL = 1:
 *   the per-row upstream gradient of a reduction='none' loss); columns C..ldd-1 are zeroed. */
int afft_softmax_ce(const float* logits, int64_t ldl, int32_t rows, int32_t C, const int64_t* labels,
                    const float* soft, int64_t lds, const uint8_t* keep, float gscale, const float* row_g,
                    float* loss_sum, void* dlogits, int64_t ldd, int32_t d_dtype, float* row_loss, void* stream);
/* The same on frames [0, frames) of `clips` clips laid out (clips, L >= frames, >= C): row r of labels / soft / keep / row_g /
 * row_loss is frame r % frames of clip r / frames, its logits sit at logits + clip * clip_stride + frame * ldl and its gradient at
 * dlogits + clip * d_clip_stride + frame * ldd.  The past / future halves of the merged classifier output (models/future_prediction.py:283-285
 * applies the same heads to both) are walked where they lie, and their gradients land side by side in ONE buffer. */
int afft_softmax_ce_frames(const float* logits, int64_t clip_stride, int64_t ldl, int32_t clips, int32_t frames, int32_t C,
                           const int64_t* labels, const float* soft, int64_t lds, const uint8_t* keep, float gscale, const float* row_g,
                           void* dlogits, int64_t d_clip_stride, int64_t ldd, int32_t d_dtype, float* row_loss, void* stream);
/* Runner._reduce_loss (common/runner.py:198-213) in one launch: term i = n[i] fp32 per-row losses at x[i] (HOST arrays of nterms <= 8
 * device pointers / counts / weights); means[i] = mean(x[i]) (may be NULL), total[0] = sum_i w[i] * means[i], in a fixed order (bit-stable).
 * afft_loss_reduce_bwd: g[i][0 .. n[i]) = g_total[0] * w[i] / n[i] (g_total NULL = 1; g[i] NULL skips a term): the upstream
 * gradients of the per-row losses, ready for afft_softmax_ce's row_g / afft_mse's g_dev. */
int afft_loss_reduce(const float* const* x, const int64_t* n, const float* w, int32_t nterms, float* means, float* total, void* stream);

int afft_loss_reduce_bwd(float* const* g, const int64_t* n, const float* w, int32_t nterms, const float* g_total, void* stream);
/* ... and ok[0] = isfinite(total[0]) && isfinite(g_total[0]) ? 1 : 0 (total: the forward's result, may be NULL): the device-side form
 * of the reference's 'The loss is NaN!' check (common/runner.py:209), written by the FIRST kernel of the backward pass and read by
 * the optimizer kernels of the same step (afft_sgd_fused_t.ok): a non-finite loss leaves parameters, momentum and images untouched. */
int afft_loss_reduce_bwd_ok(float* const* g, const int64_t* n, const float* w, int32_t nterms, const float* g_total, const float* total,
                            float* ok, void* stream);
/* Scalar reductions (the MSE loss, the gradient norm) are ORDERED: every workgroup writes one partial sum into the
 * caller-provided per-stream workspace (the same buffer as afft_gemm_t.workspace / afft_colsum: the partials sit behind its
 * AFFT_GEMM_WS_HEADER bytes of counters; at most AFFT_REDUCE_PARTIALS floats) and a second one-workgroup kernel adds them up
 * in workgroup order.  Same inputs -> same bits, on every run. */
#define AFFT_REDUCE_PARTIALS 2048
/* MSE between a[rows,d] and b[rows,d] (fp32): loss_sum += lscale * sum (a-b)^2 ;
 * da += gscale*g_dev[0]*2*(a-b) ; db -= the same (common/runner.py:164-166: both sides carry gradient;
 * g_dev = device scalar upstream gradient or NULL = 1).  da/db may be NULL.  workspace: see above (only used when
 * loss_sum != NULL and rows*d > 256). */
int afft_mse(const float* a, int64_t lda, const float* b, int64_t ldb, int32_t rows, int32_t d, float gscale,
             const float* g_dev, float lscale, float* loss_sum, float* da, int64_t ldda, float* db, int64_t lddb,
             void* workspace, int64_t workspace_bytes, void* stream);
/* loss[0] = lscale * sum (a-b)^2: overwritten, not added to (no zero-fill in front). */
int afft_mse_loss(const float* a, int64_t lda, const float* b, int64_t ldb, int32_t rows, int32_t d, float lscale, float* loss,
                  void* workspace, int64_t workspace_bytes, void* stream);
/* Backward of the MSE between n-float ranges of two (clips, len) fp32 tensors (common/runner.py:164-166 compares [:, 1:] of the
 * predicted and the observed features): da[clips, a_len] and db[clips, b_len] are WRITTEN whole -- +/- gscale*g_dev[0]*2*(a-b) on
 * [a_off, a_off+n) / [b_off, b_off+n), zeros elsewhere.  a / b rows start at clip * clip_stride; offsets, lengths and strides
 * are multiples of 4 floats; da or db may be NULL. */
int afft_mse_frames_bwd(const float* a, int64_t a_clip_stride, int32_t a_off, int32_t a_len, const float* b, int64_t b_clip_stride,
                        int32_t b_off, int32_t b_len, int32_t clips, int32_t n, float gscale, const float* g_dev, float* da, float* db,
                        void* stream);

/* ------------------------------------------------------------------ data movement / elementwise
 * fp32 [rows, cols] -> dst dtype copy; if dst_t != NULL also writes the transpose [cols, rows] (ld = ldt).
 * Used for the cached bf16 weights (W and W^T) and activation casts. Columns cols..ldd-1 of dst are zero-filled
 * when zero_pad != 0 (K padding for the fast GEMM path).  drop (may be NULL) multiplies element [r,c] by the
 * dropout / DropPath scale of index r*cols + c: the backward replay of a GEMM-epilogue dropout, or the forward
 * dropout of a GEMM input (classifier Dropout(0.2), future_prediction.py:108). */
int afft_cast(const float* src, int64_t lds, int32_t rows, int32_t cols, void* dst, int64_t ldd, int32_t dst_dtype,
              void* dst_t, int64_t ldt, int32_t zero_pad, const afft_dropout_t* drop, void* stream);
/* Fragment-packed bf16 image of a weight W [rows, cols] (fp32 masters, row stride lds; rows % 16 == 0, cols % 32 == 0): for every
 * block of 16 rows rb and every 32-deep K-step ks, the 64 lanes' MFMA operand fragments side by side,
 *   dst[((rb * (cols / 32) + ks) * 64 + lane) * 8 + j] = bf16(W[16 rb + (lane & 15)][32 ks + 8 (lane >> 4) + j]),  j = 0..7,
 * so that a wave loads the B operand of v_mfma_f32_16x16x32_bf16 for 16 output columns x 32 k with ONE fully coalesced 1-KiB
 * global load (row-major: 16 rows x 64 B per instruction, four cache lines per quad of lanes).  afft_gemm_t.b_packed. */
int afft_pack_weight(const float* src, int64_t lds, int32_t rows, int32_t cols, void* dst, void* stream);
/* Two-plane bf16 split of an fp32 matrix for afft_gemm_t.split3: hi[r, c] = bf16(src[r, c]), lo[r, c] = bf16(src[r, c] - hi[r, c]),
 * both planes [rows_pad, ldd] bf16 with everything outside [rows, cols] zero-filled (K padding of either GEMM operand role);
 * lo = hi + plane_stride elements. */
int afft_split_bf16(const float* src, int64_t lds, int32_t rows, int32_t cols, void* hi, int64_t ldd, int32_t rows_pad,
                    int64_t plane_stride, void* stream);
/* e4m3 byte image dst[r, c] = e4m3(scale * src[r, c]) (OCP e4m3fn, saturating), dst row pitch ldd BYTES, columns cols..ldd-1 and rows
 * rows..rows_pad-1 zero-filled: the weight image afft_gemm_t.b8 (scale 2^8) of a parameter that is not homed in the flat buffers.
 * With hi != NULL (fp16 [rows_pad, ldd], row pitch ldd elements) the source is split instead: hi = fp16(x) and dst = e4m3(scale * (x - hi))
 * -- the operand pair (A, a8) of afft_gemm_t.split3 = 3 with scale 2^11. */
int afft_quant_e4m3(const float* src, int64_t lds, int32_t rows, int32_t cols, float scale, void* dst, int64_t ldd, int32_t rows_pad,
                    void* hi, void* stream);
/* The same with fp16 planes (hi = fp16(x), lo = fp16(x - hi)): operands of afft_gemm_t.split3 = 2. */
int afft_split_f16(const float* src, int64_t lds, int32_t rows, int32_t cols, void* hi, int64_t ldd, int32_t rows_pad,
                   int64_t plane_stride, void* stream);
/* ModalTokenCMFuser token assembly (models/fusion.py:338-352): X[(b*T+t), s, :] for s=0 the modal token
 * (token + (t)*tok_stride_t; stride 0 = one universal token), s>=1 modality s-1 at feats[s-1] + (b*T+t)*ldf[s-1];
 * + modality_embedding[s,:] if given.  feats / ldf are HOST arrays of n_mod (<= 8) device pointers / strides. */
int afft_assemble_tokens(const float* const* feats, const int64_t* ldf, int32_t n_mod, const float* token,
                         int64_t tok_stride_t, const float* mod_embed, int32_t BT, int32_t T, int32_t d,
                         float* X, void* stream);
/* column sums: out[n] (+)= sum_m src[m, n]   (bias gradients; src dtype selectable).  No float atomics: the result is
 * bit-identical from run to run (and with it the whole training step -- every other reduction of the path is ordered too).
 * workspace (optional): scratch of `stream`; the first AFFT_GEMM_WS_HEADER bytes are not touched (the split-K scratch of the
 * same stream, afft_gemm_t.workspace, can be passed), then room for partial sums: 1 KiB per 256-column strip and block of 128
 * rows (fewer, longer row blocks if it is smaller).  Row blocks are then summed by separate workgroups and a second kernel adds
 * them up in block order.  NULL: one workgroup per column strip walks all the rows (~1.5x the time on a [5120, 8192] input). */
int afft_colsum(const void* src, int64_t lds, int32_t dtype, int32_t rows, int32_t cols, float* out,
                int32_t accumulate, void* workspace, int64_t workspace_bytes, void* stream);
/* p[0 .. bytes) = 0 by a kernel (16-byte aligned, bytes % 16 == 0): usable inside a stream capture where a large hipMemsetAsync is not. */
int afft_zero(void* p, int64_t bytes, void* stream);
/* out[clip, t, 0..C) = sum over the (at most 4) sources k with lo[k] <= t < hi[k] of src[k][clip, t + off[k], 0..C), zeros where no
 * source covers t; every output element is written exactly once.  Strides in floats (multiples of 4; C % 4 == 0).  One launch for:
 * torch.cat along the frame axis (models/future_prediction.py:161-170 builds past_futures = [z_1, z_hat_2..] this way) and its backward
 * (overlapping slice gradients are added), "token 0 of every frame" (models/fusion.py:362-365) and its zero-filled backward. */
int afft_gather_frames(float* out, int64_t out_clip_stride, int64_t out_frame_stride, int32_t clips, int32_t frames, int32_t C,
                       int32_t nsrc, const float* const* src, const int64_t* clip_stride, const int64_t* frame_stride,
                       const int32_t* lo, const int32_t* hi, const int32_t* off, void* stream);
/* y[r, :] = x[r, :] + table[(r % period), :]   (GPT-2 wpe / CA-Fuser position embedding; fp32) */
int afft_add_rows_periodic(const float* x, int64_t ldx, const float* table, int64_t ldt, int32_t rows,
                           int32_t period, int32_t d, float* y, int64_t ldy, void* stream);
/* out[(r % period), :] += sum over r of src[r, :]  (gradient of the above table) */
int afft_reduce_rows_periodic(const float* src, int64_t lds, int32_t rows, int32_t period, int32_t d,
                              float* out, int64_t ldo, void* stream);
/* Backward of a fused output activation (+ output dropout): from the upstream gradient dy (fp32) of
 *   y = drop(act(pre))   act in { NONE, GELU_ERF, GELU_TANH, RELU, SIGMOID_GATE (y = aux * sigmoid(pre)) }
 * writes dpre (dtype selectable: the operand of the dgrad / wgrad GEMMs) and, for the gate, daux (fp32) = dy*sigmoid(pre).
 * saved = pre (any dtype); for RELU `saved` may be the activated output instead (only its sign is used).
 * drop replays the forward output-dropout mask (NULL = none). */
int afft_act_bwd(int32_t act, const float* dy, int64_t lddy, const void* saved, int64_t lds, int32_t saved_dtype,
                 const float* aux, int64_t ldaux, int32_t rows, int32_t cols, const afft_dropout_t* drop,
                 void* dpre, int64_t lddp, int32_t dpre_dtype, float* daux, int64_t ldda, void* stream);
/* Softmax over the (few) columns of an fp32 [rows, n] matrix, n <= 32: the modality weights of MATT
 * (models/fusion.py:57); backward: dx = y * (dy - sum_j dy_j y_j). */
int afft_softmax_small_fwd(const float* x, int64_t ldx, int32_t rows, int32_t n, float* y, int64_t ldy, void* stream);
int afft_softmax_small_bwd(const float* y, int64_t ldy, const float* dy, int64_t lddy, int32_t rows, int32_t n,
                           float* dx, int64_t lddx, void* stream);
/* Score fusion (models/future_prediction.py:343-350): out[r, :] = sum_i w[r, i] * x_i[r, :] over n <= 8 modalities
 * (x_i fp32 [rows, cols], row stride ldx, w fp32 [rows, n] row stride ldw); backward: dx_i[r, :] = w[r, i] * dout[r, :],
 * dw[r, i] = <dout[r, :], x_i[r, :]>. */
int afft_weighted_sum_fwd(const float* const* x, int64_t ldx, const float* w, int64_t ldw, int32_t n, int32_t rows,
                          int32_t cols, float* out, int64_t ldo, void* stream);
int afft_weighted_sum_bwd(const float* const* x, int64_t ldx, const float* w, int64_t ldw, const float* dout, int64_t lddo,
                          int32_t n, int32_t rows, int32_t cols, float* const* dx, int64_t lddx, float* dw, int64_t lddw,
                          void* stream);
/* Dropout salt: a device word that every kernel XORs into its dropout / DropPath keys at kernel start.  A captured
 * hipGraph replays identical kernel arguments every step; with the salt advanced by afft_dropout_salt_step inside the
 * graph the masks still differ from step to step.  afft_set_dropout_salt(NULL) (the default) turns it off. */
int afft_set_dropout_salt(const uint32_t* salt_dev);
int afft_dropout_salt_step(uint32_t* salt_dev, void* stream);
/* MixUp with an ignore class as a GPU prologue (common/mixup.py:119-182), no host round trip:
 *   afft_mixup_plan: partner[b] = the sample b is mixed with -- the samples none of whose T past labels is ignore_cls are
 *     paired in reverse order among themselves (x[sel].flip(0)); partner[b] = b for the others and when at most one
 *     sample qualifies; labels_subclips NULL = every sample qualifies.  ignore_mask (optional) [B*T] = label == ignore_cls.
 *   afft_mixup_rows:   y[b, :] = lam * x[b, :] + (1 - lam) * x[partner[b], :]   (x fp32 [B, W]; y = x[b] when partner[b] = b)
 *   afft_mixup_labels: out[r, :] = lam * onehot(labels[r]) + (1 - lam) * onehot(labels[partner row]) with the smoothed
 *     one-hot of common/mixup.py:17-47 (ignored labels count as class 0), rows r = b * rows_per_sample + i. */
int afft_mixup_plan(const int64_t* labels_subclips, int32_t B, int32_t T, int64_t ignore_cls, int32_t* partner,
                    uint8_t* ignore_mask, void* stream);
int afft_mixup_rows(const float* x, int32_t B, int64_t W, const int32_t* partner, float lam, float* y, void* stream);
int afft_mixup_labels(const int64_t* labels, int32_t B, int32_t rows_per_sample, int32_t K, float label_smooth,
                      int64_t ignore_cls, const int32_t* partner, float lam, float* out, void* stream);
/* ZeroMaskRULSTMFeats (common/transforms.py:13-26) for a batch on the device: in every clip of x fp32 [B, T, C] exactly k
 * frames -- a uniformly random subset drawn from a counter hash of (key, clip, frame) -- are zeroed in place. */
int afft_zero_mask_frames(float* x, int32_t B, int32_t T, int64_t C, int32_t k, uint32_t key, void* stream);
/* Row softmax of wide fp32 rows: action probabilities for the verb / noun marginalisation (challenge.py:196-203). */
int afft_softmax_rows(const float* x, int64_t ldx, int32_t rows, int32_t C, float* y, int64_t ldy, void* stream);
/* Token means of the fusers without a modality token (models/fusion.py:114-116 CMFuser: mean over the M tokens of a
 * frame; :207-210 T-SA-Fuser: mean over the M modality tokens of a frame position): x fp32 [G, S, W] contiguous,
 *   afft_group_sum:   y[g, :]     = scale * sum_s x[g, s, :]
 *   afft_group_bcast: dx[g, s, :] = scale * dy[g, :]            (its backward) */
int afft_group_sum(const float* x, int32_t G, int32_t S, int64_t W, float scale, float* y, void* stream);
int afft_group_bcast(const float* dy, int32_t G, int32_t S, int64_t W, float scale, float* dx, void* stream);
/* Nesterov-momentum SGD over one flat fp32 parameter buffer (conf/opt/optimizer/sgd.yaml, train.py:262):
 *   g = gscale*g + wd*p ; buf = mom*buf + g ; p -= lr*(g + mom*buf) ; gscale = 1/world after a summing
 *   all-reduce; g may be fp32 or bf16 (bf16 gradient exchange).  p_bf16 (optional, same element offsets as p)
 *   receives the bf16 image of the updated weights: the GEMM operand copy is refreshed by the update itself.
 *   gscale_dev (optional device scalar) multiplies gscale: the gradient-clipping coefficient of afft_clip_coef. */
int afft_sgd_nesterov(float* p, const void* g, int32_t g_dtype, float* buf, void* p_bf16, int64_t n, float lr, float mom,
                      float wd, float gscale, const float* gscale_dev, int32_t first_step, void* stream);
/* p_f16 (optional, same element offsets as p): the FP16 image of the updated weights ("fp16x2" forward operands), written
 * beside p_bf16 by afft_sgd_nesterov2 / afft_sgd_nesterov_runs2 -- otherwise the functions above. */
int afft_sgd_nesterov2(float* p, const void* g, int32_t g_dtype, float* buf, void* p_bf16, void* p_f16, void* p_f8, int64_t n, float lr,
                       float mom, float wd, float gscale, const float* gscale_dev, int32_t first_step, const float* ok, void* stream);      /* p_f8: e4m3(2^8 p) bytes; ok: see afft_sgd_fused_t.ok (NULL = always) */
/* The same update over `nruns` separate runs of ONE set of flat buffers: runs = device array of nruns x {start, length}
 * (int64 elements, starts multiples of 4).  One launch for all the small parameters of a gradient bucket (LayerNorm
 * weights, biases, tokens) whose big neighbours are updated in their weight-gradient epilogues (afft_sgd_fused_t). */
int afft_sgd_nesterov_runs(float* p, const float* g, float* buf, void* p_bf16, const int64_t* runs, int32_t nruns, float lr,
                           float mom, float wd, float gscale, int32_t first_step, void* stream);
int afft_sgd_nesterov_runs2(float* p, const float* g, float* buf, void* p_bf16, void* p_f16, void* p_f8, const int64_t* runs, int32_t nruns,
                            float lr, float mom, float wd, float gscale, int32_t first_step, const float* ok, void* stream);
/* Gradient clipping by global norm (train.py:254-260, torch.nn.utils.clip_grad_norm_), without a host sync:
 *   afft_sumsq: *out += scale * sum x[i]^2 over a flat fp32/bf16 buffer (scale = gscale^2 of the optimizer), ordered
 *   through the stream's workspace (see afft_mse) - the clipping coefficient is bit-reproducible;
 *   afft_clip_coef: *coef = min(1, max_norm / (sqrt(*sumsq) + 1e-6)), *norm_out (optional) = sqrt(*sumsq). */
int afft_sumsq(const void* x, int32_t dtype, int64_t n, float scale, float* out, void* workspace, int64_t workspace_bytes,
               void* stream);
int afft_clip_coef(const float* sumsq, float max_norm, float* coef, float* norm_out, void* stream);

/* ------------------------------------------------------------------ composite entry points: one call = one sub-layer
 * SURVEY.md 8(b) "minimum exports": a whole pre-LN transformer sub-layer, forward or backward, enqueued by ONE call.
 * They launch exactly the kernels the primitive entry points above launch, in the same order, with the same arguments
 * (afft_amd/functional.py is the call-by-call version of the same sequences): the point is the host -- a training step
 * of the reference's EK100 configuration is ~370 kernel launches and becomes ~60 calls.
 *
 * bf16 speed mode only: GEMM operands are bf16 buffers whose leading dimensions are multiples of 64 and whose rows are
 * padded to multiples of 64 WITH ZEROS (they are the K dimension of the weight-gradient GEMMs); residual stream, LayerNorm
 * statistics, probabilities and every gradient of a parameter are fp32.  Weights are the bf16 IMAGES of the fp32
 * parameters ([out, in] for nn.Linear, [in, out] for HF Conv1D when `conv1d`), leading dimensions ldw*.
 * Backward: weight / bias gradients are written (acc_* = 0) or added (acc_* = 1) into the fp32 gradient buffers; NULL
 * gradient pointer = that parameter does not exist (no bias) or needs no gradient.  `aux_stream` (may equal `stream` or
 * be NULL): the weight-gradient GEMMs and bias column sums are enqueued there, ordered behind the producing kernels by
 * events; NOTHING on `stream` waits for them -- the caller joins the streams when its backward pass is over.
 * Saved by forward for backward (caller-owned, see each struct): the bf16 activations, mean / rstd, probs.            */

#define AFFT_F16X2_ONE_PASS_1 4      /* flags of afft_attn_sublayer_t.f16x2 / afft_mlp_sublayer_t.f16x2 (below) */
#define AFFT_F16X2_ONE_PASS_2 8
#define AFFT_F16X2_ONE_PASS_ATTN 16   /* afft_attn_sublayer_t only: the attention core reads q, k, v as one fp16 plane each */
typedef struct {       /* y = x + drop(proj(attention(split(qkv(LN(x))))))                                          */
  /* Block / DecoderBlock self-attention half: models/transformerblock.py:19-36,131-133,158-159 ; HF GPT2Block attn half  */
  int32_t rows, d, L, H;                 /* rows = nseq * L tokens; H heads of d / H                                   */
  int32_t conv1d, mask, mask_period;     /* AFFT_MASK_*                                                                */
  float eps, scale;
  const float* x;                        /* [rows, d] fp32, dense                                                      */
  const float* ln_w; const float* ln_b;  /* [d] (NULL: no affine)                                                      */
  const void* w_qkv; int64_t ldw_qkv; const float* b_qkv;      /* bf16 image; bias [3d] or NULL                        */
  const void* w_proj; int64_t ldw_proj; const float* b_proj;
  float p_attn; uint32_t k_attn;         /* attention-probability dropout                                              */
  afft_dropout_t out_drop;               /* projection dropout + DropPath                                              */
  /* forward outputs, saved for backward */
  void* xn; void* qkv; void* ao;         /* bf16 [rows_pad, d], [rows_pad, 3d], [rows_pad, d], dense, zero row tails   */
  float* mean; float* rstd;              /* [rows]                                                                     */
  float* probs;                          /* [nseq, H, L, L]                                                            */
  float* y;                              /* [rows, d] fp32 (forward only)                                              */
  /* backward only */
  const float* dy;                       /* [rows, d] fp32                                                             */
  void* dya;                             /* bf16 [rows_pad, d]: dy with out_drop replayed; dya_ready = 1: already holds it */
  int32_t dya_ready;                     /*   (handed over by the LayerNorm backward downstream, with the b_proj gradient) */
  void* dao; void* dqkv; void* dxn;      /* bf16 scratch [rows_pad, d], [rows_pad, 3d], [rows_pad, d] (zero row tails) */
  float* g_w_qkv; int32_t acc_w_qkv; float* g_b_qkv; int32_t acc_b_qkv;
  float* g_w_proj; int32_t acc_w_proj; float* g_b_proj; int32_t acc_b_proj;     /* g_b_proj NULL when handed over      */
  float* g_ln_w; float* g_ln_b; int32_t acc_ln;
  float* dx;                             /* [rows, d] fp32 = dy + LN'(...)                                             */
  void* dx_bf16; const afft_dropout_t* up_drop; float* up_dcol;   /* optional hand-over emission (afft_layernorm_bwd)   */
  float* ln_partial;                     /* afft_layernorm_bwd workspace                                               */
  void* gemm_ws; int64_t gemm_ws_bytes;  /* afft_gemm_t.workspace of `stream`                                          */
  void* gemm_ws_aux; int64_t gemm_ws_aux_bytes;   /* ... of `aux_stream`                                                */
  /* optional fused optimizer (afft_sgd_fused_t) per weight: that weight's gradient GEMM then runs BEHIND the data-gradient
   * GEMM that reads the weight and updates it in its epilogue; its g_w_* buffer is not written */
  const afft_sgd_fused_t* sgd_w_qkv; const afft_sgd_fused_t* sgd_w_proj;
  /* optional fragment-packed copies of the two weight images (afft_gemm_t.b_packed; nn.Linear layout only): forward GEMMs */
  const void* w_qkv_pk; const void* w_proj_pk;
  /* f16x2 != 0 ("fp16x2" precision, FORWARD only: the backward pass is the bf16 one on the *_b copies): w_qkv / w_proj point at
   * FP16 images; xn / qkv / ao are two-plane fp16 splits -- hi plane at the pointer, lo plane rows_pad * width elements behind it
   * (2x the bf16 size each) -- and xn_b / qkv_b / ao_b (optional: NULL in a forward nobody differentiates) receive the bf16 copies
   * (layout of xn / qkv / ao in the bf16 mode) that afft_attn_sublayer_bwd is then called with. */
  int32_t f16x2; void* xn_b; void* qkv_b; void* ao_b;
  /* f16x2 = 2: the lo pass of the two GEMMs on the block-scaled fp8 MFMA (afft_gemm_t.split3 = 3; nn.Linear layout, shapes for which
   * afft_gemm_lo8_ok says yes): w_qkv8 / w_proj8 = the weights' e4m3 byte images (row pitch ldw_* bytes); the lo planes of xn and ao are
   * e4m3 BYTE planes (rows_pad * d bytes, directly behind their hi planes); qkv keeps its fp16 lo plane (the attention kernel reads it). */
  const void* w_qkv8; const void* w_proj8;
  /* f16x2 may carry AFFT_F16X2_ONE_PASS_1 / _2 on top of 1 or 2: the sub-layer's first (qkv; fc1) / second (proj; fc2) GEMM runs ONE
   * fp16 pass on its operand's hi plane (afft_gemm_t.split3 = 4) and the producer of that operand writes no lo plane for it (xn: the
   * LayerNorm; ao: the attention kernel; h: the fc1 epilogue); AFFT_F16X2_ONE_PASS_ATTN: the qkv epilogue writes no lo plane and the attention
   * core multiplies the hi planes of q, k, v alone (its probabilities stay hi + lo in registers).  Which sites afford this inside the 1e-3 logits tolerance is the caller's
   * measurement (afft_amd.runtime.one_pass_sites, tools/lo_pass_sweep.py). */
  /* take > 1: only the first token of every `take` = L rows leaves the sub-layer (the SA-Fuser's last block, models/fusion.py:362-365
   * returns token 0 of every frame): attention runs over all rows, the output projection (+ residual, bias, dropout) on rows
   * 0, take, 2 take, .. only; y, dy and dya are [rows / take, d].  Backward: the projection's data gradient lands on those rows of
   * dao (the rest zero), its weight gradient reduces over rows / take rows, the residual gradient enters the LayerNorm backward on
   * those rows only.  Needs rows / take % 64 == 0; f16x2 = 2 only if the projection at rows / take rows passes afft_gemm_lo8_ok. */
  int32_t take;
} afft_attn_sublayer_t;
int afft_attn_sublayer_fwd(const afft_attn_sublayer_t* s, void* stream);
int afft_attn_sublayer_bwd(const afft_attn_sublayer_t* s, void* stream, void* aux_stream);

typedef struct {       /* y = x + drop(fc2(gelu(fc1(LN(x)))))                                                         */
  /* MLP half of Block / DecoderBlock: models/transformerblock.py:84-93,134,161 (erf GELU) ; HF GPT2MLP (gelu_new)        */
  int32_t rows, d, hidden, conv1d;
  int32_t gelu;                          /* AFFT_ACT_GELU_ERF or AFFT_ACT_GELU_TANH                                    */
  float eps;
  const float* x; const float* ln_w; const float* ln_b;
  const void* w1; int64_t ldw1; const float* b1;
  const void* w2; int64_t ldw2; const float* b2;
  afft_dropout_t out_drop;
  void* xn; void* u; void* h;            /* bf16 [rows_pad, d], [rows_pad, hidden] (pre-activation), [rows_pad, hidden];
                                          * forward-only callers may pass u = NULL: the pre-activation is then not stored     */
  float* mean; float* rstd;
  float* y;
  const float* dy; void* dya; int32_t dya_ready;
  void* du; void* dxn;                   /* bf16 scratch [rows_pad, hidden], [rows_pad, d]                              */
  float* g_w1; int32_t acc_w1; float* g_b1; int32_t acc_b1;
  float* g_w2; int32_t acc_w2; float* g_b2; int32_t acc_b2;
  float* g_ln_w; float* g_ln_b; int32_t acc_ln;
  float* dx;
  void* dx_bf16; const afft_dropout_t* up_drop; float* up_dcol;
  float* ln_partial;
  void* gemm_ws; int64_t gemm_ws_bytes; void* gemm_ws_aux; int64_t gemm_ws_aux_bytes;
  const afft_sgd_fused_t* sgd_w1; const afft_sgd_fused_t* sgd_w2;       /* as in afft_attn_sublayer_t */
  const void* w1_pk; const void* w2_pk;                                 /* as in afft_attn_sublayer_t */
  int32_t f16x2; void* xn_b; void* h_b;  /* as in afft_attn_sublayer_t: xn / h two-plane fp16 splits, w1 / w2 FP16 images, u stays bf16 */
  const void* w1_8; const void* w2_8;    /* f16x2 = 2: as in afft_attn_sublayer_t; the lo planes of xn and h are e4m3 byte planes            */
} afft_mlp_sublayer_t;
int afft_mlp_sublayer_fwd(const afft_mlp_sublayer_t* s, void* stream);
int afft_mlp_sublayer_bwd(const afft_mlp_sublayer_t* s, void* stream, void* aux_stream);

typedef struct {       /* y = x + drop(proj(attention(q = w_q LN_q(x), k = w_k LN_kv(mem), v = w_v LN_kv(mem))))      */
  /* DecoderBlock cross-attention half (CA-Fuser): models/transformerblock.py:56-76,160 ; bias-free w_q / w_k / w_v      */
  int32_t rows, d, L, H, mask, mask_period;
  float eps, scale;
  const float* x; const float* mem;      /* [rows, d] fp32 each                                                        */
  const float* nq_w; const float* nq_b; const float* nkv_w; const float* nkv_b;
  const void* w_q; const void* w_k; const void* w_v; const void* w_proj; int64_t ldw;   /* four [d, d] bf16 images     */
  const float* b_proj;
  float p_attn; uint32_t k_attn; afft_dropout_t out_drop;
  void* xq; void* mkv; void* q; void* k; void* v; void* ao;    /* bf16 [rows_pad, d] each                              */
  float* mean_q; float* rstd_q; float* mean_kv; float* rstd_kv; float* probs;
  float* y;
  const float* dy; void* dya; int32_t dya_ready;
  void* dao; void* dq; void* dk; void* dv; void* dxq;          /* bf16 scratch [rows_pad, d]                           */
  float* dmkv;                           /* fp32 scratch [rows, d]                                                     */
  float* g_w_q; int32_t acc_w_q; float* g_w_k; int32_t acc_w_k; float* g_w_v; int32_t acc_w_v;
  float* g_w_proj; int32_t acc_w_proj; float* g_b_proj; int32_t acc_b_proj;
  float* g_nq_w; float* g_nq_b; int32_t acc_nq; float* g_nkv_w; float* g_nkv_b; int32_t acc_nkv;
  float* dx; float* dmem;                /* [rows, d] fp32 each                                                        */
  void* dx_bf16; const afft_dropout_t* up_drop; float* up_dcol;
  float* ln_partial; float* ln_partial2;
  void* gemm_ws; int64_t gemm_ws_bytes; void* gemm_ws_aux; int64_t gemm_ws_aux_bytes;
  const afft_sgd_fused_t* sgd_w_q; const afft_sgd_fused_t* sgd_w_k; const afft_sgd_fused_t* sgd_w_v;
  const afft_sgd_fused_t* sgd_w_proj;                                    /* as in afft_attn_sublayer_t */
} afft_cross_attn_sublayer_t;
int afft_cross_attn_sublayer_fwd(const afft_cross_attn_sublayer_t* s, void* stream);
int afft_cross_attn_sublayer_bwd(const afft_cross_attn_sublayer_t* s, void* stream, void* aux_stream);

#ifdef __cplusplus
}
#endif
#endif /* AFFT_HIP_H */
