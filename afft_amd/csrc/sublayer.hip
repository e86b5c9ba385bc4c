// Composite entry points (include/afft_hip.h, "one call = one sub-layer"): host-side sequencing only.  Every kernel is
// launched through the primitive entry points of this library, in the order and with the arguments the call-by-call path
// (afft_amd/functional.py) uses; weight-gradient GEMMs and bias column sums go to the auxiliary stream behind an event.
#include "common.h"

namespace {

// ---- cross-stream ordering: "everything enqueued on `from` so far happens before what is enqueued on `to` next"
int stream_follows(hipStream_t to, hipStream_t from) { return afft_stream_follows(to, from); }

#define TRY(expr) do { if (int rc_ = (expr)) return rc_; } while (0)

int pad64(int n) { return (n + 63) / 64 * 64; }

struct Ws { void* p; int64_t bytes; };

afft_gemm_t gemm_base(int M, int N, int K, Ws ws) {
  afft_gemm_t g = {};
  g.M = M; g.N = N; g.K = K;
  g.dtype = AFFT_BF16;
  g.alpha = 1.0f;
  g.workspace = ws.p; g.workspace_bytes = ws.bytes;
  return g;
}

// out[rows, n_out] = epilogue(x[rows, k_in] W^T) (nn.Linear image [n_out, k_in]) or x W (Conv1D image [k_in, n_out])
afft_gemm_t lin_fwd(const void* x, int64_t ldx, int rows, int k_in, const void* W, int64_t ldw, int n_out, bool conv1d, Ws ws) {
  afft_gemm_t g = gemm_base(rows, n_out, k_in, ws);
  g.A = x; g.a_rs = ldx; g.a_cs = 1;
  g.B = W;
  if (conv1d) { g.b_rs = ldw; g.b_cs = 1; } else { g.b_rs = 1; g.b_cs = ldw; }
  return g;
}
// out[rows, k_in] = epilogue(dy[rows, n_out] W) (nn.Linear) or dy W^T (Conv1D)
afft_gemm_t lin_dgrad(const void* dy, int64_t lddy, int rows, int n_out, const void* W, int64_t ldw, int k_in, bool conv1d, Ws ws) {
  afft_gemm_t g = gemm_base(rows, k_in, n_out, ws);
  g.A = dy; g.a_rs = lddy; g.a_cs = 1;
  g.B = W;
  if (conv1d) { g.b_rs = 1; g.b_cs = ldw; } else { g.b_rs = ldw; g.b_cs = 1; }
  return g;
}
// dW (+)= dy^T x ([n_out, k_in], nn.Linear) or x^T dy ([k_in, n_out], Conv1D); the reduction runs over the padded rows
int wgrad(const void* dy, int64_t lddy, int n_out, const void* x, int64_t ldx, int k_in, int rows, bool conv1d, float* g_out,
          int acc, Ws ws, hipStream_t st, const afft_sgd_fused_t* sgd = nullptr) {
  if (!g_out) return 0;
  const void* a = conv1d ? x : dy; const void* b = conv1d ? dy : x;
  const int64_t lda = conv1d ? ldx : lddy, ldb = conv1d ? lddy : ldx;
  const int M = conv1d ? k_in : n_out, N = conv1d ? n_out : k_in;
  afft_gemm_t g = gemm_base(M, N, pad64(rows), ws);
  g.A = a; g.a_rs = 1; g.a_cs = lda;
  g.B = b; g.b_rs = ldb; g.b_cs = 1;
  g.out = g_out; g.ldo = N; g.out_dtype = AFFT_F32;
  g.accumulate = acc;
  if (sgd) {      // the update consumes the gradient in the epilogue: first (and only) contribution of the step
    AFFT_CHECK(!acc, "sublayer: a fused update needs the weight's only gradient contribution of the step");
    g.sgd = sgd;
  }
  return afft_gemm(&g, st);
}

int zero_row_tail(void* buf, int rows, int64_t width, hipStream_t st) {
  const int pr = pad64(rows);
  if (pr == rows) return 0;
  return afft_zero((char*)buf + (size_t)rows * width * 2, (int64_t)(pr - rows) * width * 2, st);      // widths are multiples of 64: 16-byte sizes
}

bool has_drop(const afft_dropout_t& d) { return d.p > 0.f || d.path_p > 0.f; }

// "fp16x2" forward: A is a two-plane fp16 split (lo plane a_lo elements behind the hi plane), the weight an FP16 image
void as_f16x2(afft_gemm_t& g, int64_t a_lo) { g.split3 = 2; g.a_lo = a_lo; g.b_lo = 0; g.b_packed = nullptr; }
// ... or ONE fp16 pass on A's hi plane (AFFT_F16X2_ONE_PASS_* sites)
void as_f16_one(afft_gemm_t& g) { g.split3 = 4; g.a_lo = 0; g.b_lo = 0; g.b_packed = nullptr; }
// ... with the lo pass on the block-scaled fp8 MFMA: a8 = the activation's e4m3 lo byte plane (row pitch = its element pitch in bytes),
// w8 = the weight's e4m3 byte image
void as_f16_lo8(afft_gemm_t& g, const void* a8, int64_t a8_ld, const void* w8, int64_t w8_ld) {
  g.split3 = 3; g.a_lo = 0; g.b_lo = 0; g.b_packed = nullptr; g.a8 = a8; g.a8_ld = a8_ld; g.b8 = w8; g.b8_ld = w8_ld;
}

}  // namespace

// ======================================================================================= self-attention sub-layer
extern "C" int afft_attn_sublayer_fwd(const afft_attn_sublayer_t* s, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AFFT_CHECK(s && s->x && s->w_qkv && s->w_proj && s->xn && s->qkv && s->ao && s->mean && s->rstd && s->probs && s->y,
             "attn_sublayer_fwd: null pointer");
  AFFT_CHECK(s->rows > 0 && s->L > 0 && s->rows % s->L == 0 && s->H > 0 && s->d % 64 == 0 && s->d % s->H == 0,
             "attn_sublayer_fwd: bad geometry (rows %d, L %d, d %d, H %d)", s->rows, s->L, s->d, s->H);
  const int R = s->rows, d = s->d;
  const int take = s->take > 1 ? s->take : 1, Ry = R / take;      // rows that leave the sub-layer (token 0 of every `take` rows)
  AFFT_CHECK(take == 1 || (take == s->L && Ry % 64 == 0), "attn_sublayer_fwd: take must be L, with rows / take a multiple of 64");
  const Ws ws = {s->gemm_ws, s->gemm_ws_bytes};
  if (s->f16x2) {
    // fp16 two-pass forward: every activation a GEMM reads is carried as hi + lo fp16 planes written by its producer (LayerNorm,
    // the qkv epilogue, the attention kernel); the bf16 copies (xn_b / qkv_b / ao_b) are what the bf16 backward reads
    const int pr = pad64(R);
    const int64_t lo1 = (int64_t)pr * d, lo3 = (int64_t)pr * 3 * d;
    const bool lo8 = (s->f16x2 & 3) == 2;      // lo planes of xn / ao as e4m3 bytes directly behind their hi planes, lo pass on the fp8 MFMA
    const bool one1 = s->f16x2 & AFFT_F16X2_ONE_PASS_1, one2 = s->f16x2 & AFFT_F16X2_ONE_PASS_2;      // qkv / proj: one fp16 pass, no lo plane of xn / ao
    const bool one_attn = s->f16x2 & AFFT_F16X2_ONE_PASS_ATTN;      // the attention core on the hi planes of q, k, v: no lo plane of qkv
    AFFT_CHECK(!lo8 || (!s->conv1d && (one1 || s->w_qkv8) && (one2 || s->w_proj8)), "attn_sublayer_fwd: f16x2 = 2 needs nn.Linear weights with e4m3 images");
    unsigned char* xn8 = (unsigned char*)s->xn + lo1 * 2;
    unsigned char* ao8 = (unsigned char*)s->ao + lo1 * 2;
    if (s->xn_b) TRY(zero_row_tail(s->xn_b, R, d, st));
    if (s->qkv_b) TRY(zero_row_tail(s->qkv_b, R, 3 * d, st));
    if (s->ao_b) TRY(zero_row_tail(s->ao_b, R, d, st));
    TRY(afft_layernorm_fwd_split(s->x, d, s->ln_w, s->ln_b, s->eps, R, d, s->xn, d, (lo8 || one1) ? 0 : lo1, s->xn_b, d, s->mean, s->rstd,
                                 (lo8 && !one1) ? xn8 : nullptr, st));
    afft_gemm_t g = lin_fwd(s->xn, d, R, d, s->w_qkv, s->ldw_qkv, 3 * d, s->conv1d, ws);
    if (one1) as_f16_one(g); else if (lo8) as_f16_lo8(g, xn8, d, s->w_qkv8, s->ldw_qkv); else as_f16x2(g, lo1);
    g.bias = s->b_qkv;
    g.out = s->qkv; g.ldo = 3 * d; g.out_dtype = AFFT_F16; g.out_lo = one_attn ? 0 : lo3;
    g.out2 = s->qkv_b; g.ldo2 = 3 * d; g.out2_dtype = AFFT_BF16;
    TRY(afft_gemm(&g, st));
    const char* q = (const char*)s->qkv;
    TRY(afft_attention_fwd_split(q, 3 * d, q + 2 * d, 3 * d, q + 4 * d, 3 * d, one_attn ? 0 : lo3, R / s->L, s->L, s->H, d / s->H, s->scale, s->mask,
                                 s->mask_period, s->p_attn, s->k_attn, s->ao, d, (lo8 || one2) ? 0 : lo1, s->ao_b, d, s->probs,
                                 (lo8 && !one2) ? ao8 : nullptr, st));
    g = lin_fwd(s->ao, (int64_t)d * take, Ry, d, s->w_proj, s->ldw_proj, d, s->conv1d, ws);
    if (one2) as_f16_one(g); else if (lo8) as_f16_lo8(g, ao8, (int64_t)d * take, s->w_proj8, s->ldw_proj); else as_f16x2(g, lo1);
    g.bias = s->b_proj;
    g.residual = s->x; g.ldres = (int64_t)d * take;
    g.drop = s->out_drop;
    g.out = s->y; g.ldo = d; g.out_dtype = AFFT_F32;
    return afft_gemm(&g, st);
  }
  TRY(zero_row_tail(s->xn, R, d, st));
  TRY(zero_row_tail(s->qkv, R, 3 * d, st));
  TRY(zero_row_tail(s->ao, R, d, st));
  TRY(afft_layernorm_fwd(s->x, d, s->ln_w, s->ln_b, s->eps, R, d, s->xn, d, AFFT_BF16, s->mean, s->rstd, st));
  afft_gemm_t g = lin_fwd(s->xn, d, R, d, s->w_qkv, s->ldw_qkv, 3 * d, s->conv1d, ws);
  g.bias = s->b_qkv;
  g.out = s->qkv; g.ldo = 3 * d; g.out_dtype = AFFT_BF16;
  if (!s->conv1d) g.b_packed = s->w_qkv_pk;
  TRY(afft_gemm(&g, st));
  const char* q = (const char*)s->qkv;
  TRY(afft_attention_fwd(q, 3 * d, q + 2 * d, 3 * d, q + 4 * d, 3 * d, AFFT_BF16, R / s->L, s->L, s->H, d / s->H, s->scale,
                         s->mask, s->mask_period, s->p_attn, s->k_attn, s->ao, d, s->probs, st));
  g = lin_fwd(s->ao, (int64_t)d * take, Ry, d, s->w_proj, s->ldw_proj, d, s->conv1d, ws);
  g.bias = s->b_proj;
  g.residual = s->x; g.ldres = (int64_t)d * take;
  g.drop = s->out_drop;
  g.out = s->y; g.ldo = d; g.out_dtype = AFFT_F32;
  if (!s->conv1d) g.b_packed = s->w_proj_pk;
  return afft_gemm(&g, st);
}

extern "C" int afft_attn_sublayer_bwd(const afft_attn_sublayer_t* s, void* stream_, void* aux_) {
  hipStream_t st = (hipStream_t)stream_, aux = aux_ ? (hipStream_t)aux_ : st;
  AFFT_CHECK(s && s->x && s->dy && s->dya && s->dao && s->dqkv && s->dxn && s->dx && s->ln_partial && s->xn && s->qkv && s->ao &&
             s->probs && s->mean && s->rstd, "attn_sublayer_bwd: null pointer");
  const int R = s->rows, d = s->d;
  const int take = s->take > 1 ? s->take : 1, Ry = R / take;      // see afft_attn_sublayer_fwd: dy / dya are [Ry, d]
  AFFT_CHECK(take == 1 || (take == s->L && Ry % 64 == 0), "attn_sublayer_bwd: take must be L, with rows / take a multiple of 64");
  const Ws ws = {s->gemm_ws, s->gemm_ws_bytes}, wsa = {aux == st ? s->gemm_ws : s->gemm_ws_aux, aux == st ? s->gemm_ws_bytes : s->gemm_ws_aux_bytes};
  const bool od = has_drop(s->out_drop);
  if (!s->dya_ready) {
    TRY(zero_row_tail(s->dya, Ry, d, st));
    TRY(afft_cast(s->dy, d, Ry, d, s->dya, d, AFFT_BF16, nullptr, 0, 0, od ? &s->out_drop : nullptr, st));
  }
  if (take > 1) {      // the projection's data gradient lands on every take-th row of dao: the rows between are zero
    TRY(afft_zero(s->dao, (int64_t)pad64(R) * d * 2, st));      // a kernel, not hipMemsetAsync: see afft_zero
  } else TRY(zero_row_tail(s->dao, R, d, st));
  TRY(zero_row_tail(s->dqkv, R, 3 * d, st));
  // A weight gradient with a fused update rewrites the weight's bf16 image: it is enqueued BEHIND the data-gradient GEMM that
  // reads that image (one event later than the plain form, which starts beside it).
  auto side_proj = [&]() -> int {
    TRY(stream_follows(aux, st));
    TRY(wgrad(s->dya, d, d, s->ao, (int64_t)d * take, d, Ry, s->conv1d, s->g_w_proj, s->acc_w_proj, wsa, aux, s->sgd_w_proj));
    if (s->g_b_proj) {
      if (od) TRY(afft_colsum(s->dya, d, AFFT_BF16, Ry, d, s->g_b_proj, s->acc_b_proj, wsa.p, wsa.bytes, aux));
      else TRY(afft_colsum(s->dy, d, AFFT_F32, Ry, d, s->g_b_proj, s->acc_b_proj, wsa.p, wsa.bytes, aux));
    }
    return 0;
  };
  if (!s->sgd_w_proj) TRY(side_proj());
  afft_gemm_t g = lin_dgrad(s->dya, d, Ry, d, s->w_proj, s->ldw_proj, d, s->conv1d, ws);
  g.out = s->dao; g.ldo = (int64_t)d * take; g.out_dtype = AFFT_BF16;
  TRY(afft_gemm(&g, st));
  if (s->sgd_w_proj) TRY(side_proj());
  const char* q = (const char*)s->qkv;
  char* dq = (char*)s->dqkv;
  TRY(afft_attention_bwd(s->dao, d, q, 3 * d, q + 2 * d, 3 * d, q + 4 * d, 3 * d, AFFT_BF16, s->probs, R / s->L, s->L, s->H,
                         d / s->H, s->scale, s->p_attn, s->k_attn, dq, 3 * d, dq + 2 * d, 3 * d, dq + 4 * d, 3 * d, st));
  auto side_qkv = [&]() -> int {
    TRY(stream_follows(aux, st));
    TRY(wgrad(s->dqkv, 3 * d, 3 * d, s->xn, d, d, R, s->conv1d, s->g_w_qkv, s->acc_w_qkv, wsa, aux, s->sgd_w_qkv));
    if (s->g_b_qkv) TRY(afft_colsum(s->dqkv, 3 * d, AFFT_BF16, R, 3 * d, s->g_b_qkv, s->acc_b_qkv, wsa.p, wsa.bytes, aux));
    return 0;
  };
  if (!s->sgd_w_qkv) TRY(side_qkv());
  g = lin_dgrad(s->dqkv, 3 * d, R, 3 * d, s->w_qkv, s->ldw_qkv, d, s->conv1d, ws);
  g.out = s->dxn; g.ldo = d; g.out_dtype = AFFT_BF16;
  TRY(afft_gemm(&g, st));
  if (s->sgd_w_qkv) TRY(side_qkv());
  return afft_layernorm_bwd_take(s->dxn, d, AFFT_BF16, s->x, d, s->ln_w, s->mean, s->rstd, R, d, s->dy, d, take, s->dx, d, s->dx_bf16,
                                 s->up_drop, s->g_ln_w, s->g_ln_b, s->acc_ln, s->up_dcol, 0, s->ln_partial, st);
}

// ======================================================================================= MLP sub-layer
extern "C" int afft_mlp_sublayer_fwd(const afft_mlp_sublayer_t* s, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AFFT_CHECK(s && s->x && s->w1 && s->w2 && s->xn && s->h && s->mean && s->rstd && s->y, "mlp_sublayer_fwd: null pointer");   // u may be NULL
  AFFT_CHECK(s->rows > 0 && s->d % 64 == 0 && s->hidden % 64 == 0, "mlp_sublayer_fwd: bad geometry");
  AFFT_CHECK(s->gelu == AFFT_ACT_GELU_ERF || s->gelu == AFFT_ACT_GELU_TANH, "mlp_sublayer_fwd: gelu must be GELU_ERF or GELU_TANH");
  const int R = s->rows, d = s->d, hd = s->hidden;
  const Ws ws = {s->gemm_ws, s->gemm_ws_bytes};
  if (s->f16x2) {      // see afft_attn_sublayer_fwd
    const int pr = pad64(R);
    const int64_t lo1 = (int64_t)pr * d, loh = (int64_t)pr * hd;
    const bool lo8 = (s->f16x2 & 3) == 2;      // see afft_attn_sublayer_fwd
    const bool one1 = s->f16x2 & AFFT_F16X2_ONE_PASS_1, one2 = s->f16x2 & AFFT_F16X2_ONE_PASS_2;      // fc1 / fc2: one fp16 pass, no lo plane of xn / h
    AFFT_CHECK(!lo8 || (!s->conv1d && (one1 || s->w1_8) && (one2 || s->w2_8)), "mlp_sublayer_fwd: f16x2 = 2 needs nn.Linear weights with e4m3 images");
    unsigned char* xn8 = (unsigned char*)s->xn + lo1 * 2;
    unsigned char* h8 = (unsigned char*)s->h + loh * 2;
    if (s->xn_b) TRY(zero_row_tail(s->xn_b, R, d, st));
    if (s->h_b) TRY(zero_row_tail(s->h_b, R, hd, st));
    if (s->u) TRY(zero_row_tail(s->u, R, hd, st));
    TRY(afft_layernorm_fwd_split(s->x, d, s->ln_w, s->ln_b, s->eps, R, d, s->xn, d, (lo8 || one1) ? 0 : lo1, s->xn_b, d, s->mean, s->rstd,
                                 (lo8 && !one1) ? xn8 : nullptr, st));
    afft_gemm_t g = lin_fwd(s->xn, d, R, d, s->w1, s->ldw1, hd, s->conv1d, ws);
    if (one1) as_f16_one(g); else if (lo8) as_f16_lo8(g, xn8, d, s->w1_8, s->ldw1); else as_f16x2(g, lo1);
    g.bias = s->b1;
    g.act = s->gelu;
    g.pre = s->u; g.ldpre = hd; g.pre_dtype = AFFT_BF16;
    g.out = s->h; g.ldo = hd; g.out_dtype = AFFT_F16;
    if (!one2) { if (lo8) g.out_lo8 = h8; else g.out_lo = loh; }      // fc2 on one pass: h goes out as its hi plane alone
    g.out2 = s->h_b; g.ldo2 = hd; g.out2_dtype = AFFT_BF16;
    TRY(afft_gemm(&g, st));
    g = lin_fwd(s->h, hd, R, hd, s->w2, s->ldw2, d, s->conv1d, ws);
    if (one2) as_f16_one(g); else if (lo8) as_f16_lo8(g, h8, hd, s->w2_8, s->ldw2); else as_f16x2(g, loh);
    g.bias = s->b2;
    g.residual = s->x; g.ldres = d;
    g.drop = s->out_drop;
    g.out = s->y; g.ldo = d; g.out_dtype = AFFT_F32;
    return afft_gemm(&g, st);
  }
  TRY(zero_row_tail(s->xn, R, d, st));
  if (s->u) TRY(zero_row_tail(s->u, R, hd, st));
  TRY(zero_row_tail(s->h, R, hd, st));
  TRY(afft_layernorm_fwd(s->x, d, s->ln_w, s->ln_b, s->eps, R, d, s->xn, d, AFFT_BF16, s->mean, s->rstd, st));
  afft_gemm_t g = lin_fwd(s->xn, d, R, d, s->w1, s->ldw1, hd, s->conv1d, ws);
  g.bias = s->b1;
  g.act = s->gelu;
  g.pre = s->u; g.ldpre = hd; g.pre_dtype = AFFT_BF16;
  g.out = s->h; g.ldo = hd; g.out_dtype = AFFT_BF16;
  if (!s->conv1d) g.b_packed = s->w1_pk;
  TRY(afft_gemm(&g, st));
  g = lin_fwd(s->h, hd, R, hd, s->w2, s->ldw2, d, s->conv1d, ws);
  g.bias = s->b2;
  g.residual = s->x; g.ldres = d;
  g.drop = s->out_drop;
  g.out = s->y; g.ldo = d; g.out_dtype = AFFT_F32;
  if (!s->conv1d) g.b_packed = s->w2_pk;
  return afft_gemm(&g, st);
}

extern "C" int afft_mlp_sublayer_bwd(const afft_mlp_sublayer_t* s, void* stream_, void* aux_) {
  hipStream_t st = (hipStream_t)stream_, aux = aux_ ? (hipStream_t)aux_ : st;
  AFFT_CHECK(s && s->x && s->dy && s->dya && s->du && s->dxn && s->dx && s->ln_partial && s->xn && s->u && s->h && s->mean && s->rstd,
             "mlp_sublayer_bwd: null pointer");
  const int R = s->rows, d = s->d, hd = s->hidden;
  const Ws ws = {s->gemm_ws, s->gemm_ws_bytes}, wsa = {aux == st ? s->gemm_ws : s->gemm_ws_aux, aux == st ? s->gemm_ws_bytes : s->gemm_ws_aux_bytes};
  const bool od = has_drop(s->out_drop);
  if (!s->dya_ready) {
    TRY(zero_row_tail(s->dya, R, d, st));
    TRY(afft_cast(s->dy, d, R, d, s->dya, d, AFFT_BF16, nullptr, 0, 0, od ? &s->out_drop : nullptr, st));
  }
  TRY(zero_row_tail(s->du, R, hd, st));
  auto side_fc2 = [&]() -> int {      // see afft_attn_sublayer_bwd for the ordering of a fused update
    TRY(stream_follows(aux, st));
    TRY(wgrad(s->dya, d, d, s->h, hd, hd, R, s->conv1d, s->g_w2, s->acc_w2, wsa, aux, s->sgd_w2));
    if (s->g_b2) {
      if (od) TRY(afft_colsum(s->dya, d, AFFT_BF16, R, d, s->g_b2, s->acc_b2, wsa.p, wsa.bytes, aux));
      else TRY(afft_colsum(s->dy, d, AFFT_F32, R, d, s->g_b2, s->acc_b2, wsa.p, wsa.bytes, aux));
    }
    return 0;
  };
  if (!s->sgd_w2) TRY(side_fc2());
  afft_gemm_t g = lin_dgrad(s->dya, d, R, d, s->w2, s->ldw2, hd, s->conv1d, ws);
  g.act = s->gelu == AFFT_ACT_GELU_ERF ? AFFT_ACT_DGELU_ERF : AFFT_ACT_DGELU_TANH;
  g.aux = s->u; g.ldaux = hd; g.aux_dtype = AFFT_BF16;
  g.out = s->du; g.ldo = hd; g.out_dtype = AFFT_BF16;
  TRY(afft_gemm(&g, st));
  if (s->sgd_w2) TRY(side_fc2());
  auto side_fc1 = [&]() -> int {
    TRY(stream_follows(aux, st));
    TRY(wgrad(s->du, hd, hd, s->xn, d, d, R, s->conv1d, s->g_w1, s->acc_w1, wsa, aux, s->sgd_w1));
    if (s->g_b1) TRY(afft_colsum(s->du, hd, AFFT_BF16, R, hd, s->g_b1, s->acc_b1, wsa.p, wsa.bytes, aux));
    return 0;
  };
  if (!s->sgd_w1) TRY(side_fc1());
  g = lin_dgrad(s->du, hd, R, hd, s->w1, s->ldw1, d, s->conv1d, ws);
  g.out = s->dxn; g.ldo = d; g.out_dtype = AFFT_BF16;
  TRY(afft_gemm(&g, st));
  if (s->sgd_w1) TRY(side_fc1());
  return afft_layernorm_bwd_take(s->dxn, d, AFFT_BF16, s->x, d, s->ln_w, s->mean, s->rstd, R, d, s->dy, d, 1, s->dx, d, s->dx_bf16, s->up_drop,
                                 s->g_ln_w, s->g_ln_b, s->acc_ln, s->up_dcol, 0, s->ln_partial, st);
}

// ======================================================================================= cross-attention sub-layer
extern "C" int afft_cross_attn_sublayer_fwd(const afft_cross_attn_sublayer_t* s, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  AFFT_CHECK(s && s->x && s->mem && s->w_q && s->w_k && s->w_v && s->w_proj && s->xq && s->mkv && s->q && s->k && s->v && s->ao &&
             s->mean_q && s->rstd_q && s->mean_kv && s->rstd_kv && s->probs && s->y, "cross_attn_sublayer_fwd: null pointer");
  AFFT_CHECK(s->rows > 0 && s->L > 0 && s->rows % s->L == 0 && s->H > 0 && s->d % 64 == 0 && s->d % s->H == 0,
             "cross_attn_sublayer_fwd: bad geometry");
  const int R = s->rows, d = s->d;
  const Ws ws = {s->gemm_ws, s->gemm_ws_bytes};
  void* bufs[6] = {s->xq, s->mkv, s->q, s->k, s->v, s->ao};
  for (void* b : bufs) TRY(zero_row_tail(b, R, d, st));
  TRY(afft_layernorm_fwd(s->x, d, s->nq_w, s->nq_b, s->eps, R, d, s->xq, d, AFFT_BF16, s->mean_q, s->rstd_q, st));
  TRY(afft_layernorm_fwd(s->mem, d, s->nkv_w, s->nkv_b, s->eps, R, d, s->mkv, d, AFFT_BF16, s->mean_kv, s->rstd_kv, st));
  const void* src[3] = {s->xq, s->mkv, s->mkv};
  const void* w[3] = {s->w_q, s->w_k, s->w_v};
  void* dst[3] = {s->q, s->k, s->v};
  for (int i = 0; i < 3; ++i) {
    afft_gemm_t g = lin_fwd(src[i], d, R, d, w[i], s->ldw, d, false, ws);
    g.out = dst[i]; g.ldo = d; g.out_dtype = AFFT_BF16;
    TRY(afft_gemm(&g, st));
  }
  TRY(afft_attention_fwd(s->q, d, s->k, d, s->v, d, AFFT_BF16, R / s->L, s->L, s->H, d / s->H, s->scale, s->mask, s->mask_period,
                         s->p_attn, s->k_attn, s->ao, d, s->probs, st));
  afft_gemm_t g = lin_fwd(s->ao, d, R, d, s->w_proj, s->ldw, d, false, ws);
  g.bias = s->b_proj;
  g.residual = s->x; g.ldres = d;
  g.drop = s->out_drop;
  g.out = s->y; g.ldo = d; g.out_dtype = AFFT_F32;
  return afft_gemm(&g, st);
}

extern "C" int afft_cross_attn_sublayer_bwd(const afft_cross_attn_sublayer_t* s, void* stream_, void* aux_) {
  hipStream_t st = (hipStream_t)stream_, aux = aux_ ? (hipStream_t)aux_ : st;
  AFFT_CHECK(s && s->x && s->mem && s->dy && s->dya && s->dao && s->dq && s->dk && s->dv && s->dxq && s->dmkv && s->dx && s->dmem &&
             s->ln_partial && s->ln_partial2, "cross_attn_sublayer_bwd: null pointer");
  const int R = s->rows, d = s->d;
  const Ws ws = {s->gemm_ws, s->gemm_ws_bytes}, wsa = {aux == st ? s->gemm_ws : s->gemm_ws_aux, aux == st ? s->gemm_ws_bytes : s->gemm_ws_aux_bytes};
  const bool od = has_drop(s->out_drop);
  if (!s->dya_ready) {
    TRY(zero_row_tail(s->dya, R, d, st));
    TRY(afft_cast(s->dy, d, R, d, s->dya, d, AFFT_BF16, nullptr, 0, 0, od ? &s->out_drop : nullptr, st));
  }
  void* bufs[4] = {s->dao, s->dq, s->dk, s->dv};
  for (void* b : bufs) TRY(zero_row_tail(b, R, d, st));
  auto side_proj = [&]() -> int {      // see afft_attn_sublayer_bwd for the ordering of a fused update
    TRY(stream_follows(aux, st));
    TRY(wgrad(s->dya, d, d, s->ao, d, d, R, false, s->g_w_proj, s->acc_w_proj, wsa, aux, s->sgd_w_proj));
    if (s->g_b_proj) {
      if (od) TRY(afft_colsum(s->dya, d, AFFT_BF16, R, d, s->g_b_proj, s->acc_b_proj, wsa.p, wsa.bytes, aux));
      else TRY(afft_colsum(s->dy, d, AFFT_F32, R, d, s->g_b_proj, s->acc_b_proj, wsa.p, wsa.bytes, aux));
    }
    return 0;
  };
  if (!s->sgd_w_proj) TRY(side_proj());
  afft_gemm_t g = lin_dgrad(s->dya, d, R, d, s->w_proj, s->ldw, d, false, ws);
  g.out = s->dao; g.ldo = d; g.out_dtype = AFFT_BF16;
  TRY(afft_gemm(&g, st));
  if (s->sgd_w_proj) TRY(side_proj());
  TRY(afft_attention_bwd(s->dao, d, s->q, d, s->k, d, s->v, d, AFFT_BF16, s->probs, R / s->L, s->L, s->H, d / s->H, s->scale,
                         s->p_attn, s->k_attn, s->dq, d, s->dk, d, s->dv, d, st));
  const bool fused_qkv = s->sgd_w_q || s->sgd_w_k || s->sgd_w_v;
  auto side_qkv = [&]() -> int {
    TRY(stream_follows(aux, st));
    TRY(wgrad(s->dq, d, d, s->xq, d, d, R, false, s->g_w_q, s->acc_w_q, wsa, aux, s->sgd_w_q));
    TRY(wgrad(s->dk, d, d, s->mkv, d, d, R, false, s->g_w_k, s->acc_w_k, wsa, aux, s->sgd_w_k));
    TRY(wgrad(s->dv, d, d, s->mkv, d, d, R, false, s->g_w_v, s->acc_w_v, wsa, aux, s->sgd_w_v));
    return 0;
  };
  if (!fused_qkv) TRY(side_qkv());
  g = lin_dgrad(s->dk, d, R, d, s->w_k, s->ldw, d, false, ws);
  g.out = s->dmkv; g.ldo = d; g.out_dtype = AFFT_F32;
  TRY(afft_gemm(&g, st));
  g = lin_dgrad(s->dv, d, R, d, s->w_v, s->ldw, d, false, ws);
  g.out = s->dmkv; g.ldo = d; g.out_dtype = AFFT_F32;
  g.accumulate = 1;
  TRY(afft_gemm(&g, st));
  g = lin_dgrad(s->dq, d, R, d, s->w_q, s->ldw, d, false, ws);
  g.out = s->dxq; g.ldo = d; g.out_dtype = AFFT_BF16;
  TRY(afft_gemm(&g, st));
  if (fused_qkv) TRY(side_qkv());
  TRY(afft_layernorm_bwd(s->dmkv, d, AFFT_F32, s->mem, d, s->nkv_w, s->mean_kv, s->rstd_kv, R, d, nullptr, s->dmem, d, nullptr, nullptr,
                         s->g_nkv_w, s->g_nkv_b, s->acc_nkv, nullptr, 0, s->ln_partial2, st));
  return afft_layernorm_bwd(s->dxq, d, AFFT_BF16, s->x, d, s->nq_w, s->mean_q, s->rstd_q, R, d, s->dy, s->dx, d, s->dx_bf16, s->up_drop,
                            s->g_nq_w, s->g_nq_b, s->acc_nq, s->up_dcol, 0, s->ln_partial, st);
}
