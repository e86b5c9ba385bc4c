// Fused softmax cross-entropy (forward + gradient in one pass over the logits) and MSE.
// Semantics: common/runner.py:13-37 (MultiDimCrossEntropy, reduction='none', ignore_index=-1 or a
// boolean row filter for soft targets), :164-166 (MSELoss, both arguments carry gradient).
#include "common.h"

namespace {

__device__ __forceinline__ float block_reduce(float v, float* sh, bool is_max) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = sh[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = is_max ? fmaxf(r, sh[w]) : r + sh[w];
  return r;
}

__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ logits, int64_t ldl, int C,
                                                         const int64_t* __restrict__ labels,
                                                         const float* __restrict__ soft, int64_t lds,
                                                         const uint8_t* __restrict__ keep, float gscale,
                                                         const float* __restrict__ row_g,
                                                         void* __restrict__ dlogits,
                                                         int64_t ldd, int d_dtype, float* __restrict__ row_loss,
                                                         int period, int64_t lgs, int64_t dgs) {
  // row r = frame r % period of clip r / period: logits at clip * lgs + frame * ldl, its gradient at clip * dgs + frame * ldd
  // (a (B, n, C) slice of a wider (B, L, C) tensor is walked where it lies; period = rows: plain [rows, C])
  __shared__ float sh[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const int clip = row / period, frame = row - clip * period;
  const float* x = logits + (int64_t)clip * lgs + (int64_t)frame * ldl;
  const int64_t drow = (int64_t)clip * dgs + (int64_t)frame * ldd;
  bool kept = true;
  int64_t lab = -1;
  if (labels) { lab = labels[row]; kept = lab != -1; }
  // a label outside [-1, C) (torch raises a device-side assert there): nothing out of bounds is read, and the row's loss
  // and gradient come out NaN so that the total cannot look healthy
  const bool bad = labels && (lab >= C || lab < -1);
  if (keep) kept = kept && keep[row] != 0;
  if (!kept) {  // uniform per block
    if (dlogits) for (int c = tid; c < ldd; c += 256) st_any(dlogits, drow + c, d_dtype, 0.f);
    if (row_loss && tid == 0) row_loss[row] = 0.f;
    return;
  }
  float m = -INFINITY;
  for (int c = tid; c < C; c += 256) m = fmaxf(m, x[c]);
  m = block_reduce(m, sh, true);
  float se = 0.f, tsum = 0.f, tx = 0.f;
  const float* t = soft ? soft + (int64_t)row * lds : nullptr;
  for (int c = tid; c < C; c += 256) {
    se += expf(x[c] - m);
    if (t) { tsum += t[c]; tx += t[c] * x[c]; }
  }
  se = block_reduce(se, sh, false);
  const float lse = m + logf(se);
  float loss;
  if (t) {
    tsum = block_reduce(tsum, sh, false);
    tx = block_reduce(tx, sh, false);
    loss = lse * tsum - tx;
  } else {
    tsum = 1.f;
    loss = bad ? NAN : lse - x[lab];
  }
  if (tid == 0 && row_loss) row_loss[row] = loss;
  if (dlogits) {
    const float inv = 1.0f / se;
    if (row_g) gscale *= row_g[row];
    for (int c = tid; c < ldd; c += 256) {
      float g = 0.f;
      if (c < C) {
        const float p = expf(x[c] - m) * inv;
        const float tc = t ? t[c] : (c == lab ? 1.f : 0.f);
        g = bad ? NAN : gscale * (p * tsum - tc);
      }
      st_any(dlogits, drow + c, d_dtype, g);
    }
  }
}

__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b,
                                                  int64_t ldb, int rows, int d, float gscale,
                                                  const float* __restrict__ g_dev, float lscale,
                                                  float* __restrict__ loss_sum, float* __restrict__ partials,
                                                  float* __restrict__ da, int64_t ldda, float* __restrict__ db,
                                                  int64_t lddb, int accumulate) {
  __shared__ float sh[4];
  float acc = 0.f;
  const int64_t total = (int64_t)rows * d;
  if (g_dev) gscale *= g_dev[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / d), c = (int)(i - (int64_t)r * d);
    const float diff = a[(int64_t)r * lda + c] - b[(int64_t)r * ldb + c];
    acc += diff * diff;
    const float g = 2.f * gscale * diff;
    if (da) da[(int64_t)r * ldda + c] += g;
    if (db) db[(int64_t)r * lddb + c] -= g;
  }
  if (!loss_sum) return;
  acc = block_reduce(acc, sh, false);
  // one workgroup: it IS the sum.  More: the partial goes to the caller's scratch and ordered_sum_kernel adds them up in
  // workgroup order behind this kernel (no float atomics: the loss scalar has the same bits on every run)
  if (threadIdx.x == 0) {
    if (gridDim.x == 1) *loss_sum = (accumulate ? *loss_sum : 0.f) + acc * lscale; else partials[blockIdx.x] = acc;
  }
}

// Backward of the MSE between frame ranges of two (clips, frames, C) tensors: the FULL gradients of both are written in one pass
// -- 2 g (a - b) inside the ranges, zeros outside -- so that no fill kernel runs in front and no slice-backward behind.
__global__ __launch_bounds__(256) void mse_frames_bwd_kernel(const float* __restrict__ a, int64_t a_cs, int a_off, int a_len,
                                                             const float* __restrict__ b, int64_t b_cs, int b_off, int b_len,
                                                             int clips, int n, float gscale, const float* __restrict__ g_dev,
                                                             float* __restrict__ da, float* __restrict__ db) {
  if (g_dev) gscale *= g_dev[0];
  const int clip = blockIdx.y;
  const float* ar = a + (int64_t)clip * a_cs;
  const float* br = b + (int64_t)clip * b_cs;
  float* dar = da ? da + (int64_t)clip * a_len : nullptr;
  float* dbr = db ? db + (int64_t)clip * b_len : nullptr;
  const int span = max(a_len, b_len);
  for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < span; i += gridDim.x * 1024) {      // lengths and offsets are multiples of 4
    if (dar && i < a_len) {
      const int j = i - a_off;
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
      if (j >= 0 && j < n) {
        const f32x4 x = *(const f32x4*)(ar + i), y = *(const f32x4*)(br + b_off + j);
        for (int e = 0; e < 4; ++e) g[e] = 2.f * gscale * (x[e] - y[e]);
      }
      *(f32x4*)(dar + i) = g;
    }
    if (dbr && i < b_len) {
      const int j = i - b_off;
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
      if (j >= 0 && j < n) {
        const f32x4 x = *(const f32x4*)(ar + a_off + j), y = *(const f32x4*)(br + i);
        for (int e = 0; e < 4; ++e) g[e] = -2.f * gscale * (x[e] - y[e]);
      }
      *(f32x4*)(dbr + i) = g;
    }
  }
}

// Runner._reduce_loss in one launch: term i = n[i] floats at x[i]; means[i] = their mean (lane-strided partial sums, then a fixed
// tree: the same bits on every run), total = sum_i w[i] * means[i].  One workgroup (the terms are a few thousand values).
struct LossTerms { const float* x[8]; float* g[8]; int64_t n[8]; float w[8]; int nterms; };
__global__ __launch_bounds__(256) void loss_reduce_kernel(const LossTerms t, float* __restrict__ means, float* __restrict__ total) {
  __shared__ float sh[4];
  float tot = 0.f;
  for (int i = 0; i < t.nterms; ++i) {
    float s = 0.f;
    for (int64_t j = threadIdx.x; j < t.n[i]; j += 256) s += t.x[i][j];
    s = block_reduce(s, sh, false);
    const float m = t.n[i] > 0 ? s / (float)t.n[i] : 0.f;
    if (threadIdx.x == 0 && means) means[i] = m;
    if (t.w[i] != 0.f) tot += t.w[i] * m;     // a weight of 0 DROPS the term (runner.py:205-207): its NaN / inf must not reach the total
  }
  if (threadIdx.x == 0) *total = tot;
}
// its backward: g[i][:] = g_total * w[i] / n[i]
__global__ __launch_bounds__(256) void loss_reduce_bwd_kernel(const LossTerms t, const float* __restrict__ g_total,
                                                              const float* __restrict__ total, float* __restrict__ ok) {
  const float go = g_total ? g_total[0] : 1.f;
  const int i = blockIdx.y;
  // the first kernel of a backward pass: "is this step finite" for the optimizer kernels that follow it (afft_sgd_fused_t.ok)
  if (ok && blockIdx.x == 0 && i == 0 && threadIdx.x == 0) *ok = (!total || isfinite(total[0])) && isfinite(go) ? 1.f : 0.f;
  if (!t.g[i] || t.n[i] <= 0) return;
  const float v = go * t.w[i] / (float)t.n[i];
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < t.n[i]; j += (int64_t)gridDim.x * 256) t.g[i][j] = v;
}

}  // namespace

extern "C" int afft_loss_reduce(const float* const* x, const int64_t* n, const float* w, int32_t nterms, float* means, float* total,
                                void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(x && n && w && total, "loss_reduce: null pointer");
  AFFT_CHECK(nterms >= 1 && nterms <= 8, "loss_reduce: 1..8 terms (got %d)", nterms);
  LossTerms t = {};
  t.nterms = nterms;
  for (int i = 0; i < nterms; ++i) { AFFT_CHECK(x[i] || n[i] == 0, "loss_reduce: null term"); t.x[i] = x[i]; t.n[i] = n[i]; t.w[i] = w[i]; }
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, stream, t, means, total);
  AFFT_LAUNCH_CHECK();
  return 0;
}

extern "C" int afft_loss_reduce_bwd(float* const* g, const int64_t* n, const float* w, int32_t nterms, const float* g_total,
                                    void* stream_) {
  return afft_loss_reduce_bwd_ok(g, n, w, nterms, g_total, nullptr, nullptr, stream_);
}

extern "C" int afft_loss_reduce_bwd_ok(float* const* g, const int64_t* n, const float* w, int32_t nterms, const float* g_total,
                                       const float* total, float* ok, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(g && n && w, "loss_reduce_bwd: null pointer");
  AFFT_CHECK(nterms >= 1 && nterms <= 8, "loss_reduce_bwd: 1..8 terms (got %d)", nterms);
  LossTerms t = {};
  t.nterms = nterms;
  int64_t nmax = 1;
  for (int i = 0; i < nterms; ++i) { t.g[i] = g[i]; t.n[i] = n[i]; t.w[i] = w[i]; nmax = n[i] > nmax ? n[i] : nmax; }
  const int gx = (int)((nmax + 255) / 256 > 64 ? 64 : (nmax + 255) / 256);
  hipLaunchKernelGGL(loss_reduce_bwd_kernel, dim3(gx, nterms), dim3(256), 0, stream, t, g_total, total, ok);
  AFFT_LAUNCH_CHECK();
  return 0;
}

extern "C" int afft_softmax_ce(const float* logits, int64_t ldl, int32_t rows, int32_t C, const int64_t* labels,
                               const float* soft, int64_t lds, const uint8_t* keep, float gscale, const float* row_g,
                               float* loss_sum, void* dlogits, int64_t ldd, int32_t d_dtype, float* row_loss,
                               void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(logits, "softmax_ce: null logits");
  AFFT_CHECK((labels != nullptr) != (soft != nullptr), "softmax_ce: give exactly one of labels / soft targets");
  AFFT_CHECK(C > 0 && ldl >= C && (!dlogits || ldd >= C), "softmax_ce: bad sizes");
  AFFT_CHECK(!loss_sum || row_loss, "softmax_ce: loss_sum is the ordered sum of row_loss: give row_loss as well");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(softmax_ce_kernel, dim3(rows), dim3(256), 0, stream, logits, ldl, C, labels, soft, lds, keep,
                     gscale, row_g, dlogits, ldd, d_dtype, row_loss, rows, (int64_t)0, (int64_t)0);
  AFFT_LAUNCH_CHECK();
  if (loss_sum) {
    hipLaunchKernelGGL(ordered_sum_kernel, dim3(1), dim3(256), 0, stream, row_loss, (int64_t)rows, 1.0f, loss_sum, 1);
    AFFT_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int afft_softmax_ce_frames(const float* logits, int64_t clip_stride, int64_t ldl, int32_t clips, int32_t frames, int32_t C,
                                      const int64_t* labels, const float* soft, int64_t lds, const uint8_t* keep, float gscale,
                                      const float* row_g, void* dlogits, int64_t d_clip_stride, int64_t ldd, int32_t d_dtype,
                                      float* row_loss, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(logits, "softmax_ce_frames: null logits");
  AFFT_CHECK((labels != nullptr) != (soft != nullptr), "softmax_ce_frames: give exactly one of labels / soft targets");
  AFFT_CHECK(C > 0 && ldl >= C && (!dlogits || ldd >= C) && frames > 0, "softmax_ce_frames: bad sizes");
  if (clips == 0) return 0;
  hipLaunchKernelGGL(softmax_ce_kernel, dim3(clips * frames), dim3(256), 0, stream, logits, ldl, C, labels, soft, lds, keep,
                     gscale, row_g, dlogits, ldd, d_dtype, row_loss, frames, clip_stride, d_clip_stride);
  AFFT_LAUNCH_CHECK();
  return 0;
}

extern "C" int afft_mse_frames_bwd(const float* a, int64_t a_clip_stride, int32_t a_off, int32_t a_len, const float* b,
                                   int64_t b_clip_stride, int32_t b_off, int32_t b_len, int32_t clips, int32_t n, float gscale,
                                   const float* g_dev, float* da, float* db, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(a && b, "mse_frames_bwd: null pointer");
  AFFT_CHECK(((a_off | a_len | b_off | b_len | n) & 3) == 0 && (a_clip_stride & 3) == 0 && (b_clip_stride & 3) == 0,
             "mse_frames_bwd: offsets, lengths and clip strides must be multiples of 4 floats");
  AFFT_CHECK(a_off >= 0 && b_off >= 0 && a_off + n <= a_len && b_off + n <= b_len, "mse_frames_bwd: the frame range leaves the tensor");
  AFFT_CHECK((((uintptr_t)a | (uintptr_t)b | (uintptr_t)da | (uintptr_t)db) & 15) == 0, "mse_frames_bwd: 16-byte aligned pointers");
  if (clips == 0 || (!da && !db)) return 0;
  const int span = a_len > b_len ? a_len : b_len;
  int gx = (span / 4 + 255) / 256;
  if (gx > 64) gx = 64;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(mse_frames_bwd_kernel, dim3(gx, clips), dim3(256), 0, stream, a, a_clip_stride, a_off, a_len, b, b_clip_stride,
                     b_off, b_len, clips, n, gscale, g_dev, da, db);
  AFFT_LAUNCH_CHECK();
  return 0;
}

static int mse_launch(const float* a, int64_t lda, const float* b, int64_t ldb, int32_t rows, int32_t d,
                      float gscale, const float* g_dev, float lscale, float* loss_sum, float* da, int64_t ldda,
                      float* db, int64_t lddb, void* workspace, int64_t workspace_bytes, int accumulate, hipStream_t stream) {
  AFFT_CHECK(a && b, "mse: null pointer");
  if (rows == 0 || d == 0) return 0;
  const int64_t total = (int64_t)rows * d;
  int grid = (int)((total + 255) / 256);
  if (grid > AFFT_REDUCE_PARTIALS) grid = AFFT_REDUCE_PARTIALS;
  AFFT_CHECK(!loss_sum || grid == 1 || (workspace && workspace_bytes >= AFFT_GEMM_WS_HEADER + 4 * (int64_t)grid),
             "mse: the loss sum needs the stream's workspace (header + AFFT_REDUCE_PARTIALS floats)");
  float* scratch = workspace ? (float*)((char*)workspace + AFFT_GEMM_WS_HEADER) : nullptr;
  hipLaunchKernelGGL(mse_kernel, dim3(grid), dim3(256), 0, stream, a, lda, b, ldb, rows, d, gscale, g_dev, lscale, loss_sum,
                     scratch, da, ldda, db, lddb, accumulate);
  AFFT_LAUNCH_CHECK();
  if (loss_sum && grid > 1) {
    hipLaunchKernelGGL(ordered_sum_kernel, dim3(1), dim3(256), 0, stream, scratch, (int64_t)grid, lscale, loss_sum, accumulate);
    AFFT_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int afft_mse(const float* a, int64_t lda, const float* b, int64_t ldb, int32_t rows, int32_t d,
                        float gscale, const float* g_dev, float lscale, float* loss_sum, float* da, int64_t ldda,
                        float* db, int64_t lddb, void* workspace, int64_t workspace_bytes, void* stream_) {
  return mse_launch(a, lda, b, ldb, rows, d, gscale, g_dev, lscale, loss_sum, da, ldda, db, lddb, workspace, workspace_bytes, 1,
                    (hipStream_t)stream_);
}

extern "C" int afft_mse_loss(const float* a, int64_t lda, const float* b, int64_t ldb, int32_t rows, int32_t d, float lscale,
                             float* loss, void* workspace, int64_t workspace_bytes, void* stream_) {
  AFFT_CHECK(loss, "mse_loss: null output");
  return mse_launch(a, lda, b, ldb, rows, d, 0.f, nullptr, lscale, loss, nullptr, 0, nullptr, 0, workspace, workspace_bytes, 0,
                    (hipStream_t)stream_);
}
