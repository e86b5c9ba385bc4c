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
                                                         int64_t ldd, int d_dtype, float* __restrict__ row_loss) {
  __shared__ float sh[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* x = logits + (int64_t)row * ldl;
  bool kept = true;
  int64_t lab = -1;
  if (labels) { lab = labels[row]; kept = lab != -1; }
  // a label outside [-1, C) (torch raises a device-side assert there): nothing out of bounds is read, and the row's loss
  // and gradient come out NaN so that the total cannot look healthy
  const bool bad = labels && (lab >= C || lab < -1);
  if (keep) kept = kept && keep[row] != 0;
  if (!kept) {  // uniform per block
    if (dlogits) for (int c = tid; c < ldd; c += 256) st_any(dlogits, (int64_t)row * ldd + c, d_dtype, 0.f);
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
      st_any(dlogits, (int64_t)row * ldd + c, d_dtype, g);
    }
  }
}

__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b,
                                                  int64_t ldb, int rows, int d, float gscale,
                                                  const float* __restrict__ g_dev, float lscale,
                                                  float* __restrict__ loss_sum, float* __restrict__ partials,
                                                  float* __restrict__ da, int64_t ldda, float* __restrict__ db,
                                                  int64_t lddb) {
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
    if (gridDim.x == 1) *loss_sum += acc * lscale; else partials[blockIdx.x] = acc;
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
__global__ __launch_bounds__(256) void loss_reduce_bwd_kernel(const LossTerms t, const float* __restrict__ g_total) {
  const float go = g_total ? g_total[0] : 1.f;
  const int i = blockIdx.y;
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
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(g && n && w, "loss_reduce_bwd: null pointer");
  AFFT_CHECK(nterms >= 1 && nterms <= 8, "loss_reduce_bwd: 1..8 terms (got %d)", nterms);
  LossTerms t = {};
  t.nterms = nterms;
  int64_t nmax = 1;
  for (int i = 0; i < nterms; ++i) { t.g[i] = g[i]; t.n[i] = n[i]; t.w[i] = w[i]; nmax = n[i] > nmax ? n[i] : nmax; }
  const int gx = (int)((nmax + 255) / 256 > 64 ? 64 : (nmax + 255) / 256);
  hipLaunchKernelGGL(loss_reduce_bwd_kernel, dim3(gx, nterms), dim3(256), 0, stream, t, g_total);
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
                     gscale, row_g, dlogits, ldd, d_dtype, row_loss);
  AFFT_LAUNCH_CHECK();
  if (loss_sum) {
    hipLaunchKernelGGL(ordered_sum_kernel, dim3(1), dim3(256), 0, stream, row_loss, (int64_t)rows, 1.0f, loss_sum, 1);
    AFFT_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int afft_mse(const float* a, int64_t lda, const float* b, int64_t ldb, int32_t rows, int32_t d,
                        float gscale, const float* g_dev, float lscale, float* loss_sum, float* da, int64_t ldda,
                        float* db, int64_t lddb, void* workspace, int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(a && b, "mse: null pointer");
  if (rows == 0 || d == 0) return 0;
  const int64_t total = (int64_t)rows * d;
  int grid = (int)((total + 255) / 256);
  if (grid > AFFT_REDUCE_PARTIALS) grid = AFFT_REDUCE_PARTIALS;
  AFFT_CHECK(!loss_sum || grid == 1 || (workspace && workspace_bytes >= AFFT_GEMM_WS_HEADER + 4 * (int64_t)grid),
             "mse: the loss sum needs the stream's workspace (header + AFFT_REDUCE_PARTIALS floats)");
  float* scratch = workspace ? (float*)((char*)workspace + AFFT_GEMM_WS_HEADER) : nullptr;
  hipLaunchKernelGGL(mse_kernel, dim3(grid), dim3(256), 0, stream, a, lda, b, ldb, rows, d, gscale, g_dev, lscale, loss_sum,
                     scratch, da, ldda, db, lddb);
  AFFT_LAUNCH_CHECK();
  if (loss_sum && grid > 1) {
    hipLaunchKernelGGL(ordered_sum_kernel, dim3(1), dim3(256), 0, stream, scratch, (int64_t)grid, lscale, loss_sum, 1);
    AFFT_LAUNCH_CHECK();
  }
  return 0;
}
