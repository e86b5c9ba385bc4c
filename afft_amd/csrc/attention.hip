// Small-sequence attention, forward and backward, generic over storage type (fp32 parity mode / bf16).
// One workgroup per (sequence, head); L <= 32 tokens, so the whole score matrix lives in LDS and the
// mask is applied in-register while the scores are produced.  HBM-bound by construction (reads q,k,v once
// through L1/L2, writes out once): attention is < 0.2 % of the path's FLOPs.
//   softmax(q k^T * hd^-0.5 + mask) v : models/transformerblock.py:24-33,64-73 ; HF GPT-2 eager attention.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int LMAX = 32;

__device__ __forceinline__ bool masked(int mask, int i, int j) {
  return (mask == AFFT_MASK_DIAG && i == j) || (mask == AFFT_MASK_CAUSAL && j > i);
}

template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ q, int64_t ldq, const T* __restrict__ k,
                                                       int64_t ldk, const T* __restrict__ v, int64_t ldv, int L, int H,
                                                       int hd, float scale, int mask, unsigned dthresh, unsigned dkey,
                                                       float dinv, T* __restrict__ out, int64_t ldo,
                                                       float* __restrict__ probs) {
  __shared__ float sc[LMAX][LMAX + 1];
  const int seq = blockIdx.x / H, h = blockIdx.x % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = (int64_t)seq * L;
  const T* qh = q + row0 * ldq + (int64_t)h * hd;
  const T* kh = k + row0 * ldk + (int64_t)h * hd;
  const T* vh = v + row0 * ldv + (int64_t)h * hd;
  // scores: one wave per (i, j) pair, lanes stride the head dimension
  for (int idx = wave; idx < L * L; idx += 4) {
    const int i = idx / L, j = idx - i * L;
    float s = 0.f;
    if (!masked(mask, i, j)) {
      for (int c = lane; c < hd; c += 64) s += Elem<T>::ld(qh + i * ldq + c) * Elem<T>::ld(kh + j * ldk + c);
      s = wave_sum(s) * scale;
    } else {
      s = -INFINITY;
    }
    if (lane == 0) sc[i][j] = s;
  }
  __syncthreads();
  if (tid < L) {
    const int i = tid;
    float m = -INFINITY;
    for (int j = 0; j < L; ++j) m = fmaxf(m, sc[i][j]);
    float sum = 0.f;
    for (int j = 0; j < L; ++j) {
      const float e = sc[i][j] == -INFINITY ? 0.f : expf(sc[i][j] - m);
      sc[i][j] = e;
      sum += e;
    }
    const float inv = 1.0f / sum;
    float* pr = probs ? probs + (((int64_t)seq * H + h) * L + i) * L : nullptr;
    const unsigned base = (unsigned)((((int64_t)seq * H + h) * L + i) * L);
    for (int j = 0; j < L; ++j) {
      const float p = sc[i][j] * inv;
      if (pr) pr[j] = p;  // pre-dropout probabilities (backward regenerates the mask)
      sc[i][j] = (dthresh && !drop_keep(dkey, base + j, dthresh)) ? 0.f : p * dinv;
    }
  }
  __syncthreads();
  for (int c = tid; c < hd; c += 256) {
    float vc[LMAX];
#pragma unroll
    for (int j = 0; j < LMAX; ++j) vc[j] = j < L ? Elem<T>::ld(vh + j * ldv + c) : 0.f;
    for (int i = 0; i < L; ++i) {
      float o = 0.f;
#pragma unroll
      for (int j = 0; j < LMAX; ++j) o += (j < L ? sc[i][j] : 0.f) * vc[j];
      Elem<T>::st(out + (row0 + i) * ldo + (int64_t)h * hd + c, o);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const T* __restrict__ dout, int64_t lddo, const T* __restrict__ q,
                                                       int64_t ldq, const T* __restrict__ k, int64_t ldk,
                                                       const T* __restrict__ v, int64_t ldv,
                                                       const float* __restrict__ probs, int L, int H, int hd, float scale,
                                                       unsigned dthresh, unsigned dkey, float dinv,
                                                       T* __restrict__ dq, int64_t lddq, T* __restrict__ dk, int64_t lddk,
                                                       T* __restrict__ dv, int64_t lddv) {
  __shared__ float pp[LMAX][LMAX + 1];  // probabilities (pre-dropout), later the dropped-out P' used by dV
  __shared__ float ds[LMAX][LMAX + 1];  // dP, then dS*scale
  const int seq = blockIdx.x / H, h = blockIdx.x % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = (int64_t)seq * L;
  const T* doh = dout + row0 * lddo + (int64_t)h * hd;
  const T* qh = q + row0 * ldq + (int64_t)h * hd;
  const T* kh = k + row0 * ldk + (int64_t)h * hd;
  const T* vh = v + row0 * ldv + (int64_t)h * hd;
  const float* pr = probs + ((int64_t)seq * H + h) * L * L;
  for (int idx = tid; idx < L * L; idx += 256) pp[idx / L][idx % L] = pr[idx];
  // dP[i][j] = sum_c dO[i][c] v[j][c]
  for (int idx = wave; idx < L * L; idx += 4) {
    const int i = idx / L, j = idx - i * L;
    float s = 0.f;
    for (int c = lane; c < hd; c += 64) s += Elem<T>::ld(doh + i * lddo + c) * Elem<T>::ld(vh + j * ldv + c);
    s = wave_sum(s);
    if (lane == 0) ds[i][j] = s;
  }
  __syncthreads();
  if (tid < L) {
    const int i = tid;
    const unsigned base = (unsigned)((((int64_t)seq * H + h) * L + i) * L);
    float dot = 0.f;
    for (int j = 0; j < L; ++j) {
      // ds holds dP' (gradient of the dropped-out probabilities); dP = dP' * m/(1-p)
      const float m = (dthresh && !drop_keep(dkey, base + j, dthresh)) ? 0.f : dinv;
      ds[i][j] *= m;
      dot += pp[i][j] * ds[i][j];
    }
    for (int j = 0; j < L; ++j) {
      const float m = (dthresh && !drop_keep(dkey, base + j, dthresh)) ? 0.f : dinv;
      ds[i][j] = pp[i][j] * (ds[i][j] - dot) * scale;
      pp[i][j] *= m;  // P' for dV
    }
  }
  __syncthreads();
  for (int c = tid; c < hd; c += 256) {
    float a[LMAX];
    // dV[j][c] = sum_i P[i][j] dO[i][c]
#pragma unroll
    for (int i = 0; i < LMAX; ++i) a[i] = i < L ? Elem<T>::ld(doh + i * lddo + c) : 0.f;
    for (int j = 0; j < L; ++j) {
      float o = 0.f;
#pragma unroll
      for (int i = 0; i < LMAX; ++i) o += (i < L ? pp[i][j] : 0.f) * a[i];
      Elem<T>::st(dv + (row0 + j) * lddv + (int64_t)h * hd + c, o);
    }
    // dQ[i][c] = sum_j dS[i][j] k[j][c]
#pragma unroll
    for (int j = 0; j < LMAX; ++j) a[j] = j < L ? Elem<T>::ld(kh + j * ldk + c) : 0.f;
    for (int i = 0; i < L; ++i) {
      float o = 0.f;
#pragma unroll
      for (int j = 0; j < LMAX; ++j) o += (j < L ? ds[i][j] : 0.f) * a[j];
      Elem<T>::st(dq + (row0 + i) * lddq + (int64_t)h * hd + c, o);
    }
    // dK[j][c] = sum_i dS[i][j] q[i][c]
#pragma unroll
    for (int i = 0; i < LMAX; ++i) a[i] = i < L ? Elem<T>::ld(qh + i * ldq + c) : 0.f;
    for (int j = 0; j < L; ++j) {
      float o = 0.f;
#pragma unroll
      for (int i = 0; i < LMAX; ++i) o += (i < L ? ds[i][j] : 0.f) * a[i];
      Elem<T>::st(dk + (row0 + j) * lddk + (int64_t)h * hd + c, o);
    }
  }
}

}  // namespace

// bf16 MFMA path (attention_mfma.hip); returns -1 when the shape is not handled
int afft_attention_mfma(bool backward, const void* dout, int64_t lddo, const void* q, int64_t ldq, const void* k,
                        int64_t ldk, const void* v, int64_t ldv, float* probs, int nseq, int L, int H, int hd,
                        float scale, int mask, float drop_p, unsigned drop_key, void* out, int64_t ldo, void* dq,
                        int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, hipStream_t stream);
static bool use_mfma_attention() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("AFFT_ATTN_GENERIC"); v = (e && e[0] == '1') ? 0 : 1; }
  return v == 1;
}

extern "C" int afft_attention_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                  int32_t dtype, int32_t nseq, int32_t L, int32_t H, int32_t hd, float scale,
                                  int32_t mask, float drop_p, uint32_t drop_key, void* out, int64_t ldo, float* probs,
                                  void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(q && k && v && out, "attention_fwd: null pointer");
  AFFT_CHECK(L >= 1 && L <= LMAX, "attention_fwd: sequence length %d outside 1..%d", L, LMAX);
  AFFT_CHECK(mask >= AFFT_MASK_NONE && mask <= AFFT_MASK_CAUSAL, "attention_fwd: bad mask %d", mask);
  AFFT_CHECK(!(mask == AFFT_MASK_DIAG && L == 1), "attention_fwd: diagonal mask with L=1 masks every key");
  AFFT_CHECK(drop_p >= 0.f && drop_p < 1.f, "attention_fwd: dropout p outside [0,1)");
  if (nseq == 0) return 0;
  if (dtype == AFFT_BF16 && use_mfma_attention()) {
    const int rc = afft_attention_mfma(false, nullptr, 0, q, ldq, k, ldk, v, ldv, probs, nseq, L, H, hd, scale, mask,
                                       drop_p, drop_key, out, ldo, nullptr, 0, nullptr, 0, nullptr, 0, stream);
    if (rc >= 0) return rc;
  }
  afft_dropout_t dd = {drop_p, drop_key, 0.f, 0u, 1};
  const DropParams dp = make_drop(&dd);
  const dim3 grid(nseq * H), block(256);
  if (dtype == AFFT_F32)
    hipLaunchKernelGGL(attn_fwd_kernel<float>, grid, block, 0, stream, (const float*)q, ldq, (const float*)k, ldk,
                       (const float*)v, ldv, L, H, hd, scale, mask, dp.thresh, dp.key, dp.inv_keep, (float*)out, ldo, probs);
  else if (dtype == AFFT_BF16)
    hipLaunchKernelGGL(attn_fwd_kernel<bf16_t>, grid, block, 0, stream, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk,
                       (const bf16_t*)v, ldv, L, H, hd, scale, mask, dp.thresh, dp.key, dp.inv_keep, (bf16_t*)out, ldo, probs);
  else AFFT_CHECK(false, "attention_fwd: bad dtype %d", dtype);
  AFFT_LAUNCH_CHECK();
  return 0;
}

extern "C" int afft_attention_bwd(const void* dout, int64_t lddo, const void* q, int64_t ldq, const void* k, int64_t ldk,
                                  const void* v, int64_t ldv, int32_t dtype, const float* probs, int32_t nseq, int32_t L,
                                  int32_t H, int32_t hd, float scale, float drop_p, uint32_t drop_key, void* dq,
                                  int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(dout && q && k && v && probs && dq && dk && dv, "attention_bwd: null pointer");
  AFFT_CHECK(L >= 1 && L <= LMAX, "attention_bwd: sequence length %d outside 1..%d", L, LMAX);
  AFFT_CHECK(drop_p >= 0.f && drop_p < 1.f, "attention_bwd: dropout p outside [0,1)");
  if (nseq == 0) return 0;
  if (dtype == AFFT_BF16 && use_mfma_attention()) {
    const int rc = afft_attention_mfma(true, dout, lddo, q, ldq, k, ldk, v, ldv, const_cast<float*>(probs), nseq, L, H, hd,
                                       scale, 0, drop_p, drop_key, nullptr, 0, dq, lddq, dk, lddk, dv, lddv, stream);
    if (rc >= 0) return rc;
  }
  afft_dropout_t dd = {drop_p, drop_key, 0.f, 0u, 1};
  const DropParams dp = make_drop(&dd);
  const dim3 grid(nseq * H), block(256);
  if (dtype == AFFT_F32)
    hipLaunchKernelGGL(attn_bwd_kernel<float>, grid, block, 0, stream, (const float*)dout, lddo, (const float*)q, ldq,
                       (const float*)k, ldk, (const float*)v, ldv, probs, L, H, hd, scale, dp.thresh, dp.key, dp.inv_keep,
                       (float*)dq, lddq, (float*)dk, lddk, (float*)dv, lddv);
  else if (dtype == AFFT_BF16)
    hipLaunchKernelGGL(attn_bwd_kernel<bf16_t>, grid, block, 0, stream, (const bf16_t*)dout, lddo, (const bf16_t*)q, ldq,
                       (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, probs, L, H, hd, scale, dp.thresh, dp.key, dp.inv_keep,
                       (bf16_t*)dq, lddq, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv);
  else AFFT_CHECK(false, "attention_bwd: bad dtype %d", dtype);
  AFFT_LAUNCH_CHECK();
  return 0;
}
