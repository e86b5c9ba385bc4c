// Small-sequence attention, forward and backward, generic over storage type (fp32 parity mode / bf16).
// One workgroup per (sequence, head); L <= 128 tokens (the T-SA-Fuser attends over M*T tokens: models/fusion.py:121-215),
// so the whole score matrix lives in LDS (template LM = 32 / 64 / 128 rows) and the mask is applied in-register while
// the scores are produced.  HBM-bound by construction (reads q,k,v once
// through L1/L2, writes out once): attention is < 0.2 % of the path's FLOPs.
//   softmax(q k^T * hd^-0.5 + mask) v : models/transformerblock.py:24-33,64-73 ; HF GPT-2 eager attention.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int LMAX = 128;

__device__ __forceinline__ bool masked(int mask, int period, int i, int j) {
  return (mask == AFFT_MASK_DIAG && i == j) || (mask == AFFT_MASK_CAUSAL && j > i) ||
         (mask == AFFT_MASK_BLOCKCAUSAL && (j % period) > (i % period));
}

template <typename T, int LM>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ q, int64_t ldq, const T* __restrict__ k,
                                                       int64_t ldk, const T* __restrict__ v, int64_t ldv, int L, int H,
                                                       int hd, float scale, int mask, int period, unsigned dthresh,
                                                       unsigned dkey, float dinv, const unsigned* __restrict__ salt,
                                                       T* __restrict__ out, int64_t ldo, float* __restrict__ probs,
                                                       const float* __restrict__ addm) {      // addm: additive fp32 [L][L] table or null
  if (salt) dkey ^= *salt;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float (*sc)[LM + 1] = reinterpret_cast<float (*)[LM + 1]>(smem_raw);
  const int seq = blockIdx.x / H, h = blockIdx.x % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = (int64_t)seq * L;
  const T* qh = q + row0 * ldq + (int64_t)h * hd;
  const T* kh = k + row0 * ldk + (int64_t)h * hd;
  const T* vh = v + row0 * ldv + (int64_t)h * hd;
  // scores: a wave owns query rows i = wave, wave + 4, ...: the row is read once into registers (lanes stride the head
  // dimension) and dotted with 4 key rows at a time, so 4 x hd/64 independent loads are in flight per reduction
  // (one (i, j) pair per iteration was a chain of dependent L2 round trips: 1.8 ms per launch at L = 64, hd = 512)
  constexpr int NQ = 16;                               // hd <= 64 * NQ = 1024
  for (int i = wave; i < L; i += 4) {
    float qv[NQ];
#pragma unroll
    for (int t = 0; t < NQ; ++t) qv[t] = (lane + 64 * t) < hd ? Elem<T>::ld(qh + i * ldq + lane + 64 * t) : 0.f;
    for (int j0 = 0; j0 < L; j0 += 4) {
      float acc4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u;
        if (j < L && !masked(mask, period, i, j)) {
#pragma unroll
          for (int t = 0; t < NQ; ++t)
            if ((lane + 64 * t) < hd) acc4[u] += qv[t] * Elem<T>::ld(kh + j * ldk + lane + 64 * t);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u;
        if (j < L) {
          float sv = masked(mask, period, i, j) ? -INFINITY : wave_sum(acc4[u]) * scale;
          if (addm) sv += addm[i * L + j];      // models/transformerblock.py:27-28: attn = attn + attn_mask (any values, -inf included)
          if (lane == 0) sc[i][j] = sv;
        }
      }
    }
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
    float vc[LM];
#pragma unroll
    for (int j = 0; j < LM; ++j) vc[j] = j < L ? Elem<T>::ld(vh + j * ldv + c) : 0.f;
    for (int i = 0; i < L; ++i) {
      float o = 0.f;
#pragma unroll
      for (int j = 0; j < LM; ++j) o += (j < L ? sc[i][j] : 0.f) * vc[j];
      Elem<T>::st(out + (row0 + i) * ldo + (int64_t)h * hd + c, o);
    }
  }
}

template <typename T, int LM>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const T* __restrict__ dout, int64_t lddo, const T* __restrict__ q,
                                                       int64_t ldq, const T* __restrict__ k, int64_t ldk,
                                                       const T* __restrict__ v, int64_t ldv,
                                                       const float* __restrict__ probs, int L, int H, int hd, float scale,
                                                       unsigned dthresh, unsigned dkey, float dinv,
                                                       const unsigned* __restrict__ salt,
                                                       T* __restrict__ dq, int64_t lddq, T* __restrict__ dk, int64_t lddk,
                                                       T* __restrict__ dv, int64_t lddv) {
  if (salt) dkey ^= *salt;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float (*pp)[LM + 1] = reinterpret_cast<float (*)[LM + 1]>(smem_raw);                  // probabilities (pre-dropout), later the dropped-out P' used by dV
  float (*ds)[LM + 1] = reinterpret_cast<float (*)[LM + 1]>(smem_raw) + LM;             // dP, then dS*scale
  const int seq = blockIdx.x / H, h = blockIdx.x % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = (int64_t)seq * L;
  const T* doh = dout + row0 * lddo + (int64_t)h * hd;
  const T* qh = q + row0 * ldq + (int64_t)h * hd;
  const T* kh = k + row0 * ldk + (int64_t)h * hd;
  const T* vh = v + row0 * ldv + (int64_t)h * hd;
  const float* pr = probs + ((int64_t)seq * H + h) * L * L;
  for (int idx = tid; idx < L * L; idx += 256) pp[idx / L][idx % L] = pr[idx];
  // dP[i][j] = sum_c dO[i][c] v[j][c]: same row-in-registers, 4-keys-at-a-time scheme as the forward scores
  constexpr int NQ = 16;
  for (int i = wave; i < L; i += 4) {
    float dv_[NQ];
#pragma unroll
    for (int t = 0; t < NQ; ++t) dv_[t] = (lane + 64 * t) < hd ? Elem<T>::ld(doh + i * lddo + lane + 64 * t) : 0.f;
    for (int j0 = 0; j0 < L; j0 += 4) {
      float acc4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u;
        if (j < L) {
#pragma unroll
          for (int t = 0; t < NQ; ++t)
            if ((lane + 64 * t) < hd) acc4[u] += dv_[t] * Elem<T>::ld(vh + j * ldv + lane + 64 * t);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u;
        if (j < L) {
          const float sv = wave_sum(acc4[u]);
          if (lane == 0) ds[i][j] = sv;
        }
      }
    }
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
    float a[LM];
    // dV[j][c] = sum_i P[i][j] dO[i][c]
#pragma unroll
    for (int i = 0; i < LM; ++i) a[i] = i < L ? Elem<T>::ld(doh + i * lddo + c) : 0.f;
    for (int j = 0; j < L; ++j) {
      float o = 0.f;
#pragma unroll
      for (int i = 0; i < LM; ++i) o += (i < L ? pp[i][j] : 0.f) * a[i];
      Elem<T>::st(dv + (row0 + j) * lddv + (int64_t)h * hd + c, o);
    }
    // dQ[i][c] = sum_j dS[i][j] k[j][c]
#pragma unroll
    for (int j = 0; j < LM; ++j) a[j] = j < L ? Elem<T>::ld(kh + j * ldk + c) : 0.f;
    for (int i = 0; i < L; ++i) {
      float o = 0.f;
#pragma unroll
      for (int j = 0; j < LM; ++j) o += (j < L ? ds[i][j] : 0.f) * a[j];
      Elem<T>::st(dq + (row0 + i) * lddq + (int64_t)h * hd + c, o);
    }
    // dK[j][c] = sum_i dS[i][j] q[i][c]
#pragma unroll
    for (int i = 0; i < LM; ++i) a[i] = i < L ? Elem<T>::ld(qh + i * ldq + c) : 0.f;
    for (int j = 0; j < L; ++j) {
      float o = 0.f;
#pragma unroll
      for (int i = 0; i < LM; ++i) o += (i < L ? ds[i][j] : 0.f) * a[i];
      Elem<T>::st(dk + (row0 + j) * lddk + (int64_t)h * hd + c, o);
    }
  }
}



template <typename T, int LM>
int launch_fwd(dim3 grid, hipStream_t stream, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
               int L, int H, int hd, float scale, int mask, int period, const DropParams& dp, void* out, int64_t ldo,
               float* probs, const float* addm = nullptr) {
  constexpr size_t lds = sizeof(float) * LM * (LM + 1);
  auto kern = attn_fwd_kernel<T, LM>;
  static std::atomic<uint64_t> attr_done{0};
  if (lds > 48 * 1024)
    if (int rc = afft_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, &attr_done)) return rc;
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, (const T*)q, ldq, (const T*)k, ldk, (const T*)v, ldv, L, H, hd, scale,
                     mask, period, dp.thresh, dp.key, dp.inv_keep, dp.salt, (T*)out, ldo, probs, addm);
  return 0;
}
template <typename T, int LM>
int launch_bwd(dim3 grid, hipStream_t stream, const void* dout, int64_t lddo, const void* q, int64_t ldq, const void* k,
               int64_t ldk, const void* v, int64_t ldv, const float* probs, int L, int H, int hd, float scale,
               const DropParams& dp, void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv) {
  constexpr size_t lds = sizeof(float) * 2 * LM * (LM + 1);
  auto kern = attn_bwd_kernel<T, LM>;
  static std::atomic<uint64_t> attr_done{0};
  if (lds > 48 * 1024)
    if (int rc = afft_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, &attr_done)) return rc;
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, (const T*)dout, lddo, (const T*)q, ldq, (const T*)k, ldk, (const T*)v, ldv,
                     probs, L, H, hd, scale, dp.thresh, dp.key, dp.inv_keep, dp.salt, (T*)dq, lddq, (T*)dk, lddk, (T*)dv, lddv);
  return 0;
}

}  // namespace

// bf16 MFMA path (attention_mfma.hip); returns -1 when the shape is not handled
int afft_attention_mfma(bool backward, const void* dout, int64_t lddo, const void* q, int64_t ldq, const void* k,
                        int64_t ldk, const void* v, int64_t ldv, float* probs, int nseq, int L, int H, int hd,
                        float scale, int mask, float drop_p, unsigned drop_key, void* out, int64_t ldo, void* dq,
                        int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, hipStream_t stream,
                        int planes = 0, int64_t in_lo = 0, int64_t out_lo = 0, void* out_b = nullptr, int64_t ldob = 0, void* out_lo8 = nullptr);
static bool use_mfma_attention() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("AFFT_ATTN_GENERIC"); v = (e && e[0] == '1') ? 0 : 1; }
  return v == 1;
}

extern "C" int afft_attention_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                  int32_t dtype, int32_t nseq, int32_t L, int32_t H, int32_t hd, float scale,
                                  int32_t mask, int32_t mask_period, float drop_p, uint32_t drop_key, void* out,
                                  int64_t ldo, float* probs, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(q && k && v && out, "attention_fwd: null pointer");
  AFFT_CHECK(L >= 1 && L <= LMAX, "attention_fwd: sequence length %d outside 1..%d", L, LMAX);
  AFFT_CHECK(mask >= AFFT_MASK_NONE && mask <= AFFT_MASK_BLOCKCAUSAL, "attention_fwd: bad mask %d", mask);
  AFFT_CHECK(mask != AFFT_MASK_BLOCKCAUSAL || (mask_period >= 1 && L % mask_period == 0),
             "attention_fwd: block-causal mask needs a period that divides L (L=%d, period=%d)", L, mask_period);
  AFFT_CHECK(!(mask == AFFT_MASK_DIAG && L == 1), "attention_fwd: diagonal mask with L=1 masks every key");
  AFFT_CHECK(drop_p >= 0.f && drop_p < 1.f, "attention_fwd: dropout p outside [0,1)");
  AFFT_CHECK(hd >= 1 && hd <= 1024, "attention_fwd: head dimension %d outside 1..1024", hd);
  if (nseq == 0) return 0;
  const int64_t es_ = dtype == AFFT_F32 ? 4 : 2, rw_ = (int64_t)nseq * L * H * hd, pb_ = probs ? (int64_t)nseq * H * L * L * 4 : 0;
  AfftKernelScope ktrace(AFFT_K_ATTN_FWD, nseq * L, H * hd, 4 * es_ * rw_ + pb_, 4 * (int64_t)nseq * H * L * L * hd, stream);
  if (dtype == AFFT_BF16 && use_mfma_attention()) {
    const int rc = afft_attention_mfma(false, nullptr, 0, q, ldq, k, ldk, v, ldv, probs, nseq, L, H, hd, scale,
                                       mask | (mask == AFFT_MASK_BLOCKCAUSAL ? mask_period << 8 : 0),
                                       drop_p, drop_key, out, ldo, nullptr, 0, nullptr, 0, nullptr, 0, stream);
    if (rc >= 0) return rc;
  }
  afft_dropout_t dd = {drop_p, drop_key, 0.f, 0u, 1};
  const DropParams dp = make_drop(&dd);
  const dim3 grid(nseq * H);
  AFFT_CHECK(dtype == AFFT_F32 || dtype == AFFT_BF16, "attention_fwd: bad dtype %d", dtype);
  int rc;
#define FWD(T, LM) launch_fwd<T, LM>(grid, stream, q, ldq, k, ldk, v, ldv, L, H, hd, scale, mask, mask_period, dp, out, ldo, probs)
  if (dtype == AFFT_F32) rc = L <= 32 ? FWD(float, 32) : L <= 64 ? FWD(float, 64) : FWD(float, 128);
  else rc = L <= 32 ? FWD(bf16_t, 32) : L <= 64 ? FWD(bf16_t, 64) : FWD(bf16_t, 128);
#undef FWD
  if (rc) return rc;
  AFFT_LAUNCH_CHECK();
  return 0;
}

extern "C" int afft_attention_fwd_table(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                        int32_t dtype, int32_t nseq, int32_t L, int32_t H, int32_t hd, float scale,
                                        const float* mask_table, float drop_p, uint32_t drop_key, void* out, int64_t ldo,
                                        float* probs, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(q && k && v && out && mask_table, "attention_fwd_table: null pointer");
  AFFT_CHECK(L >= 1 && L <= LMAX, "attention_fwd_table: sequence length %d outside 1..%d", L, LMAX);
  AFFT_CHECK(drop_p >= 0.f && drop_p < 1.f, "attention_fwd_table: dropout p outside [0,1)");
  AFFT_CHECK(hd >= 1 && hd <= 1024, "attention_fwd_table: head dimension %d outside 1..1024", hd);
  AFFT_CHECK(dtype == AFFT_F32 || dtype == AFFT_BF16, "attention_fwd_table: bad dtype %d", dtype);
  if (nseq == 0) return 0;
  const int64_t es_ = dtype == AFFT_F32 ? 4 : 2, rw_ = (int64_t)nseq * L * H * hd, pb_ = probs ? (int64_t)nseq * H * L * L * 4 : 0;
  AfftKernelScope ktrace(AFFT_K_ATTN_FWD, nseq * L, H * hd, 4 * es_ * rw_ + pb_, 4 * (int64_t)nseq * H * L * L * hd, stream);
  afft_dropout_t dd = {drop_p, drop_key, 0.f, 0u, 1};
  const DropParams dp = make_drop(&dd);
  const dim3 grid(nseq * H);
  int rc;
#define FWD(T, LM) launch_fwd<T, LM>(grid, stream, q, ldq, k, ldk, v, ldv, L, H, hd, scale, AFFT_MASK_NONE, 0, dp, out, ldo, probs, mask_table)
  if (dtype == AFFT_F32) rc = L <= 32 ? FWD(float, 32) : L <= 64 ? FWD(float, 64) : FWD(float, 128);
  else rc = L <= 32 ? FWD(bf16_t, 32) : L <= 64 ? FWD(bf16_t, 64) : FWD(bf16_t, 128);
#undef FWD
  if (rc) return rc;
  AFFT_LAUNCH_CHECK();
  return 0;
}

extern "C" int afft_attention_fwd_split(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, int64_t in_lo,
                                        int32_t nseq, int32_t L, int32_t H, int32_t hd, float scale, int32_t mask, int32_t mask_period,
                                        float drop_p, uint32_t drop_key, void* out_hi, int64_t ldo, int64_t out_lo, void* out_bf16,
                                        int64_t ldob, float* probs, void* out_lo8, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(q && k && v && out_hi, "attention_fwd_split: null pointer");
  AFFT_CHECK(L >= 1 && L <= 64, "attention_fwd_split: sequence length %d outside 1..64 (MFMA path only)", L);
  AFFT_CHECK(mask >= AFFT_MASK_NONE && mask <= AFFT_MASK_BLOCKCAUSAL, "attention_fwd_split: bad mask %d", mask);
  AFFT_CHECK(mask != AFFT_MASK_BLOCKCAUSAL || (mask_period >= 1 && L % mask_period == 0),
             "attention_fwd_split: block-causal mask needs a period that divides L (L=%d, period=%d)", L, mask_period);
  AFFT_CHECK(!(mask == AFFT_MASK_DIAG && L == 1), "attention_fwd_split: diagonal mask with L=1 masks every key");
  AFFT_CHECK(drop_p >= 0.f && drop_p < 1.f, "attention_fwd_split: dropout p outside [0,1)");
  AFFT_CHECK(in_lo >= 0, "attention_fwd_split: in_lo is the distance to the inputs' lo planes (0: one fp16 plane each)");
  AFFT_CHECK(!out_lo8 || (out_lo == 0 && (((uintptr_t)out_lo8) & 3) == 0), "attention_fwd_split: out_lo8 excludes out_lo and must be 4-byte aligned");
  if (nseq == 0) return 0;
  const int64_t rw_ = (int64_t)nseq * L * H * hd, pb_ = probs ? (int64_t)nseq * H * L * L * 4 : 0;
  AfftKernelScope ktrace(AFFT_K_ATTN_FWD, nseq * L, H * hd, (3 * (in_lo ? 4 : 2) + (out_lo ? 4 : out_lo8 ? 3 : 2) + (out_bf16 ? 2 : 0)) * rw_ + pb_,
                         3 * 4 * (int64_t)nseq * H * L * L * hd, stream);
  const int rc = afft_attention_mfma(false, nullptr, 0, q, ldq, k, ldk, v, ldv, probs, nseq, L, H, hd, scale,
                                     mask | (mask == AFFT_MASK_BLOCKCAUSAL ? mask_period << 8 : 0), drop_p, drop_key, out_hi, ldo,
                                     nullptr, 0, nullptr, 0, nullptr, 0, stream, 1, in_lo, out_lo, out_bf16, ldob, out_lo8);
  AFFT_CHECK(rc >= 0, "attention_fwd_split: shape not handled by the MFMA path (hd %d must be a multiple of 64 and <= 1024, 16-byte aligned rows)", hd);
  return rc;
}

extern "C" int afft_attention_bwd(const void* dout, int64_t lddo, const void* q, int64_t ldq, const void* k, int64_t ldk,
                                  const void* v, int64_t ldv, int32_t dtype, const float* probs, int32_t nseq, int32_t L,
                                  int32_t H, int32_t hd, float scale, float drop_p, uint32_t drop_key, void* dq,
                                  int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(dout && q && k && v && probs && dq && dk && dv, "attention_bwd: null pointer");
  AFFT_CHECK(L >= 1 && L <= LMAX, "attention_bwd: sequence length %d outside 1..%d", L, LMAX);
  AFFT_CHECK(drop_p >= 0.f && drop_p < 1.f, "attention_bwd: dropout p outside [0,1)");
  AFFT_CHECK(hd >= 1 && hd <= 1024, "attention_bwd: head dimension %d outside 1..1024", hd);
  if (nseq == 0) return 0;
  const int64_t es_ = dtype == AFFT_F32 ? 4 : 2, rw_ = (int64_t)nseq * L * H * hd, pb_ = (int64_t)nseq * H * L * L * 4;
  AfftKernelScope ktrace(AFFT_K_ATTN_BWD, nseq * L, H * hd, 7 * es_ * rw_ + pb_, 8 * (int64_t)nseq * H * L * L * hd, stream);
  if (dtype == AFFT_BF16 && use_mfma_attention()) {
    const int rc = afft_attention_mfma(true, dout, lddo, q, ldq, k, ldk, v, ldv, const_cast<float*>(probs), nseq, L, H, hd,
                                       scale, 0, drop_p, drop_key, nullptr, 0, dq, lddq, dk, lddk, dv, lddv, stream);
    if (rc >= 0) return rc;
  }
  afft_dropout_t dd = {drop_p, drop_key, 0.f, 0u, 1};
  const DropParams dp = make_drop(&dd);
  const dim3 grid(nseq * H);
  AFFT_CHECK(dtype == AFFT_F32 || dtype == AFFT_BF16, "attention_bwd: bad dtype %d", dtype);
  int rc;
#define BWD(T, LM) launch_bwd<T, LM>(grid, stream, dout, lddo, q, ldq, k, ldk, v, ldv, probs, L, H, hd, scale, dp, dq, lddq, dk, lddk, dv, lddv)
  if (dtype == AFFT_F32) rc = L <= 32 ? BWD(float, 32) : L <= 64 ? BWD(float, 64) : BWD(float, 128);
  else rc = L <= 32 ? BWD(bf16_t, 32) : L <= 64 ? BWD(bf16_t, 64) : BWD(bf16_t, 128);
#undef BWD
  if (rc) return rc;
  AFFT_LAUNCH_CHECK();
  return 0;
}
