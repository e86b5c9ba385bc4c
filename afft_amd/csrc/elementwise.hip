// HBM-bound data movement: dtype casts (+ transposed copy through an LDS tile), SA-Fuser token
// assembly, bias-gradient column sums, periodic (position / token) row tables, Nesterov SGD.
#include <stdarg.h>

#include <cstdlib>
#include <mutex>
#include <vector>

#include "common.h"

// ---- error plumbing shared by every translation unit
static thread_local char g_err[512] = "";
void afft_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* afft_last_error(void) { return g_err; }

#include <mutex>
int afft_ensure_dynamic_lds(const void* kern, size_t bytes, std::atomic<uint64_t>* done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  const uint64_t bit = dev < 64 ? (1ull << dev) : 0;     // devices >= 64: set the attribute on every launch
  if (bit && (done->load(std::memory_order_acquire) & bit)) return 0;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
    afft_set_error("afft: cannot reserve %zu bytes of dynamic LDS on device %d", bytes, dev);
    (void)hipGetLastError();
    return 2;
  }
  done->fetch_or(bit, std::memory_order_release);
  return 0;
}
extern "C" int afft_version(void) { return 1; }

// ---- kernel trace (include/afft_hip.h: afft_kernel_trace_begin / _end)
std::atomic<int> g_afft_ktrace_open{0};
namespace {
struct KTraceRec { afft_kernel_trace_rec_t r; hipEvent_t a, b; };
std::mutex g_ktrace_mu;
std::vector<KTraceRec>* g_ktrace = nullptr;
size_t g_ktrace_cap = 0;
}  // namespace
void afft_ktrace_push(int kind, int rows, int width, int64_t bytes, int64_t flops, hipStream_t, hipEvent_t a, hipEvent_t b) {
  std::lock_guard<std::mutex> lk(g_ktrace_mu);
  if (!g_ktrace || g_ktrace->size() >= g_ktrace_cap) { (void)hipEventDestroy(a); (void)hipEventDestroy(b); return; }
  g_ktrace->push_back(KTraceRec{afft_kernel_trace_rec_t{kind, rows, width, 0, bytes, flops, 0.f}, a, b});
}
extern "C" int afft_kernel_trace_begin(int32_t capacity) {
  std::lock_guard<std::mutex> lk(g_ktrace_mu);
  AFFT_CHECK(!g_ktrace, "afft_kernel_trace_begin: a trace is already open");
  AFFT_CHECK(capacity > 0, "afft_kernel_trace_begin: capacity must be positive");
  g_ktrace = new std::vector<KTraceRec>();
  g_ktrace->reserve(capacity);
  g_ktrace_cap = (size_t)capacity;
  g_afft_ktrace_open.store(1);
  return 0;
}
extern "C" int afft_kernel_trace_end(afft_kernel_trace_rec_t* out, int32_t capacity) {
  std::vector<KTraceRec>* tr;
  {
    std::lock_guard<std::mutex> lk(g_ktrace_mu);
    g_afft_ktrace_open.store(0);
    tr = g_ktrace;
    g_ktrace = nullptr;
  }
  if (!tr) { afft_set_error("afft_kernel_trace_end: no trace is open"); return -1; }
  int n = 0;
  for (KTraceRec& t : *tr) {
    float ms = 0.f;
    if (hipEventSynchronize(t.b) == hipSuccess && hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess && out && n < capacity) {
      t.r.ms = ms;
      out[n++] = t.r;
    }
    (void)hipEventDestroy(t.a);
    (void)hipEventDestroy(t.b);
  }
  (void)hipGetLastError();
  delete tr;
  return n;
}

namespace {

// 64x64 tile per workgroup; coalesced read along cols, coalesced writes along cols (dst) and rows (dst_t)
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, int64_t lds_, int rows, int cols,
                                                   void* __restrict__ dst, int64_t ldd, int dst_dtype,
                                                   void* __restrict__ dst_t, int64_t ldt, int pad_cols,
                                                   const DropParams drop_) {
  const DropParams drop = with_salt(drop_);
  __shared__ float tile[64][65];
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int rr = ty; rr < 64; rr += 4) {
    const int r = r0 + rr, c = c0 + tx;
    float v = 0.f;
    if (r < rows && c < cols) {
      v = src[(int64_t)r * lds_ + c];
      v *= drop_elem_scale(drop, (unsigned)r * (unsigned)cols + (unsigned)c) * drop_row_scale(drop, r);
    }
    tile[rr][tx] = v;
    if (dst && r < rows && c < pad_cols) st_any(dst, (int64_t)r * ldd + c, dst_dtype, v);
  }
  if (dst_t) {
    __syncthreads();
    for (int cc = ty; cc < 64; cc += 4) {
      const int c = c0 + cc, r = r0 + tx;
      if (c < cols && r < rows) st_any(dst_t, (int64_t)c * ldt + r, dst_dtype, tile[tx][cc]);
    }
  }
}

// dst-only copy/cast, 4 elements per thread (cols % 4 == 0, aligned rows), optional dropout replay
__global__ __launch_bounds__(256) void cast_rows4_kernel(const float* __restrict__ src, int64_t lds_, int rows, int cols,
                                                         void* __restrict__ dst, int64_t ldd, int dst_dtype,
                                                         const DropParams drop_) {
  const DropParams drop = with_salt(drop_);
  const int nq = cols >> 2;
  const int64_t total = (int64_t)rows * nq;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / nq), c = (int)(i - (int64_t)r * nq) * 4;
    const float4 t = *(const float4*)(src + (int64_t)r * lds_ + c);
    float v[4] = {t.x, t.y, t.z, t.w};
    if (drop.thresh || drop.path_thresh) {
      const float rs = drop_row_scale(drop, r);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] *= rs * drop_elem_scale(drop, (unsigned)r * (unsigned)cols + (unsigned)(c + k));
    }
    store4(dst, (int64_t)r * ldd + c, dst_dtype, v);
  }
}

// two-plane bf16 split x = hi + lo of an fp32 matrix (bf16x3 GEMM operands), zero-filled out to [rows_pad, ldd];
// 8 columns per thread: two 16-byte loads when the source allows, one 16-byte store per plane
template <bool F16>   // F16: fp16 planes (hi = fp16(x), lo = fp16(x - hi)) for the fp16 two-pass GEMM mode
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ src, int64_t lds_, int rows, int cols,
                                                         bf16_t* __restrict__ hi, int64_t ldd, int rows_pad,
                                                         int64_t plane_stride, int src_vec) {
  const int nq = (int)(ldd >> 3);
  const int64_t total = (int64_t)rows_pad * nq;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / nq), c = (int)(i - (int64_t)r * nq) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
      const float* p = src + (int64_t)r * lds_ + c;
      if (src_vec && c + 7 < cols) load8(p, 0, AFFT_F32, v);
      else
#pragma unroll
        for (int k = 0; k < 8; ++k) if (c + k < cols) v[k] = p[k];
    }
    if constexpr (F16) {
      typedef __attribute__((ext_vector_type(8))) _Float16 h8;
      h8 hv, lv;
#pragma unroll
      for (int k = 0; k < 8; ++k) { hv[k] = (_Float16)v[k]; lv[k] = (_Float16)(v[k] - (float)hv[k]); }
      *(h8*)(hi + (int64_t)r * ldd + c) = hv;
      *(h8*)(hi + plane_stride + (int64_t)r * ldd + c) = lv;
    } else {
      float h[8], l[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { h[k] = bf2f(f2bf(v[k])); l[k] = v[k] - h[k]; }
      store8(hi, (int64_t)r * ldd + c, AFFT_BF16, h);
      store8(hi + plane_stride, (int64_t)r * ldd + c, AFFT_BF16, l);
    }
  }
}

struct Modal8 { const float* p[8]; int64_t ld[8]; };

__global__ __launch_bounds__(256) void assemble_kernel(Modal8 mods, int S, const float* __restrict__ token,
                                                       int64_t tok_stride_t, const float* __restrict__ mod_embed,
                                                       int T, int d, float* __restrict__ X) {
  const int row = blockIdx.x;  // row = bt*S + s
  const int bt = row / S, s = row - bt * S;
  const float* src = s == 0 ? token + (int64_t)(bt % T) * tok_stride_t : mods.p[s - 1] + (int64_t)bt * mods.ld[s - 1];
  const float* emb = mod_embed ? mod_embed + (int64_t)s * d : nullptr;
  float* dst = X + (int64_t)row * d;
  for (int c = threadIdx.x * 4; c < d; c += 1024) {
    float4 v = *(const float4*)(src + c);
    if (emb) {
      const float4 e = *(const float4*)(emb + c);
      v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
    }
    *(float4*)(dst + c) = v;
  }
}

// Column sums without float atomics, so that two runs of a training step give bit-identical bias gradients.
// Form 1 (caller gave a workspace): the rows are cut into blocks; workgroup (strip x, row block y) of the first kernel sums its
// rows of a 256-column strip and parks the partial sums in the workspace, a second kernel adds the partials up in block order
// (a kernel boundary between writer and reader: no cross-XCD visibility question) -- as many workgroups as the atomic form had,
// and its memset traded for the second launch.
template <bool V4>
__global__ __launch_bounds__(256) void colsum_blocks_kernel(const void* __restrict__ src, int64_t lds_, int dtype, int rows,
                                                            int cols, int rows_per_block, int accumulate,
                                                            float* __restrict__ out, float* __restrict__ part) {
  __shared__ float sh[4][256];
  const int tid = threadIdx.x;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  const int cc = blockIdx.x * 256 + tid;
  float t = 0.f;
  if (V4) {      // 64 column-quads x 4 row-lanes
    const int qd = tid & 63, rl = tid >> 6;
    const int c = (blockIdx.x * 64 + qd) * 4;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < cols)
      for (int r = r0 + rl; r < r1; r += 4) {
        float v[4];
        load4(src, (int64_t)r * lds_ + c, dtype, v);
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
      }
#pragma unroll
    for (int k = 0; k < 4; ++k) sh[rl][qd * 4 + k] = s[k];
    __syncthreads();
    t = (sh[0][tid] + sh[1][tid]) + (sh[2][tid] + sh[3][tid]);
  } else if (cc < cols) {
    for (int r = r0; r < r1; ++r) t += ld_any(src, (int64_t)r * lds_ + cc, dtype);
  }
  if (gridDim.y > 1) part[((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * 256 + tid] = t;
  else if (cc < cols) out[cc] = accumulate ? out[cc] + t : t;
}

__global__ __launch_bounds__(256) void colsum_combine_kernel(const float* __restrict__ part, int nby, int cols, int accumulate,
                                                             float* __restrict__ out) {
  const int cc = blockIdx.x * 256 + threadIdx.x;
  if (cc >= cols) return;
  const float* p = part + (int64_t)blockIdx.x * nby * 256 + threadIdx.x;
  float sum = 0.f;
  int y = 0;
  for (; y + 7 < nby; y += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(y + u) * 256];
#pragma unroll
    for (int u = 0; u < 8; ++u) sum += v[u];
  }
  for (; y < nby; ++y) sum += p[y * 256];
  out[cc] = accumulate ? out[cc] + sum : sum;
}

// Form 2 (no workspace): every output column is summed by ONE workgroup over all rows -- fewer workgroups, ~1.5x the time on the
// step's [5120, 8192] sums.  Generic form: one thread per column walks all rows.
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ src, int64_t lds_, int dtype, int rows,
                                                     int cols, int accumulate, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int r = 0;
  for (; r + 3 < rows; r += 4) {
    s0 += ld_any(src, (int64_t)r * lds_ + c, dtype);
    s1 += ld_any(src, (int64_t)(r + 1) * lds_ + c, dtype);
    s2 += ld_any(src, (int64_t)(r + 2) * lds_ + c, dtype);
    s3 += ld_any(src, (int64_t)(r + 3) * lds_ + c, dtype);
  }
  for (; r < rows; ++r) s0 += ld_any(src, (int64_t)r * lds_ + c, dtype);
  const float t = (s0 + s1) + (s2 + s3);
  out[c] = accumulate ? out[c] + t : t;
}

// vectorized form (cols % 4 == 0, 4-wide aligned rows): a workgroup owns QUADS column-quads (128-byte row segments: 16 quads
// of bf16, 8 of fp32) and ALL rows, RL = 256 / QUADS row lanes x 4 rows in flight each; the lane partials are added in lane order.
template <int QUADS>
__global__ __launch_bounds__(256) void colsum4_kernel(const void* __restrict__ src, int64_t lds_, int dtype, int rows,
                                                      int cols, int accumulate, float* __restrict__ out) {
  constexpr int RL = 256 / QUADS, CW = QUADS * 4;
  __shared__ float sh[RL][CW];
  const int qd = threadIdx.x % QUADS, rl = threadIdx.x / QUADS;
  const int c = (blockIdx.x * QUADS + qd) * 4;
  float s[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int k = 0; k < 4; ++k) s[u][k] = 0.f;
  if (c < cols) {
    int r = rl;
    for (; r + 3 * RL < rows; r += 4 * RL) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v[4];
        load4(src, (int64_t)(r + RL * u) * lds_ + c, dtype, v);
#pragma unroll
        for (int k = 0; k < 4; ++k) s[u][k] += v[k];
      }
    }
    for (; r < rows; r += RL) {
      float v[4];
      load4(src, (int64_t)r * lds_ + c, dtype, v);
#pragma unroll
      for (int k = 0; k < 4; ++k) s[0][k] += v[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) sh[rl][qd * 4 + k] = (s[0][k] + s[1][k]) + (s[2][k] + s[3][k]);
  __syncthreads();
  const int cc = blockIdx.x * CW + threadIdx.x;
  if (threadIdx.x < CW && cc < cols) {
    float t = 0.f;
#pragma unroll
    for (int l = 0; l < RL; ++l) t += sh[l][threadIdx.x];
    out[cc] = accumulate ? out[cc] + t : t;
  }
}

// out[clip, t, :] = sum over the sources that cover frame t of src[clip, t + off, :], zeros where none does: concatenation along the
// frame axis, its backward (slices that overlap are added), "token 0 of every frame" and its zero-filled backward -- one launch each,
// every output element written exactly once (no fill in front, no add behind).
struct FrameSrcs { const float* p[4]; int64_t clip_stride[4]; int64_t frame_stride[4]; int lo[4], hi[4], off[4]; int n; };
__global__ __launch_bounds__(256) void gather_frames_kernel(const FrameSrcs s, float* __restrict__ out, int64_t out_clip_stride,
                                                           int64_t out_frame_stride, int frames, int C4) {
  const int clip = blockIdx.y;
  const int total = frames * C4;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int t = i / C4, c = (i - t * C4) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < s.n && t >= s.lo[k] && t < s.hi[k]) {
        const f32x4 x = *(const f32x4*)(s.p[k] + (int64_t)clip * s.clip_stride[k] + (int64_t)(t + s.off[k]) * s.frame_stride[k] + c);
        v[0] += x[0]; v[1] += x[1]; v[2] += x[2]; v[3] += x[3];
      }
    *(f32x4*)(out + (int64_t)clip * out_clip_stride + (int64_t)t * out_frame_stride + c) = v;
  }
}


__global__ __launch_bounds__(256) void add_rows_periodic_kernel(const float* __restrict__ x, int64_t ldx,
                                                                const float* __restrict__ table, int64_t ldt,
                                                                int period, int d, float* __restrict__ y, int64_t ldy) {
  const int row = blockIdx.x;
  const float* xr = x + (int64_t)row * ldx;
  const float* tr = table + (int64_t)(row % period) * ldt;
  float* yr = y + (int64_t)row * ldy;
  for (int c = threadIdx.x * 4; c < d; c += 1024) {
    const float4 a = *(const float4*)(xr + c);
    const float4 b = *(const float4*)(tr + c);
    *(float4*)(yr + c) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

__global__ __launch_bounds__(256) void reduce_rows_periodic_kernel(const float* __restrict__ src, int64_t lds_, int rows,
                                                                   int period, int d, float* __restrict__ out,
                                                                   int64_t ldo) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int p = blockIdx.y;
  if (c >= d) return;
  float s = 0.f;
  for (int r = p; r < rows; r += period) s += src[(int64_t)r * lds_ + c];
  out[(int64_t)p * ldo + c] += s;
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const void* __restrict__ g_, int g_dtype,
                                                  float* __restrict__ buf, bf16_t* __restrict__ p16, bf16_t* __restrict__ p16h,
                                                  unsigned char* __restrict__ p8, int64_t n, float lr,
                                                  float mom, float wd, float gscale, const float* __restrict__ gscale_dev,
                                                  int first, const float* __restrict__ ok) {
  if (ok && *ok == 0.f) return;            // non-finite loss: the step is a no-op (afft_sgd_fused_t.ok)
  if (gscale_dev) gscale *= *gscale_dev;   // clip coefficient computed on the device (afft_clip_coef)
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    if (i + 3 < n) {
      float4 pv = *(float4*)(p + i);
      float g4[4];
      load4(g_, i, g_dtype, g4);
      const float4 gv = make_float4(g4[0], g4[1], g4[2], g4[3]);
      float4 bv = (first & AFFT_SGD_FIRST_STEP) ? make_float4(0.f, 0.f, 0.f, 0.f) : *(float4*)(buf + i);
      float gg[4] = {gv.x, gv.y, gv.z, gv.w};
      float bb[4] = {bv.x, bv.y, bv.z, bv.w};
      float pp[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) sgd_update(pp[r], bb[r], gg[r], lr, mom, wd, gscale, first);
      *(float4*)(buf + i) = make_float4(bb[0], bb[1], bb[2], bb[3]);
      *(float4*)(p + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
      if (p16) store4(p16, i, AFFT_BF16, pp);
      if (p16h) store4(p16h, i, AFFT_F16, pp);
      if (p8) store_e4m3<4>(p8, i, pp, 256.0f);
    } else {
      for (int64_t j = i; j < n; ++j) {
        float pj = p[j], bj = (first & AFFT_SGD_FIRST_STEP) ? 0.f : buf[j];
        sgd_update(pj, bj, ld_any(g_, j, g_dtype), lr, mom, wd, gscale, first);
        buf[j] = bj;
        p[j] = pj;
        if (p16) p16[j] = f2bf(pj);
        if (p16h) p16h[j] = f2h(pj);
        if (p8) p8[j] = f2e4m3(pj * 256.0f);
      }
    }
  }
}

// the same update over a table of runs {start, length}: block b owns run b (runs are short: biases, LayerNorm weights)
__global__ __launch_bounds__(256) void sgd_runs_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                       bf16_t* __restrict__ p16, bf16_t* __restrict__ p16h, unsigned char* __restrict__ p8,
                                                       const int64_t* __restrict__ runs, float lr, float mom,
                                                       float wd, float gscale, int first, const float* __restrict__ ok) {
  if (ok && *ok == 0.f) return;
  const int64_t s0 = runs[2 * blockIdx.x], len = runs[2 * blockIdx.x + 1];
  for (int64_t j = s0 + threadIdx.x; j < s0 + len; j += 256) {
    float pj = p[j], bj = (first & AFFT_SGD_FIRST_STEP) ? 0.f : buf[j];
    sgd_update(pj, bj, g[j], lr, mom, wd, gscale, first);
    buf[j] = bj;
    p[j] = pj;
    if (p16) p16[j] = f2bf(pj);
    if (p16h) p16h[j] = f2h(pj);
    if (p8) p8[j] = f2e4m3(pj * 256.0f);
  }
}

}  // namespace

extern "C" int afft_sgd_nesterov_runs2(float* p, const float* g, float* buf, void* p_bf16, void* p_f16, void* p_f8, const int64_t* runs, int32_t nruns,
                                       float lr, float mom, float wd, float gscale, int32_t first_step, const float* ok, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(p && g && buf && (runs || nruns == 0), "sgd_runs: null pointer");
  if (nruns <= 0) return 0;
  hipLaunchKernelGGL(sgd_runs_kernel, dim3(nruns), dim3(256), 0, stream, p, g, buf, (bf16_t*)p_bf16, (bf16_t*)p_f16, (unsigned char*)p_f8, runs, lr, mom, wd, gscale,
                     first_step, ok);
  AFFT_LAUNCH_CHECK();
  return 0;
}
extern "C" int afft_sgd_nesterov_runs(float* p, const float* g, float* buf, void* p_bf16, const int64_t* runs, int32_t nruns,
                                      float lr, float mom, float wd, float gscale, int32_t first_step, void* stream_) {
  return afft_sgd_nesterov_runs2(p, g, buf, p_bf16, nullptr, nullptr, runs, nruns, lr, mom, wd, gscale, first_step, nullptr, stream_);
}

namespace {
// fp32 [rows, cols] -> fragment-packed bf16 image (include/afft_hip.h: afft_pack_weight).  One thread per 8-element fragment:
// reads 32 contiguous bytes of a row, writes the 16-byte fragment; consecutive threads walk the k-chunks of one row, so reads
// are coalesced along the row and a wave's writes fall into 16 rows x 4 lane groups of one or two 1-KiB blocks.
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ src, int64_t lds_, int rows, int cols,
                                                          bf16_t* __restrict__ dst) {
  const int chunks = cols >> 3;
  const int64_t total = (int64_t)rows * chunks;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int m = (int)(i / chunks), n = (int)(i % chunks) << 3;
    const float4 a = *(const float4*)(src + (int64_t)m * lds_ + n), b = *(const float4*)(src + (int64_t)m * lds_ + n + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    store8(dst, packed_frag(m, n, cols), AFFT_BF16, v);
  }
}
}  // namespace

extern "C" int afft_pack_weight(const float* src, int64_t lds_, int32_t rows, int32_t cols, void* dst, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(src && dst, "pack_weight: null pointer");
  AFFT_CHECK(rows > 0 && cols > 0 && rows % 16 == 0 && cols % 32 == 0 && lds_ % 4 == 0, "pack_weight: needs rows %% 16 == 0, cols %% 32 == 0 (got %d x %d)", rows, cols);
  AFFT_CHECK(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "pack_weight: buffers must be 16-byte aligned");
  const int64_t total = (int64_t)rows * (cols >> 3);
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, stream, src, lds_, rows, cols, (bf16_t*)dst);
  AFFT_LAUNCH_CHECK();
  return 0;
}

extern "C" int afft_cast(const float* src, int64_t lds_, int32_t rows, int32_t cols, void* dst, int64_t ldd,
                         int32_t dst_dtype, void* dst_t, int64_t ldt, int32_t zero_pad, const afft_dropout_t* drop,
                         void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(src && (dst || dst_t), "cast: null pointer");
  AFFT_CHECK(!dst || ldd >= cols, "cast: ldd < cols");
  AFFT_CHECK(!dst_t || ldt >= rows, "cast: ldt < rows");
  if (rows == 0 || cols == 0) return 0;
  const int pad_cols = (dst && zero_pad) ? (int)ldd : cols;
  if (dst && !dst_t && pad_cols == cols && cols % 4 == 0 && lds_ % 4 == 0 && ldd % 4 == 0 &&
      (((uintptr_t)src) & 15) == 0 && (((uintptr_t)dst) & (dst_dtype == AFFT_F32 ? 15 : 7)) == 0) {
    const int64_t total = (int64_t)rows * (cols / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(cast_rows4_kernel, dim3((int)blocks), dim3(256), 0, stream, src, lds_, rows, cols, dst, ldd,
                       dst_dtype, make_drop(drop));
    AFFT_LAUNCH_CHECK();
    return 0;
  }
  dim3 grid((pad_cols + 63) / 64, (rows + 63) / 64);
  hipLaunchKernelGGL(cast_kernel, grid, dim3(256), 0, stream, src, lds_, rows, cols, dst, ldd, dst_dtype, dst_t, ldt,
                     pad_cols, make_drop(drop));
  AFFT_LAUNCH_CHECK();
  return 0;
}

static int split_planes(bool f16, const float* src, int64_t lds_, int32_t rows, int32_t cols, void* hi, int64_t ldd,
                        int32_t rows_pad, int64_t plane_stride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(src && hi, "split_bf16: null pointer");
  AFFT_CHECK(rows >= 0 && cols >= 0 && rows_pad >= rows && ldd >= cols, "split_bf16: bad sizes");
  AFFT_CHECK(ldd % 8 == 0 && plane_stride % 8 == 0 && (((uintptr_t)hi) & 15) == 0, "split_bf16: planes must be 16-byte aligned with ldd %% 8 == 0");
  AFFT_CHECK(plane_stride >= (int64_t)rows_pad * ldd, "split_bf16: planes overlap");
  if (rows_pad == 0 || ldd == 0) return 0;
  const int src_vec = lds_ % 4 == 0 && (((uintptr_t)src) & 15) == 0;
  int64_t blocks = ((int64_t)rows_pad * (ldd / 8) + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (f16) hipLaunchKernelGGL(split_bf16_kernel<true>, dim3((int)blocks), dim3(256), 0, stream, src, lds_, rows, cols, (bf16_t*)hi, ldd,
                              rows_pad, plane_stride, src_vec);
  else hipLaunchKernelGGL(split_bf16_kernel<false>, dim3((int)blocks), dim3(256), 0, stream, src, lds_, rows, cols, (bf16_t*)hi, ldd,
                          rows_pad, plane_stride, src_vec);
  AFFT_LAUNCH_CHECK();
  return 0;
}

namespace {
// e4m3 byte image (optionally of the fp16 split's lo part): 8 columns per thread, zero-filled out to [rows_pad, ldd]
__global__ __launch_bounds__(256) void quant_e4m3_kernel(const float* __restrict__ src, int64_t lds_, int rows, int cols, float scale,
                                                         unsigned char* __restrict__ dst, int64_t ldd, int rows_pad, bf16_t* __restrict__ hi,
                                                         int src_vec) {
  const int nq = (int)(ldd >> 3);
  const int64_t total = (int64_t)rows_pad * nq;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / nq), c = (int)(i - (int64_t)r * nq) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
      const float* p = src + (int64_t)r * lds_ + c;
      if (src_vec && c + 7 < cols) load8(p, 0, AFFT_F32, v);
      else
#pragma unroll
        for (int k = 0; k < 8; ++k) if (c + k < cols) v[k] = p[k];
    }
    if (hi) {
      f16x8_v hv;
#pragma unroll
      for (int k = 0; k < 8; ++k) { hv[k] = (_Float16)v[k]; v[k] -= (float)hv[k]; }
      *(f16x8_v*)(hi + (int64_t)r * ldd + c) = hv;
    }
    store_e4m3<8>(dst, (int64_t)r * ldd + c, v, scale);
  }
}
}  // namespace

extern "C" int afft_quant_e4m3(const float* src, int64_t lds_, int32_t rows, int32_t cols, float scale, void* dst, int64_t ldd,
                               int32_t rows_pad, void* hi, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(src && dst, "quant_e4m3: null pointer");
  AFFT_CHECK(rows >= 0 && cols >= 0 && rows_pad >= rows && ldd >= cols && ldd % 8 == 0, "quant_e4m3: bad sizes (ldd %% 8 == 0)");
  AFFT_CHECK((((uintptr_t)dst) & 7) == 0 && (((uintptr_t)hi) & 15) == 0, "quant_e4m3: misaligned planes");
  if (rows_pad == 0 || ldd == 0) return 0;
  const int src_vec = lds_ % 4 == 0 && (((uintptr_t)src) & 15) == 0;
  int64_t blocks = ((int64_t)rows_pad * (ldd / 8) + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(quant_e4m3_kernel, dim3((int)blocks), dim3(256), 0, stream, src, lds_, rows, cols, scale, (unsigned char*)dst, ldd,
                     rows_pad, (bf16_t*)hi, src_vec);
  AFFT_LAUNCH_CHECK();
  return 0;
}

extern "C" int afft_split_bf16(const float* src, int64_t lds_, int32_t rows, int32_t cols, void* hi, int64_t ldd,
                               int32_t rows_pad, int64_t plane_stride, void* stream_) {
  return split_planes(false, src, lds_, rows, cols, hi, ldd, rows_pad, plane_stride, stream_);
}

extern "C" int afft_split_f16(const float* src, int64_t lds_, int32_t rows, int32_t cols, void* hi, int64_t ldd,
                              int32_t rows_pad, int64_t plane_stride, void* stream_) {
  return split_planes(true, src, lds_, rows, cols, hi, ldd, rows_pad, plane_stride, stream_);
}

extern "C" int afft_assemble_tokens(const float* const* feats, const int64_t* ldf, int32_t n_mod, const float* token,
                                    int64_t tok_stride_t, const float* mod_embed, int32_t BT, int32_t T, int32_t d,
                                    float* X, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(feats && ldf && token && X, "assemble_tokens: null pointer");
  AFFT_CHECK(n_mod >= 1 && n_mod <= 8, "assemble_tokens: %d modalities (1..8 supported)", n_mod);
  AFFT_CHECK(d % 4 == 0, "assemble_tokens: d %% 4 != 0");
  Modal8 m;
  for (int i = 0; i < 8; ++i) { m.p[i] = i < n_mod ? feats[i] : nullptr; m.ld[i] = i < n_mod ? ldf[i] : 0; }
  for (int i = 0; i < n_mod; ++i) AFFT_CHECK(m.p[i] && m.ld[i] % 4 == 0, "assemble_tokens: modality %d pointer/stride", i);
  if (BT == 0) return 0;
  const int S = n_mod + 1;
  hipLaunchKernelGGL(assemble_kernel, dim3(BT * S), dim3(256), 0, stream, m, S, token, tok_stride_t, mod_embed, T, d, X);
  AFFT_LAUNCH_CHECK();
  return 0;
}

extern "C" int afft_colsum(const void* src, int64_t lds_, int32_t dtype, int32_t rows, int32_t cols, float* out,
                           int32_t accumulate, void* workspace, int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(src && out, "colsum: null pointer");
  AFFT_CHECK(rows >= 0 && cols >= 0 && lds_ >= cols, "colsum: bad sizes");
  AFFT_CHECK(!workspace || (((uintptr_t)workspace) & 15) == 0, "colsum: workspace must be 16-byte aligned");
  if (cols == 0) return 0;
#ifdef AFFT_EXPERIMENT_SKIP_COLSUM      // what the bias-gradient column sums cost inside the step (wrong bias gradients: timing only; needs make DIAG=1)
  if (rows > 256) return 0;
#endif
  if (rows == 0) {
    if (!accumulate && hipMemsetAsync(out, 0, sizeof(float) * cols, stream) != hipSuccess) { afft_set_error("colsum: memset failed"); return 2; }
    return 0;
  }
  const bool v4 = cols % 4 == 0 && lds_ % 4 == 0 && (((uintptr_t)src) & (dtype == AFFT_F32 ? 15 : 7)) == 0;
  const int strips = (cols + 255) / 256;
  // row blocks the workspace has room for: 1 KiB of partials per (strip, block), behind the header the split-K GEMMs of the
  // same stream keep their counters in (so that one scratch buffer per stream serves both)
  int64_t room = 0;
  if (workspace && workspace_bytes > AFFT_GEMM_WS_HEADER) room = (workspace_bytes - AFFT_GEMM_WS_HEADER) / ((int64_t)strips * 1024);
  int rpb = v4 ? 128 : 64;
  int nby = (rows + rpb - 1) / rpb;
  if (nby > 1 && room < nby) {
    if (room >= 2) { rpb = (int)((rows + room - 1) / room); rpb = (rpb + 3) / 4 * 4; nby = (rows + rpb - 1) / rpb; }
    else nby = 0;     // no (usable) workspace
  }
  if (nby >= 1) {
    float* part = workspace ? (float*)((char*)workspace + AFFT_GEMM_WS_HEADER) : nullptr;
    dim3 grid(strips, nby);
    if (v4) hipLaunchKernelGGL(colsum_blocks_kernel<true>, grid, dim3(256), 0, stream, src, lds_, dtype, rows, cols, rpb, accumulate, out, part);
    else hipLaunchKernelGGL(colsum_blocks_kernel<false>, grid, dim3(256), 0, stream, src, lds_, dtype, rows, cols, rpb, accumulate, out, part);
    AFFT_LAUNCH_CHECK();
    if (nby > 1) {
      hipLaunchKernelGGL(colsum_combine_kernel, dim3(strips), dim3(256), 0, stream, part, nby, cols, accumulate, out);
      AFFT_LAUNCH_CHECK();
    }
    return 0;
  }
  if (v4 && dtype == AFFT_BF16) hipLaunchKernelGGL(colsum4_kernel<16>, dim3((cols + 63) / 64), dim3(256), 0, stream, src, lds_, dtype, rows, cols, accumulate, out);
  else if (v4) hipLaunchKernelGGL(colsum4_kernel<8>, dim3((cols + 31) / 32), dim3(256), 0, stream, src, lds_, dtype, rows, cols, accumulate, out);
  else hipLaunchKernelGGL(colsum_kernel, dim3((cols + 255) / 256), dim3(256), 0, stream, src, lds_, dtype, rows, cols, accumulate, out);
  AFFT_LAUNCH_CHECK();
  return 0;
}

// zero fill by a kernel (16-byte stores): a hipMemsetAsync of this size inside a stream capture became a memset node that crashed
// hipGraph instantiation on ROCm 7.0 depending on the buffer's address (found in round 5)
__global__ __launch_bounds__(256) void zero16_kernel(uint4* __restrict__ p, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) p[i] = make_uint4(0, 0, 0, 0);
}
extern "C" int afft_zero(void* p, int64_t bytes, void* stream_) {
  AFFT_CHECK(p && bytes >= 0 && bytes % 16 == 0 && ((uintptr_t)p & 15) == 0, "afft_zero: a 16-byte aligned pointer and a multiple of 16 bytes");
  if (bytes == 0) return 0;
  const int64_t n16 = bytes / 16;
  const int grid = (int)std::min<int64_t>((n16 + 255) / 256, 2048);
  hipLaunchKernelGGL(zero16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream_, (uint4*)p, n16);
  AFFT_LAUNCH_CHECK();
  return 0;
}

extern "C" int afft_gather_frames(float* out, int64_t out_clip_stride, int64_t out_frame_stride, int32_t clips, int32_t frames,
                                  int32_t C, int32_t nsrc, const float* const* src, const int64_t* clip_stride,
                                  const int64_t* frame_stride, const int32_t* lo, const int32_t* hi, const int32_t* off, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(out && nsrc >= 0 && nsrc <= 4 && C > 0 && C % 4 == 0, "gather_frames: bad argument (at most 4 sources, C a multiple of 4)");
  AFFT_CHECK(out_clip_stride % 4 == 0 && out_frame_stride % 4 == 0 && ((uintptr_t)out & 15) == 0, "gather_frames: unaligned output");
  FrameSrcs s{};
  s.n = nsrc;
  for (int k = 0; k < nsrc; ++k) {
    AFFT_CHECK(src[k] && clip_stride[k] % 4 == 0 && frame_stride[k] % 4 == 0 && ((uintptr_t)src[k] & 15) == 0, "gather_frames: unaligned source");
    AFFT_CHECK(lo[k] >= 0 && hi[k] <= frames && lo[k] + off[k] >= 0, "gather_frames: a source's frame range leaves the output or starts before the source");
    s.p[k] = src[k]; s.clip_stride[k] = clip_stride[k]; s.frame_stride[k] = frame_stride[k]; s.lo[k] = lo[k]; s.hi[k] = hi[k]; s.off[k] = off[k];
  }
  if (clips == 0 || frames == 0) return 0;
  const int total = frames * (C / 4);
  int gx = (total + 255) / 256;
  const int want = (2048 + clips - 1) / clips;      // ~2048 workgroups in all
  if (gx > want) gx = want < 1 ? 1 : want;
  hipLaunchKernelGGL(gather_frames_kernel, dim3(gx, clips), dim3(256), 0, stream, s, out, out_clip_stride, out_frame_stride, frames, C / 4);
  AFFT_LAUNCH_CHECK();
  return 0;
}

extern "C" int afft_add_rows_periodic(const float* x, int64_t ldx, const float* table, int64_t ldt, int32_t rows,
                                      int32_t period, int32_t d, float* y, int64_t ldy, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(x && table && y && period > 0, "add_rows_periodic: bad argument");
  AFFT_CHECK(d % 4 == 0 && ldx % 4 == 0 && ldt % 4 == 0 && ldy % 4 == 0, "add_rows_periodic: sizes must be multiples of 4");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(add_rows_periodic_kernel, dim3(rows), dim3(256), 0, stream, x, ldx, table, ldt, period, d, y, ldy);
  AFFT_LAUNCH_CHECK();
  return 0;
}

extern "C" int afft_reduce_rows_periodic(const float* src, int64_t lds_, int32_t rows, int32_t period, int32_t d,
                                         float* out, int64_t ldo, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(src && out && period > 0, "reduce_rows_periodic: bad argument");
  if (rows == 0 || d == 0) return 0;
  // dense rows: row r = k * period + p is row k, columns p*d.. of the [rows/period, period*d] view, so the sums are the
  // column sums of that view (many workgroups, coalesced); period 1 is a plain column sum at any row stride
  if (period == 1) return afft_colsum(src, lds_, AFFT_F32, rows, d, out, 1, nullptr, 0, stream_);
  if (lds_ == d && ldo == d && rows % period == 0)
    return afft_colsum(src, (int64_t)period * d, AFFT_F32, rows / period, period * d, out, 1, nullptr, 0, stream_);
  dim3 grid((d + 255) / 256, period);
  hipLaunchKernelGGL(reduce_rows_periodic_kernel, grid, dim3(256), 0, stream, src, lds_, rows, period, d, out, ldo);
  AFFT_LAUNCH_CHECK();
  return 0;
}

// ---- dropout salt (see common.h): set once, advanced by a one-thread kernel at the start of every step
__global__ void salt_step_kernel(unsigned* salt) { *salt = mix32(*salt + 0x9E3779B9u); }

// ---- MixUp with an ignore class as a GPU prologue (common/mixup.py:119-182), no host round trip
// plan: sample b takes part iff none of its T past labels is the ignore class; the participants are mixed with the
// participants in reverse order (x[sel] * lam + x[sel].flip(0) * (1 - lam)); partner[b] = b when b does not take part
// or when at most one sample does (:156-158: no mixing then).  One workgroup; B is a batch size (<= 4096).
__global__ __launch_bounds__(256) void mixup_plan_kernel(const int64_t* __restrict__ sub, int B, int T, int64_t ignore_cls,
                                                         int* __restrict__ partner, uint8_t* __restrict__ ign) {
  extern __shared__ int sh[];          // [B] flags, then ranks
  for (int b = threadIdx.x; b < B; b += 256) {
    int ok = 1;
    if (sub)
      for (int t = 0; t < T; ++t) {
        const bool ig = sub[(int64_t)b * T + t] == ignore_cls;
        if (ign) ign[(int64_t)b * T + t] = ig ? 1 : 0;
        ok &= ig ? 0 : 1;
      }
    sh[b] = ok;
  }
  __syncthreads();
  __shared__ int count;
  if (threadIdx.x == 0) {
    int c = 0;
    for (int b = 0; b < B; ++b) { const int f = sh[b]; sh[b] = f ? c : -1; c += f; }
    count = c;
  }
  __syncthreads();
  // rank r of the participants pairs with rank count-1-r: find it by a second pass (B is small)
  for (int b = threadIdx.x; b < B; b += 256) {
    int pb = b;
    if (count > 1 && sh[b] >= 0) {
      const int want = count - 1 - sh[b];
      for (int j = 0; j < B; ++j)
        if (sh[j] == want) { pb = j; break; }
    }
    partner[b] = pb;
  }
}
__global__ __launch_bounds__(256) void mixup_rows_kernel(const float* __restrict__ x, int64_t W, const int* __restrict__ partner,
                                                         float lam, float* __restrict__ y) {
  const int b = blockIdx.y, pb = partner[b];
  const float* xa = x + (int64_t)b * W;
  const float* xb = x + (int64_t)pb * W;
  float* yo = y + (int64_t)b * W;
  const float oml = 1.0f - lam;
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < W; c += (int64_t)gridDim.x * 256)
    yo[c] = pb == b ? xa[c] : xa[c] * lam + xb[c] * oml;
}
// soft labels of rows r = b * rps + i: lam * onehot(l[r]) + (1 - lam) * onehot(l[partner row]), smoothed one-hot
// (common/mixup.py:17-47: off value ls / K, on value 1 - ls + ls / K); ignored labels count as class 0 (:150-151)
__global__ __launch_bounds__(256) void mixup_labels_kernel(const int64_t* __restrict__ labels, int rps, int K, float ls,
                                                           int64_t ignore_cls, const int* __restrict__ partner, float lam,
                                                           float* __restrict__ out) {
  const int r = blockIdx.x, b = r / rps, i = r - b * rps, pb = partner[b];
  int64_t la = labels[r], lb = labels[(int64_t)pb * rps + i];
  if (la == ignore_cls) la = 0;
  if (lb == ignore_cls) lb = 0;
  const float off = ls / (float)K, on = 1.0f - ls + off, oml = 1.0f - lam;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float a = k == la ? on : off, bb = k == lb ? on : off;
    out[(int64_t)r * K + k] = pb == b ? a : a * lam + bb * oml;
  }
}
// ZeroMaskRULSTMFeats (common/transforms.py:13-26) for a whole batch on the device: in every clip exactly k of the T
// frames, a uniformly random subset, are set to zero.  Frame t of clip b is masked iff fewer than k frames of the clip
// have a smaller hash(key, b, t) (ties broken by index): no host RNG, reproducible from the key.
__global__ __launch_bounds__(256) void zero_mask_frames_kernel(float* __restrict__ x, int T, int64_t C, int k, unsigned key) {
  const int b = blockIdx.y, t = blockIdx.x;
  const unsigned mine = mix32(key ^ mix32((unsigned)b * 0x9E3779B1u + (unsigned)t));
  int smaller = 0;
  for (int j = 0; j < T; ++j) {
    const unsigned h = mix32(key ^ mix32((unsigned)b * 0x9E3779B1u + (unsigned)j));
    smaller += (h < mine || (h == mine && j < t)) ? 1 : 0;
  }
  if (smaller >= k) return;
  float* row = x + ((int64_t)b * T + t) * C;
  for (int64_t c = threadIdx.x; c < C; c += 256) row[c] = 0.f;
}

extern "C" int afft_zero_mask_frames(float* x, int32_t B, int32_t T, int64_t C, int32_t k, uint32_t key, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(x && k >= 0 && k <= T, "zero_mask_frames: bad argument (k = %d, T = %d)", k, T);
  if (B == 0 || T == 0 || C == 0 || k == 0) return 0;
  hipLaunchKernelGGL(zero_mask_frames_kernel, dim3(T, B), dim3(256), 0, stream, x, T, C, k, key);
  AFFT_LAUNCH_CHECK();
  return 0;
}

// softmax over wide fp32 rows (one workgroup per row): class probabilities for verb / noun marginalisation
// (challenge.py:196-203)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, int64_t ldx, int C,
                                                           float* __restrict__ y, int64_t ldy) {
  __shared__ float sh[4];
  const float* xr = x + (int64_t)blockIdx.x * ldx;
  float* yr = y + (int64_t)blockIdx.x * ldy;
  float m = -INFINITY;
  for (int c = threadIdx.x; c < C; c += 256) m = fmaxf(m, xr[c]);
  for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  __syncthreads();
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) s += expf(xr[c] - m);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  const float inv = 1.0f / ((sh[0] + sh[1]) + (sh[2] + sh[3]));
  for (int c = threadIdx.x; c < C; c += 256) yr[c] = expf(xr[c] - m) * inv;
}

const unsigned* g_afft_drop_salt = nullptr;

extern "C" int afft_set_dropout_salt(const uint32_t* salt_dev) {
  g_afft_drop_salt = salt_dev;
  return 0;
}
extern "C" int afft_dropout_salt_step(uint32_t* salt_dev, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(salt_dev, "dropout_salt_step: null pointer");
  hipLaunchKernelGGL(salt_step_kernel, dim3(1), dim3(1), 0, stream, salt_dev);
  AFFT_LAUNCH_CHECK();
  return 0;
}

extern "C" int afft_mixup_plan(const int64_t* labels_subclips, int32_t B, int32_t T, int64_t ignore_cls, int32_t* partner,
                               uint8_t* ignore_mask, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(partner && B >= 1 && B <= 4096 && T >= 0, "mixup_plan: bad argument (B = %d, at most 4096)", B);
  hipLaunchKernelGGL(mixup_plan_kernel, dim3(1), dim3(256), sizeof(int) * B, stream, labels_subclips, B, T, ignore_cls, partner,
                     ignore_mask);
  AFFT_LAUNCH_CHECK();
  return 0;
}
extern "C" int afft_mixup_rows(const float* x, int32_t B, int64_t W, const int32_t* partner, float lam, float* y, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(x && y && partner, "mixup_rows: null pointer");
  if (B == 0 || W == 0) return 0;
  int64_t gx = (W + 255) / 256;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(mixup_rows_kernel, dim3((int)gx, B), dim3(256), 0, stream, x, W, partner, lam, y);
  AFFT_LAUNCH_CHECK();
  return 0;
}
extern "C" int afft_mixup_labels(const int64_t* labels, int32_t B, int32_t rows_per_sample, int32_t K, float label_smooth,
                                 int64_t ignore_cls, const int32_t* partner, float lam, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(labels && partner && out && K >= 1 && rows_per_sample >= 1, "mixup_labels: bad argument");
  if (B == 0) return 0;
  hipLaunchKernelGGL(mixup_labels_kernel, dim3(B * rows_per_sample), dim3(256), 0, stream, labels, rows_per_sample, K, label_smooth,
                     ignore_cls, partner, lam, out);
  AFFT_LAUNCH_CHECK();
  return 0;
}
extern "C" int afft_softmax_rows(const float* x, int64_t ldx, int32_t rows, int32_t C, float* y, int64_t ldy, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(x && y && C >= 1, "softmax_rows: bad argument");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, stream, x, ldx, C, y, ldy);
  AFFT_LAUNCH_CHECK();
  return 0;
}

// backward of y = drop(act(pre)) (see afft_act_bwd in the header)
__global__ __launch_bounds__(256) void act_bwd_kernel(int act, const float* __restrict__ dy, int64_t lddy,
                                                      const void* __restrict__ saved, int64_t lds_, int sdt,
                                                      const float* __restrict__ aux, int64_t ldaux, int rows, int cols,
                                                      const DropParams drop_, void* __restrict__ dpre, int64_t lddp, int pdt,
                                                      float* __restrict__ daux, int64_t ldda) {
  const DropParams drop = with_salt(drop_);
  const int64_t n = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
    float g = dy[(int64_t)r * lddy + c];
    if (drop.thresh || drop.path_thresh) g *= drop_row_scale(drop, r) * drop_elem_scale(drop, (unsigned)r * (unsigned)cols + (unsigned)c);
    const float p = ld_any(saved, (int64_t)r * lds_ + c, sdt);
    float dp;
    switch (act) {
      case AFFT_ACT_GELU_ERF: dp = g * dgelu_erf_f(p); break;
      case AFFT_ACT_GELU_TANH: dp = g * dgelu_tanh_f(p); break;
      case AFFT_ACT_RELU: dp = p > 0.f ? g : 0.f; break;
      case AFFT_ACT_SIGMOID_GATE: {
        const float sg = 1.0f / (1.0f + __expf(-p));
        const float a = aux[(int64_t)r * ldaux + c];
        dp = g * a * sg * (1.0f - sg);
        if (daux) daux[(int64_t)r * ldda + c] = g * sg;
        break;
      }
      default: dp = g;
    }
    st_any(dpre, (int64_t)r * lddp + c, pdt, dp);
  }
}

// softmax over n <= 32 columns, one thread per row
__global__ __launch_bounds__(256) void softmax_small_fwd_kernel(const float* __restrict__ x, int64_t ldx, int rows, int n,
                                                                float* __restrict__ y, int64_t ldy) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float v[32];
  float m = -INFINITY;
  for (int j = 0; j < n; ++j) { v[j] = x[(int64_t)r * ldx + j]; m = fmaxf(m, v[j]); }
  float s = 0.f;
  for (int j = 0; j < n; ++j) { v[j] = expf(v[j] - m); s += v[j]; }
  const float inv = 1.0f / s;
  for (int j = 0; j < n; ++j) y[(int64_t)r * ldy + j] = v[j] * inv;
}
__global__ __launch_bounds__(256) void softmax_small_bwd_kernel(const float* __restrict__ y, int64_t ldy,
                                                                const float* __restrict__ dy, int64_t lddy, int rows, int n,
                                                                float* __restrict__ dx, int64_t lddx) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float dot = 0.f;
  for (int j = 0; j < n; ++j) dot += dy[(int64_t)r * lddy + j] * y[(int64_t)r * ldy + j];
  for (int j = 0; j < n; ++j) dx[(int64_t)r * lddx + j] = y[(int64_t)r * ldy + j] * (dy[(int64_t)r * lddy + j] - dot);
}

// score fusion: one workgroup per row
struct WsPtrs { const float* x[8]; float* dx[8]; };
__global__ __launch_bounds__(256) void weighted_sum_fwd_kernel(WsPtrs p, int64_t ldx, const float* __restrict__ w, int64_t ldw,
                                                               int n, int cols, float* __restrict__ out, int64_t ldo) {
  const int r = blockIdx.x;
  float wr[8];
  for (int i = 0; i < n; ++i) wr[i] = w[(int64_t)r * ldw + i];
  for (int c = threadIdx.x; c < cols; c += 256) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += wr[i] * p.x[i][(int64_t)r * ldx + c];
    out[(int64_t)r * ldo + c] = s;
  }
}
__global__ __launch_bounds__(256) void weighted_sum_bwd_kernel(WsPtrs p, int64_t ldx, const float* __restrict__ w, int64_t ldw,
                                                               const float* __restrict__ dout, int64_t lddo, int n, int cols,
                                                               int64_t lddx, float* __restrict__ dw, int64_t lddw) {
  __shared__ float sh[8][4];
  const int r = blockIdx.x;
  float wr[8], acc[8];
  for (int i = 0; i < n; ++i) { wr[i] = w[(int64_t)r * ldw + i]; acc[i] = 0.f; }
  for (int c = threadIdx.x; c < cols; c += 256) {
    const float g = dout[(int64_t)r * lddo + c];
    for (int i = 0; i < n; ++i) {
      acc[i] += g * p.x[i][(int64_t)r * ldx + c];
      if (p.dx[i]) p.dx[i][(int64_t)r * lddx + c] = wr[i] * g;
    }
  }
  for (int i = 0; i < n; ++i) {
    const float s = wave_sum(acc[i]);
    if ((threadIdx.x & 63) == 0) sh[i][threadIdx.x >> 6] = s;
  }
  __syncthreads();
  if (threadIdx.x < n && dw) dw[(int64_t)r * lddw + threadIdx.x] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

extern "C" int afft_act_bwd(int32_t act, const float* dy, int64_t lddy, const void* saved, int64_t lds_, int32_t saved_dtype,
                            const float* aux, int64_t ldaux, int32_t rows, int32_t cols, const afft_dropout_t* drop,
                            void* dpre, int64_t lddp, int32_t dpre_dtype, float* daux, int64_t ldda, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(dy && dpre, "act_bwd: null pointer");
  AFFT_CHECK(act == AFFT_ACT_NONE || saved, "act_bwd: the activation needs its saved input");
  AFFT_CHECK(act != AFFT_ACT_SIGMOID_GATE || aux, "act_bwd: the gate needs aux");
  AFFT_CHECK(act == AFFT_ACT_NONE || act == AFFT_ACT_GELU_ERF || act == AFFT_ACT_GELU_TANH || act == AFFT_ACT_RELU ||
             act == AFFT_ACT_SIGMOID_GATE, "act_bwd: bad activation %d", act);
  if (rows == 0 || cols == 0) return 0;
  int64_t blocks = ((int64_t)rows * cols + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  const void* sv = saved ? saved : (const void*)dy;
  hipLaunchKernelGGL(act_bwd_kernel, dim3((int)blocks), dim3(256), 0, stream, act, dy, lddy, sv, saved ? lds_ : lddy,
                     saved ? saved_dtype : AFFT_F32, aux, ldaux, rows, cols, make_drop(drop), dpre, lddp, dpre_dtype, daux, ldda);
  AFFT_LAUNCH_CHECK();
  return 0;
}
extern "C" int afft_softmax_small_fwd(const float* x, int64_t ldx, int32_t rows, int32_t n, float* y, int64_t ldy, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(x && y && n >= 1 && n <= 32, "softmax_small_fwd: bad argument (n = %d, at most 32)", n);
  if (rows == 0) return 0;
  hipLaunchKernelGGL(softmax_small_fwd_kernel, dim3((rows + 255) / 256), dim3(256), 0, stream, x, ldx, rows, n, y, ldy);
  AFFT_LAUNCH_CHECK();
  return 0;
}
extern "C" int afft_softmax_small_bwd(const float* y, int64_t ldy, const float* dy, int64_t lddy, int32_t rows, int32_t n,
                                      float* dx, int64_t lddx, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(y && dy && dx && n >= 1 && n <= 32, "softmax_small_bwd: bad argument (n = %d, at most 32)", n);
  if (rows == 0) return 0;
  hipLaunchKernelGGL(softmax_small_bwd_kernel, dim3((rows + 255) / 256), dim3(256), 0, stream, y, ldy, dy, lddy, rows, n, dx, lddx);
  AFFT_LAUNCH_CHECK();
  return 0;
}
extern "C" int afft_weighted_sum_fwd(const float* const* x, int64_t ldx, const float* w, int64_t ldw, int32_t n, int32_t rows,
                                     int32_t cols, float* out, int64_t ldo, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(x && w && out && n >= 1 && n <= 8, "weighted_sum_fwd: bad argument (n = %d, at most 8)", n);
  if (rows == 0 || cols == 0) return 0;
  WsPtrs p;
  for (int i = 0; i < 8; ++i) { p.x[i] = i < n ? x[i] : nullptr; p.dx[i] = nullptr; }
  hipLaunchKernelGGL(weighted_sum_fwd_kernel, dim3(rows), dim3(256), 0, stream, p, ldx, w, ldw, n, cols, out, ldo);
  AFFT_LAUNCH_CHECK();
  return 0;
}
extern "C" int afft_weighted_sum_bwd(const float* const* x, int64_t ldx, const float* w, int64_t ldw, const float* dout,
                                     int64_t lddo, int32_t n, int32_t rows, int32_t cols, float* const* dx, int64_t lddx,
                                     float* dw, int64_t lddw, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(x && w && dout && n >= 1 && n <= 8, "weighted_sum_bwd: bad argument (n = %d, at most 8)", n);
  if (rows == 0 || cols == 0) return 0;
  WsPtrs p;
  for (int i = 0; i < 8; ++i) { p.x[i] = i < n ? x[i] : nullptr; p.dx[i] = (i < n && dx) ? dx[i] : nullptr; }
  hipLaunchKernelGGL(weighted_sum_bwd_kernel, dim3(rows), dim3(256), 0, stream, p, ldx, w, ldw, dout, lddo, n, cols, lddx, dw, lddw);
  AFFT_LAUNCH_CHECK();
  return 0;
}

// y[g, :] = scale * sum_s x[g, s, :]  (x fp32 [G, S, W], W % 4 == 0): the token mean of the fusers without a modality
// token (models/fusion.py:114-116 CMFuser, :207-210 T-SA-Fuser); and its backward, a broadcast
__global__ __launch_bounds__(256) void group_sum_kernel(const float* __restrict__ x, int S, int64_t W, float scale,
                                                        float* __restrict__ y) {
  const int64_t g = blockIdx.y;
  const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (c >= W) return;
  const float* xg = x + g * S * W + c;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < S; ++s) {
    const float4 t = *(const float4*)(xg + (int64_t)s * W);
    a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
  }
  *(float4*)(y + g * W + c) = make_float4(a.x * scale, a.y * scale, a.z * scale, a.w * scale);
}
__global__ __launch_bounds__(256) void group_bcast_kernel(const float* __restrict__ dy, int S, int64_t W, float scale,
                                                          float* __restrict__ dx) {
  const int64_t g = blockIdx.y;
  const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (c >= W) return;
  float4 t = *(const float4*)(dy + g * W + c);
  t = make_float4(t.x * scale, t.y * scale, t.z * scale, t.w * scale);
  for (int s = 0; s < S; ++s) *(float4*)(dx + (g * S + s) * W + c) = t;
}

extern "C" int afft_group_sum(const float* x, int32_t G, int32_t S, int64_t W, float scale, float* y, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(x && y && S >= 1, "group_sum: bad argument");
  AFFT_CHECK(W % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0, "group_sum: W must be a multiple of 4, buffers 16-byte aligned");
  if (G == 0 || W == 0) return 0;
  hipLaunchKernelGGL(group_sum_kernel, dim3((unsigned)((W / 4 + 255) / 256), G), dim3(256), 0, stream, x, S, W, scale, y);
  AFFT_LAUNCH_CHECK();
  return 0;
}
extern "C" int afft_group_bcast(const float* dy, int32_t G, int32_t S, int64_t W, float scale, float* dx, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(dy && dx && S >= 1, "group_bcast: bad argument");
  AFFT_CHECK(W % 4 == 0 && (((uintptr_t)dy | (uintptr_t)dx) & 15) == 0, "group_bcast: W must be a multiple of 4, buffers 16-byte aligned");
  if (G == 0 || W == 0) return 0;
  hipLaunchKernelGGL(group_bcast_kernel, dim3((unsigned)((W / 4 + 255) / 256), G), dim3(256), 0, stream, dy, S, W, scale, dx);
  AFFT_LAUNCH_CHECK();
  return 0;
}

// sum of squares of a flat gradient buffer (fp32 or bf16): one partial per workgroup into the caller's scratch, added up in
// workgroup order by ordered_sum_kernel behind it (no float atomics: the clipping coefficient, hence the parameters, are
// bit-reproducible)
__global__ __launch_bounds__(256) void sumsq_kernel(const void* __restrict__ x, int dtype, int64_t n,
                                                    float* __restrict__ partials) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    if (i + 3 < n) {
      float v[4];
      load4(x, i, dtype, v);
      s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    } else {
      for (int64_t j = i; j < n; ++j) { const float v = ld_any(x, j, dtype); s += v * v; }
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float max_norm, float* __restrict__ coef,
                                 float* __restrict__ norm_out) {
  const float norm = sqrtf(*sumsq);
  if (norm_out) *norm_out = norm;
  const float c = max_norm / (norm + 1e-6f);    // torch.nn.utils.clip_grad_norm_: clamped to 1
  *coef = c < 1.0f ? c : 1.0f;
}

extern "C" int afft_sumsq(const void* x, int32_t dtype, int64_t n, float scale, float* out, void* workspace,
                          int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(x && out, "sumsq: null pointer");
  AFFT_CHECK(dtype == AFFT_F32 || dtype == AFFT_BF16, "sumsq: bad dtype");
  AFFT_CHECK((((uintptr_t)x) & 15) == 0, "sumsq: buffer must be 16-byte aligned");
  if (n == 0) return 0;
  int64_t blocks = (n + 1023) / 1024;
  if (blocks > AFFT_REDUCE_PARTIALS) blocks = AFFT_REDUCE_PARTIALS;
  AFFT_CHECK(workspace && workspace_bytes >= AFFT_GEMM_WS_HEADER + 4 * blocks,
             "sumsq: needs the stream's workspace (header + AFFT_REDUCE_PARTIALS floats)");
  float* scratch = (float*)((char*)workspace + AFFT_GEMM_WS_HEADER);
  hipLaunchKernelGGL(sumsq_kernel, dim3((int)blocks), dim3(256), 0, stream, x, dtype, n, scratch);
  AFFT_LAUNCH_CHECK();
  hipLaunchKernelGGL(ordered_sum_kernel, dim3(1), dim3(256), 0, stream, scratch, blocks, scale, out, 1);
  AFFT_LAUNCH_CHECK();
  return 0;
}

extern "C" int afft_clip_coef(const float* sumsq, float max_norm, float* coef, float* norm_out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(sumsq && coef && max_norm > 0.f, "clip_coef: bad argument");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, stream, sumsq, max_norm, coef, norm_out);
  AFFT_LAUNCH_CHECK();
  return 0;
}

extern "C" int afft_sgd_nesterov2(float* p, const void* g, int32_t g_dtype, float* buf, void* p_bf16, void* p_f16, void* p_f8, int64_t n, float lr,
                                  float mom, float wd, float gscale, const float* gscale_dev, int32_t first_step, const float* ok, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(p && g && buf, "sgd: null pointer");
  AFFT_CHECK(((uintptr_t)p & 15) == 0 && ((uintptr_t)g & 15) == 0 && ((uintptr_t)buf & 15) == 0, "sgd: buffers must be 16-byte aligned");
  AFFT_CHECK(g_dtype == AFFT_F32 || g_dtype == AFFT_BF16, "sgd: bad gradient dtype");
  if (n == 0) return 0;
  int64_t blocks = (n + 1023) / 1024;
  // Grid cap: one 256-thread block per CU.  The update runs beside the backward GEMMs, whose workgroups need a CU's whole
  // register file, so every CU an update wave sits on is a CU without a GEMM tile (4096 blocks: 17.4 ms/step, 256: 16.9,
  // 128: 17.1; tools/sgd_blocks.sh).  AFFT_SGD_BLOCKS overrides.
  static const int64_t max_blocks = [] {
    const char* e = getenv("AFFT_SGD_BLOCKS");
    const long v = e ? atol(e) : 0;
    return (int64_t)(v > 0 ? v : 256);
  }();
  if (blocks > max_blocks) blocks = max_blocks;
  hipLaunchKernelGGL(sgd_kernel, dim3((int)blocks), dim3(256), 0, stream, p, g, g_dtype, buf, (bf16_t*)p_bf16, (bf16_t*)p_f16, (unsigned char*)p_f8, n, lr, mom, wd,
                     gscale, gscale_dev, first_step, ok);
  AFFT_LAUNCH_CHECK();
  return 0;
}
extern "C" int afft_sgd_nesterov(float* p, const void* g, int32_t g_dtype, float* buf, void* p_bf16, int64_t n, float lr, float mom,
                                 float wd, float gscale, const float* gscale_dev, int32_t first_step, void* stream_) {
  return afft_sgd_nesterov2(p, g, g_dtype, buf, p_bf16, nullptr, nullptr, n, lr, mom, wd, gscale, gscale_dev, first_step, nullptr, stream_);
}
