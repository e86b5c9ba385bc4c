// Shared device/host helpers for the gfx950 kernels (wave64, MFMA, LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/afft_hip.h"

typedef unsigned short bf16_t;  // raw bf16 bits in memory
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define AFFT_LDS __attribute__((address_space(3)))
#define AFFT_GLOBAL __attribute__((address_space(1)))

void afft_set_error(const char* fmt, ...);

#define AFFT_CHECK(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      afft_set_error(__VA_ARGS__);       \
      return 1;                          \
    }                                    \
  } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device), from any host thread.  `done` is the
// call site's own bit set (one bit per device ordinal); the attribute is a property of the function on a device.
#include <atomic>
int afft_ensure_dynamic_lds(const void* kern, size_t bytes, std::atomic<uint64_t>* done);

// measurement hook for the non-GEMM kernels (afft_kernel_trace_begin / _end, elementwise.hip): a scope object in an entry point
// brackets everything the call enqueues on `stream` with an event pair while a trace is open; free otherwise (one relaxed load)
extern std::atomic<int> g_afft_ktrace_open;
void afft_ktrace_push(int kind, int rows, int width, int64_t bytes, int64_t flops, hipStream_t stream, hipEvent_t a, hipEvent_t b);
struct AfftKernelScope {
  hipEvent_t a = nullptr, b = nullptr;
  hipStream_t st; int kind, rows, width; int64_t bytes, flops;
  AfftKernelScope(int kind_, int rows_, int width_, int64_t bytes_, int64_t flops_, hipStream_t s)
      : st(s), kind(kind_), rows(rows_), width(width_), bytes(bytes_), flops(flops_) {
    if (!g_afft_ktrace_open.load(std::memory_order_relaxed)) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { (void)hipGetLastError(); a = b = nullptr; return; }
    (void)hipEventRecord(a, st);
  }
  ~AfftKernelScope() {
    if (!a) return;
    (void)hipEventRecord(b, st);
    afft_ktrace_push(kind, rows, width, bytes, flops, st, a, b);
  }
};

// cross-stream ordering: "everything enqueued on `from` so far happens before what is enqueued on `to` next"
static inline int afft_stream_follows(hipStream_t to, hipStream_t from) {
  if (to == from) return 0;
  static thread_local hipEvent_t ev = nullptr;     // re-recording is safe: a wait captures the record that precedes it
  if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
    afft_set_error("hipEventCreate failed");
    (void)hipGetLastError();
    return 2;
  }
  if (hipEventRecord(ev, from) != hipSuccess || hipStreamWaitEvent(to, ev, 0) != hipSuccess) {
    afft_set_error("event record / wait failed");
    (void)hipGetLastError();
    return 2;
  }
  return 0;
}

#define AFFT_LAUNCH_CHECK()                                                     \
  do {                                                                          \
    hipError_t e_ = hipGetLastError();                                          \
    if (e_ != hipSuccess) {                                                     \
      afft_set_error("%s:%d HIP launch error: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return 2;                                                                 \
    }                                                                           \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950
  return __builtin_bit_cast(unsigned short, (__bf16)f);
}

// fp16 planes of the "fp16x2" precision (AFFT_F16): raw fp16 bits in memory, RNE conversion (v_cvt_f16_f32)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_v;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_v;
__device__ __forceinline__ unsigned short f2h(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
__device__ __forceinline__ float h2f(unsigned short v) { return (float)__builtin_bit_cast(_Float16, v); }

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

__device__ __forceinline__ float ld_any(const void* p, int64_t idx, int dtype) {
  return dtype == AFFT_F32 ? ((const float*)p)[idx] : dtype == AFFT_F16 ? h2f(((const bf16_t*)p)[idx]) : bf2f(((const bf16_t*)p)[idx]);
}
__device__ __forceinline__ void st_any(void* p, int64_t idx, int dtype, float v) {
  if (dtype == AFFT_F32) ((float*)p)[idx] = v; else if (dtype == AFFT_F16) ((bf16_t*)p)[idx] = f2h(v); else ((bf16_t*)p)[idx] = f2bf(v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Second stage of every scalar reduction on the path (MSE, sum of squares, summed row losses): ONE workgroup adds n partial
// sums in an order that depends on n only - lane t takes t, t + 256, ... front to back, then a fixed tree - and does
// *out (+)= scale * total.  No float atomics anywhere: the same inputs give the same bits on every run.
static __global__ __launch_bounds__(256) void ordered_sum_kernel(const float* __restrict__ partials, int64_t n, float scale,
                                                                 float* __restrict__ out, int accumulate) {
  __shared__ float sh_[4];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += partials[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh_[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = ((sh_[0] + sh_[1]) + (sh_[2] + sh_[3])) * scale;
    *out = accumulate ? *out + t : t;
  }
}

// ---- counter-based dropout RNG: keep(idx) is a pure function of (key, element index), so backward regenerates
//      the forward mask instead of storing it. key is drawn per call site and step on the host.
__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ bool drop_keep(unsigned key, unsigned idx, unsigned thresh) {
  return mix32(mix32(idx) ^ key) >= thresh;   // P(keep) = 1 - thresh / 2^32
}
struct DropParams {
  unsigned thresh;      // element dropout: p * 2^32 (0 = off)
  unsigned key;
  float inv_keep;       // 1 / (1 - p)
  unsigned path_thresh; // DropPath per row group (0 = off)
  unsigned path_key;
  float path_inv_keep;
  int path_group;       // rows per sample (frame / clip)
  const unsigned* salt; // device word XOR-ed into the keys at kernel start (afft_set_dropout_salt), or NULL
};
// Process-wide device pointer to a 32-bit "salt" (afft_set_dropout_salt): a captured hipGraph replays the same kernel
// arguments every step, so what changes the masks from step to step has to live in device memory.
extern const unsigned* g_afft_drop_salt;
inline DropParams make_drop(const afft_dropout_t* d) {
  DropParams o = {0u, 0u, 1.0f, 0u, 0u, 1.0f, 1, nullptr};
  if (!d) return o;
  if (d->p > 0.f || d->path_p > 0.f) o.salt = g_afft_drop_salt;
  if (d->p > 0.f) {
    double t = (double)d->p * 4294967296.0;
    o.thresh = t >= 4294967295.0 ? 0xffffffffu : (unsigned)t;
    o.key = d->key;
    o.inv_keep = 1.0f / (1.0f - d->p);
  }
  if (d->path_p > 0.f) {
    double t = (double)d->path_p * 4294967296.0;
    o.path_thresh = t >= 4294967295.0 ? 0xffffffffu : (unsigned)t;
    o.path_key = d->path_key;
    o.path_inv_keep = 1.0f / (1.0f - d->path_p);
    o.path_group = d->path_group > 0 ? d->path_group : 1;
  }
  return o;
}
// call once at kernel start: folds the device salt into the keys
__device__ __forceinline__ DropParams with_salt(DropParams d) {
  if (d.salt) {
    const unsigned s = *d.salt;
    d.key ^= s;
    d.path_key ^= mix32(s ^ 0x5bd1e995u);
  }
  return d;
}
__device__ __forceinline__ float drop_row_scale(const DropParams& d, int m) {
  if (!d.path_thresh) return 1.0f;
  return drop_keep(d.path_key, (unsigned)(m / d.path_group), d.path_thresh) ? d.path_inv_keep : 0.0f;
}
__device__ __forceinline__ float drop_elem_scale(const DropParams& d, unsigned idx) {
  if (!d.thresh) return 1.0f;
  return drop_keep(d.key, idx, d.thresh) ? d.inv_keep : 0.0f;
}

// ---- activations (nn.GELU exact erf: models/transformerblock.py:119 ; HF gelu_new: GPT2MLP)
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_erf_f(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}
__device__ __forceinline__ float gelu_tanh_f(float x) {
  float u = 0.79788456080286536f * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.0f + tanhf(u));
}
__device__ __forceinline__ float dgelu_tanh_f(float x) {
  float x2 = x * x;
  float u = 0.79788456080286536f * (x + 0.044715f * x * x2);
  float t = tanhf(u);
  return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * 0.79788456080286536f * (1.0f + 3.0f * 0.044715f * x2);
}

// ---- the same activations for the epilogues of the bf16-operand GEMM kernels (flag AFFT_ACT_FAST on the activation code): a
//      library erff / tanhf is 45-80 VALU instructions per element with divergent branches, and an epilogue applies it to 128
//      elements per thread -- measured +64 us on the 177-us fc1 GEMM, +106 us on its data gradient, +140 us on the predictor's.
//      erf: Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7) on v_rcp_f32 / v_exp_f32; tanh(u) = 1 - 2 / (1 + exp(2u)).  Absolute
//      errors of ~2e-7, far inside what the bf16 operands of these kernels leave; the exact-fp32 GEMM keeps the library functions.
// Every rounding is pinned (explicit fma / mul / add, as in sgd_update): the 4-wide and 8-wide epilogues, packed or scalar
// instructions, composite or call-by-call launches all give bit-identical results.
__device__ __forceinline__ float erf_as7126(float z, float e /* = exp(-z*z) */) {
  const float t = __builtin_amdgcn_rcpf(__fmaf_rn(0.3275911f, fabsf(z), 1.0f));
  float p = __fmaf_rn(1.061405429f, t, -1.453152027f);
  p = __fmaf_rn(p, t, 1.421413741f);
  p = __fmaf_rn(p, t, -0.284496736f);
  p = __fmaf_rn(p, t, 0.254829592f);
  return copysignf(__fmaf_rn(-__fmul_rn(p, t), e, 1.0f), z);
}
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = __fmul_rn(x, 0.70710678118654752f);
  const float hx = __fmul_rn(0.5f, x);
  return __fmaf_rn(hx, erf_as7126(z, __expf(-__fmul_rn(z, z))), hx);
}
__device__ __forceinline__ float dgelu_erf_fast(float x) {
  const float z = __fmul_rn(x, 0.70710678118654752f);
  const float e = __expf(-__fmul_rn(z, z));               // exp(-x^2 / 2): shared by erf and the density term
  return __fmaf_rn(__fmul_rn(x, 0.39894228040143268f), e, __fmaf_rn(0.5f, erf_as7126(z, e), 0.5f));
}
__device__ __forceinline__ float tanh_fast(float u) {
  return __fmaf_rn(-2.0f, __builtin_amdgcn_rcpf(__fadd_rn(1.0f, __expf(__fmul_rn(2.0f, u)))), 1.0f);
}
__device__ __forceinline__ float gelu_new_arg(float x, float x2) {   // sqrt(2/pi) (x + 0.044715 x^3)
  return __fmul_rn(0.79788456080286536f, __fmaf_rn(__fmul_rn(0.044715f, x2), x, x));
}
__device__ __forceinline__ float gelu_tanh_fast(float x) {
  const float hx = __fmul_rn(0.5f, x);
  return __fmaf_rn(hx, tanh_fast(gelu_new_arg(x, __fmul_rn(x, x))), hx);
}
__device__ __forceinline__ float dgelu_tanh_fast(float x) {
  const float x2 = __fmul_rn(x, x);
  const float t = tanh_fast(gelu_new_arg(x, x2));
  const float sech2 = __fmaf_rn(-t, t, 1.0f);
  const float du = __fmul_rn(0.79788456080286536f, __fmaf_rn(3.0f * 0.044715f, x2, 1.0f));
  return __fmaf_rn(__fmul_rn(__fmul_rn(0.5f, x), sech2), du, __fmaf_rn(0.5f, t, 0.5f));
}
constexpr int AFFT_ACT_FAST = 0x100;

// ---- momentum-SGD update of one element (torch.optim.SGD with dampening 0), shared by the stand-alone update kernels and the
//      fused GEMM epilogue: every rounding is pinned (explicit fma / mul), so both give bit-identical parameters.
//      flags: AFFT_SGD_FIRST_STEP (the momentum buffer starts as the gradient), AFFT_SGD_PLAIN_MOMENTUM (nesterov = False:
//      p -= lr * buf instead of p -= lr * (g + mom * buf)); include/afft_hip.h.
__device__ __forceinline__ void sgd_update(float& p, float& buf, float g, float lr, float mom, float wd, float gscale, int flags) {
  const float gg = __fmaf_rn(g, gscale, __fmul_rn(wd, p));
  const float bb = (flags & AFFT_SGD_FIRST_STEP) ? gg : __fmaf_rn(mom, buf, gg);
  buf = bb;
  p = __fmaf_rn(-lr, (flags & AFFT_SGD_PLAIN_MOMENTUM) ? bb : __fmaf_rn(mom, bb, gg), p);
}
struct SgdEpi { float* p; float* buf; bf16_t* p16; float lr, mom, wd, gscale; int first; bf16_t* p16k; bf16_t* p16h; unsigned char* p8; const float* ok; };   // p16h: fp16 image (row-major, like p16); p8: e4m3(2^8 p) bytes
// element offset of the 8-element fragment that holds W[m][n .. n + 7] (n % 8 == 0) in the fragment-packed image of a [rows, ld] weight
// (afft_pack_weight, include/afft_hip.h)
__device__ __forceinline__ int64_t packed_frag(int m, int n, int64_t ld) {
  return (((int64_t)(m >> 4) * (ld >> 5) + (n >> 5)) * 64 + (m & 15) + 16 * ((n >> 3) & 3)) * 8;
}

// ---- GEMM epilogue shared by the bf16 fast path and the fp32 path ------------------------------
struct EpiParams {
  int M, N;
  float alpha;
  const float* bias;
  int act;
  const void* aux; int64_t ldaux; int aux_dtype;
  void* pre; int64_t ldpre; int pre_dtype;
  const float* rowscale;
  const float* residual; int64_t ldres;
  int accumulate;
  void* out; int64_t ldo; int out_dtype;
  int64_t out_lo;   // != 0 (out_dtype AFFT_F16): out is the HI plane of a two-plane fp16 split, the lo plane sits out_lo elements behind it
  unsigned char* out_lo8;   // != NULL (out_dtype AFFT_F16): the lo part goes out as e4m3(2^11 (v - hi)) bytes, row pitch ldo bytes
  void* out2; int64_t ldo2; int out2_dtype;
  int vec4;  // host-verified: every ld % 4 == 0 and bases 16-byte aligned -> 4-wide accesses legal
  int vec8;  // host-verified: every ld % 8 == 0 and bases 16-byte aligned -> 8-wide accesses legal (16-B bf16 stores)
  DropParams drop;  // dropout on the (activated) GEMM output + DropPath row scale, before the residual add
  SgdEpi sgd;       // sgd.p != NULL: the result is a weight gradient consumed by the update of p (layout of out); nothing is stored to out
};

__device__ __forceinline__ void store4(void* base, int64_t idx, int dtype, const float (&v)[4]) {
  if (dtype == AFFT_F32) {
    *(float4*)((float*)base + idx) = make_float4(v[0], v[1], v[2], v[3]);
  } else if (dtype == AFFT_F16) {
    f16x4_v h;
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = (_Float16)v[r];
    *(f16x4_v*)((bf16_t*)base + idx) = h;
  } else {
    uint2 u;
    u.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
    u.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
    *(uint2*)((bf16_t*)base + idx) = u;
  }
}
__device__ __forceinline__ void load4(const void* base, int64_t idx, int dtype, float (&v)[4]) {
  if (dtype == AFFT_F32) {
    float4 t = *(const float4*)((const float*)base + idx);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else if (dtype == AFFT_F16) {
    const f16x4_v h = *(const f16x4_v*)((const bf16_t*)base + idx);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (float)h[r];
  } else {
    uint2 u = *(const uint2*)((const bf16_t*)base + idx);
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
  }
}

// two-plane fp16 split of N values: hi = fp16(v) at idx, lo = fp16(v - hi) lo_off elements behind it (the A operand of the
// fp16 two-pass GEMM, afft_gemm_t.split3 = 2, written by the producing kernel instead of a separate split pass)
template <int N>
__device__ __forceinline__ void store_split(void* base, int64_t idx, int64_t lo_off, const float (&v)[N]) {
  typedef __attribute__((ext_vector_type(N))) _Float16 hN;
  hN h, l;
#pragma unroll
  for (int r = 0; r < N; ++r) { h[r] = (_Float16)v[r]; l[r] = (_Float16)(v[r] - (float)h[r]); }
  *(hN*)((bf16_t*)base + idx) = h;
  *(hN*)((bf16_t*)base + idx + lo_off) = l;
}

// v_cvt_pk_fp8_f32 does NOT saturate: a value beyond e4m3's 448 becomes the NaN code (found the hard way: a weight of 2.0 times the
// image scale 2^8 turned every logit into NaN).  Everything headed for an e4m3 byte is clamped to +-448 first (NaN stays NaN).
__device__ __forceinline__ float e4m3_clamp(float x) { return __builtin_amdgcn_fmed3f(x, -448.0f, 448.0f); }
template <bool HI>
__device__ __forceinline__ int pk_e4m3_(float a, float b, int old) { return __builtin_amdgcn_cvt_pk_fp8_f32(e4m3_clamp(a), e4m3_clamp(b), old, HI); }
#define pk_e4m3(a, b, old, hi) pk_e4m3_<hi>(a, b, old)
template <int N>      // dst[idx .. idx + N) = e4m3(scale * v): the weight byte image (scale 2^8)
__device__ __forceinline__ void store_e4m3(unsigned char* dst, int64_t idx, const float (&v)[N], float scale) {
  static_assert(N == 4 || N == 8, "4 or 8 values");
  int w0 = 0;
  w0 = pk_e4m3(v[0] * scale, v[1] * scale, w0, false);
  w0 = pk_e4m3(v[2] * scale, v[3] * scale, w0, true);
  if constexpr (N == 4) {
    *(int*)(dst + idx) = w0;
  } else {
    int w1 = 0;
    w1 = pk_e4m3(v[4] * scale, v[5] * scale, w1, false);
    w1 = pk_e4m3(v[6] * scale, v[7] * scale, w1, true);
    *(int2*)(dst + idx) = make_int2(w0, w1);
  }
}
// hi = fp16(v) at idx of `base`; lo = e4m3(2^11 (v - hi)) at byte idx of `lo8` (the A operand of the fp16 + fp8 GEMM, afft_gemm_t.split3 = 3).
// v_cvt_pk_fp8_f32 is the OCP e4m3fn conversion on gfx950 (saturating); |v - hi| <= 2^-11 |v| ... 2^11 (v - hi) <= |v| / 2.
template <int N>
__device__ __forceinline__ void store_split8(void* base, unsigned char* lo8, int64_t idx, const float (&v)[N]) {
  static_assert(N == 4 || N == 8, "4 or 8 values");
  typedef __attribute__((ext_vector_type(N))) _Float16 hN;
  hN h;
  float l[N];
#pragma unroll
  for (int r = 0; r < N; ++r) { h[r] = (_Float16)v[r]; l[r] = (v[r] - (float)h[r]) * 2048.0f; }
  *(hN*)((bf16_t*)base + idx) = h;
  int w0 = 0;
  w0 = pk_e4m3(l[0], l[1], w0, false);
  w0 = pk_e4m3(l[2], l[3], w0, true);
  if constexpr (N == 4) {
    *(int*)(lo8 + idx) = w0;
  } else {
    int w1 = 0;
    w1 = pk_e4m3(l[4], l[5], w1, false);
    w1 = pk_e4m3(l[6], l[7], w1, true);
    *(int2*)(lo8 + idx) = make_int2(w0, w1);
  }
}
__device__ __forceinline__ unsigned char f2e4m3(float f) { return (unsigned char)(pk_e4m3(f, 0.f, 0, false) & 0xff); }

__host__ __device__ __forceinline__ bool act_needs_aux(int act) {
  act &= 0xff;
  return act == AFFT_ACT_DGELU_ERF || act == AFFT_ACT_DGELU_TANH || act == AFFT_ACT_SIGMOID_GATE;
}
__device__ __forceinline__ float apply_act(int act, float v, float aux) {
  switch (act) {
    case AFFT_ACT_GELU_ERF: return gelu_erf_f(v);
    case AFFT_ACT_GELU_TANH: return gelu_tanh_f(v);
    case AFFT_ACT_DGELU_ERF: return v * dgelu_erf_f(aux);
    case AFFT_ACT_DGELU_TANH: return v * dgelu_tanh_f(aux);
    case AFFT_ACT_GELU_ERF | AFFT_ACT_FAST: return gelu_erf_fast(v);
    case AFFT_ACT_GELU_TANH | AFFT_ACT_FAST: return gelu_tanh_fast(v);
    case AFFT_ACT_DGELU_ERF | AFFT_ACT_FAST: return __fmul_rn(v, dgelu_erf_fast(aux));
    case AFFT_ACT_DGELU_TANH | AFFT_ACT_FAST: return __fmul_rn(v, dgelu_tanh_fast(aux));
    case AFFT_ACT_RELU | AFFT_ACT_FAST: return v > 0.f ? v : 0.f;
    case AFFT_ACT_SIGMOID_GATE | AFFT_ACT_FAST: return aux / (1.0f + __expf(-v));
    case AFFT_ACT_RELU: return v > 0.f ? v : 0.f;
    case AFFT_ACT_SIGMOID_GATE: return aux / (1.0f + __expf(-v));
    default: return v;
  }
}

// the same for N elements with the switch taken ONCE (the compiler does not unswitch the unrolled element loop by itself: it
// left a scalar compare-and-branch chain per element, about as expensive as the activation): one straight-line block per
// activation, whose N independent chains also hide the quarter-rate v_exp_f32 / v_rcp_f32 behind each other
template <int N>
__device__ __forceinline__ void apply_act_n(int act, float (&v)[N], const float (&a)[N]) {
#define AFFT_ACT_CASE(code, expr)            \
  case code:                                 \
    _Pragma("unroll") for (int r = 0; r < N; ++r) v[r] = (expr); \
    break;
  switch (act) {
    AFFT_ACT_CASE(AFFT_ACT_GELU_ERF, gelu_erf_f(v[r]))
    AFFT_ACT_CASE(AFFT_ACT_GELU_TANH, gelu_tanh_f(v[r]))
    AFFT_ACT_CASE(AFFT_ACT_DGELU_ERF, v[r] * dgelu_erf_f(a[r]))
    AFFT_ACT_CASE(AFFT_ACT_DGELU_TANH, v[r] * dgelu_tanh_f(a[r]))
    AFFT_ACT_CASE(AFFT_ACT_GELU_ERF | AFFT_ACT_FAST, gelu_erf_fast(v[r]))
    AFFT_ACT_CASE(AFFT_ACT_GELU_TANH | AFFT_ACT_FAST, gelu_tanh_fast(v[r]))
    AFFT_ACT_CASE(AFFT_ACT_DGELU_ERF | AFFT_ACT_FAST, __fmul_rn(v[r], dgelu_erf_fast(a[r])))
    AFFT_ACT_CASE(AFFT_ACT_DGELU_TANH | AFFT_ACT_FAST, __fmul_rn(v[r], dgelu_tanh_fast(a[r])))
    case AFFT_ACT_RELU | AFFT_ACT_FAST:
    AFFT_ACT_CASE(AFFT_ACT_RELU, v[r] > 0.f ? v[r] : 0.f)
    case AFFT_ACT_SIGMOID_GATE | AFFT_ACT_FAST:
    AFFT_ACT_CASE(AFFT_ACT_SIGMOID_GATE, a[r] / (1.0f + __expf(-v[r])))
    default: break;
  }
#undef AFFT_ACT_CASE
}

// v[0..3] = accumulators for C[m, n..n+3]
// dp = with_salt(e.drop), computed once per thread by the caller
__device__ __forceinline__ void epilogue4(const EpiParams& e, const DropParams& dp, int m, int n, float (&v)[4]) {
  if (m >= e.M || n >= e.N) return;
  const bool full = e.vec4 && (n + 3 < e.N);
  if (e.sgd.p) {     // fused optimizer: v is the gradient of p[m, n .. n+3]
    if (e.sgd.ok && *e.sgd.ok == 0.f) return;      // non-finite loss: the step is a no-op (afft_sgd_fused_t.ok)
    const int64_t idx = (int64_t)m * e.ldo + n;
    if (full) {
      float pv[4], bv[4];
      load4(e.sgd.p, idx, AFFT_F32, pv);
      load4(e.sgd.buf, idx, AFFT_F32, bv);
#pragma unroll
      for (int r = 0; r < 4; ++r) sgd_update(pv[r], bv[r], v[r] * e.alpha, e.sgd.lr, e.sgd.mom, e.sgd.wd, e.sgd.gscale, e.sgd.first);
      store4(e.sgd.p, idx, AFFT_F32, pv);
      store4(e.sgd.buf, idx, AFFT_F32, bv);
      if (e.sgd.p16) store4(e.sgd.p16, idx, AFFT_BF16, pv);
      if (e.sgd.p16h) store4(e.sgd.p16h, idx, AFFT_F16, pv);
      if (e.sgd.p8) store_e4m3<4>(e.sgd.p8, idx, pv, 256.0f);
      if (e.sgd.p16k) store4(e.sgd.p16k, packed_frag(m, n & ~7, e.ldo) + (n & 7), AFFT_BF16, pv);
    } else {
      for (int r = 0; r < 4 && n + r < e.N; ++r) {
        float pv = e.sgd.p[idx + r], bv = e.sgd.buf[idx + r];
        sgd_update(pv, bv, v[r] * e.alpha, e.sgd.lr, e.sgd.mom, e.sgd.wd, e.sgd.gscale, e.sgd.first);
        e.sgd.p[idx + r] = pv; e.sgd.buf[idx + r] = bv;
        if (e.sgd.p16) e.sgd.p16[idx + r] = f2bf(pv);
        if (e.sgd.p16h) e.sgd.p16h[idx + r] = f2h(pv);
        if (e.sgd.p8) e.sgd.p8[idx + r] = f2e4m3(pv * 256.0f);
        if (e.sgd.p16k) e.sgd.p16k[packed_frag(m, (n + r) & ~7, e.ldo) + ((n + r) & 7)] = f2bf(pv);
      }
    }
    return;
  }
  const float rs = (e.rowscale ? e.rowscale[m] : 1.0f) * drop_row_scale(dp, m);
  const bool scaled = e.rowscale || dp.path_thresh;
  if (full) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= e.alpha;
    if (e.bias) {
      float4 b = *(const float4*)(e.bias + n);
      v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (e.pre) store4(e.pre, (int64_t)m * e.ldpre + n, e.pre_dtype, v);
    if (e.act != AFFT_ACT_NONE) {
      float a[4] = {0.f, 0.f, 0.f, 0.f};
      if (act_needs_aux(e.act)) load4(e.aux, (int64_t)m * e.ldaux + n, e.aux_dtype, a);
      apply_act_n<4>(e.act, v, a);
    }
    if (dp.thresh) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= drop_elem_scale(dp, (unsigned)m * (unsigned)e.N + (unsigned)(n + r));
    }
    if (scaled) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= rs;
    }
    if (e.residual) {
      float4 t = *(const float4*)(e.residual + (int64_t)m * e.ldres + n);
      v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
    }
    if (e.accumulate) {
      float4 t = *(const float4*)((const float*)e.out + (int64_t)m * e.ldo + n);
      v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
    }
    if (e.out_lo8) store_split8<4>(e.out, e.out_lo8, (int64_t)m * e.ldo + n, v);
    else if (e.out_lo) store_split<4>(e.out, (int64_t)m * e.ldo + n, e.out_lo, v);
    else store4(e.out, (int64_t)m * e.ldo + n, e.out_dtype, v);
    if (e.out2) store4(e.out2, (int64_t)m * e.ldo2 + n, e.out2_dtype, v);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int nn = n + r;
      if (nn >= e.N) break;
      float x = v[r] * e.alpha;
      if (e.bias) x += e.bias[nn];
      if (e.pre) st_any(e.pre, (int64_t)m * e.ldpre + nn, e.pre_dtype, x);
      float a = 0.f;
      if (act_needs_aux(e.act)) a = ld_any(e.aux, (int64_t)m * e.ldaux + nn, e.aux_dtype);
      x = apply_act(e.act, x, a);
      x *= drop_elem_scale(dp, (unsigned)m * (unsigned)e.N + (unsigned)nn);
      x *= rs;
      if (e.residual) x += e.residual[(int64_t)m * e.ldres + nn];
      if (e.accumulate) x += ((const float*)e.out)[(int64_t)m * e.ldo + nn];
      st_any(e.out, (int64_t)m * e.ldo + nn, e.out_dtype, x);
      if (e.out_lo) ((bf16_t*)e.out)[(int64_t)m * e.ldo + nn + e.out_lo] = f2h(x - h2f(f2h(x)));
      if (e.out_lo8) e.out_lo8[(int64_t)m * e.ldo + nn] = f2e4m3((x - h2f(f2h(x))) * 2048.0f);
      if (e.out2) st_any(e.out2, (int64_t)m * e.ldo2 + nn, e.out2_dtype, x);
    }
  }
}

// 8-wide accesses: one 16-byte access per bf16 tensor (half the store instructions of two 4-wide calls -- the
// epilogue of a 256x256 tile is bound by store issue and by every CU writing at once), two per fp32 tensor.
__device__ __forceinline__ void store8(void* base, int64_t idx, int dtype, const float (&v)[8]) {
  if (dtype == AFFT_F32) {
    *(float4*)((float*)base + idx) = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)((float*)base + idx + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else if (dtype == AFFT_F16) {
    f16x8_v h;
#pragma unroll
    for (int r = 0; r < 8; ++r) h[r] = (_Float16)v[r];
    *(f16x8_v*)((bf16_t*)base + idx) = h;
  } else {
    uint4 u;
    u.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
    u.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
    u.z = (unsigned)f2bf(v[4]) | ((unsigned)f2bf(v[5]) << 16);
    u.w = (unsigned)f2bf(v[6]) | ((unsigned)f2bf(v[7]) << 16);
    *(uint4*)((bf16_t*)base + idx) = u;
  }
}
__device__ __forceinline__ void load8(const void* base, int64_t idx, int dtype, float (&v)[8]) {
  if (dtype == AFFT_F32) {
    const float4 a = *(const float4*)((const float*)base + idx), b = *(const float4*)((const float*)base + idx + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else if (dtype == AFFT_F16) {
    const f16x8_v h = *(const f16x8_v*)((const bf16_t*)base + idx);
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = (float)h[r];
  } else {
    const uint4 u = *(const uint4*)((const bf16_t*)base + idx);
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
  }
}

// v[0..7] = accumulators for C[m, n..n+7], n % 8 == 0
__device__ __forceinline__ void epilogue8(const EpiParams& e, const DropParams& dp, int m, int n, float (&v)[8]) {
  if (m >= e.M || n >= e.N) return;
  if (!(e.vec8 && n + 7 < e.N)) {
    float lo[4] = {v[0], v[1], v[2], v[3]}, hi[4] = {v[4], v[5], v[6], v[7]};
    epilogue4(e, dp, m, n, lo);
    epilogue4(e, dp, m, n + 4, hi);
    return;
  }
  if (e.sgd.p) {     // fused optimizer, 8 parameters per lane: 16-byte accesses throughout
    if (e.sgd.ok && *e.sgd.ok == 0.f) return;      // non-finite loss: the step is a no-op (afft_sgd_fused_t.ok)
    const int64_t idx = (int64_t)m * e.ldo + n;
    float pv[8], bv[8];
    load8(e.sgd.p, idx, AFFT_F32, pv);
    load8(e.sgd.buf, idx, AFFT_F32, bv);
#pragma unroll
    for (int r = 0; r < 8; ++r) sgd_update(pv[r], bv[r], v[r] * e.alpha, e.sgd.lr, e.sgd.mom, e.sgd.wd, e.sgd.gscale, e.sgd.first);
    store8(e.sgd.p, idx, AFFT_F32, pv);
    store8(e.sgd.buf, idx, AFFT_F32, bv);
    if (e.sgd.p16) store8(e.sgd.p16, idx, AFFT_BF16, pv);
    if (e.sgd.p16h) store8(e.sgd.p16h, idx, AFFT_F16, pv);
    if (e.sgd.p8) store_e4m3<8>(e.sgd.p8, idx, pv, 256.0f);
    if (e.sgd.p16k) store8(e.sgd.p16k, packed_frag(m, n, e.ldo), AFFT_BF16, pv);
    return;
  }
  const float rs = (e.rowscale ? e.rowscale[m] : 1.0f) * drop_row_scale(dp, m);
  const bool scaled = e.rowscale || dp.path_thresh;
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] *= e.alpha;
  if (e.bias) {
    float b[8];
    load8(e.bias, n, AFFT_F32, b);
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] += b[r];
  }
  if (e.pre) store8(e.pre, (int64_t)m * e.ldpre + n, e.pre_dtype, v);
  if (e.act != AFFT_ACT_NONE) {
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (act_needs_aux(e.act)) load8(e.aux, (int64_t)m * e.ldaux + n, e.aux_dtype, a);
    apply_act_n<8>(e.act, v, a);
  }
  if (dp.thresh) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] *= drop_elem_scale(dp, (unsigned)m * (unsigned)e.N + (unsigned)(n + r));
  }
  if (scaled) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] *= rs;
  }
  if (e.residual) {
    float t[8];
    load8(e.residual, (int64_t)m * e.ldres + n, AFFT_F32, t);
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] += t[r];
  }
  if (e.accumulate) {
    float t[8];
    load8(e.out, (int64_t)m * e.ldo + n, AFFT_F32, t);
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] += t[r];
  }
  if (e.out_lo8) store_split8<8>(e.out, e.out_lo8, (int64_t)m * e.ldo + n, v);
  else if (e.out_lo) store_split<8>(e.out, (int64_t)m * e.ldo + n, e.out_lo, v);
  else store8(e.out, (int64_t)m * e.ldo + n, e.out_dtype, v);
  if (e.out2) store8(e.out2, (int64_t)m * e.ldo2 + n, e.out2_dtype, v);
}
