// LayerNorm forward / backward (HBM-bound; one wave per row, 16-byte accesses, wave-level reductions).
// nn.LayerNorm semantics (biased variance, eps inside the sqrt): models/fusion.py:281,362,
// models/transformerblock.py:122,127,150-152 (eps 1e-6) and HF GPT-2 ln_1/ln_2/ln_f (eps 1e-5).
#include <type_traits>

#include "common.h"

namespace {

constexpr int LN_WAVES = 4;  // rows per workgroup step

template <int NV>  // float4 chunks per lane: d <= 256*NV
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ w, const float* __restrict__ b,
                                                     float eps, int rows, int d, void* __restrict__ y, int64_t ldy,
                                                     int y_dtype, float* __restrict__ mean, float* __restrict__ rstd,
                                                     int64_t y_lo, bf16_t* __restrict__ y2, int64_t ldy2, unsigned char* __restrict__ y_lo8) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * LN_WAVES + wave;
  if (row >= rows) return;
  const float* xr = x + (int64_t)row * ldx;
  const int nq = d >> 2;  // float4 count (d % 4 == 0 checked on the host)
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NV; ++t) {
    const int q = lane + 64 * t;
    v[t] = q < nq ? *(const float4*)(xr + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[t].x + v[t].y) + (v[t].z + v[t].w);
  }
  const float mu = wave_sum(s) / (float)d;
  float ss = 0.f;
#pragma unroll
  for (int t = 0; t < NV; ++t) {
    const int q = lane + 64 * t;
    if (q < nq) {
      const float a = v[t].x - mu, bq = v[t].y - mu, c = v[t].z - mu, e = v[t].w - mu;
      ss += (a * a + bq * bq) + (c * c + e * e);
    }
  }
  const float rs = rsqrtf(wave_sum(ss) / (float)d + eps);
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
#pragma unroll
  for (int t = 0; t < NV; ++t) {
    const int q = lane + 64 * t;
    if (q < nq) {
      float o[4] = {(v[t].x - mu) * rs, (v[t].y - mu) * rs, (v[t].z - mu) * rs, (v[t].w - mu) * rs};
      if (w) {
        const float4 ww = *(const float4*)(w + 4 * q);
        o[0] *= ww.x; o[1] *= ww.y; o[2] *= ww.z; o[3] *= ww.w;
      }
      if (b) {
        const float4 bb = *(const float4*)(b + 4 * q);
        o[0] += bb.x; o[1] += bb.y; o[2] += bb.z; o[3] += bb.w;
      }
      if (y_lo8) store_split8<4>(y, y_lo8, (int64_t)row * ldy + 4 * q, o);  // fp16 hi + e4m3 lo (the fp8 lo pass)
      else if (y_lo) store_split<4>(y, (int64_t)row * ldy + 4 * q, y_lo, o);     // fp16x2 operand planes (afft_layernorm_fwd_split)
      else store4(y, (int64_t)row * ldy + 4 * q, y_dtype, o);
      if (y2) store4(y2, (int64_t)row * ldy2 + 4 * q, AFFT_BF16, o);
    }
  }
}

// backward: a workgroup is 16 waves = 4 row groups x 4 column slices.  Wave (rg, cs) owns the columns of slice cs
// (d/4 wide) of rows base + rg, base += 4 * gridDim.x: a quarter of a row per wave keeps the per-lane state small
// (16 waves per CU and more stay resident, their loads overlap), the two per-row sums are combined over the 4 slices
// through a double-buffered LDS cell with one barrier per row step, and the dw/db column partials stay in registers
// until the end, when the 4 row groups add them into one [2][d] LDS slab in a fixed order (deterministic).
constexpr int LNB_RG = 4, LNB_CS = 4;

template <int NV, bool DY32>  // float4 chunks per lane inside a slice: d/4 <= LNB_CS * 64 * NV; DY32: dy is fp32 (else bf16)
__global__ __launch_bounds__(1024) void ln_bwd_kernel(const void* __restrict__ dy, int64_t lddy,
                                                      const float* __restrict__ x, int64_t ldx,
                                                      const float* __restrict__ w, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, int rows, int d,
                                                      const float* __restrict__ dx_in, float* __restrict__ dx_out,
                                                      int64_t lddx, bf16_t* __restrict__ dx_bf16,
                                                      const DropParams drop_, int nslab, float* __restrict__ partial,
                                                      int64_t lddx_in, int in_take) {
  const DropParams drop = with_salt(drop_);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* slab = (float*)smem;                      // [nslab][d]: dw, db (, column sums of the masked copy)
  float* cell = slab + nslab * d;                  // [2 buffers][LNB_RG][LNB_CS][2]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rg = wave >> 2, cs = wave & 3;
  const int nq = d >> 2, qs = (nq + LNB_CS - 1) / LNB_CS;
  const int q0 = cs * qs, q1 = min(nq, q0 + qs);
  float4 pw[NV], pb[NV], pc[NV], ww[NV];
#pragma unroll
  for (int t = 0; t < NV; ++t) {
    pw[t] = pb[t] = pc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int q = q0 + lane + 64 * t;
    ww[t] = (w && q < q1) ? *(const float4*)(w + 4 * q) : make_float4(1.f, 1.f, 1.f, 1.f);
  }
  const float inv_d = 1.0f / (float)d;
  int buf = 0;
  // Software pipeline: the loads of row step i + 1 (dy, x, the incoming dx, the row's statistics) are issued BEFORE step i is
  // computed, so they are in flight across its reductions, its barrier and its stores -- a workgroup runs only ~5 row steps at
  // 5120 rows, each of which used to expose one full HBM round trip (2.3 TB/s alone).
  struct Pre { float4 x; typename std::conditional<DY32, uint4, uint2>::type dy; };
  Pre nxt[NV];
  float nmu = 0.f, nrs = 0.f;
  auto prefetch = [&](int row) {
    if (row >= rows) return;
    nmu = mean[row];
    nrs = rstd[row];
    const float* xr = x + (int64_t)row * ldx;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
      const int q = q0 + lane + 64 * t;
      if (q < q1) {
        nxt[t].x = *(const float4*)(xr + 4 * q);
        if constexpr (DY32) nxt[t].dy = *(const uint4*)((const float*)dy + (int64_t)row * lddy + 4 * q);
        else nxt[t].dy = *(const uint2*)((const bf16_t*)dy + (int64_t)row * lddy + 4 * q);
      }
    }
  };
  constexpr bool PIPE = NV <= 2;      // d <= 2048: the second register set fits (no spills at 16 waves per CU); wider rows load in place
  if (PIPE) prefetch(blockIdx.x * LNB_RG + rg);
  for (int base = blockIdx.x * LNB_RG; base < rows; base += gridDim.x * LNB_RG, buf ^= 1) {
    const int row = base + rg;
    const bool live = row < rows;
    float g[NV][4], xh[NV][4];
    float4 din[NV];
    float c1 = 0.f, c2 = 0.f, rs = 0.f;
    if (!PIPE) prefetch(row);
    Pre cur[NV];
#pragma unroll
    for (int t = 0; t < NV; ++t) cur[t] = nxt[t];
    const float mu = nmu;
    rs = nrs;
    if (PIPE) prefetch(row + gridDim.x * LNB_RG);       // the next row step of this wave: in flight from here on
    // in_take > 1: the incoming dx is [rows / in_take, d] and belongs to rows 0, in_take, ..: the others take none
    const bool has_in = dx_in && (in_take <= 1 || row % in_take == 0);
    if (live && dx_in) {                      // this step's incoming dx is only needed behind the barrier: its latency hides under the reductions
      const int64_t in_row = in_take > 1 ? row / in_take : row;
#pragma unroll
      for (int t = 0; t < NV; ++t) {
        const int q = q0 + lane + 64 * t;
        din[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < q1 && has_in) din[t] = *(const float4*)(dx_in + in_row * lddx_in + 4 * q);
      }
    }
    if (live) {
#pragma unroll
      for (int t = 0; t < NV; ++t) {
        const int q = q0 + lane + 64 * t;
        if (q < q1) {
          float dyv[4];
          if constexpr (DY32) {
            dyv[0] = __uint_as_float(cur[t].dy.x); dyv[1] = __uint_as_float(cur[t].dy.y);
            dyv[2] = __uint_as_float(cur[t].dy.z); dyv[3] = __uint_as_float(cur[t].dy.w);
          } else {
            dyv[0] = __uint_as_float(cur[t].dy.x << 16); dyv[1] = __uint_as_float(cur[t].dy.x & 0xffff0000u);
            dyv[2] = __uint_as_float(cur[t].dy.y << 16); dyv[3] = __uint_as_float(cur[t].dy.y & 0xffff0000u);
          }
          const float4 xv = cur[t].x;
          xh[t][0] = (xv.x - mu) * rs; xh[t][1] = (xv.y - mu) * rs; xh[t][2] = (xv.z - mu) * rs; xh[t][3] = (xv.w - mu) * rs;
          g[t][0] = dyv[0] * ww[t].x; g[t][1] = dyv[1] * ww[t].y; g[t][2] = dyv[2] * ww[t].z; g[t][3] = dyv[3] * ww[t].w;
          pw[t].x += dyv[0] * xh[t][0]; pw[t].y += dyv[1] * xh[t][1]; pw[t].z += dyv[2] * xh[t][2]; pw[t].w += dyv[3] * xh[t][3];
          pb[t].x += dyv[0]; pb[t].y += dyv[1]; pb[t].z += dyv[2]; pb[t].w += dyv[3];
#pragma unroll
          for (int r = 0; r < 4; ++r) { c1 += g[t][r]; c2 += g[t][r] * xh[t][r]; }
        }
      }
    }
    c1 = wave_sum(c1);
    c2 = wave_sum(c2);
    float* my = cell + ((buf * LNB_RG + rg) * LNB_CS) * 2;
    if (lane == 0) { my[cs * 2] = c1; my[cs * 2 + 1] = c2; }
    __syncthreads();
    c1 = ((my[0] + my[2]) + (my[4] + my[6])) * inv_d;
    c2 = ((my[1] + my[3]) + (my[5] + my[7])) * inv_d;
    if (live) {
#pragma unroll
      for (int t = 0; t < NV; ++t) {
        const int q = q0 + lane + 64 * t;
        if (q < q1) {
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = rs * (g[t][r] - c1 - xh[t][r] * c2);
          const int64_t idx = (int64_t)row * lddx + 4 * q;
          if (dx_in) { o[0] += din[t].x; o[1] += din[t].y; o[2] += din[t].z; o[3] += din[t].w; }
          *(float4*)(dx_out + idx) = make_float4(o[0], o[1], o[2], o[3]);
          if (dx_bf16 || nslab == 3) {   // the copy the upstream GEMMs consume: their output-dropout mask replayed
            if (drop.thresh || drop.path_thresh) {
              const float prs = drop_row_scale(drop, row);
#pragma unroll
              for (int r = 0; r < 4; ++r) o[r] *= prs * drop_elem_scale(drop, (unsigned)row * (unsigned)d + (unsigned)(4 * q + r));
            }
            if (dx_bf16) store4(dx_bf16, idx, AFFT_BF16, o);
            pc[t].x += o[0]; pc[t].y += o[1]; pc[t].z += o[2]; pc[t].w += o[3];
          }
        }
      }
    }
  }
  // the 4 row groups add their column partials into the slab one after the other
  for (int r = 0; r < LNB_RG; ++r) {
    if (rg == r) {
#pragma unroll
      for (int t = 0; t < NV; ++t) {
        const int q = q0 + lane + 64 * t;
        if (q < q1) {
          float4 a = pw[t], c = pb[t], e = pc[t];
          if (r) {
            const float4 sa = *(const float4*)(slab + 4 * q), sc = *(const float4*)(slab + d + 4 * q);
            a.x += sa.x; a.y += sa.y; a.z += sa.z; a.w += sa.w;
            c.x += sc.x; c.y += sc.y; c.z += sc.z; c.w += sc.w;
            if (nslab == 3) {
              const float4 se = *(const float4*)(slab + 2 * d + 4 * q);
              e.x += se.x; e.y += se.y; e.z += se.z; e.w += se.w;
            }
          }
          *(float4*)(slab + 4 * q) = a;
          *(float4*)(slab + d + 4 * q) = c;
          if (nslab == 3) *(float4*)(slab + 2 * d + 4 * q) = e;
        }
      }
    }
    __syncthreads();
  }
  for (int c = threadIdx.x; c < nslab * d; c += 1024) partial[(int64_t)blockIdx.x * nslab * d + c] = slab[c];
}

// 64 columns x 16 part-lanes per workgroup: coalesced 256-B reads, 16 loads in flight per column, fixed summation order
// (deterministic)
constexpr int LNR_PL = 16;
__global__ __launch_bounds__(64 * LNR_PL) void ln_bwd_reduce_kernel(const float* __restrict__ partial, int nparts, int d, int nslab,
                                                                   float* __restrict__ dw, float* __restrict__ db, int accumulate,
                                                                   float* __restrict__ dcol, int dcol_accumulate) {
  __shared__ float sh[LNR_PL][64];
  const int col = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + col;
  float s = 0.f;
  if (c < nslab * d)
    for (int p = pl; p < nparts; p += LNR_PL) s += partial[(int64_t)p * nslab * d + c];
  sh[pl][col] = s;
  __syncthreads();
  if (pl == 0 && c < nslab * d) {
    s = 0.f;
#pragma unroll
    for (int k = 0; k < LNR_PL; k += 4) s += (sh[k][col] + sh[k + 1][col]) + (sh[k + 2][col] + sh[k + 3][col]);
    const int which = c / d, k = c - which * d;
    float* dst = which == 0 ? dw : which == 1 ? db : dcol;
    const int acc = which == 2 ? dcol_accumulate : accumulate;
    if (dst) dst[k] = acc ? dst[k] + s : s;
  }
}

int pick_nv(int d) {
  const int nq = d / 4;
  for (int nv = 1; nv <= 16; nv *= 2)
    if (nq <= 64 * nv) return nv;
  return 0;
}

}  // namespace

extern "C" int afft_layernorm_bwd_nparts(int32_t rows) {
  int n = ((rows + LNB_RG - 1) / LNB_RG + 1) / 2;   // >= 2 row steps per workgroup amortise its column-partial slab
  return n < 1 ? 1 : (n > 256 ? 256 : n);           // one workgroup (16 waves) per CU at most
}

static int ln_fwd_launch(const float* x, int64_t ldx, const float* w, const float* b, float eps, int32_t rows, int32_t d, void* y,
                         int64_t ldy, int32_t y_dtype, int64_t y_lo, void* y2, int64_t ldy2, float* mean, float* rstd, hipStream_t stream,
                         void* y_lo8 = nullptr) {
  AFFT_CHECK(x && y, "layernorm_fwd: null pointer");
  AFFT_CHECK(d > 0 && d % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldy2 % 4 == 0 && y_lo % 4 == 0,
             "layernorm_fwd: d/ld must be multiples of 4 (d=%d)", d);
  AFFT_CHECK(y_dtype >= AFFT_F32 && y_dtype <= AFFT_F16, "layernorm_fwd: bad y_dtype %d", y_dtype);
  const int nv = pick_nv(d);
  AFFT_CHECK(nv != 0, "layernorm_fwd: d=%d exceeds 4096", d);
  if (rows == 0) return 0;
  AfftKernelScope ktrace(AFFT_K_LN_FWD, rows, d,
                         (int64_t)rows * d * (4 + (y_dtype == AFFT_F32 ? 4 : 2) + (y_lo ? 2 : 0) + (y_lo8 ? 1 : 0) + (y2 ? 2 : 0)) + (int64_t)rows * 8, 0, stream);
  const int grid = (rows + LN_WAVES - 1) / LN_WAVES;
#define LN_FWD(NV) hipLaunchKernelGGL(ln_fwd_kernel<NV>, dim3(grid), dim3(256), 0, stream, x, ldx, w, b, eps, rows, d, y, ldy, y_dtype, mean, rstd, y_lo, (bf16_t*)y2, ldy2, (unsigned char*)y_lo8)
  switch (nv) { case 1: LN_FWD(1); break; case 2: LN_FWD(2); break; case 4: LN_FWD(4); break; case 8: LN_FWD(8); break; default: LN_FWD(16); }
#undef LN_FWD
  AFFT_LAUNCH_CHECK();
  return 0;
}

extern "C" int afft_layernorm_fwd(const float* x, int64_t ldx, const float* w, const float* b, float eps,
                                  int32_t rows, int32_t d, void* y, int64_t ldy, int32_t y_dtype, float* mean,
                                  float* rstd, void* stream_) {
  return ln_fwd_launch(x, ldx, w, b, eps, rows, d, y, ldy, y_dtype, 0, nullptr, 0, mean, rstd, (hipStream_t)stream_);
}

extern "C" int afft_layernorm_fwd_split(const float* x, int64_t ldx, const float* w, const float* b, float eps, int32_t rows, int32_t d,
                                        void* y_hi, int64_t ldy, int64_t y_lo, void* y_bf16, int64_t ldyb, float* mean, float* rstd,
                                        void* y_lo8, void* stream_) {
  AFFT_CHECK(!y_lo8 || y_lo == 0, "layernorm_fwd_split: give the lo part as an fp16 plane (y_lo) or as an e4m3 byte plane (y_lo8), not both");
  return ln_fwd_launch(x, ldx, w, b, eps, rows, d, y_hi, ldy, AFFT_F16, y_lo, y_bf16, ldyb, mean, rstd, (hipStream_t)stream_, y_lo8);
}

extern "C" int afft_layernorm_bwd(const void* dy, int64_t lddy, int32_t dy_dtype, const float* x, int64_t ldx,
                                  const float* w, const float* mean, const float* rstd, int32_t rows, int32_t d,
                                  const float* dx_in, float* dx_out, int64_t lddx, void* dx_bf16,
                                  const afft_dropout_t* copy_drop, float* dw, float* db, int32_t accumulate,
                                  float* dcol, int32_t dcol_accumulate, float* partial, void* stream_) {
  return afft_layernorm_bwd_take(dy, lddy, dy_dtype, x, ldx, w, mean, rstd, rows, d, dx_in, lddx, 1, dx_out, lddx, dx_bf16, copy_drop, dw, db,
                                 accumulate, dcol, dcol_accumulate, partial, stream_);
}

extern "C" int afft_layernorm_bwd_take(const void* dy, int64_t lddy, int32_t dy_dtype, const float* x, int64_t ldx, const float* w,
                                       const float* mean, const float* rstd, int32_t rows, int32_t d, const float* dx_in, int64_t lddx_in,
                                       int32_t in_take, float* dx_out, int64_t lddx, void* dx_bf16, const afft_dropout_t* copy_drop,
                                       float* dw, float* db, int32_t accumulate, float* dcol, int32_t dcol_accumulate, float* partial,
                                       void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(in_take >= 1 && lddx_in % 4 == 0, "layernorm_bwd: bad in_take / lddx_in");
  AFFT_CHECK(dy && x && mean && rstd && dx_out && partial, "layernorm_bwd: null pointer");
  AFFT_CHECK(d > 0 && d % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0, "layernorm_bwd: d/ld must be multiples of 4");
  AFFT_CHECK(d <= 2048, "layernorm_bwd: d=%d exceeds 2048 (the widest stream of the path; the 4096-wide instantiation spilled 101-125 registers and was removed in round 6)", d);
  if (rows == 0) return 0;
  AfftKernelScope ktrace(AFFT_K_LN_BWD, rows, d,
                         (int64_t)rows * d * ((dy_dtype == AFFT_F32 ? 4 : 2) + 4 + (dx_in ? 4 : 0) + 4 + (dx_bf16 ? 2 : 0)) + (int64_t)rows * 8, 0, stream);
  const int qs = (d / 4 + LNB_CS - 1) / LNB_CS;
  const int nv = qs <= 64 ? 1 : 2;
  const int grid = afft_layernorm_bwd_nparts(rows);
  const int nslab = dcol ? 3 : 2;
  const DropParams drop = make_drop(copy_drop);
  const size_t lds = (size_t)nslab * d * sizeof(float) + 2 * LNB_RG * LNB_CS * 2 * sizeof(float);
  AFFT_CHECK(dy_dtype == AFFT_F32 || dy_dtype == AFFT_BF16, "layernorm_bwd: dy is fp32 or bf16");
#define LN_BWD(NV, F) hipLaunchKernelGGL((ln_bwd_kernel<NV, F>), dim3(grid), dim3(1024), lds, stream, dy, lddy, x, ldx, w, mean, rstd, rows, d, dx_in, dx_out, lddx, (bf16_t*)dx_bf16, drop, nslab, partial, lddx_in, in_take)
  if (dy_dtype == AFFT_F32) { if (nv == 1) LN_BWD(1, true); else LN_BWD(2, true); }
  else { if (nv == 1) LN_BWD(1, false); else LN_BWD(2, false); }
#undef LN_BWD
  AFFT_LAUNCH_CHECK();
  if (dw || db || dcol) {
    // (round 5: this launch on the auxiliary stream -- it feeds the optimizer, not the chain -- measured 0.2 ms per step SLOWER)
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((nslab * d + 63) / 64), dim3(64 * LNR_PL), 0, stream, partial, grid, d, nslab, dw, db,
                       accumulate, dcol, dcol_accumulate);
    AFFT_LAUNCH_CHECK();
  }
  return 0;
}
