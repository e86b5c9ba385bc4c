// GEMM + fused epilogue for gfx950.
//
//  * gemm_bf16_kernel<NT|TN>: bf16 MFMA (v_mfma_f32_16x16x32_bf16), 128x128x64 tiles, 4 waves (2x2),
//    wave tile 64x64 = 4x4 MFMA tiles, operands staged HBM->LDS with 16-byte LDS-DMA
//    (global_load_lds_dwordx4), XOR-swizzled LDS images (conflict-free ds_read_b128 for the
//    k-contiguous NT image, conflict-free ds_read_b64_tr_b16 for the k-strided TN image),
//    double-buffered, XCD-aware grouped tile order.
//  * gemm_f32_kernel: exact fp32 (v_mfma_f32_32x32x2_f32), any strides / sizes; the parity mode
//    and the fallback for shapes the fast path does not take.
//
// Replaces nn.Linear / HF Conv1D forward, dgrad and wgrad on the AFFT path (see include/afft_hip.h).
#include <type_traits>

#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile
constexpr int GROUP_M = 8;

struct GemmFast {
  const bf16_t* A; int64_t lda;  // NT: A[M][K] ; TN: A[K][M]
  const bf16_t* B; int64_t ldb;  // NT: B[N][K] ; TN: B[K][N]
  int K;
  int tiles_m, tiles_n;
  EpiParams e;
};

__device__ __forceinline__ void tile_coords(int tiles_m, int tiles_n, int& tm, int& tn) {
  // XCD-aware remap: workgroups b and b+8 share an XCD (and its L2); give each XCD a contiguous
  // chunk of the tile list, then walk that chunk in GROUP_M-tall column groups so that co-resident
  // tiles share A row-panels and B column-panels in that XCD's L2.
  const int nwg = tiles_m * tiles_n;
  const int bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int width = GROUP_M * tiles_n;
  const int group = id / width;
  const int first_m = group * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int in_group = id - group * width;
  tm = first_m + in_group % gsz;
  tn = in_group / gsz;
}

// ----- NT image: tile [128 rows][64 k] bf16, 128 B per row, 16-B chunk c of row r stored at chunk c ^ ((r>>1)&7)
__device__ __forceinline__ void stage_nt(const bf16_t* __restrict__ G, int64_t ld, int row0, int nrows, int k0,
                                         char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int j = wave * 4 + jj;          // 1-KiB piece = 8 rows
    const int row = j * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int grow = row0 + row;
    grow = grow < nrows ? grow : nrows - 1;  // tail rows: re-read a valid row, result discarded by the epilogue
    const bf16_t* src = G + (int64_t)grow * ld + k0 + chunk * 8;
    __builtin_amdgcn_global_load_lds((const AFFT_GLOBAL void*)src, (AFFT_LDS void*)(lds_tile + j * 1024), 16, 0, 0);
  }
}
__device__ __forceinline__ bf16x8 frag_nt(const char* lds_tile, int row, int chunk) {
  return *(const bf16x8*)(lds_tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

// ----- TN image: tile [64 k][128 cols] bf16, 256 B per row, 32-B unit u of row r stored at unit u ^ f(r),
//       f(r) = (r&3) | ((r>>3)&1)<<2  -> the two 4-row blocks a 32-lane half reads by ds_read_b64_tr_b16 hit 8 distinct units
__device__ __forceinline__ int tn_f(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }
__device__ __forceinline__ void stage_tn(const bf16_t* __restrict__ G, int64_t ld, int col0, int k0,
                                         char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int j = wave * 4 + jj;          // 1-KiB piece = 4 rows of 256 B
    const int row = j * 4 + (lane >> 4);
    const int c16 = lane & 15;
    const int src_c16 = (((c16 >> 1) ^ tn_f(row)) << 1) | (c16 & 1);
    int64_t col = col0 + src_c16 * 8;
    col = col < ld - 8 ? col : ld - 8;     // tail columns: stay inside the row, result discarded
    const bf16_t* src = G + (int64_t)(k0 + row) * ld + col;
    __builtin_amdgcn_global_load_lds((const AFFT_GLOBAL void*)src, (AFFT_LDS void*)(lds_tile + j * 1024), 16, 0, 0);
  }
}
// fragment for MFMA 16x16x32: lane (g = lane>>4, r = lane&15) needs tile[k = kb + 8g + j][col0 + r], j = 0..7
__device__ __forceinline__ bf16x8 frag_tn(const char* lds_tile, int kb, int unit, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int r0 = kb + 8 * g + q, r1 = r0 + 4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (AFFT_LDS bf16x4*)(lds_tile + r0 * 256 + ((unit ^ tn_f(r0)) << 5) + p * 8));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (AFFT_LDS bf16x4*)(lds_tile + r1 * 256 + ((unit ^ tn_f(r1)) << 5) + p * 8));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int TN>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmFast g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A 16K | B 16K]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  int tm, tn;
  tile_coords(g.tiles_m, g.tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int M = g.e.M, N = g.e.N;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int buf, int kt) {
    char* a = smem + buf * (2 * TILE_BYTES);
    char* b = a + TILE_BYTES;
    if (TN) {
      stage_tn(g.A, g.lda, m0, kt * BK, a, wave, lane);
      stage_tn(g.B, g.ldb, n0, kt * BK, b, wave, lane);
    } else {
      stage_nt(g.A, g.lda, m0, M, kt * BK, a, wave, lane);
      stage_nt(g.B, g.ldb, n0, N, kt * BK, b, wave, lane);
    }
  };

  const int nk = g.K / BK;
  stage(0, 0);
  __syncthreads();  // an LDS-DMA is in flight -> the compiler's fence waits vmcnt(0) here
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* a = smem + cur * (2 * TILE_BYTES);
    const char* b = a + TILE_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[4], bfr[4];
      if (TN) {
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = frag_tn(a, 32 * s, wr * 4 + i, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = frag_tn(b, 32 * s, wc * 4 + j, lane);
      } else {
        const int chunk = s * 4 + (lane >> 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = frag_nt(a, wr * 64 + i * 16 + (lane & 15), chunk);
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = frag_nt(b, wc * 64 + j * 16 + (lane & 15), chunk);
      }
      // operands swapped on purpose: D[row = n][col = m] -> each lane owns 4 consecutive n of one row m
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  static_for<0, 16>([&](auto idx) {
    constexpr int i = decltype(idx)::value >> 2, j = decltype(idx)::value & 3;
    const int m = m0 + wr * 64 + i * 16 + (lane & 15);
    const int n = n0 + wc * 64 + j * 16 + 4 * (lane >> 4);
    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
    epilogue4(g.e, m, n, v);
  });
}

// ---------------------------------------------------------------------------------------------
// exact-fp32 path: 64x64x16 tiles, 4 waves (2x2), each wave one 32x32 tile on v_mfma_f32_32x32x2_f32.
struct GemmF32 {
  const void* A; int64_t a_rs, a_cs;
  const void* B; int64_t b_rs, b_cs;
  int K;
  int tiles_m;
  EpiParams e;
};

template <typename T>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmF32 g) {
  __shared__ float As[16][68];  // [k][m]
  __shared__ float Bs[16][68];  // [k][n]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tm = blockIdx.x % g.tiles_m, tn = blockIdx.x / g.tiles_m;
  const int m0 = tm * 64, n0 = tn * 64;
  const int M = g.e.M, N = g.e.N, K = g.K;
  const T* A = (const T*)g.A;
  const T* B = (const T*)g.B;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const bool a_kfast = (g.a_cs == 1);  // consecutive threads along the contiguous dimension
  const bool b_kfast = (g.b_rs == 1);
  for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int ka, ma, kb, nb;
      if (a_kfast) { ka = tid & 15; ma = (tid >> 4) + 16 * r; } else { ma = tid & 63; ka = (tid >> 6) + 4 * r; }
      if (b_kfast) { kb = tid & 15; nb = (tid >> 4) + 16 * r; } else { nb = tid & 63; kb = (tid >> 6) + 4 * r; }
      float av = 0.f, bv = 0.f;
      if (m0 + ma < M && k0 + ka < K) av = Elem<T>::ld(A + (int64_t)(m0 + ma) * g.a_rs + (int64_t)(k0 + ka) * g.a_cs);
      if (n0 + nb < N && k0 + kb < K) bv = Elem<T>::ld(B + (int64_t)(k0 + kb) * g.b_rs + (int64_t)(n0 + nb) * g.b_cs);
      As[ka][ma] = av;
      Bs[kb][nb] = bv;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int k = 2 * kk + (lane >> 5);
      const float bside = Bs[k][wc * 32 + (lane & 31)];
      const float aside = As[k][wr * 32 + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bside, aside, acc, 0, 0, 0);  // D[row = n][col = m]
    }
    __syncthreads();
  }
  const int m = m0 + wr * 32 + (lane & 31);
  static_for<0, 4>([&](auto idx) {
    constexpr int gq = decltype(idx)::value;
    const int n = n0 + wc * 32 + 8 * gq + 4 * (lane >> 5);
    float v[4] = {acc[4 * gq + 0], acc[4 * gq + 1], acc[4 * gq + 2], acc[4 * gq + 3]};
    epilogue4(g.e, m, n, v);
  });
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

extern "C" int afft_gemm(const afft_gemm_t* d, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(d != nullptr, "afft_gemm: null descriptor");
  AFFT_CHECK(d->M >= 0 && d->N >= 0 && d->K >= 0, "afft_gemm: negative size");
  AFFT_CHECK(d->dtype == AFFT_F32 || d->dtype == AFFT_BF16, "afft_gemm: bad dtype %d", d->dtype);
  AFFT_CHECK(d->A && d->B && d->out, "afft_gemm: null operand");
  AFFT_CHECK(!d->accumulate || d->out_dtype == AFFT_F32, "afft_gemm: accumulate needs an fp32 output");
  AFFT_CHECK(d->act < AFFT_ACT_DGELU_ERF || d->aux, "afft_gemm: DGELU needs aux");
  if (d->M == 0 || d->N == 0) return 0;

  EpiParams e;
  e.M = d->M; e.N = d->N; e.alpha = d->alpha; e.bias = d->bias; e.act = d->act;
  e.aux = d->aux; e.ldaux = d->ldaux; e.aux_dtype = d->aux_dtype;
  e.pre = d->pre; e.ldpre = d->ldpre; e.pre_dtype = d->pre_dtype;
  e.rowscale = d->rowscale; e.residual = d->residual; e.ldres = d->ldres;
  e.accumulate = d->accumulate;
  e.out = d->out; e.ldo = d->ldo; e.out_dtype = d->out_dtype;
  e.out2 = d->out2; e.ldo2 = d->ldo2; e.out2_dtype = d->out2_dtype;
  AFFT_CHECK(d->drop.p >= 0.f && d->drop.p < 1.f && d->drop.path_p >= 0.f && d->drop.path_p < 1.f, "afft_gemm: dropout p outside [0,1)");
  e.drop = make_drop(&d->drop);
  auto ok4 = [](const void* p, int64_t ld, int dtype) {
    if (!p) return true;
    const uintptr_t align = dtype == AFFT_F32 ? 16 : 8;
    return (ld % 4 == 0) && ((((uintptr_t)p) & (align - 1)) == 0);
  };
  e.vec4 = ok4(d->out, d->ldo, d->out_dtype) && ok4(d->out2, d->ldo2, d->out2_dtype) &&
           ok4(d->pre, d->ldpre, d->pre_dtype) && ok4(d->aux, d->ldaux, d->aux_dtype) &&
           ok4(d->residual, d->ldres, AFFT_F32) && ok4(d->bias, 4, AFFT_F32);

  const bool nt = d->a_cs == 1 && d->b_rs == 1;
  const bool tn = d->a_rs == 1 && d->b_cs == 1;
  bool fast = d->dtype == AFFT_BF16 && d->K >= BK && d->K % BK == 0 && aligned16(d->A) && aligned16(d->B);
  if (fast && nt) fast = (d->a_rs % 8 == 0) && (d->b_cs % 8 == 0);
  else if (fast && tn) fast = (d->a_cs % 8 == 0) && (d->b_rs % 8 == 0) && d->a_cs >= 8 && d->b_rs >= 8;
  else fast = false;

  if (fast) {
    GemmFast g;
    g.A = (const bf16_t*)d->A; g.B = (const bf16_t*)d->B;
    g.lda = nt ? d->a_rs : d->a_cs;
    g.ldb = nt ? d->b_cs : d->b_rs;
    g.K = d->K;
    g.tiles_m = (d->M + BM - 1) / BM; g.tiles_n = (d->N + BN - 1) / BN;
    g.e = e;
    const int grid = g.tiles_m * g.tiles_n;
    const size_t lds = 4 * TILE_BYTES;
    if (nt) hipLaunchKernelGGL(gemm_bf16_kernel<0>, dim3(grid), dim3(256), lds, stream, g);
    else hipLaunchKernelGGL(gemm_bf16_kernel<1>, dim3(grid), dim3(256), lds, stream, g);
    AFFT_LAUNCH_CHECK();
    return 0;
  }
  GemmF32 g;
  g.A = d->A; g.a_rs = d->a_rs; g.a_cs = d->a_cs;
  g.B = d->B; g.b_rs = d->b_rs; g.b_cs = d->b_cs;
  g.K = d->K; g.tiles_m = (d->M + 63) / 64;
  g.e = e;
  const int grid = g.tiles_m * ((d->N + 63) / 64);
  if (d->dtype == AFFT_F32) hipLaunchKernelGGL(gemm_f32_kernel<float>, dim3(grid), dim3(256), 0, stream, g);
  else hipLaunchKernelGGL(gemm_f32_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, g);
  AFFT_LAUNCH_CHECK();
  return 0;
}
