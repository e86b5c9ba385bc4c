// Shared tile machinery of the bf16 MFMA GEMM kernels (gemm.hip, gemm_pp.hip): LDS operand images,
// LDS-DMA staging, fragment reads, XCD-aware tile order.
#pragma once
#include <type_traits>

#include "common.h"

namespace afft_gemm_detail {


constexpr int BK = 64;
constexpr int GROUP_M = 8;

struct GemmFast {
  const bf16_t* A; int64_t lda;  // k-contiguous: A[M][K] ; k-strided: A[K][M]
  const bf16_t* B; int64_t ldb;  // k-contiguous: B[N][K] ; k-strided: B[K][N]
  int K;
  int tiles_m, tiles_n;
  int splitk;   // K is cut into `splitk` slices along gridDim.y (128x128 kernel only); 1 = off
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

// ----- k-contiguous image: tile [ROWS][64 k] bf16, 128 B per row; 16-B chunk c of row r lives at chunk c ^ ((r>>1)&7)
template <int ROWS, int NWAVES>
__device__ __forceinline__ void stage_kc(const bf16_t* __restrict__ G, int64_t ld, int row0, int nrows, int k0,
                                         char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int jj = 0; jj < ROWS / 8 / NWAVES; ++jj) {
    const int j = wave + jj * NWAVES;     // 1-KiB piece = 8 rows
    const int row = j * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int grow = row0 + row;
    grow = grow < nrows ? grow : nrows - 1;  // tail rows: re-read a valid row, result discarded by the epilogue
    const bf16_t* src = G + (int64_t)grow * ld + k0 + chunk * 8;
    __builtin_amdgcn_global_load_lds((const AFFT_GLOBAL void*)src, (AFFT_LDS void*)(lds_tile + j * 1024), 16, 0, 0);
  }
}
__device__ __forceinline__ bf16x8 frag_kc(const char* lds_tile, int row, int chunk) {
  return *(const bf16x8*)(lds_tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

// ----- k-strided image: tile [64 k][COLS] bf16, 2*COLS B per row; 32-B unit u of row r lives at unit u ^ f(r),
//       f(r) = (r&3) | ((r>>3)&1)<<2 : the two 4-row blocks a 32-lane half reads by ds_read_b64_tr_b16 hit 8 distinct units
__device__ __forceinline__ int ks_f(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }
template <int COLS, int NWAVES>
__device__ __forceinline__ void stage_ks(const bf16_t* __restrict__ G, int64_t ld, int col0, int k0,
                                         char* lds_tile, int wave, int lane) {
  constexpr int CH = COLS / 8;            // 16-B chunks per row (16 or 32)
  constexpr int RPP = 64 / CH;            // rows per 1-KiB piece
#pragma unroll
  for (int jj = 0; jj < COLS / 8 / NWAVES; ++jj) {
    const int j = wave + jj * NWAVES;
    const int row = j * RPP + lane / CH;
    const int c16 = lane % CH;
    const int src_c16 = (((c16 >> 1) ^ ks_f(row)) << 1) | (c16 & 1);
    int64_t col = col0 + src_c16 * 8;
    col = col < ld - 8 ? col : ld - 8;     // tail columns: stay inside the row, result discarded
    const bf16_t* src = G + (int64_t)(k0 + row) * ld + col;
    __builtin_amdgcn_global_load_lds((const AFFT_GLOBAL void*)src, (AFFT_LDS void*)(lds_tile + j * 1024), 16, 0, 0);
  }
}
// fragment for MFMA 16x16x32: lane (g = lane>>4, r = lane&15) needs tile[k = kb + 8g + j][16*unit + r], j = 0..7
template <int COLS>
__device__ __forceinline__ bf16x8 frag_ks(const char* lds_tile, int kb, int unit, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int r0 = kb + 8 * g + q, r1 = r0 + 4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (AFFT_LDS bf16x4*)(lds_tile + r0 * (2 * COLS) + ((unit ^ ks_f(r0)) << 5) + p * 8));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (AFFT_LDS bf16x4*)(lds_tile + r1 * (2 * COLS) + ((unit ^ ks_f(r1)) << 5) + p * 8));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt_only() {
  static_assert(N >= 0 && N < 64, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt immediate");
  // lgkmcnt(0): this wave's LDS reads of the stage about to be refilled have returned before it signals the barrier
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}


}  // namespace afft_gemm_detail
