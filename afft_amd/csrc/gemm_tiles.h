// Shared tile machinery of the bf16 MFMA GEMM kernels (gemm.hip, gemm_pp.hip): LDS operand images,
// LDS-DMA staging, fragment reads, XCD-aware tile order.
#pragma once
#include <type_traits>

#include "common.h"

namespace afft_gemm_detail {


template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

constexpr int BK = 64;
#ifndef AFFT_GROUP_M
#define AFFT_GROUP_M 6     // tile rows per column group inside an XCD's chunk.  Alone every path shape is flat over 4..12 (round 1 took 8); inside
#endif                     // the step, beside the other stream, 4-6 are 1-2.3 % ahead of 8 on cfg2 / EK100 / cfg4 (profiles/r02_knob_sweeps.txt)
constexpr int GROUP_M = AFFT_GROUP_M;

struct GemmFast {
  const bf16_t* A; int64_t lda;  // k-contiguous: A[M][K] ; k-strided: A[K][M]
  const bf16_t* B; int64_t ldb;  // k-contiguous: B[N][K] ; k-strided: B[K][N]
  int K;
  int tiles_m, tiles_n;
  int splitk;   // 128x128 kernel: K is cut into `splitk` slices along gridDim.y; 1 = off
  float* ws;    // split-K: fp32 partial tiles [tile][slice][128*128]
  int* counters;  // split-K: arrivals per tile (zero between launches)
  // bf16x3 (X3 = 1 instantiations): K counts the K-tiles of all three segments, nk_seg of them per segment; segment 0 reads
  // (A hi, B hi), 1 (A lo, B hi), 2 (A hi, B lo); the lo planes sit a_lo / b_lo elements behind A / B.
  // fp16x2 (X3 = 2): fp16 planes, the first TWO segments only (A = hi + lo, B rounded once), v_mfma_f32_16x16x32_f16
  int nk_seg;
  int64_t a_lo, b_lo;
  // fp16 + fp8 (X3 = 3, "fp16x2" with the lo pass on the block-scaled fp8 MFMA): segment 0 = nk_seg K-tiles of (A hi fp16, B fp16
  // image); segment 1 = nk_seg / 2 K-tiles of 128 k each of (A8 = e4m3(2^11 (a - hi)), B8 = e4m3(2^8 b)), byte planes addressed as
  // bf16_t arrays of half the length (a K-tile is 128 B per row either way): lda8 / ldb8 are their row pitches in bf16_t units
  const bf16_t* A8; const bf16_t* B8; int64_t lda8, ldb8;
  EpiParams e;
};

// global K-tile index -> K offset inside the segment and the operand planes of that segment (wave-uniform SALU work)
// the MFMA of every bf16-path kernel: 16-bit operands by the instantiation's plane format (X3 = 2: fp16, else bf16)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
// the lo pass of X3 = 3: one v_mfma_scale_f32_16x16x128_f8f6f4 per pair of 16-byte fragments (k-substeps 0 and 1 of a 128-byte row:
// lane group g holds bytes 16g.. and 64 + 16g.. of its row in BOTH operands, so the k sets pair up whatever order the instruction
// walks them in); e4m3 operands with constant block scales 2^-8 (srcA = the weight fragment) and 2^-11 (srcB = the activation's)
// The fragments of such a kernel are kept as PAIRS (k-substeps 0 and 1 side by side in one 8-register tuple): the block-scaled MFMA
// takes the tuple whole, the fp16 MFMA of the first segment takes its halves (sub-registers: no copies).  Assembling the tuple from
// two separately allocated 4-register fragments at the point of use cost 208 spilled registers in the first build.
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(16))) short bf16x16;
__device__ __forceinline__ bf16x16 frag_pair(bf16x8 s0, bf16x8 s1) { return __builtin_shufflevector(s0, s1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15); }
__device__ __forceinline__ bf16x8 frag_half(bf16x16 p, int s) {
  return s ? __builtin_shufflevector(p, p, 8, 9, 10, 11, 12, 13, 14, 15) : __builtin_shufflevector(p, p, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ f32x4 mfma_lo8(bf16x16 b, bf16x16 a, f32x4 c) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(__builtin_bit_cast(i32x8_t, b), __builtin_bit_cast(i32x8_t, a), c, 0, 0, 0, 127 - 8, 0,
                                                          127 - 11);
}

template <int X3>
__device__ __forceinline__ f32x4 mfma16(bf16x8 b, bf16x8 a, f32x4 c) {
  if constexpr (X3 == 2 || X3 == 3) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, b), __builtin_bit_cast(f16x8_t, a), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c, 0, 0, 0);
}

template <int X3>
__device__ __forceinline__ void seg_operands(const GemmFast& g, int kt, int& k0, const bf16_t*& A, const bf16_t*& B) {
  A = g.A; B = g.B;
  if constexpr (X3 == 3) {
    if (kt >= g.nk_seg) { kt -= g.nk_seg; A = g.A8; B = g.B8; }
  } else if constexpr (X3) {
    const int s1 = kt >= g.nk_seg, s2 = kt >= 2 * g.nk_seg;
    kt -= (s1 + s2) * g.nk_seg;
    if (s1 && !s2) A += g.a_lo;
    if (s2) B += g.b_lo;
  }
  k0 = kt * BK;
}

// XCD-aware tile order: workgroups b and b+8 share an XCD (and its L2); each XCD gets a contiguous chunk of the tile list
// (xcd_chunk), and a chunk is walked in GROUP_M-tall column groups (tile_from_id) so that co-resident tiles share A
// row-panels and B column-panels in that XCD's L2.
__device__ __forceinline__ void xcd_chunk(int ntiles, int xcd, int& first, int& count) {
  const int q = ntiles >> 3, r = ntiles & 7;
  first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  count = q + (xcd < r ? 1 : 0);
}
__device__ __forceinline__ void tile_from_id(int tiles_m, int tiles_n, int id, int& tm, int& tn) {
  const int width = GROUP_M * tiles_n;
  const int group = id / width;
  const int first_m = group * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int in_group = id - group * width;
  tm = first_m + in_group % gsz;
  tn = in_group / gsz;
}
__device__ __forceinline__ void tile_coords(int tiles_m, int tiles_n, int bid, int& tm, int& tn) {
  int first, count;
  xcd_chunk(tiles_m * tiles_n, bid & 7, first, count);
  tile_from_id(tiles_m, tiles_n, first + (bid >> 3), tm, tn);
}

// ----- LDS-DMA staging, scalar-base form: global_load_lds_dwordx4 voff32, s[base:base+1] with M0 = LDS destination.
// A 1-KiB piece is one wave-instruction (64 lanes x 16 B, lane-linear in LDS).  Everything lane-dependent in a source
// address is ONE loop-invariant 32-bit VGPR per operand (row-in-piece * pitch + swizzled 16-byte chunk); the piece's
// first row, the K offset and the edge clamps are wave-uniform SALU work.  (Per-lane 64-bit pointers, which the
// __builtin_amdgcn_global_load_lds form needs, cost 2 VGPRs per piece kind and 2 VALU per piece.)
// Pieces are dealt to the NW waves of the workgroup round-robin: wave w stages pieces w, w + NW, ...
//
// k-contiguous image: tile [ROWS][64 k] bf16, 128 B per row; 16-B chunk c of row r lives at chunk c ^ ((r>>1)&7);
//   piece j = rows 8j..8j+7, lane -> row (lane>>3), chunk (lane&7)
// k-strided image: tile [64 k][128 cols] bf16, 256 B per row; 32-B unit u of row r lives at unit u ^ f(r),
//   f(r) = (r&3) | ((r>>3)&1)<<2 (the two 4-row blocks a 32-lane half reads by ds_read_b64_tr_b16 hit 8 distinct units);
//   piece j = k-rows 4j..4j+3, lane -> row (lane>>4), 16-B chunk (lane&15)
struct LaneOffsets {
  unsigned kc_row, kc_chunk16;   // k-contiguous: row inside the piece, byte offset of the (swizzled) source chunk
  unsigned ks_row, ks_c16;       // k-strided:    k-row inside the piece, byte offset of the (swizzled) source chunk
};
__device__ __forceinline__ LaneOffsets lane_offsets(int wave, int lane) {   // NW even: (j & 1), ((j >> 1) & 1) = wave's
  LaneOffsets o;
  o.kc_row = lane >> 3;
  o.kc_chunk16 = (unsigned)(((lane & 7) ^ (((wave & 1) << 2) + (lane >> 4))) << 4);
  o.ks_row = lane >> 4;
  o.ks_c16 = (unsigned)((((((lane & 15) >> 1) ^ ((lane >> 4) | (((wave >> 1) & 1) << 2))) << 1) | (lane & 1)) << 4);
  return o;
}
// LDS destination = lds_wave (the one live scalar: the tile ring's LDS address + this wave's 1-KiB lane of every piece) + off (a
// plain number: ring slot, piece).  Round 2 passed the sum: the compiler then kept one precomputed SGPR per (slot, piece) --
// 16-22 of them -- alive across the main loop, and in the looping instantiations (stream-K, capped grid) spilled them to VGPR
// lanes and read them back between the LDS-DMA issues (22 v_readlane per two K-tiles).  An immediate is rematerialised instead.
__device__ __forceinline__ void glds16(const char* sbase, unsigned voff, unsigned lds_wave, unsigned off) {
  asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_wave), "s"(off) : "memory", "scc");
}
// 4 bytes per lane (256 B per wave-instruction): the cheapest operation that still counts in vmcnt (K-loop overshoot)
__device__ __forceinline__ void glds4(const char* sbase, unsigned voff, unsigned lds_wave, unsigned off) {
  asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_wave), "s"(off) : "memory", "scc");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(AFFT_LDS const char*)p; }

// source of piece jj of this wave: wave-uniform base pointer + per-lane byte offset
template <int NW>
__device__ __forceinline__ void src_kc_piece(const bf16_t* __restrict__ G, int64_t ld, unsigned ld2, unsigned voff_full,
                                             const LaneOffsets& lo, int row0, int nrows, int k0, int wave, int jj,
                                             const char*& sbase, unsigned& voff) {
  const int j = wave + jj * NW;
  const int pb = min(row0 + j * 8, nrows - 1);          // tail pieces re-read valid rows; the epilogue drops them
  const int lim = nrows - 1 - pb;
  sbase = (const char*)(G + (int64_t)pb * ld + k0);
  voff = voff_full;
  if (lim < 7) voff = min(lo.kc_row, (unsigned)lim) * ld2 + lo.kc_chunk16;
}
template <int NW>
__device__ __forceinline__ void src_ks_piece(const bf16_t* __restrict__ G, int64_t ld, unsigned ld2, unsigned voff_full,
                                             const LaneOffsets& lo, int col0, int k0, int wave, int jj,
                                             const char*& sbase, unsigned& voff) {
  const int limc = ((int)ld - 8 - col0) * 2;              // last 16-byte chunk that stays inside the row
  const int j = wave + jj * NW;
  sbase = (const char*)(G + (int64_t)(k0 + j * 4) * ld + col0);
  voff = voff_full;
  if (limc < 240) voff = lo.ks_row * ld2 + min(lo.ks_c16, (unsigned)max(limc, 0));
}
// rows row0.. of a k-contiguous operand G[nrows][ld], K offset k0 -> PIECES pieces of this wave at byte `dst` of the ring
// (lds_wave = LDS address of the ring + wave * 1024)
template <int NW>
__device__ __forceinline__ void stage_kc_piece(const bf16_t* __restrict__ G, int64_t ld, unsigned ld2, unsigned voff_full,
                                               const LaneOffsets& lo, int row0, int nrows, int k0, unsigned dst, int wave, int jj,
                                               unsigned lds_wave) {
  const char* sbase; unsigned voff;
  src_kc_piece<NW>(G, ld, ld2, voff_full, lo, row0, nrows, k0, wave, jj, sbase, voff);
  glds16(sbase, voff, lds_wave, dst + jj * NW * 1024);
}
template <int NW, int PIECES>
__device__ __forceinline__ void stage_kc(const bf16_t* __restrict__ G, int64_t ld, unsigned ld2, unsigned voff_full,
                                         const LaneOffsets& lo, int row0, int nrows, int k0, unsigned dst, int wave, unsigned lds_wave) {
#pragma unroll
  for (int jj = 0; jj < PIECES; ++jj) stage_kc_piece<NW>(G, ld, ld2, voff_full, lo, row0, nrows, k0, dst, wave, jj, lds_wave);
}
// columns col0..col0+127 of a k-strided operand G[K][ld], K rows k0..k0+63
template <int NW>
__device__ __forceinline__ void stage_ks_piece(const bf16_t* __restrict__ G, int64_t ld, unsigned ld2, unsigned voff_full,
                                               const LaneOffsets& lo, int col0, int k0, unsigned dst, int wave, int jj,
                                               unsigned lds_wave) {
  const char* sbase; unsigned voff;
  src_ks_piece<NW>(G, ld, ld2, voff_full, lo, col0, k0, wave, jj, sbase, voff);
  glds16(sbase, voff, lds_wave, dst + jj * NW * 1024);
}
template <int NW, int PIECES>
__device__ __forceinline__ void stage_ks(const bf16_t* __restrict__ G, int64_t ld, unsigned ld2, unsigned voff_full,
                                         const LaneOffsets& lo, int col0, int k0, unsigned dst, int wave, unsigned lds_wave) {
#pragma unroll
  for (int jj = 0; jj < PIECES; ++jj) stage_ks_piece<NW>(G, ld, ld2, voff_full, lo, col0, k0, dst, wave, jj, lds_wave);
}
__device__ __forceinline__ bf16x8 frag_kc(const char* lds_tile, int row, int chunk) {
  return *(const bf16x8*)(lds_tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

__device__ __forceinline__ int ks_f(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }
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

// ----- split-K hand-over ("last arriver"): every K-segment of a tile parks its fp32 partial (NV f32x4 per thread,
// NT threads) in the workspace and counts itself in; the segment whose count completes the tile adds the partials in
// segment order -- bitwise the same sum whoever it is -- and returns true (it runs the epilogue); the others return false.
// No workgroup waits for another, so nothing can deadlock, and there is no release/acquire fence (an agent-scope release
// writes the whole L2 back and an acquire invalidates it: +30 us per GEMM measured).  The form is the guide's split-K seam
// (MI355X_MICROARCH.md, visibility: first row of the sc1 table): EVERY payload store a 16-byte write-through buffer store,
// every storing wave drains (s_waitcnt vmcnt(0)) before the workgroup barrier, ONE lane adds to the tile's counter (agent
// scope), the workgroup whose add came last reads EVERY payload byte with 16-byte sc loads behind a barrier.  Round 2 stored
// the partials as 4-byte system-scope atomics: one fabric write per dword, ~6x the time per byte of a 16-byte store.
// slot(sl): wave-uniform base pointer of segment sl's partial [NV][NT] x 16 B.  smem: 4 bytes of LDS nobody else touches
// between the two barriers inside.
#ifndef AFFT_HANDOFF_AUX
#define AFFT_HANDOFF_AUX 17      // cache policy of the payload stores and loads: 16 = sc1 (agent), 17 = sc0 sc1 (system)
#endif
#ifndef AFFT_HANDOFF_DIAG
#define AFFT_HANDOFF_DIAG 0      // diagnostic builds only (wrong results): 1 = no payload stores, 2 = no payload loads
#endif
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int NV, int NT, class SlotFn>
__device__ __forceinline__ bool handoff_combine(f32x4 (&acc)[NV], SlotFn&& slot, int* counter, int S, int me, int tid, char* smem) {
  constexpr int SLICE_BYTES = NV * NT * 16;
  // Segment 0 looks before it parks: if every other segment has already arrived it is the last one -- it keeps its
  // accumulators, adds the others to them in K order (a + b = b + a bitwise) and nothing of its own goes to memory.
  bool last_without_parking = false;
  if (me == 0) {
    if (tid == 0) *(volatile int*)smem = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    last_without_parking = __builtin_amdgcn_readfirstlane(*(volatile int*)smem) == S - 1;
    __syncthreads();
  }
  if (!last_without_parking) {
    {
      const __amdgpu_buffer_rsrc_t mine = __builtin_amdgcn_make_buffer_rsrc(slot(me), 0, SLICE_BYTES, 0x00020000);
      static_for<0, NV>([&](auto idx) {
        constexpr int v = decltype(idx)::value;
        if (!(AFFT_HANDOFF_DIAG & 1))
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[v]), mine, tid * 16, v * NT * 16, AFFT_HANDOFF_AUX);
      });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's partial has been written through
    __syncthreads();
    if (tid == 0) *(volatile int*)smem = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int arrived = __builtin_amdgcn_readfirstlane(*(volatile int*)smem);    // workgroup-uniform: callers branch on the result
    if (arrived != S - 1) return false;
  }
  // every parked segment is read back from the workspace, this workgroup's own too (the same bits as its registers): no
  // second set of NV accumulators -> no spills (256 accumulator + sum registers did not fit beside the 256x256 kernel's 128)
  for (int sl = last_without_parking ? 1 : 0; sl < S; ++sl) {
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(slot(sl), 0, SLICE_BYTES, 0x00020000);
    static_for<0, NV>([&](auto idx) {
      constexpr int v = decltype(idx)::value;
      f32x4 t = acc[v];
      if (!(AFFT_HANDOFF_DIAG & 2))
      t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(src, tid * 16, v * NT * 16, AFFT_HANDOFF_AUX));
      acc[v] = sl == 0 ? t : acc[v] + t;
    });
  }
  if (tid == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
  return true;
}

// plain split-K: the S slices of tile `tile` sit side by side in the workspace
template <int NV, int NT>
__device__ __forceinline__ bool splitk_combine(f32x4 (&acc)[NV], float* ws, int* counters, int tile, int S, int me, int tid,
                                               char* smem) {
  float* tile_ws = ws + (int64_t)tile * S * (NV * NT * 4);
  return handoff_combine<NV, NT>(acc, [&](int sl) { return tile_ws + (int64_t)sl * (NV * NT * 4); }, counters + tile, S, me, tid, smem);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt_only() {
  static_assert(N >= 0 && N < 64, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt immediate");
  // lgkmcnt(0): this wave's LDS reads of the stage about to be refilled have returned before it signals the barrier
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}


}  // namespace afft_gemm_detail
