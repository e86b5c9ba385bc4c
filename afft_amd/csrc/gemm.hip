// GEMM + fused epilogue for gfx950.
//
//  * gemm_bf16_kernel<WM, WN, STAGES, A_KS, B_KS>: bf16 MFMA (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
//    A workgroup of WM x WN waves owns a (64*WM) x (64*WN) output tile (each wave 64x64 = 4x4 MFMA tiles) and
//    walks K in 64-deep steps through a STAGES-deep LDS ring that is filled by 16-byte LDS-DMA
//    (global_load_lds_dwordx4) running STAGES-1 K-steps ahead: the loop waits with a COUNTED s_waitcnt vmcnt(N)
//    (never 0 while more tiles are to come) and one raw s_barrier per K-step, so HBM/L2 latency hides under
//    the MFMAs of the tiles in between.
//      operand images in LDS (XOR-swizzled on the DMA source address, LDS writes stay lane-linear):
//        k-contiguous operand ("row-major [rows][K]"): [rows][64 k], 128 B rows, conflict-free ds_read_b128
//        k-strided operand    ("[K][cols]"):           [64 k][cols], conflict-free ds_read_b64_tr_b16 (transposed read)
//      layouts: NT (A[M][K], B[N][K])  forward of nn.Linear, dgrad of HF Conv1D
//               NN (A[M][K], B[K][N])  dgrad of nn.Linear, forward of HF Conv1D
//               TN (A[K][M], B[K][N])  every weight gradient (reduction over the rows of both operands)
//      shapes:  256x128 tile, 8 waves, 3 stages (144 KiB LDS, 1 workgroup/CU)   -- large GEMMs
//               128x128 tile, 4 waves, 2 stages ( 64 KiB LDS, 2 workgroups/CU)  -- small grids / tails
//    XCD-aware grouped tile order (workgroups b and b+8 share an XCD and its L2).
//    X3 instantiations (afft_gemm_t.split3, "bf16x3"): A and B are two-plane hi / lo splits of fp32 matrices; the K loop runs
//    three segments (hi*hi, lo*hi, hi*lo) with the operand planes chosen per K-step on the SALU -- fp32-grade products.
//    Every epilogue can end in the fused optimizer update instead of a store (afft_sgd_fused_t, common.h: SgdEpi).
//    Dispatch (choose_variant / choose_splitk): 256x256 ping-pong tiles (gemm_pp.hip) when the utilisation of their last
//    round beats that of the 128x128 grid (x1.25); split-K into caller-provided scratch (afft_gemm_t.workspace).
//  * gemm_f32_kernel: exact fp32 (v_mfma_f32_32x32x2_f32), any strides / sizes; the parity mode
//    and the fallback for shapes the fast path does not take.
//
// Replaces nn.Linear / HF Conv1D forward, dgrad and wgrad on the AFFT path (see include/afft_hip.h).
#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "gemm_tiles.h"

using namespace afft_gemm_detail;

int afft_gemm_launch_pp(int a_ks, int b_ks, afft_gemm_detail::GemmFast& g, hipStream_t stream, int x3);
bool afft_gemm_pp2_takes(int M, int N, int K, int x3);      // gemm_pp.hip: the launch runs gemm_bf16_pp2_kernel (whole tiles, even K-tile count; plain bf16 or the NT fp16 + fp8 forward; K = the caller's K)
int afft_gemm_launch_bd(int rows160, int packed, afft_gemm_detail::GemmFast& g, hipStream_t stream);
#ifdef AFFT_EXPERIMENT_Q4      // tools/experiments/gemm_q4.hip (tools/experiments/build_q4.sh): the four-quadrant kernel, variant 11
int afft_gemm_launch_q4(afft_gemm_detail::GemmFast& g, hipStream_t stream);
#endif

#ifndef AFFT_G128_EPI_UNROLL
#define AFFT_G128_EPI_UNROLL 1   // 2 (what pays in gemm_pp.hip) measured 1-3 % slower on the EK100-width and cfg4 steps
#endif

namespace {

template <int WM, int WN, int STAGES, bool A_KS, bool B_KS, bool SPLITK, int X3 = 0>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN >= 8) ? 2 : 2) void gemm_bf16_kernel(const GemmFast g) {
  constexpr int NW = WM * WN;
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int LPT = (BM / 8 + BN / 8) / NW;   // LDS-DMA instructions per wave per K-step
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "pieces must divide evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [STAGES][A image | B image]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  int tm, tn;
  tile_coords(g.tiles_m, g.tiles_n, blockIdx.x, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int M = g.e.M, N = g.e.N;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK / g.splitk;          // K-steps of this slice
  const int ktbase = blockIdx.y * nk;          // split-K: slice z = blockIdx.y
  static_assert(BM == 128 && BN == 128, "staging helpers assume 128-row / 128-column operand tiles");
  const unsigned lds_wave = lds_addr(smem) + wave * 1024;
  const unsigned lda2 = (unsigned)(g.lda * 2), ldb2 = (unsigned)(g.ldb * 2);
  const LaneOffsets lo = lane_offsets(wave, lane);
  const unsigned voffA = A_KS ? lo.ks_row * lda2 + lo.ks_c16 : lo.kc_row * lda2 + lo.kc_chunk16;
  const unsigned voffB = B_KS ? lo.ks_row * ldb2 + lo.ks_c16 : lo.kc_row * ldb2 + lo.kc_chunk16;
  auto stage = [&](int kt) {
    const unsigned a = (kt % STAGES) * STAGE_BYTES;      // bytes into the ring; lds_wave carries the ring's address
    const unsigned b = a + A_BYTES;
    int k0; const bf16_t *Ap, *Bp;
    seg_operands<X3>(g, ktbase + kt, k0, Ap, Bp);
    if constexpr (A_KS) stage_ks<NW, 16 / NW>(Ap, g.lda, lda2, voffA, lo, m0, k0, a, wave, lds_wave);
    else stage_kc<NW, 16 / NW>(Ap, g.lda, lda2, voffA, lo, m0, M, k0, a, wave, lds_wave);
    if constexpr (B_KS) stage_ks<NW, 16 / NW>(Bp, g.ldb, ldb2, voffB, lo, n0, k0, b, wave, lds_wave);
    else stage_kc<NW, 16 / NW>(Bp, g.ldb, ldb2, voffB, lo, n0, N, k0, b, wave, lds_wave);
  };
  // wait until all but the `ahead` most recently issued tiles of this wave have landed, then rendezvous
  auto wait_tiles_then_barrier = [&](int ahead) {
    if constexpr (STAGES >= 4) { if (ahead >= 2) wait_vmcnt<2 * LPT>(); else if (ahead == 1) wait_vmcnt<LPT>(); else wait_vmcnt<0>(); }
    else if constexpr (STAGES == 3) { if (ahead >= 1) wait_vmcnt<LPT>(); else wait_vmcnt<0>(); }
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
  };

  constexpr int D = STAGES - 1;  // prefetch distance in K-steps
#pragma unroll
  for (int t = 0; t < D; ++t)
    if (t < nk) stage(t);
  wait_tiles_then_barrier(min(D, nk) - 1);   // tile 0 landed everywhere

  for (int kt = 0; kt < nk; ++kt) {
    if (kt + D < nk) stage(kt + D);         // refills the stage read in iteration kt-1 (all waves passed its barrier)
    const char* a = smem + (kt % STAGES) * STAGE_BYTES;
    const char* b = a + A_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[4], bfr[4];
      const int chunk = s * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (A_KS) af[i] = frag_ks<BM>(a, 32 * s, wr * 4 + i, lane);
        else af[i] = frag_kc(a, wr * 64 + i * 16 + (lane & 15), chunk);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (B_KS) bfr[j] = frag_ks<BN>(b, 32 * s, wc * 4 + j, lane);
        else bfr[j] = frag_kc(b, wc * 64 + j * 16 + (lane & 15), chunk);
      }
      // operands swapped on purpose: D[row = n][col = m] -> each lane owns 4 consecutive n of one row m
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = mfma16<X3>(bfr[j], af[i], acc[i][j]);
    }
    if (kt + 1 < nk) {
      // tile kt+1 must have landed; tiles kt+2 .. min(kt+D, nk-1) may stay in flight across the barrier
      const int ahead = min(kt + D, nk - 1) - (kt + 1);
      wait_tiles_then_barrier(ahead);
    }
  }

  // Split-K (small grids): see splitk_combine (gemm_tiles.h); the last-arriving slice runs the ordinary epilogue.
  if constexpr (SPLITK) {   // its own instantiation: the combine code doubles the register footprint of the plain kernel
    __syncthreads();   // every wave is done reading the ring -> smem is free
    if (!splitk_combine<16, 64 * NW>(reinterpret_cast<f32x4(&)[16]>(acc), g.ws, g.counters, blockIdx.x, g.splitk, blockIdx.y, tid, smem))
      return;
  }

  // Epilogue through LDS (same scheme as gemm_pp.hip): accumulators -> fp32 [BM][BN] image with a 16-byte row pad
  // (conflict-free scatter), then whole rows per wave with 16-byte LDS reads and fully coalesced global accesses.
  constexpr int ESTRIDE = BN * 4 + 16;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ... and its reads have returned (see gemm_bf16_g2_kernel: s_barrier alone does not say so)
  __builtin_amdgcn_s_barrier();   // every wave is done reading the ring
  static_for<0, 16>([&](auto idx) {
    constexpr int i = decltype(idx)::value >> 2, j = decltype(idx)::value & 3;
    const int row = wr * 64 + i * 16 + (lane & 15);
    const int col = wc * 64 + j * 16 + 4 * (lane >> 4);
    *(f32x4*)(smem + row * ESTRIDE + col * 4) = acc[i][j];
  });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const DropParams dp = with_salt(g.e.drop);
  constexpr int LPR = BN / 8;            // lanes per row (8 columns each: 16-byte bf16 stores)
  constexpr int RPI = 64 / LPR;          // rows per wave-iteration
#pragma unroll AFFT_G128_EPI_UNROLL      // row steps in flight per thread, as in gemm_pp.hip
  for (int it = 0; it < BM / NW / RPI; ++it) {
    const int row = wave * (BM / NW) + it * RPI + lane / LPR;
    const int c8 = lane % LPR;
    const f32x4 t0 = *(const f32x4*)(smem + row * ESTRIDE + c8 * 32);
    const f32x4 t1 = *(const f32x4*)(smem + row * ESTRIDE + c8 * 32 + 16);
    float o[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
    epilogue8(g.e, dp, m0 + row, n0 + 8 * c8, o);
  }
}

// ---------------------------------------------------------------------------------------------
// Round 6: gemm_bf16_kernel<2, 2, 2, ..> with a loop that holds nothing but the schedule ("g2"), for what the small-grid launches of
// the step are -- plain bf16, whole 128x128 tiles, an even number of K-tiles per slice: the predictor's M = B*T rows, the EK100 widths,
// the reference's batch of 16.  Same tile, ring (2 stages), images, split-K seam and epilogue; what changes is what gemm_pp.hip's pp2
// kernel changed: the K loop runs over K-tile PAIRS without a branch (the last pair, which has no successor to stage, is peeled), the 8
// LDS-DMA sources of a wave are 8 loop-invariant SGPR bases + one running VGPR per operand (the general kernel recomputes row clamps and
// 64-bit products per piece and K-tile), ring stage and fragment index are instruction offsets of a handful of precomputed LDS addresses.
template <int OFF>
__device__ __forceinline__ void glds16c(const char* sbase, unsigned voff, unsigned lds_wave) {
  asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_wave), "n"(OFF) : "memory", "scc");
}

// F16: the operands are fp16 (one fp16 pass of the "fp16x2" forward, afft_gemm_t.split3 = 4: the predictor's GEMMs) -- the MFMA form is all that changes.
template <bool A_KS, bool B_KS, bool SPLITK, bool F16 = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16_g2_kernel(const GemmFast g) {
  constexpr int A_BYTES = 128 * BK * 2, STAGE_BYTES = 2 * A_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A image | B image]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  int tm, tn;
  tile_coords(g.tiles_m, g.tiles_n, blockIdx.x, tm, tn);
  const int m0 = tm * 128, n0 = tn * 128;
  const int nk = g.K / BK / g.splitk;          // K-tiles of this slice: even, >= 2 (g2_takes)
  const int ktbase = blockIdx.y * nk;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned lds_wave = lds_addr(smem) + wave * 1024;
  const unsigned lda2 = (unsigned)(g.lda * 2), ldb2 = (unsigned)(g.ldb * 2);
  const LaneOffsets lo = lane_offsets(wave, lane);
  const unsigned stepA = A_KS ? 64u * lda2 : 128u, stepB = B_KS ? 64u * ldb2 : 128u;     // bytes per K-tile
  unsigned vA = (A_KS ? lo.ks_row * lda2 + lo.ks_c16 : lo.kc_row * lda2 + lo.kc_chunk16) + (unsigned)ktbase * stepA;
  unsigned vB = (B_KS ? lo.ks_row * ldb2 + lo.ks_c16 : lo.kc_row * ldb2 + lo.kc_chunk16) + (unsigned)ktbase * stepB;
  const char* bA[4];
  const char* bB[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {      // piece j = wave + 4 jj: rows 8j.. (k-contiguous) / k-rows 4j.. (k-strided) of the 128-wide tile
    const int j = wave + 4 * jj;
    bA[jj] = A_KS ? (const char*)(g.A + (int64_t)(4 * j) * g.lda + m0) : (const char*)(g.A + (int64_t)(m0 + 8 * j) * g.lda);
    bB[jj] = B_KS ? (const char*)(g.B + (int64_t)(4 * j) * g.ldb + n0) : (const char*)(g.B + (int64_t)(n0 + 8 * j) * g.ldb);
  }
  auto stage = [&](auto pc) {      // the next K-tile into ring stage P
    constexpr int P = decltype(pc)::value;
    static_for<0, 4>([&](auto jc) { glds16c<P * STAGE_BYTES + decltype(jc)::value * 4096>(bA[decltype(jc)::value], vA, lds_wave); });
    static_for<0, 4>([&](auto jc) { glds16c<P * STAGE_BYTES + A_BYTES + decltype(jc)::value * 4096>(bB[decltype(jc)::value], vB, lds_wave); });
    vA += stepA;
    vB += stepB;
  };
  // fragment addresses (see gemm_pp.hip pp2): k-contiguous image -> one register per k-substep, fragment index = +2 KiB; k-strided
  // image -> one register per fragment, k-substep = +8 KiB, second 4-row block = +1 KiB
  constexpr int NA = A_KS ? 4 : 2, NB = B_KS ? 4 : 2;
  unsigned adA[NA], adB[NB];
  {
    const unsigned l0 = lds_addr(smem);
    const int g4 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3, r16 = lane & 15;
#pragma unroll
    for (int x = 0; x < NA; ++x) {
      unsigned a;
      if constexpr (A_KS) a = (8 * g4 + q4) * 256 + ((((wr * 4 + x) ^ ks_f(8 * g4 + q4))) << 5) + p4 * 8;
      else { const int row = wr * 64 + r16; a = row * 128 + ((((x * 4 + g4) ^ ((row >> 1) & 7))) << 4); }
      adA[x] = l0 + a;
      asm volatile("" : "+v"(adA[x]));
    }
#pragma unroll
    for (int x = 0; x < NB; ++x) {
      unsigned b;
      if constexpr (B_KS) b = (8 * g4 + q4) * 256 + ((((wc * 4 + x) ^ ks_f(8 * g4 + q4))) << 5) + p4 * 8;
      else { const int row = wc * 64 + r16; b = row * 128 + ((((x * 4 + g4) ^ ((row >> 1) & 7))) << 4); }
      adB[x] = l0 + A_BYTES + b;
      asm volatile("" : "+v"(adB[x]));
    }
  }
  auto rd128 = [](unsigned addr, auto offc) -> bf16x8 { return *(AFFT_LDS const bf16x8*)(size_t)(addr + decltype(offc)::value); };
  auto rdtr = [](unsigned addr, auto offc) -> bf16x8 {
    const bf16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AFFT_LDS bf16x4*)(size_t)(addr + decltype(offc)::value));
    const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AFFT_LDS bf16x4*)(size_t)(addr + decltype(offc)::value + 1024));
    bf16x8 f;
    f[0] = lo4[0]; f[1] = lo4[1]; f[2] = lo4[2]; f[3] = lo4[3];
    f[4] = hi4[0]; f[5] = hi4[1]; f[6] = hi4[2]; f[7] = hi4[3];
    return f;
  };
  auto ktile = [&](auto pc, auto lastc) {
    constexpr int P = decltype(pc)::value;
    constexpr bool LAST = decltype(lastc)::value;
    if constexpr (!LAST) stage(std::integral_constant<int, 1 - P>{});      // refills the stage read one K-tile ago (every wave passed its barrier)
    static_for<0, 2>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      bf16x8 af[4], bfr[4];
      static_for<0, 4>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (A_KS) af[i] = rdtr(adA[i], std::integral_constant<int, P * STAGE_BYTES + s * 8192>{});
        else af[i] = rd128(adA[s], std::integral_constant<int, P * STAGE_BYTES + i * 2048>{});
        if constexpr (B_KS) bfr[i] = rdtr(adB[i], std::integral_constant<int, P * STAGE_BYTES + s * 8192>{});
        else bfr[i] = rd128(adB[s], std::integral_constant<int, P * STAGE_BYTES + i * 2048>{});
      });
      // operands swapped on purpose: D[row = n][col = m] -> each lane owns 4 consecutive n of one row m
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<F16 ? 2 : 0>(bfr[j], af[i], acc[i][j]);
    });
    if constexpr (!LAST) {
      wait_vmcnt<0>();      // the next K-tile has landed (this wave's pieces) and this wave's reads of the current one have returned
      __builtin_amdgcn_s_barrier();
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  stage(I0{});
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  for (int pr = 0; pr < nk / 2 - 1; ++pr) {
    ktile(I0{}, std::false_type{});
    ktile(I1{}, std::false_type{});
  }
  ktile(I0{}, std::false_type{});
  ktile(I1{}, std::true_type{});

  if constexpr (SPLITK) {
    __syncthreads();   // every wave is done reading the ring -> smem is free
    if (!splitk_combine<16, 256>(reinterpret_cast<f32x4(&)[16]>(acc), g.ws, g.counters, blockIdx.x, g.splitk, blockIdx.y, tid, smem)) return;
  }
  constexpr int ESTRIDE = 128 * 4 + 16;
  // Every wave is done READING the ring -- its LDS reads have RETURNED, not merely been issued: s_barrier does not wait for lgkmcnt, and
  // the compiler is free to sink the last K-tile's MFMAs (register-only) below the barrier together with the waits it puts in front of
  // them.  Without this wait a wave could pass the barrier with fragment reads of the last K-tile in flight while another wave already
  // scattered its accumulators into the same LDS bytes: the first build of this kernel did exactly that -- rare per launch, but a
  // training step has hundreds, and ~half of the bench processes ended with a NaN loss (profiles/r06_g2_epilogue_race.txt).
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  static_for<0, 16>([&](auto idx) {
    constexpr int i = decltype(idx)::value >> 2, j = decltype(idx)::value & 3;
    const int row = wr * 64 + i * 16 + (lane & 15);
    const int col = wc * 64 + j * 16 + 4 * (lane >> 4);
    *(f32x4*)(smem + row * ESTRIDE + col * 4) = acc[i][j];
  });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const DropParams dp = with_salt(g.e.drop);
#pragma unroll AFFT_G128_EPI_UNROLL
  for (int it = 0; it < 8; ++it) {       // 4 rows per wave-iteration, 16 lanes (8 columns each) per row
    const int row = wave * 32 + it * 4 + lane / 16;
    const int c8 = lane % 16;
    const f32x4 t0 = *(const f32x4*)(smem + row * ESTRIDE + c8 * 32);
    const f32x4 t1 = *(const f32x4*)(smem + row * ESTRIDE + c8 * 32 + 16);
    float o[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
    epilogue8(g.e, dp, m0 + row, n0 + 8 * c8, o);
  }
}

#ifndef AFFT_G2
#define AFFT_G2 1      // 0: every 128x128 launch on gemm_bf16_kernel (A/B builds)
#endif
bool g2_shape(int M, int N, int K, int splitk) {
  return AFFT_G2 && M % 128 == 0 && N % 128 == 0 && K % (BK * splitk) == 0 && (K / BK / splitk) % 2 == 0 && K / BK / splitk >= 2;
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
  const DropParams dp = with_salt(g.e.drop);
  static_for<0, 4>([&](auto idx) {
    constexpr int gq = decltype(idx)::value;
    const int n = n0 + wc * 32 + 8 * gq + 4 * (lane >> 5);
    float v[4] = {acc[4 * gq + 0], acc[4 * gq + 1], acc[4 * gq + 2], acc[4 * gq + 3]};
    epilogue4(g.e, dp, m, n, v);
  });
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

int g_splitk_mode = 1;   // 128x128 kernel: 0 off, 1 auto, 2 / 4: force that many slices wherever the shape allows (tests, tuning)
// CUs of the current device (tile-shape choice): asked once per device ordinal
int cu_count() {
  static std::atomic<int> cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 256; }
  int n = cached[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
    cached[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

// Split-K workspace: provided by the caller per launch (afft_gemm_t.workspace, private to the stream): AFFT_GEMM_WS_HEADER
// bytes of arrival counters (zero between launches) followed by the fp32 partial tiles.  Nothing is allocated here.
constexpr int kMaxSplitTiles = AFFT_GEMM_WS_HEADER / (int)sizeof(int);

// ---- measurement hook: event pairs around fast-path launches while a trace is open (afft_gemm_trace_begin / _end)
struct TraceRec { afft_gemm_trace_rec_t r; hipEvent_t a, b; };
std::mutex g_trace_mu;
std::vector<TraceRec>* g_trace = nullptr;
size_t g_trace_cap = 0;

int g_variant = 0;  // 0 auto, 1 = 128x128 tile, 2 = 256x128 tile, 3 = 256x256 ping-pong (tuning / tests)

template <int WM, int WN, int STAGES, bool A_KS, bool B_KS, bool SPLITK, int X3 = 0>
int launch_fast(GemmFast& g, hipStream_t stream) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr size_t ring = (size_t)STAGES * (BM + BN) * BK * 2, epi = (size_t)BM * (BN * 4 + 16);
  constexpr size_t lds = ring > epi ? ring : epi;
  g.tiles_m = (g.e.M + BM - 1) / BM;
  g.tiles_n = (g.e.N + BN - 1) / BN;
  if constexpr (WM == 2 && WN == 2 && STAGES == 2 && (X3 == 0 || X3 == 2)) {
    const bool fits32 = (A_KS ? (int64_t)(g.K + 8) * g.lda * 2 : (8 * g.lda + g.K) * 2) < (1LL << 32) &&
                        (B_KS ? (int64_t)(g.K + 8) * g.ldb * 2 : (8 * g.ldb + g.K) * 2) < (1LL << 32);      // the running K offset is a 32-bit VGPR
    const bool one_segment = X3 == 0 || g.K == g.nk_seg * BK;      // X3 = 2: only the one-pass form (split3 = 4) -- the kernel has no operand planes
    if (fits32 && one_segment && g2_shape(g.e.M, g.e.N, g.K, g.splitk)) {      // whole tiles, even K-tile count per slice: the steady-state kernel
      auto k2 = gemm_bf16_g2_kernel<A_KS, B_KS, SPLITK, X3 == 2>;
      static std::atomic<uint64_t> attr2_done{0};
      if (int rc = afft_ensure_dynamic_lds(reinterpret_cast<const void*>(k2), lds, &attr2_done)) return rc;
      hipLaunchKernelGGL(k2, dim3(g.tiles_m * g.tiles_n, g.splitk), dim3(256), lds, stream, g);
      AFFT_LAUNCH_CHECK();
      return 0;
    }
  }
  auto kern = gemm_bf16_kernel<WM, WN, STAGES, A_KS, B_KS, SPLITK, X3>;
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = afft_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, &attr_done)) return rc;
  hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n, g.splitk), dim3(64 * WM * WN), lds, stream, g);
  AFFT_LAUNCH_CHECK();
  return 0;
}

int choose_variant(int M, int N, int K, bool A_KS, bool B_KS);
bool bd_packed_wins(int M, int N, int K);
extern int g_bd_mode;

// K-slices afft_gemm will use for a fast-path problem (1 = no split-K)
int choose_splitk(int variant, int M, int N, int K) {
  if (!g_splitk_mode) return 1;
  const int nk = K / BK;
  if (variant == 3) return 1;     // 256x256 tiles never split K (stream-K was built, measured slower on this power-limited part and removed: profiles/HISTORY.md)
  // 128x128 tiles (2 workgroups per CU = 512 slots): a grid that leaves slots empty is bound by the LDS fill rate of the CUs
  // that have a workgroup -- more workgroups pulling is the lever.  Cut K so that tiles x slices approaches 512, keeping at
  // least 16 K-steps (K = 1024) per slice: <= 128 tiles -> 2 slices (4 when K >= 6144), and -- round 2, measured on the
  // K = 5120 weight gradients of the d = 1024 and d = 2048 models (profiles/r02_gemm_ek100_shapes.txt: 3072x1024x5120
  // 70 -> 49 us, 4096x1024x5120 74 -> 54 us, 1024x1024x5120 39 -> 30 us) -- <= 256 tiles with K >= 4096 -> 2 slices,
  // <= 64 tiles with K >= 4096 -> 4.
  const int64_t t1 = (int64_t)((M + 127) / 128) * ((N + 127) / 128);
  int s = 1;
  if (g_splitk_mode == 1) {
    if (t1 <= 128 && nk >= 32) s = (nk >= 96 && t1 * 4 <= 512) ? 4 : 2;
    if (t1 <= 64 && nk >= 64) s = 4;
    if (t1 > 128 && t1 <= 256 && nk >= 64) s = 2;
  }
  else if (t1 * g_splitk_mode <= kMaxSplitTiles && nk >= 2 * g_splitk_mode) s = g_splitk_mode;
  while (s > 1 && nk % s != 0) s >>= 1;
  return (s > 1 && t1 <= kMaxSplitTiles) ? s : 1;
}

// bytes of partial tiles (without the counter header) split-K needs for this problem; 0 = it does not split
int64_t splitk_bytes(int variant, int M, int N, int K, int* slices) {
  const int s = variant >= 4 ? 1 : choose_splitk(variant, M, N, K);
  if (slices) *slices = s;
  if (s <= 1) return 0;
  const int64_t tiles = (int64_t)((M + 127) / 128) * ((N + 127) / 128);
  return tiles * s * 128 * 128 * (int64_t)sizeof(float);
}

template <bool A_KS, bool B_KS>
int launch_layout_impl(GemmFast& g, hipStream_t stream, const afft_gemm_t* d);

template <bool A_KS, bool B_KS>
int launch_layout(GemmFast& g, hipStream_t stream, const afft_gemm_t* d) {
  if (!g_trace) return launch_layout_impl<A_KS, B_KS>(g, stream, d);
  TraceRec t = {};
  {
    std::lock_guard<std::mutex> lk(g_trace_mu);
    if (!g_trace || g_trace->size() >= g_trace_cap) return launch_layout_impl<A_KS, B_KS>(g, stream, d);
  }
  if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) { (void)hipGetLastError(); return launch_layout_impl<A_KS, B_KS>(g, stream, d); }
  (void)hipEventRecord(t.a, stream);
  const int rc = launch_layout_impl<A_KS, B_KS>(g, stream, d);
  (void)hipEventRecord(t.b, stream);
  int variant = choose_variant(g.e.M, g.e.N, g.K, A_KS, B_KS);
  if (!A_KS && !B_KS && d->b_packed && !d->split3 && g_variant == 0 && g.ldb == d->K && bd_packed_wins(g.e.M, g.e.N, g.K)) variant = 10;
  if (variant == 3 && afft_gemm_pp2_takes(g.e.M, g.e.N, d->K, d->split3)) variant = 13;      // the steady-state 256x256 kernel (gemm_bf16_pp2_kernel)
  t.r = afft_gemm_trace_rec_t{g.e.M, g.e.N, d->K, A_KS, B_KS, variant, g.splitk, d->split3, d->sgd != nullptr, 0.f};
  std::lock_guard<std::mutex> lk(g_trace_mu);
  if (g_trace) g_trace->push_back(t);
  return rc;
}

// "B direct" kernel on a fragment-packed weight (afft_gemm_t.b_packed, gemm_bd.hip) against the 256x256 ping-pong kernel: a cost
// model in us fitted to both kernels alone on one MI355X (profiles/r04_gemm_bd.txt).  Ping-pong: one workgroup per CU, a K-tile
// of a full round costs ~1.9 us, of a last round with <= 160 busy CUs 1.4 us (the part is power-limited), + 10 us; B-direct
// 160x256 tiles: 1.09 us per K-tile and round + 8.8 us per round (prologue drain + epilogue, one workgroup per CU and no
// overlap between tiles).  Taken only when its grid is ONE round (N = 2048 outputs of M = 5120 rows: 256 tiles on 256 CUs where
// 256-row tiles give 160): inside the model's forward pass the multi-round shapes measured slower than the ping-pong kernel
// (fc1 with its GELU epilogue 229 vs 201 us: four rounds of epilogues with nothing beside them), the one-round shapes faster
// (fc2 162 vs 184 us, projection 64 vs 67 us; profiles/r04_gemm_bd.txt "in the step").
// AFFT_BD_MODE: 0 = never, 1 = by the model (default), 2 = whenever the shape is eligible.
int g_bd_mode = [] { const char* e = getenv("AFFT_BD_MODE"); return e ? atoi(e) : 1; }();   // declared above
bool bd_packed_wins(int M, int N, int K) {
  if (g_bd_mode == 0 || N % 16 != 0 || N < 256 || K % 64 != 0) return false;
  const int ncu = cu_count(), nk = K / BK;
  const int64_t t160 = (int64_t)((M + 159) / 160) * ((N + 255) / 256), t256 = (int64_t)((M + 255) / 256) * ((N + 255) / 256);
  if (t256 < 160) return false;                      // small grids: the 128x128 kernel's territory
  if (g_bd_mode >= 2) return true;
  const int64_t r160 = (t160 + ncu - 1) / ncu, r256 = (t256 + ncu - 1) / ncu;
  if (r160 != 1) return false;
  const int64_t busy = t256 - (r256 - 1) * ncu;
  const double last = 1.4 + 0.5 * (double)std::max<int64_t>(0, busy - 160) / 96.0;
  const double pp_us = nk * ((double)(r256 - 1) * 1.9 + last) + 10.0;
  const double bd_us = (double)r160 * (nk * 1.09 + 8.8);
  return bd_us < pp_us;
}

template <bool A_KS, bool B_KS>
int launch_layout_impl(GemmFast& g, hipStream_t stream, const afft_gemm_t* d) {
  if constexpr (!A_KS && !B_KS) {
    if (d->b_packed && !d->split3 && g_variant == 0 && g.ldb == d->K && bd_packed_wins(g.e.M, g.e.N, g.K)) {
      g.splitk = 1; g.ws = nullptr; g.counters = nullptr;
      g.B = (const bf16_t*)d->b_packed;
      return afft_gemm_launch_bd(1, 1, g, stream);
    }
  }
  const int variant = choose_variant(g.e.M, g.e.N, g.K, A_KS, B_KS);
  g.splitk = 1;
  g.ws = nullptr;
  g.counters = nullptr;
  if (d->split3 == 3) {     // fp16 hi pass + fp8 lo pass: NT on the 256x256 kernel (callers ask afft_gemm_lo8_ok first)
    if constexpr (!A_KS && !B_KS) {
      if (variant == 3) return afft_gemm_launch_pp(A_KS, B_KS, g, stream, 3);
    }
    afft_set_error("afft_gemm: split3 = 3 needs the NT layout and a problem the 256x256 kernel takes (afft_gemm_lo8_ok)");
    return 1;
  }
  if (d->split3 == 2 || d->split3 == 4) {     // fp16 two-pass / one pass (g.K = one segment) (forward layouts only): same tile choice as bf16x3
    if constexpr (!A_KS) {
      if (variant == 3) return afft_gemm_launch_pp(A_KS, B_KS, g, stream, 2);
      // small grids (the predictor's M = B*T rows): split-K over the 2K-long loop -- with two slices one workgroup runs the hi
      // pass of a tile and another its lo pass, and the last to arrive adds them (slice order: bitwise repeatable)
      int s2 = 1;
      const int64_t need2 = splitk_bytes(variant, g.e.M, g.e.N, g.K, &s2);
      if (s2 > 1 && d->workspace && d->workspace_bytes >= need2 + AFFT_GEMM_WS_HEADER) {
        g.counters = (int*)d->workspace;
        g.ws = (float*)((char*)d->workspace + AFFT_GEMM_WS_HEADER);
        g.splitk = s2;
        return launch_fast<2, 2, 2, A_KS, B_KS, true, 2>(g, stream);
      }
      return launch_fast<2, 2, 2, A_KS, B_KS, false, 2>(g, stream);
    } else {
      afft_set_error("afft_gemm: the fp16 two-pass mode (split3 = 2) is built for the forward layouts only (A k-contiguous)");
      return 1;
    }
  }
  if (d->split3) {     // bf16x3: 256x256 tiles once the grid fills the chip, else 128x128; no split-K
    if (variant == 3) return afft_gemm_launch_pp(A_KS, B_KS, g, stream, 1);
    return launch_fast<2, 2, 2, A_KS, B_KS, false, 1>(g, stream);
  }
  int s = 1;
  const int64_t need = splitk_bytes(variant, g.e.M, g.e.N, g.K, &s);
  if (s > 1 && d->workspace && d->workspace_bytes >= need + AFFT_GEMM_WS_HEADER) {   // else: run unsplit
    g.counters = (int*)d->workspace;
    g.ws = (float*)((char*)d->workspace + AFFT_GEMM_WS_HEADER);
    g.splitk = s;
  }
  if (variant == 3) return afft_gemm_launch_pp(A_KS, B_KS, g, stream, 0);
#ifdef AFFT_EXPERIMENT_Q4
  if (variant == 11) { g.splitk = 1; g.ws = nullptr; g.counters = nullptr; return afft_gemm_launch_q4(g, stream); }
#endif
  if (variant >= 7 && variant <= 10) return afft_gemm_launch_bd(variant == 8 || variant == 10, variant >= 9, g, stream);
  if (variant == 4) return launch_fast<2, 2, 4, A_KS, B_KS, false>(g, stream);
  if (g.splitk > 1) return launch_fast<2, 2, 2, A_KS, B_KS, true>(g, stream);
  return launch_fast<2, 2, 2, A_KS, B_KS, false>(g, stream);
}

// B-direct kernels (gemm_bd.hip): k-contiguous operands only, whole 16-column blocks
bool bd_ok(int N, bool A_KS, bool B_KS) { return !A_KS && !B_KS && N >= 16 && N % 16 == 0; }

int choose_variant(int M, int N, int K, bool A_KS, bool B_KS) {
  if (g_variant >= 7 && g_variant <= 10) { if (bd_ok(N, A_KS, B_KS)) return g_variant; }
#ifdef AFFT_EXPERIMENT_Q4
  else if (g_variant == 11) { if (!A_KS && !B_KS) return 11; }      // four-quadrant kernel: k-contiguous operands only
#endif
  else if (g_variant != 0) return g_variant;
  // measured (profiles/r01_gemm_variants_bench2.txt): the 256x256 ping-pong kernel (1 workgroup/CU) wins once its
  // grid covers >= ~60 % of the CUs; below that (GPT-2's M = 1024 GEMMs, small weight gradients) two independent
  // 128x128 workgroups per CU win.  The 256x128 3-stage shape (variant 2) never wins and is kept for reference.
  const int64_t t3 = (int64_t)((M + 255) / 256) * ((N + 255) / 256);
  if (t3 < 160) return 1;
  // Both shapes waste the slots of their last, partial round (256 slots of one 256x256 tile, 512 of two 128x128 tiles per
  // CU); per FLOP the big tile is ~1.25x as efficient.  Round 2 (profiles/r02_gemm_ek100_shapes.txt): 5120x4096x1024 is 320
  // big tiles = 1.25 rounds (57 us on 128x128 tiles, 66 us on 256x256), 5120x3072x1024 is 240 = one nearly full round (45
  // vs 38 us); 5120x2048x2048 (160 big tiles) 60 vs 55 us.
  const int64_t t1 = (int64_t)((M + 127) / 128) * ((N + 127) / 128);
  const int ncu = cu_count();
  const double u3 = (double)t3 / ((double)ncu * ((t3 + ncu - 1) / ncu)), u1 = (double)t1 / (2.0 * ncu * ((t1 + 2 * ncu - 1) / (2 * ncu)));
  return u3 * 1.25 >= u1 ? 3 : 1;
}

}  // namespace

extern "C" int afft_gemm_variant_for(int M, int N, int K, int a_kstrided, int b_kstrided) {
  return choose_variant(M, N, K, a_kstrided != 0, b_kstrided != 0);
}

extern "C" int afft_gemm_lo8_ok(int M, int N, int K) {
  return (g_variant == 0 || g_variant == 3) && K % 128 == 0 && K >= 128 && choose_variant(M, N, 2 * K, false, false) == 3 ? 1 : 0;
}

extern "C" int afft_gemm_packed_wanted(int M, int N, int K) {
  return (g_variant == 0 && bd_packed_wins(M, N, K)) ? 1 : 0;
}

extern "C" int64_t afft_gemm_workspace_bytes(int M, int N, int K, int a_kstrided, int b_kstrided) {
  const int v = choose_variant(M, N, K, a_kstrided != 0, b_kstrided != 0);
  const int64_t b = splitk_bytes(v, M, N, K, nullptr);
  return b ? b + AFFT_GEMM_WS_HEADER : 0;
}

extern "C" int afft_gemm_splitk_for(int M, int N, int K, int a_kstrided, int b_kstrided) {
  const int v = choose_variant(M, N, K, a_kstrided != 0, b_kstrided != 0);
  return v >= 4 ? 1 : choose_splitk(v, M, N, K);
}

extern "C" int afft_gemm_trace_begin(int32_t capacity) {
  std::lock_guard<std::mutex> lk(g_trace_mu);
  AFFT_CHECK(!g_trace, "afft_gemm_trace_begin: a trace is already open");
  AFFT_CHECK(capacity > 0, "afft_gemm_trace_begin: capacity must be positive");
  g_trace = new std::vector<TraceRec>();
  g_trace->reserve(capacity);
  g_trace_cap = (size_t)capacity;
  return 0;
}

extern "C" int afft_gemm_trace_end(afft_gemm_trace_rec_t* out, int32_t capacity) {
  std::vector<TraceRec>* tr;
  {
    std::lock_guard<std::mutex> lk(g_trace_mu);
    tr = g_trace;
    g_trace = nullptr;
  }
  if (!tr) { afft_set_error("afft_gemm_trace_end: no trace is open"); return -1; }
  int n = 0;
  for (TraceRec& t : *tr) {
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

extern "C" int afft_set_gemm_splitk(int mode) {
  if (mode != 0 && mode != 1 && mode != 2 && mode != 4) { afft_set_error("afft_set_gemm_splitk: %d is not 0, 1, 2 or 4", mode); return 1; }
  g_splitk_mode = mode;
  return 0;
}

extern "C" int afft_set_gemm_variant(int v) {
#ifdef AFFT_EXPERIMENT_Q4
  if (v == 11) { g_variant = v; return 0; }
#endif
  if (v != 0 && v != 1 && v != 3 && v != 4 && !(v >= 7 && v <= 10)) { afft_set_error("afft_set_gemm_variant: %d is not 0 (auto), 1 (128x128), 3 (256x256 ping-pong), 4 (128x128, 4 stages), 7 / 8 (B-direct 256x256 / 160x256 tiles on NT layouts, B row-major; other layouts as auto), 9 / 10 (the same, B points at a fragment-packed image)", v); return 1; }
  g_variant = v;
  return 0;
}

extern "C" int afft_gemm(const afft_gemm_t* d, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AFFT_CHECK(d != nullptr, "afft_gemm: null descriptor");
  AFFT_CHECK(d->M >= 0 && d->N >= 0 && d->K >= 0, "afft_gemm: negative size");
  AFFT_CHECK(d->dtype == AFFT_F32 || d->dtype == AFFT_BF16, "afft_gemm: bad dtype %d", d->dtype);
  AFFT_CHECK(d->A && d->B && d->out, "afft_gemm: null operand");
  AFFT_CHECK(!d->accumulate || d->out_dtype == AFFT_F32, "afft_gemm: accumulate needs an fp32 output");
  AFFT_CHECK(d->act >= AFFT_ACT_NONE && d->act <= AFFT_ACT_SIGMOID_GATE, "afft_gemm: bad activation %d", d->act);
  AFFT_CHECK(!act_needs_aux(d->act) || d->aux, "afft_gemm: this activation needs aux");
  if (d->M == 0 || d->N == 0) return 0;

  EpiParams e;
  e.M = d->M; e.N = d->N; e.alpha = d->alpha; e.bias = d->bias; e.act = d->act;
  e.aux = d->aux; e.ldaux = d->ldaux; e.aux_dtype = d->aux_dtype;
  e.pre = d->pre; e.ldpre = d->ldpre; e.pre_dtype = d->pre_dtype;
  e.rowscale = d->rowscale; e.residual = d->residual; e.ldres = d->ldres;
  e.accumulate = d->accumulate;
  e.out = d->out; e.ldo = d->ldo; e.out_dtype = d->out_dtype;
  AFFT_CHECK(d->out_dtype >= AFFT_F32 && d->out_dtype <= AFFT_F16, "afft_gemm: bad out_dtype %d", d->out_dtype);
  AFFT_CHECK(!d->out_lo || (d->out_dtype == AFFT_F16 && !d->accumulate && d->out_lo % 8 == 0),
             "afft_gemm: out_lo (two-plane fp16 output) needs out_dtype AFFT_F16, no accumulate, and a 16-byte aligned plane offset");
  e.out_lo = d->out_lo;
  AFFT_CHECK(!d->out_lo8 || (d->out_dtype == AFFT_F16 && !d->out_lo && !d->accumulate && (((uintptr_t)d->out_lo8) & 7) == 0 && d->ldo % 8 == 0),
             "afft_gemm: out_lo8 (fp16 hi + e4m3 lo output) needs out_dtype AFFT_F16, no out_lo, no accumulate, 8-byte aligned rows");
  e.out_lo8 = (unsigned char*)d->out_lo8;
  e.out2 = d->out2; e.ldo2 = d->ldo2; e.out2_dtype = d->out2_dtype;
  AFFT_CHECK(d->drop.p >= 0.f && d->drop.p < 1.f && d->drop.path_p >= 0.f && d->drop.path_p < 1.f, "afft_gemm: dropout p outside [0,1)");
  e.drop = make_drop(&d->drop);
  e.sgd = SgdEpi{nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, 1.f, 0, nullptr, nullptr, nullptr, nullptr};
  if (d->sgd) {
    AFFT_CHECK(d->sgd->p && d->sgd->buf, "afft_gemm: fused update without parameter / momentum buffers");
    AFFT_CHECK(!d->accumulate && !d->bias && d->act == AFFT_ACT_NONE && !d->residual && !d->rowscale && !d->pre && !d->out2 &&
               d->drop.p == 0.f && d->drop.path_p == 0.f, "afft_gemm: a fused update takes the plain product (no other epilogue stage)");
    AFFT_CHECK(!d->sgd->p_pk16 || (d->ldo % 32 == 0 && d->M % 16 == 0 && ((uintptr_t)d->sgd->p_pk16 & 15) == 0),
               "afft_gemm: a fragment-packed image needs a [16 a, 32 b] weight");
    e.sgd = SgdEpi{d->sgd->p, d->sgd->buf, (bf16_t*)d->sgd->p_bf16, d->sgd->lr, d->sgd->mom, d->sgd->wd, d->sgd->gscale,
                   d->sgd->first_step, (bf16_t*)d->sgd->p_pk16, (bf16_t*)d->sgd->p_f16, (unsigned char*)d->sgd->p_f8, d->sgd->ok};
  }
  auto ok4 = [](const void* p, int64_t ld, int dtype) {
    if (!p) return true;
    const uintptr_t align = dtype == AFFT_F32 ? 16 : 8;
    return (ld % 4 == 0) && ((((uintptr_t)p) & (align - 1)) == 0);
  };
  auto ok8 = [](const void* p, int64_t ld) { return !p || ((ld % 8 == 0) && ((((uintptr_t)p) & 15) == 0)); };
  e.vec8 = ok8(d->out, d->ldo) && ok8(d->out2, d->ldo2) && ok8(d->pre, d->ldpre) && ok8(d->aux, d->ldaux) &&
           ok8(d->residual, d->ldres) && ok8(d->bias, 8) && ok8(e.sgd.p, d->ldo) && ok8(e.sgd.buf, d->ldo) && ok8(e.sgd.p16, d->ldo) && ok8(e.sgd.p16h, d->ldo);
  e.vec4 = ok4(e.sgd.p, d->ldo, AFFT_F32) && ok4(e.sgd.buf, d->ldo, AFFT_F32) && ok4(e.sgd.p16, d->ldo, AFFT_BF16) && ok4(e.sgd.p16h, d->ldo, AFFT_BF16) &&
           ok4(d->out, d->ldo, d->out_dtype) && ok4(d->out2, d->ldo2, d->out2_dtype) &&
           ok4(d->pre, d->ldpre, d->pre_dtype) && ok4(d->aux, d->ldaux, d->aux_dtype) &&
           ok4(d->residual, d->ldres, AFFT_F32) && ok4(d->bias, 4, AFFT_F32);

  // operand layouts of the MFMA fast path: k-contiguous (unit stride along k) or k-strided (unit stride along m / n)
  AFFT_CHECK(!d->b_packed || (((uintptr_t)d->b_packed & 15) == 0), "afft_gemm: b_packed must be 16-byte aligned");
  const bool a_kc = d->a_cs == 1, a_ks = d->a_rs == 1;
  const bool b_kc = d->b_rs == 1, b_ks = d->b_cs == 1;
  bool fast = d->dtype == AFFT_BF16 && d->K >= BK && d->K % BK == 0 && aligned16(d->A) && aligned16(d->B);
  const int64_t lda = a_kc ? d->a_rs : d->a_cs, ldb = b_kc ? d->b_cs : d->b_rs;
  fast = fast && (a_kc || a_ks) && (b_kc || b_ks) && lda % 8 == 0 && ldb % 8 == 0 && lda >= 8 && ldb >= 8;
  fast = fast && !(a_ks && !a_kc && b_kc && !b_ks);   // (A k-strided, B k-contiguous) does not occur on the path

  AFFT_CHECK(d->split3 >= 0 && d->split3 <= 4, "afft_gemm: split3 is 0, 1 (bf16x3), 2 (fp16 two-pass), 3 (fp16 + fp8 lo pass) or 4 (one fp16 pass)");
  AFFT_CHECK(d->split3 != 3 || (d->a8 && d->b8 && d->K % 128 == 0 && d->a8_ld % 16 == 0 && d->b8_ld % 16 == 0 && d->a8_ld >= d->K && d->b8_ld >= d->K &&
                                aligned16(d->a8) && aligned16(d->b8)),
             "afft_gemm: split3 = 3 needs the two e4m3 byte planes (16-byte aligned rows) and K %% 128 == 0");
  AFFT_CHECK(!d->split3 || fast, "afft_gemm: split3 needs 16-bit planes in a fast-path layout (K %% 64 == 0, 16-byte aligned rows)");
  if (fast) {
    GemmFast g;
    g.A = (const bf16_t*)d->A; g.B = (const bf16_t*)d->B;
    g.lda = lda; g.ldb = ldb;
    g.K = d->split3 == 3 ? d->K + d->K / 2 : d->split3 == 2 ? 2 * d->K : d->split3 == 1 ? 3 * d->K : d->K;      // in 64-wide K-tiles of 128 B per row
    g.nk_seg = d->K / BK;
    g.a_lo = d->a_lo; g.b_lo = d->b_lo;
    g.A8 = (const bf16_t*)d->a8; g.B8 = (const bf16_t*)d->b8; g.lda8 = d->a8_ld / 2; g.ldb8 = d->b8_ld / 2;
    g.e = e;
    // bf16-operand kernels: activation math on v_exp_f32 / v_rcp_f32 (common.h: AFFT_ACT_FAST, |error| ~2e-7) instead of the
    // library's erff / tanhf, whose 45-80 instructions per element showed as +64..140 us per launch; AFFT_EXACT_ACT=1 keeps them
    static const bool exact_act = [] { const char* v = getenv("AFFT_EXACT_ACT"); return v && v[0] == '1'; }();
    if (g.e.act != AFFT_ACT_NONE && !exact_act) g.e.act |= AFFT_ACT_FAST;
    const bool A_KS = !a_kc, B_KS = !b_kc;
    if (!A_KS && !B_KS) return launch_layout<false, false>(g, stream, d);
    if (!A_KS && B_KS) return launch_layout<false, true>(g, stream, d);
    return launch_layout<true, true>(g, stream, d);
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
