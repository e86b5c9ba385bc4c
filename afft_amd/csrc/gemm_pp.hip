// 256x256x64 ping-pong bf16 MFMA GEMM (see the comment above the kernel). Split from gemm.hip to keep build times down.
#include "gemm_tiles.h"

using namespace afft_gemm_detail;

#ifndef AFFT_PP_DMA_FIRST
#define AFFT_PP_DMA_FIRST 1   // issue the LDS-DMA before the fragment reads of an L segment: NN +4 %, TN +1-5 %, NT +-1 % with the
                              // scalar-base staging (with per-lane pointers it had cost the k-strided layouts 30 %)
#endif
#ifndef AFFT_PP_PRIO
#define AFFT_PP_PRIO 1        // experiment: 1 = MFMA segments at raised priority, 0 = no priority change, 2 = L segments raised
#endif
#ifndef AFFT_PP_CLAMP
#define AFFT_PP_CLAMP 2       // 2 = branch-free wait: 256-byte dummy LDS-DMA past the end of K (see issue()); 1 = re-read the last K-tile; 0 = guarded issue + run-time wait selection
#endif
#ifndef AFFT_PP_EPI_UNROLL
#define AFFT_PP_EPI_UNROLL 2  // row steps of the epilogue in flight per thread: 2 takes 8-12 us off the GELU epilogues of a 5120x8192 output, 4 loses again
#endif
#ifndef AFFT_PP_DMA_IN_C
#define AFFT_PP_DMA_IN_C 0    // experiment: issue a phase's two LDS-DMA instructions INSIDE its MFMA segment (after AFFT_PP_DMA_AT MFMAs: the
#endif                        // matrix pipe has queued work, the VMEM issue is free) instead of in the L segment, the longer of the two
#ifndef AFFT_PP_DMA_AT
#define AFFT_PP_DMA_AT 4
#endif
#ifndef AFFT_PP_DIAG
#define AFFT_PP_DIAG 0    // diagnostic builds only (wrong results): 1 = no fragment reads, 2 = no LDS-DMA in the loop, 4 = no MFMA,
                          // 8 = no global accesses in the epilogue, 16 = no epilogue at all
#endif

#if !defined(AFFT_DIAG_BUILD) && (AFFT_PP_DIAG || defined(AFFT_PP_SAMETILE) || defined(AFFT_PP_STAMP) || AFFT_HANDOFF_DIAG)
#error "AFFT_PP_DIAG / AFFT_PP_SAMETILE / AFFT_PP_STAMP / AFFT_HANDOFF_DIAG give wrong results or change the kernel's signature: diagnostic builds only (make DIAG=1 DIAGFLAGS=-D...)"
#endif

namespace {

// ---------------------------------------------------------------------------------------------
// 256x256x64 "ping-pong" kernel: 8 waves, one workgroup per CU, 128 KiB LDS.
//
// Why: a 128x128 tile needs 1 byte of L2->LDS fill per 64 FLOP and the per-CU LDS-fill path tops out near
// 70 GB/s, which caps that shape around 1.1 PFLOP/s (profiles/r01_gemm_pmc_*.txt); 256x256 halves the bytes.
// With one workgroup per CU the two waves that share a SIMD must not do the same thing at the same time, so the
// 8 waves form two groups (waves 0-3 / 4-7, one of each per SIMD) that run the SAME phase program one barrier
// apart: while one group issues its 16 MFMAs of a phase, the other reads the next phase's fragments from LDS and
// issues LDS-DMA.  Per K-tile (64 deep) a wave runs 4 phases = the 4 quadrants of its 128x64 output, and its rows /
// columns are interleaved over the two 128-row halves of the A and B tiles, so a K-tile is consumed half-tile by
// half-tile (A0,B0 | B1 | A1 | B0 of the next K-tile) and staged half-tile by half-tile, LEAD = 6 half-tiles (12 LDS-DMA
// instructions per wave) ahead, behind a counted s_waitcnt vmcnt(6) that is the same in every phase: past the end of K
// the stream goes on with 256-byte dummy transfers (AFFT_PP_CLAMP), so no phase needs a run-time choice of the wait.
// What bounds it (tools/fill_bench.hip): not the fill -- LDS-DMA sustains 15 TB/s beside 1.9 PFLOP/s of
// register-operand MFMAs -- but the LDS -> register leg (the MFMAs wait for their ds_read fragments; a bare loop with the
// same reads reaches 1.4 PFLOP/s) and the power-limited clock (1.85 GHz inside the loop).
//
//   stream of half-tiles (16 KiB each): index m = 4*kt + q, q: 0 = A rows 0-127, 1 = B rows 0-127,
//   2 = B rows 128-255, 3 = A rows 128-255; ring slot = ((kt & 1) * 4 + q).
//   phase n = 4*kt + p:  L(n): issue half-tile n + LEAD, read the fragments phase n needs, wait until half-tile
//   n + 3 has landed (this wave's pieces), s_barrier;  C(n): 16 MFMAs, s_barrier.
//   slot 2n:   group 0 runs L(n),  group 1 runs C(n-1)      slot 2n+1: group 0 runs C(n),  group 1 runs L(n)
//   RAW: half-tile m is first read in L(m - (m&3 ? 1 : 0) ...) >= two slots after every wave's wait for it;
//   WAR: half-tile m overwrites m - 8.  Half-tile h is read in L(h) (q = 0), L(h-1) (q = 2, 3) or L(h-2) (q = 1), and a wave only
//   waits for those LDS reads at the head of its next C segment -- group 1 holds h's data in registers when the barrier that
//   ends slot 2h+2 falls (q = 0; earlier for the others).  The overwriting transfer is issued in L(h + 8 - LEAD), by group 0 in
//   slot 2(h + 8 - LEAD): LEAD = 6 -> slot 2h+4, two barriers later.  LEAD = 7 (rounds 2-3) put it in slot 2h+2, the very slot
//   in which group 1 is still waiting for its reads of A rows 0-127: ordered by latency alone, and once in a few thousand
//   launches inside the training step (LDS busy with a co-resident kernel of the other stream) the transfer won -- a few rows
//   of one tile came out with one K-tile of the wrong operand (profiles/r04_pp_war_race.txt).  LEAD = 6 is also as fast or faster.
#ifdef AFFT_PP_STAMP   // diagnostic build only: per-segment cycle sums of workgroup 0 (see tools/pp_stamp.py)
unsigned long long* g_pp_stamp = nullptr;
#define STAMP(var) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); } while (0)
#else
#define STAMP(var) do { } while (0)
#endif

// Epilogue of the 256x256 kernels through LDS (the K-loop's ring is free): two passes of 128 rows.  Accumulators are scattered into
// an fp32 [128][256] image (row stride 1040 B = 1 KiB + 16 B, so the 16 rows a lane group writes fall in 16 different 16-byte
// slots), then every wave walks 16 whole rows: one conflict-free 16-byte LDS read per lane and fully coalesced global accesses
// (bias, residual, pre-activation, outputs) in a runtime loop -- no 32-fold unrolled epilogue, no 32-byte store segments.
__device__ __forceinline__ void pp_epilogue(const GemmFast& g, char* smem, f32x4 (&acc)[2][2][4][2], int m0, int n0, int lane, int wave,
                                            int gp, int wc) {
  constexpr int ESTRIDE = 1040;
  const DropParams dp = with_salt(g.e.drop);
  if (AFFT_PP_DIAG & 16) return;
  static_for<0, 2>([&](auto ihc) {
    constexpr int ih = decltype(ihc)::value;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS reads of the ring have RETURNED (s_barrier alone does not say so: gemm.hip, g2 kernel)
    __builtin_amdgcn_s_barrier();   // pass 0: every wave is done with the ring; pass 1: pass 0 has been read back
    static_for<0, 16>([&](auto idx) {
      constexpr int v = decltype(idx)::value;
      constexpr int jh = v >> 3, i = (v >> 1) & 3, j = v & 1;
      const int row = gp * 64 + i * 16 + (lane & 15);
      const int col = jh * 128 + wc * 32 + j * 16 + 4 * (lane >> 4);
      *(f32x4*)(smem + row * ESTRIDE + col * 4) = acc[ih][jh][i][j];
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll AFFT_PP_EPI_UNROLL
    for (int rr = 0; rr < 8; ++rr) {     // two rows per step: a lane owns 8 consecutive columns (16-byte bf16 stores)
      const int row = wave * 16 + rr * 2 + (lane >> 5);
      const int c8 = lane & 31;
      const f32x4 t0 = *(const f32x4*)(smem + row * ESTRIDE + c8 * 32);
      const f32x4 t1 = *(const f32x4*)(smem + row * ESTRIDE + c8 * 32 + 16);
      float o[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
      if (!(AFFT_PP_DIAG & 8)) epilogue8(g.e, dp, m0 + ih * 128 + row, n0 + 8 * c8, o);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  });
}

// X3: 1 = bf16x3 operand planes, 2 = fp16 planes, two passes (gemm_tiles.h: seg_operands, mfma16).
template <bool A_KS, bool B_KS, int X3 = 0>
__global__ __launch_bounds__(512, 2) void gemm_bf16_pp_kernel(const GemmFast g
#ifdef AFFT_PP_STAMP
    , unsigned long long* stamp_out
#endif
) {
  constexpr int HB = 128 * BK * 2;   // half-tile bytes (16 KiB)
#ifndef AFFT_PP_LEAD
#define AFFT_PP_LEAD 6
#endif
  constexpr int LEAD = AFFT_PP_LEAD;             // half-tiles of look-ahead of the LDS-DMA stream
#ifndef AFFT_PP_ALLOW_RACY_LEAD                  // (diagnostic builds reproduce the round-3 schedule with -DAFFT_PP_LEAD=7 -DAFFT_PP_ALLOW_RACY_LEAD)
  static_assert(LEAD >= 3 && LEAD <= 6, "LEAD = 7 refills a ring slot in the slot in which its last reader still waits for its LDS reads (see WAR above)");
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gp = wave >> 2, wc = wave & 3;
  const int lane = lane0;
  int tm, tn;
  tile_coords(g.tiles_m, g.tiles_n, blockIdx.x, tm, tn);
  const int m0 = tm * 256, n0 = tn * 256;
#ifdef AFFT_PP_SAMETILE   // diagnostic build only: every workgroup stages tile (0,0) -> all L2 hits (results are wrong)
  const int m0l = 0, n0l = 0;
#else
  const int m0l = m0, n0l = n0;
#endif
  const int M = g.e.M, N = g.e.N;
  const int nk = g.K / BK, NH = 4 * nk; (void)NH;

  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 aF[4][2], bF[2][2][2];   // A fragments of the live half; B fragments of both halves
  bf16x16 aP[4], bP[2][2];        // X3 = 3: the same as pairs (k-substeps 0 | 1 in one 8-register tuple, see gemm_tiles.h)
  (void)aP; (void)bP;
  if (AFFT_PP_DIAG & 1) {
    for (int i = 0; i < 4; ++i) for (int s = 0; s < 2; ++s) for (int e = 0; e < 8; ++e) aF[i][s][e] = (short)(lane * 37 + i);
    for (int a = 0; a < 2; ++a) for (int j = 0; j < 2; ++j) for (int s = 0; s < 2; ++s) for (int e = 0; e < 8; ++e) bF[a][j][s][e] = (short)(lane * 11 + j);
  }

  // LDS-DMA staging (gemm_tiles.h): one lane-offset VGPR per operand, scalar bases
  const unsigned lds_wave = lds_addr(smem) + wave * 1024;
  const unsigned lda2 = (unsigned)(g.lda * 2), ldb2 = (unsigned)(g.ldb * 2);
  const LaneOffsets lo = lane_offsets(wave, lane);
  const unsigned voffA = A_KS ? lo.ks_row * lda2 + lo.ks_c16 : lo.kc_row * lda2 + lo.kc_chunk16;
  const unsigned voffB = B_KS ? lo.ks_row * ldb2 + lo.ks_c16 : lo.kc_row * ldb2 + lo.kc_chunk16;
  const unsigned lda8_2 = X3 == 3 ? (unsigned)(g.lda8 * 2) : 0u, ldb8_2 = X3 == 3 ? (unsigned)(g.ldb8 * 2) : 0u;
  const unsigned voffA8 = lo.kc_row * lda8_2 + lo.kc_chunk16, voffB8 = lo.kc_row * ldb8_2 + lo.kc_chunk16;
  (void)voffA8; (void)voffB8;
  bool in_loop = false; (void)in_loop;
  auto issue = [&](int m, int q) {   // q = m & 3 (compile-time at every call site)
#if AFFT_PP_CLAMP == 2
    // past the end of K every wave still issues its two vmcnt-counted operations per half-tile, but as 256-byte dummies
    // (first bytes of A into the dead ring slot): the number of LDS-DMA instructions in flight is the same in every phase
    // -> one constant counted wait instead of a run-time selection, at 1/4 of the overshoot traffic of re-reading a K-tile
    if (m >= NH) {
      const unsigned dd = (((m >> 2) & 1) * 4 + q) * HB;
      glds4((const char*)g.A, (unsigned)lane * 4, lds_wave, dd);
      glds4((const char*)g.A, (unsigned)lane * 4, lds_wave, dd + 256);
      return;
    }
    const int kt = m >> 2;
#elif AFFT_PP_CLAMP
    // past the end of K the stream re-reads the last K-tile into the (dead) ring slot: no guard, and the same number of
    // LDS-DMA instructions in flight in every phase -> one constant counted wait instead of a scalar branch chain
    const int kt = min(m >> 2, nk - 1);
#else
    if (m >= NH) return;
    const int kt = m >> 2;
#endif
    if ((AFFT_PP_DIAG & 2) && in_loop) return;
    const unsigned dst = (((m >> 2) & 1) * 4 + q) * HB;      // bytes into the ring; lds_wave carries the ring's address
    int k0; const bf16_t *Ap, *Bp;
    seg_operands<X3>(g, kt, k0, Ap, Bp);
    const bool lo8 = X3 == 3 && kt >= g.nk_seg;      // the fp8 segment: byte planes, their own row pitches (one staging call either way)
    if (q == 0 || q == 3) {
      const int r0 = m0l + (q == 3 ? 128 : 0);
      if constexpr (A_KS) stage_ks<8, 2>(Ap, g.lda, lda2, voffA, lo, r0, k0, dst, wave, lds_wave);
      else stage_kc<8, 2>(Ap, lo8 ? g.lda8 : g.lda, lo8 ? lda8_2 : lda2, lo8 ? voffA8 : voffA, lo, r0, M, k0, dst, wave, lds_wave);
    } else {
      const int c0 = n0l + (q == 2 ? 128 : 0);
      if constexpr (B_KS) stage_ks<8, 2>(Bp, g.ldb, ldb2, voffB, lo, c0, k0, dst, wave, lds_wave);
      else stage_kc<8, 2>(Bp, lo8 ? g.ldb8 : g.ldb, lo8 ? ldb8_2 : ldb2, lo8 ? voffB8 : voffB, lo, c0, N, k0, dst, wave, lds_wave);
    }
  };
  auto load_a = [&](int kt, int ih) {
    if (AFFT_PP_DIAG & 1) return;
    const char* base = smem + ((kt & 1) * 4 + (ih ? 3 : 0)) * HB;
    if constexpr (X3 == 3) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        aP[i] = frag_pair(frag_kc(base, gp * 64 + i * 16 + (lane & 15), lane >> 4), frag_kc(base, gp * 64 + i * 16 + (lane & 15), 4 + (lane >> 4)));
      return;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (A_KS) aF[i][s] = frag_ks<128>(base, 32 * s, gp * 4 + i, lane);
        else aF[i][s] = frag_kc(base, gp * 64 + i * 16 + (lane & 15), s * 4 + (lane >> 4));
      }
  };
  auto load_b = [&](int kt, int jh, auto slotc) {   // B half jh of K-tile kt -> fragment slot
    constexpr int slot = decltype(slotc)::value;
    if (AFFT_PP_DIAG & 1) return;
    const char* base = smem + ((kt & 1) * 4 + 1 + jh) * HB;
    if constexpr (X3 == 3) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        bP[slot][j] = frag_pair(frag_kc(base, wc * 32 + j * 16 + (lane & 15), lane >> 4), frag_kc(base, wc * 32 + j * 16 + (lane & 15), 4 + (lane >> 4)));
      return;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (B_KS) bF[slot][j][s] = frag_ks<128>(base, 32 * s, wc * 2 + j, lane);
        else bF[slot][j][s] = frag_kc(base, wc * 32 + j * 16 + (lane & 15), s * 4 + (lane >> 4));
      }
  };
  // end of an L segment: this wave's pieces of every half-tile <= n + 3 have landed
  unsigned long long sL = 0, sW = 0, sB1 = 0, sC = 0, sB2 = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
  (void)sL; (void)sW; (void)sB1; (void)sC; (void)sB2; (void)t0; (void)t1; (void)t2; (void)t3; (void)t4; (void)t5;
  auto wait_then_barrier = [&](int n) {
    STAMP(t1);
#if AFFT_PP_CLAMP
    const int out = LEAD - 3 - (AFFT_PP_DMA_IN_C ? 1 : 0); (void)n;    // DMA in the C segment: the newest half-tile is issued one segment later
#else
    const int last = min(n + LEAD, NH - 1);
    const int out = last - (n + 3);           // half-tiles allowed to stay in flight
#endif
    // no lgkmcnt here: the LDS reads of this segment only have to be back before this wave's own MFMAs (the
    // compiler's wait after the barrier), so their latency overlaps the barrier; the ring slot they read is not
    // refilled until >= 3 barriers later, each of which this wave passes with lgkmcnt already drained.
    if (out >= 4) wait_vmcnt_only<8>();
    else if (out == 3) wait_vmcnt_only<6>();
    else if (out == 2) wait_vmcnt_only<4>();
    else if (out == 1) wait_vmcnt_only<2>();
    else wait_vmcnt_only<0>();
    STAMP(t2);
    __builtin_amdgcn_s_barrier();
    STAMP(t3);
    sL += t1 - t0; sW += t2 - t1; sB1 += t3 - t2;
  };
  auto compute = [&](auto ihc, auto jhc, auto slotc, int dma_m, int dma_q, auto lo8c) {
    constexpr int ih = decltype(ihc)::value, jh = decltype(jhc)::value, slot = decltype(slotc)::value;
    constexpr bool lo8 = decltype(lo8c)::value;      // compile-time: with both MFMA forms behind a run-time branch in one loop body the
                                                     // register allocator spilled 241 registers; two loops, one form each, spill none
    (void)dma_m; (void)dma_q;
    __builtin_amdgcn_sched_barrier(0);
    if (AFFT_PP_PRIO == 1) __builtin_amdgcn_s_setprio(1);
    if constexpr (X3 == 3) {
      if constexpr (lo8) {      // fp8 segment: 8 block-scaled MFMAs of 128 k instead of 16 of 32 k (the same fragment pairs, taken whole)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[ih][jh][i][j] = mfma_lo8(bP[slot][j], aP[i], acc[ih][jh][i][j]);
      } else {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[ih][jh][i][j] = mfma16<3>(frag_half(bP[slot][j], s), frag_half(aP[i], s), acc[ih][jh][i][j]);
      }
    } else
    if (!(AFFT_PP_DIAG & 4))
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (AFFT_PP_DMA_IN_C && (s * 8 + i * 2 + j) == AFFT_PP_DMA_AT) {
            __builtin_amdgcn_sched_barrier(0);
            issue(dma_m, dma_q);
            __builtin_amdgcn_sched_barrier(0);
          }
          acc[ih][jh][i][j] = mfma16<X3>(bF[slot][j][s], aF[i][s], acc[ih][jh][i][j]);
        }
    if (AFFT_PP_PRIO == 1) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(t4);
    __builtin_amdgcn_s_barrier();
    STAMP(t5);
    sC += t4 - t3; sB2 += t5 - t4; t0 = t5;
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // prologue = "L(-1)": the half-tiles every later wait rule assumes are already issued
  static_for<0, LEAD>([&](auto mc) { issue(decltype(mc)::value, decltype(mc)::value & 3); });
  wait_then_barrier(-1);
  load_b(0, 0, I0{});                          // K-tile 0's first B fragments (later ones are read a phase early)
  if (gp == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one slot behind group 0
  STAMP(t0);
  sL = sW = sB1 = sC = sB2 = 0;
  unsigned long long tstart = t0; (void)tstart;

  // One K-tile = 4 phases = the 4 quadrants of the wave's 128x64 output, in the order (0,0) (0,1) (1,1) (1,0).
  // The two B fragment slots swap roles every K-tile (P = parity: B half 0 lives in slot P, half 1 in slot 1-P) so
  // that the next K-tile's first B fragments can be read during phase 3, whose own operands are all resident:
  // LDS reads per phase 8 / 4 / 8 / 4 instead of 12 / 4 / 8 / 0 (the L segment has to fit under 16 MFMAs).
  in_loop = true;
  auto ktile = [&](auto Pc, int kt, auto lo8) {
    using SP = std::integral_constant<int, decltype(Pc)::value>;
    using SQ = std::integral_constant<int, 1 - decltype(Pc)::value>;
    const int n = 4 * kt;
    auto L = [&](auto reads, int m, int q) {
      if (AFFT_PP_PRIO == 2) __builtin_amdgcn_s_setprio(1);
      if (AFFT_PP_DMA_IN_C) reads();
      else if (AFFT_PP_DMA_FIRST) { issue(m, q); reads(); } else { reads(); issue(m, q); }
      if (AFFT_PP_PRIO == 2) __builtin_amdgcn_s_setprio(0);
    };
    L([&] { load_a(kt, 0); }, n + 0 + LEAD, (0 + LEAD) & 3);            wait_then_barrier(n + 0); compute(I0{}, I0{}, SP{}, n + 0 + LEAD, (0 + LEAD) & 3, lo8);
    L([&] { load_b(kt, 1, SQ{}); }, n + 1 + LEAD, (1 + LEAD) & 3);      wait_then_barrier(n + 1); compute(I0{}, I1{}, SQ{}, n + 1 + LEAD, (1 + LEAD) & 3, lo8);
    L([&] { load_a(kt, 1); }, n + 2 + LEAD, (2 + LEAD) & 3);            wait_then_barrier(n + 2); compute(I1{}, I1{}, SQ{}, n + 2 + LEAD, (2 + LEAD) & 3, lo8);
    L([&] { load_b(kt + 1, 0, SQ{}); }, n + 3 + LEAD, (3 + LEAD) & 3);  wait_then_barrier(n + 3); compute(I1{}, I0{}, SP{}, n + 3 + LEAD, (3 + LEAD) & 3, lo8);
  };
  const int nk_first = X3 == 3 ? g.nk_seg : nk;      // X3 = 3: fp16 segment (an even number of K-tiles: K % 128 == 0), then the fp8 segment
  for (int kt = 0; kt < nk_first; kt += 2) {
    ktile(I0{}, kt, std::false_type{});
    if (kt + 1 < nk_first) ktile(I1{}, kt + 1, std::false_type{});
  }
  if constexpr (X3 == 3) {
    for (int kt = nk_first; kt < nk; kt += 2) {
      ktile(I0{}, kt, std::true_type{});
      if (kt + 1 < nk) ktile(I1{}, kt + 1, std::true_type{});
    }
  }
  if (gp == 0) __builtin_amdgcn_s_barrier();
#if AFFT_PP_CLAMP
  wait_vmcnt_only<0>();            // the overshoot LDS-DMA has landed before the ring is reused (split-K combine / epilogue image)
#endif
#ifdef AFFT_PP_STAMP
  if (stamp_out && blockIdx.x == 0 && lane == 0) {
    unsigned long long* o = stamp_out + wave * 8;
    o[0] = sL; o[1] = sW; o[2] = sB1; o[3] = sC; o[4] = sB2; o[5] = t0 - tstart; o[6] = (unsigned long long)nk;
  }
#endif

  pp_epilogue(g, smem, acc, m0, n0, lane, wave, gp, wc);
}

// ---------------------------------------------------------------------------------------------
// Round 6: the same ping-pong schedule with a steady-state loop that contains nothing but the schedule ("pp2").
// gemm_bf16_pp_kernel above serves every shape (edge tiles, odd K-tile counts, operand planes); its loop pays for that: a run-time
// "past the end of K?" test and a dummy transfer beside every LDS-DMA issue (16 branches per two K-tiles), 64-bit source-address
// arithmetic per piece (scalar multiplies in the k-strided layouts), and fragment registers the allocator renames from phase to
// phase (a false write-after-write wait in the middle of a phase's LDS reads).  This kernel takes only what the hot shapes are --
// plain bf16, whole 256x256 tiles (M, N % 256 == 0), an even number >= 4 of K-tiles -- and in exchange:
//   * the loop over K-tile PAIRS is branch-free: the last pair, whose look-ahead would run past the end of K, is peeled (TAIL
//     instantiations of the same phase code: fewer issues, the counted waits draining 6 -> 4 -> 2 -> 0);
//   * every source address is (one of 8 loop-invariant wave-uniform bases, in SGPRs) + (one VGPR per operand: the lane's offset
//     inside a piece plus the running K offset, advanced by ONE v_add per operand and K-tile) -- no address arithmetic per piece;
//   * LDS destinations are immediates of the M0 write.
// Schedule, ring, images, fragment reads, hazards (RAW / WAR derivation) and epilogue are gemm_bf16_pp_kernel's, LEAD = 6.
// AFFT_PP2_WAIT = 1: the counted wait only in phases 0 and 2 of a K-tile.  Deadlines per half-tile h = 4j + q (wait in L(w), w <= ..):
// q = 0: 4j - 1, q = 1: 4j - 2, q = 2: 4j, q = 3: 4j + 1 (read phase minus one, see RAW above).  The wait of L(4j - 2) [phase 2 of
// K-tile j - 1] retires h <= 4j + 1 (q = 0, 1 of K-tile j: in time), the wait of L(4j) [phase 0] retires h <= 4j + 3 (q = 2, 3: in
// time): each with 3 half-tiles = 6 operations allowed in flight -- the waits of phases 1 and 3 only ever asked for data whose
// deadline is one phase later, so dropping them relaxes the schedule without touching a deadline.
#ifndef AFFT_PP2_WAIT
#define AFFT_PP2_WAIT 1
#endif
#ifndef AFFT_PP2_PRIO
#define AFFT_PP2_PRIO 1       // MFMA segments at raised priority
#endif
#ifndef AFFT_PP2_DMA_FIRST
#define AFFT_PP2_DMA_FIRST 1  // a phase's LDS-DMA before its fragment reads
#endif
template <int OFF>
__device__ __forceinline__ void glds16i(const char* sbase, unsigned voff, unsigned lds_wave) {
  asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_wave), "n"(OFF) : "memory", "scc");
}

// X3 = 3 (NT only): the "fp16x2" forward GEMM -- nk_seg K-tiles of (A hi fp16, B fp16 image) on v_mfma_f32_16x16x32_f16, then nk_seg / 2 K-tiles
// of 128 k each of the e4m3 byte planes (A8, B8: their own bases and row pitches) on the block-scaled fp8 MFMA (gemm_tiles.h: mfma_lo8).  The
// LDS-DMA stream runs 1.5 K-tiles ahead of the MFMAs, so it changes planes inside the fp16 segment's LAST PAIR of K-tiles: in K-tile nk_seg - 2
// (TAIL code 3, "TRANS") phase 0 stages the last fp16 B half and then switches B's bases and lane offset, phase 1 does the same for A, phases
// 2 and 3 already stage the first fp8 K-tile -- all at compile-time positions.  Fragments are kept as PAIRS (k-substeps 0 | 1 in one
// 8-register tuple): the fp8 MFMA takes the tuple whole, the fp16 MFMA its halves (sub-registers, no copies).
template <bool A_KS, bool B_KS, int X3 = 0>
__global__ __launch_bounds__(512, 2) void gemm_bf16_pp2_kernel(const GemmFast g) {
  static_assert(X3 != 3 || (!A_KS && !B_KS), "the fp16 + fp8 forward is NT only");
  constexpr int HB = 128 * BK * 2;   // half-tile bytes (16 KiB)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gp = wave >> 2, wc = wave & 3;
  int tm, tn;
  tile_coords(g.tiles_m, g.tiles_n, blockIdx.x, tm, tn);
  const int m0 = tm * 256, n0 = tn * 256;
  const int npairs = (X3 == 3 ? g.nk_seg * BK : g.K) / (2 * BK);      // K-tile pairs: of all segments (X3 = 0, 1, 2: >= 2) / of the fp16 segment (X3 = 3: >= 1)

  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 aF[4][2], bF[2][2][2];   // A fragments of the live half; B fragments of both halves
  bf16x16 aP[4], bP[2][2];        // X3 = 3: the same as pairs
  (void)aF; (void)bF; (void)aP; (void)bP;

  const unsigned lds_wave = lds_addr(smem) + wave * 1024;
  const unsigned lda2 = (unsigned)(g.lda * 2), ldb2 = (unsigned)(g.ldb * 2);
  const LaneOffsets lo = lane_offsets(wave, lane);
  unsigned vA = A_KS ? lo.ks_row * lda2 + lo.ks_c16 : lo.kc_row * lda2 + lo.kc_chunk16;   // lane offset + running K offset
  unsigned vB = B_KS ? lo.ks_row * ldb2 + lo.ks_c16 : lo.kc_row * ldb2 + lo.kc_chunk16;
  const unsigned stepA = A_KS ? 64u * lda2 : 128u, stepB = B_KS ? 64u * ldb2 : 128u;     // bytes per K-tile
  // the 8 source bases of this wave: [operand half][piece]; piece j = wave + 8 jj covers rows 8j.. (k-contiguous) / k-rows 4j.. (k-strided)
  const char* bA[2][2];
  const char* bB[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = wave + 8 * jj;
      bA[h][jj] = A_KS ? (const char*)(g.A + (int64_t)(4 * j) * g.lda + m0 + 128 * h) : (const char*)(g.A + (int64_t)(m0 + 128 * h + 8 * j) * g.lda);
      bB[h][jj] = B_KS ? (const char*)(g.B + (int64_t)(4 * j) * g.ldb + n0 + 128 * h) : (const char*)(g.B + (int64_t)(n0 + 128 * h + 8 * j) * g.ldb);
    }
  // X3 = 3: the LDS-DMA stream moves on to the e4m3 byte planes (K offset back to 0, their own row pitch)
  bool sw_now = false;      // wave-uniform: this pair of K-tiles is the fp16 segment's last (set by the pair loop)
  // K advance after a staged K-tile's closing half-tile, by the parity of the K-tile that ISSUES it (first / second of a pair).  Plain steps, except:
  // X3 = 1 / 2 -- at a segment's end the first K-tile of a pair jumps to the next pair of operand planes (a_lo / b_lo elements away, K offset back
  // to 0): a different addend, no branch; X3 = 3 -- 0 in the last pair of the fp8 segment (below)
  unsigned incA[2] = {stepA, stepA}, incB[2] = {stepB, stepB};
  auto switch_a = [&] {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) bA[h][jj] = (const char*)(g.A8 + (int64_t)(m0 + 128 * h + 8 * (wave + 8 * jj)) * g.lda8);
    vA = lo.kc_row * (unsigned)(g.lda8 * 2) + lo.kc_chunk16;
  };
  auto switch_b = [&] {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) bB[h][jj] = (const char*)(g.B8 + (int64_t)(n0 + 128 * h + 8 * (wave + 8 * jj)) * g.ldb8);
    vB = lo.kc_row * (unsigned)(g.ldb8 * 2) + lo.kc_chunk16;
  };
  // half-tile kind q (0 = A rows 0-127, 1 = B 0-127, 2 = B 128-255, 3 = A 128-255) of a K-tile of parity KP -> ring slot KP * 4 + q
  // SW: this is the first segment's last half-tile of the operand: switch planes instead of stepping on
  auto issue = [&](auto qc, auto kpc, auto swc) {
    constexpr int q = decltype(qc)::value, KP = decltype(kpc)::value;
    constexpr bool SW = decltype(swc)::value;
    constexpr int dst = (KP * 4 + q) * HB;
    if constexpr (q == 0 || q == 3) {
      glds16i<dst>(bA[q == 3][0], vA, lds_wave);
      glds16i<dst + 8192>(bA[q == 3][1], vA, lds_wave);
      if constexpr (q == 3) { if constexpr (SW) { if (sw_now) switch_a(); else vA += incA[1 - KP]; } else vA += incA[1 - KP]; }      // A's second half closes its K-tile
    } else {
      glds16i<dst>(bB[q == 2][0], vB, lds_wave);
      glds16i<dst + 8192>(bB[q == 2][1], vB, lds_wave);
      if constexpr (q == 2) { if constexpr (SW) { if (sw_now) switch_b(); else vB += incB[1 - KP]; } else vB += incB[1 - KP]; }
    }
  };
  // Fragment reads: every lane-dependent part of an LDS address is one of a few VGPRs computed ONCE (made opaque, so that the compiler
  // neither re-derives them per phase nor hoists one register per (ring slot, fragment) -- 36 address registers and, in the TN
  // instantiation, 17 spills in the first build); ring slot, k-substep and fragment index are instruction offsets (< 64 KiB: one
  // register set per half of the ring).  k-contiguous image (gemm_tiles.h frag_kc): address(row, chunk) = row * 128 +
  // ((chunk ^ ((row >> 1) & 7)) << 4) -- fragment i / j adds 16 rows = 2 KiB, the k-substep flips bit 2 of the chunk (a lane-dependent
  // +-64 B: one register per substep).  k-strided image (frag_ks): both 4-row blocks and both k-substeps of a fragment are fixed
  // distances (1 KiB, 8 KiB) from one address that depends on the 32-byte unit = fragment index (one register per fragment).
  constexpr int NA = A_KS ? 4 : 2, NB = 2;
  unsigned adA[2][NA], adB[2][NB];      // [ring half][..] LDS byte addresses of ring slot 0 / 4
  {
    const unsigned l0 = lds_addr(smem);
    const int g4 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3, r16 = lane & 15;
#pragma unroll
    for (int x = 0; x < NA; ++x) {
      unsigned a;
      if constexpr (A_KS) a = (8 * g4 + q4) * 256 + ((((gp * 4 + x) ^ ks_f(8 * g4 + q4))) << 5) + p4 * 8;
      else { const int row = gp * 64 + r16; a = row * 128 + ((((x * 4 + g4) ^ ((row >> 1) & 7))) << 4); }
      adA[0][x] = l0 + a; adA[1][x] = l0 + a + 4 * HB;
      asm volatile("" : "+v"(adA[0][x]), "+v"(adA[1][x]));
    }
#pragma unroll
    for (int x = 0; x < NB; ++x) {
      unsigned b;
      if constexpr (B_KS) b = (8 * g4 + q4) * 256 + ((((wc * 2 + x) ^ ks_f(8 * g4 + q4))) << 5) + p4 * 8;
      else { const int row = wc * 32 + r16; b = row * 128 + ((((x * 4 + g4) ^ ((row >> 1) & 7))) << 4); }
      adB[0][x] = l0 + b; adB[1][x] = l0 + b + 4 * HB;
      asm volatile("" : "+v"(adB[0][x]), "+v"(adB[1][x]));
    }
  }
  auto rd128 = [](unsigned addr, auto offc) -> bf16x8 {
    return *(AFFT_LDS const bf16x8*)(size_t)(addr + decltype(offc)::value);
  };
  auto rdtr = [](unsigned addr, auto offc) -> bf16x8 {      // one k-strided fragment: two transposing 8-byte reads 4 k-rows (1 KiB) apart
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AFFT_LDS bf16x4*)(size_t)(addr + decltype(offc)::value));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AFFT_LDS bf16x4*)(size_t)(addr + decltype(offc)::value + 1024));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
  };
  // PAIR (X3 = 3, fp8 K-tiles): both k-substeps of a fragment side by side in one 8-register tuple, as the block-scaled MFMA takes them; the fp16
  // K-tiles keep the two substeps as separate 4-register fragments (pairs everywhere spilled 80 registers: 8-register tuples fragment the file)
  auto load_a = [&](auto kpc, auto ihc, auto pairc) {
    constexpr int KP = decltype(kpc)::value, ih = decltype(ihc)::value;
    constexpr bool PAIR = decltype(pairc)::value;
    constexpr int so = (ih ? 3 : 0) * HB;       // slot offset inside the ring half
    static_for<0, 2>([&](auto sc) {
      static_for<0, 4>([&](auto ic) {
        constexpr int s = decltype(sc)::value, i = decltype(ic)::value;
        if constexpr (PAIR) { if constexpr (s == 0) aP[i] = frag_pair(rd128(adA[KP][0], std::integral_constant<int, so + i * 2048>{}),
                                                                         rd128(adA[KP][1], std::integral_constant<int, so + i * 2048>{})); }
        else if constexpr (A_KS) aF[i][s] = rdtr(adA[KP][i], std::integral_constant<int, so + s * 8192>{});
        else aF[i][s] = rd128(adA[KP][s], std::integral_constant<int, so + i * 2048>{});
      });
    });
  };
  auto load_b = [&](auto kpc, auto jhc, auto slotc, auto pairc) {   // B half jh of a K-tile of parity KP -> fragment slot
    constexpr int KP = decltype(kpc)::value, jh = decltype(jhc)::value, slot = decltype(slotc)::value;
    constexpr bool PAIR = decltype(pairc)::value;
    constexpr int so = (1 + jh) * HB;
    static_for<0, 2>([&](auto sc) {
      static_for<0, 2>([&](auto jc) {
        constexpr int s = decltype(sc)::value, j = decltype(jc)::value;
        if constexpr (PAIR) { if constexpr (s == 0) bP[slot][j] = frag_pair(rd128(adB[KP][0], std::integral_constant<int, so + j * 2048>{}),
                                                                               rd128(adB[KP][1], std::integral_constant<int, so + j * 2048>{})); }
        else if constexpr (B_KS) bF[slot][j][s] = rdtr(adB[KP][j], std::integral_constant<int, so + s * 8192>{});
        else bF[slot][j][s] = rd128(adB[KP][s], std::integral_constant<int, so + j * 2048>{});
      });
    });
  };
  auto compute = [&](auto ihc, auto jhc, auto slotc, auto lo8c) {
    constexpr int ih = decltype(ihc)::value, jh = decltype(jhc)::value, slot = decltype(slotc)::value;
    constexpr bool LO8 = decltype(lo8c)::value;      // compile-time: one MFMA form per K-tile instantiation (a run-time choice spilled 241 registers in round 5)
    __builtin_amdgcn_sched_barrier(0);
    if (AFFT_PP2_PRIO) __builtin_amdgcn_s_setprio(1);
    if constexpr (X3 == 3 && LO8) {       // 8 block-scaled fp8 MFMAs of 128 k
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[ih][jh][i][j] = mfma_lo8(bP[slot][j], aP[i], acc[ih][jh][i][j]);
    } else if constexpr (X3 == 3) {      // fp16 segment: A in 4-register fragments, B the halves of its pairs (sub-registers)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[ih][jh][i][j] = mfma16<3>(frag_half(bP[slot][j], s), aF[i][s], acc[ih][jh][i][j]);
    } else {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[ih][jh][i][j] = mfma16<X3>(bF[slot][j][s], aF[i][s], acc[ih][jh][i][j]);
    }
    if (AFFT_PP2_PRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;

  // prologue = "L(-1)": half-tiles 0..5 (K-tile 0 whole, A0 and B0 of K-tile 1)
  using NoSw = std::false_type;
  issue(I0{}, I0{}, NoSw{}); issue(I1{}, I0{}, NoSw{}); issue(I2{}, I0{}, NoSw{}); issue(I3{}, I0{}, NoSw{}); issue(I0{}, I1{}, NoSw{}); issue(I1{}, I1{}, NoSw{});
  wait_vmcnt_only<6>();
  __builtin_amdgcn_s_barrier();
  load_b(I0{}, I0{}, I0{}, std::integral_constant<bool, X3 == 3>{});  // K-tile 0's first B fragments (later ones are read a phase early)
  if (gp == 1) __builtin_amdgcn_s_barrier();    // group 1 runs one slot behind group 0

  // One K-tile of parity P (ring half P): 4 phases = the 4 quadrants of the wave's 128x64 output, in the order (0,0) (0,1) (1,1) (1,0);
  // phase p stages half-tile 4 kt + p + 6 = kind (p + 2) & 3 of K-tile kt + 1 (p = 0, 1: ring half 1 - P) / kt + 2 (p = 2, 3: half P).
  // TAIL: 0 = steady state, 1 = K-tile nk - 2 (only phases 0, 1 still stage), 2 = K-tile nk - 1 (nothing left to stage or to prefetch)
  // TAIL code 3 ("TRANS", X3 = 3): the steady state's schedule, but phases 0 and 1 stage the first segment's LAST B / A half and switch planes
  auto ktile = [&](auto Pc, auto tailc, auto lo8c) {
    constexpr int P = decltype(Pc)::value, TC = decltype(tailc)::value;
    constexpr int TAIL = TC >= 3 ? 0 : TC;      // 3 = steady state whose phases 0 / 1 may switch planes (sw_now)
    using SW = std::integral_constant<bool, TC == 3>;
    using LO8 = decltype(lo8c);
    using PR = std::integral_constant<bool, X3 == 3 && LO8::value>;      // this K-tile's A fragments are pairs (fp8 K-tiles)
    using PRB = std::integral_constant<bool, X3 == 3>;                   // B fragments: pairs in every K-tile (they are read a K-tile early, across the segment boundary)
    using KP = std::integral_constant<int, P>;
    using KQ = std::integral_constant<int, 1 - P>;
    using SP = KP;      // fragment slot of B half 0 (the two slots swap roles every K-tile)
    using SQ = KQ;
    constexpr bool W = AFFT_PP2_WAIT != 0;
    // phase 0
    if constexpr (TAIL < 2 && AFFT_PP2_DMA_FIRST) issue(I2{}, KQ{}, SW{});
    load_a(KP{}, I0{}, PR{});
    if constexpr (TAIL < 2 && !AFFT_PP2_DMA_FIRST) issue(I2{}, KQ{}, SW{});
    if constexpr (TAIL == 0 || TAIL == 1) wait_vmcnt_only<6>(); else wait_vmcnt_only<0>();
    __builtin_amdgcn_s_barrier();
    compute(I0{}, I0{}, SP{}, LO8{});
    // phase 1
    if constexpr (TAIL < 2 && AFFT_PP2_DMA_FIRST) issue(I3{}, KQ{}, SW{});
    load_b(KP{}, I1{}, SQ{}, PRB{});
    if constexpr (TAIL < 2 && !AFFT_PP2_DMA_FIRST) issue(I3{}, KQ{}, SW{});
    if constexpr (TAIL == 0) { if constexpr (!W) wait_vmcnt_only<6>(); } else if constexpr (TAIL == 1) wait_vmcnt_only<6>();
    __builtin_amdgcn_s_barrier();
    compute(I0{}, I1{}, SQ{}, LO8{});
    // phase 2
    if constexpr (TAIL == 0 && AFFT_PP2_DMA_FIRST) issue(I0{}, KP{}, NoSw{});
    load_a(KP{}, I1{}, PR{});
    if constexpr (TAIL == 0 && !AFFT_PP2_DMA_FIRST) issue(I0{}, KP{}, NoSw{});
    if constexpr (TAIL == 0) wait_vmcnt_only<6>(); else if constexpr (TAIL == 1) wait_vmcnt_only<4>();
    __builtin_amdgcn_s_barrier();
    compute(I1{}, I1{}, SQ{}, LO8{});
    // phase 3
    if constexpr (TAIL == 0 && AFFT_PP2_DMA_FIRST) issue(I1{}, KP{}, NoSw{});
    if constexpr (TAIL < 2) load_b(KQ{}, I0{}, SQ{}, PRB{});
    if constexpr (TAIL == 0 && !AFFT_PP2_DMA_FIRST) issue(I1{}, KP{}, NoSw{});
    if constexpr (TAIL == 0) { if constexpr (!W) wait_vmcnt_only<6>(); } else if constexpr (TAIL == 1) wait_vmcnt_only<2>();
    __builtin_amdgcn_s_barrier();
    compute(I1{}, I0{}, SP{}, LO8{});
  };
  using F16 = std::false_type;
  using F8 = std::true_type;
  if constexpr (X3 == 3) {
    // every pair of the fp16 segment in ONE loop body (a separate body for the last pair made the allocator permute the accumulators
    // through scratch at its boundaries): the first K-tile of a pair carries the plane switch under a wave-uniform flag
    int sw_pair = npairs - 1;
    asm volatile("" : "+s"(sw_pair));      // opaque: the compiler would peel the last iteration into a body of its own (and permute the accumulators around it)
    for (int pr = 0; pr < npairs; ++pr) {
      sw_now = pr == sw_pair;
      ktile(I0{}, I3{}, F16{});
      ktile(I1{}, I0{}, F16{});
    }
    // fp8 segment: nk_seg / 2 K-tiles of 128 k (even: launch_pp2), EVERY pair in the steady-state body -- a peeled tail made the allocator
    // permute the accumulators through scratch (the instantiation sits at 256 registers).  The look-ahead of the last pair stays inside K: the
    // running offsets stop advancing, so its six surplus half-tiles re-read the last K-tile (L2 hits, 96 KiB per tile) into ring slots that
    // nobody reads any more -- the same slots at the same phases as in every other pair, so the WAR analysis is unchanged -- and the counted
    // wait stays vmcnt(6) throughout; the surplus is drained before the epilogue reuses the ring.
    const int npairs8 = g.nk_seg / 4;
    int last8 = npairs8 - 1;
    asm volatile("" : "+s"(last8));
    for (int pr = 0; pr < npairs8; ++pr) {
      incA[0] = incA[1] = pr == last8 ? 0u : stepA;
      incB[0] = incB[1] = pr == last8 ? 0u : stepB;
      ktile(I0{}, I0{}, F8{});
      ktile(I1{}, I0{}, F8{});
    }
    wait_vmcnt_only<0>();
  } else {
    // X3 = 1 (bf16x3) / 2 (fp16 two-pass): segments of nk_seg K-tiles -- (A, B), (A + a_lo, B), X3 = 1 also (A, B + b_lo) -- walked as ONE stream: in the
    // pair that holds a segment's last two K-tiles the first K-tile's closing half-tiles add a jump instead of a step (gemm_tiles.h seg_operands)
    const int sp = g.nk_seg / 2;      // pairs per segment
    unsigned jA1 = stepA, jB1 = stepB, jA2 = stepA, jB2 = stepB;
    if constexpr (X3 == 1 || X3 == 2) {
      const unsigned backA = (unsigned)(g.nk_seg - 1) * stepA, backB = (unsigned)(g.nk_seg - 1) * stepB;
      jA1 = (unsigned)(g.a_lo * 2) - backA;  jB1 = 0u - backB;                              // segment 0 -> 1: A's lo planes, B rewinds
      jA2 = 0u - (unsigned)(g.a_lo * 2) - backA;  jB2 = (unsigned)(g.b_lo * 2) - backB;      // segment 1 -> 2: A back to hi, B's lo planes
    }
    for (int pr = 0; pr < npairs - 1; ++pr) {
      if constexpr (X3 == 1 || X3 == 2) {
        incA[0] = pr == sp - 1 ? jA1 : pr == 2 * sp - 1 ? jA2 : stepA;
        incB[0] = pr == sp - 1 ? jB1 : pr == 2 * sp - 1 ? jB2 : stepB;
      }
      ktile(I0{}, I0{}, F16{});
      ktile(I1{}, I0{}, F16{});
    }
    ktile(I0{}, I1{}, F16{});
    ktile(I1{}, I2{}, F16{});
  }
  if (gp == 0) __builtin_amdgcn_s_barrier();
  pp_epilogue(g, smem, acc, m0, n0, lane, wave, gp, wc);
}

#ifndef AFFT_PP2
#define AFFT_PP2 1      // 0: every shape on gemm_bf16_pp_kernel (A/B builds)
#endif
bool pp2_shape(int M, int N, int K) { return AFFT_PP2 && M % 256 == 0 && N % 256 == 0 && K % (2 * BK) == 0 && K >= 4 * BK; }
// the running K offset lives in a 32-bit VGPR (k-strided operands: K rows of `ld` elements): the whole walk must stay below 4 GiB
inline bool walk_fits32(bool ks, int K, int64_t ld) { return ks ? (int64_t)(K + 8) * ld * 2 < (1LL << 32) : (8 * ld + K) * 2 < (1LL << 32); }
template <bool A_KS, bool B_KS>
bool pp2_takes(const GemmFast& g) { return pp2_shape(g.e.M, g.e.N, g.K) && walk_fits32(A_KS, g.K, g.lda) && walk_fits32(B_KS, g.K, g.ldb); }

// fp16 + fp8 forward (X3 = 3): g.K counts both segments in 64-wide K-tiles (nk_seg + nk_seg / 2); pairs in both: nk_seg % 4 == 0, nk_seg >= 4
bool pp2x3_takes(const GemmFast& g) {
#ifdef AFFT_PP2_X3_OFF
  return false;
#else
  return AFFT_PP2 && g.e.M % 256 == 0 && g.e.N % 256 == 0 && g.nk_seg % 4 == 0 && g.nk_seg >= 4 && g.K == g.nk_seg * BK + g.nk_seg * BK / 2 &&
         walk_fits32(false, g.nk_seg * BK, g.lda) && walk_fits32(false, g.nk_seg * BK, g.ldb) && walk_fits32(false, g.nk_seg * BK, g.lda8) &&
         walk_fits32(false, g.nk_seg * BK, g.ldb8);
#endif
}

// bf16x3 / fp16 two-pass (X3 = 1 / 2): g.K counts all segments; whole tiles, an even number of K-tiles per segment (pairs never straddle a
// segment), every plane inside the 32-bit walk
template <bool A_KS, bool B_KS, int X3>
bool pp2planes_takes(const GemmFast& g) {
#ifdef AFFT_PP2_PLANES_OFF
  return false;
#else
  const int64_t span = ((int64_t)(g.a_lo > g.b_lo ? g.a_lo : g.b_lo)) * 2;
  const bool one_pass = X3 == 2 && g.K == g.nk_seg * BK && g.nk_seg >= 4;      // afft_gemm_t.split3 = 4: the first segment alone (no jump is ever taken)
  return AFFT_PP2 && g.e.M % 256 == 0 && g.e.N % 256 == 0 && g.nk_seg % 2 == 0 && g.nk_seg >= 2 && (g.K == (X3 == 1 ? 3 : 2) * g.nk_seg * BK || one_pass) &&
         span < (1LL << 30) && walk_fits32(A_KS, g.nk_seg * BK, g.lda) && walk_fits32(B_KS, g.nk_seg * BK, g.ldb) &&
         (A_KS ? (int64_t)g.nk_seg * BK * g.lda * 2 : (int64_t)8 * g.lda * 2) + span < (1LL << 31) &&
         (B_KS ? (int64_t)g.nk_seg * BK * g.ldb * 2 : (int64_t)8 * g.ldb * 2) + span < (1LL << 31);
#endif
}

template <bool A_KS, bool B_KS, int X3 = 0>
int launch_pp2(GemmFast& g, hipStream_t stream) {
  constexpr size_t lds = 128 * 1040;
  g.tiles_m = g.e.M / 256;
  g.tiles_n = g.e.N / 256;
  auto kern = gemm_bf16_pp2_kernel<A_KS, B_KS, X3>;
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = afft_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, &attr_done)) return rc;
  hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n), dim3(512), lds, stream, g);
  AFFT_LAUNCH_CHECK();
  return 0;
}

template <bool A_KS, bool B_KS, int X3 = 0>
int launch_pp(GemmFast& g, hipStream_t stream) {
  constexpr size_t lds = 128 * 1040;          // ring: 2 K-tiles x 4 half-tiles x 16 KiB = 128 KiB; epilogue image: 130 KiB
  g.tiles_m = (g.e.M + 255) / 256;
  g.tiles_n = (g.e.N + 255) / 256;
  auto kern = gemm_bf16_pp_kernel<A_KS, B_KS, X3>;
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = afft_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, &attr_done)) return rc;
  const int grid = g.tiles_m * g.tiles_n;
#ifdef AFFT_PP_STAMP
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, g, g_pp_stamp);
#else
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, g);
#endif
  AFFT_LAUNCH_CHECK();
  return 0;
}


}  // namespace

#ifdef AFFT_PP_STAMP
extern "C" void afft_debug_pp_stamp(void* p) { g_pp_stamp = (unsigned long long*)p; }
#endif

bool afft_gemm_pp2_takes(int M, int N, int K, int x3) {      // K: the caller's K (one segment)
  if (x3 == 3) { const int ns = K / BK; return AFFT_PP2 && M % 256 == 0 && N % 256 == 0 && ns % 4 == 0 && ns >= 4; }
  if (x3 == 1 || x3 == 2) { const int ns = K / BK; return AFFT_PP2 && M % 256 == 0 && N % 256 == 0 && ns % 2 == 0 && ns >= 2; }
  if (x3 == 4) { const int ns = K / BK; return AFFT_PP2 && M % 256 == 0 && N % 256 == 0 && ns % 2 == 0 && ns >= 4; }
  return x3 == 0 && pp2_shape(M, N, K);
}

int afft_gemm_launch_pp(int a_ks, int b_ks, afft_gemm_detail::GemmFast& g, hipStream_t stream, int x3) {
#ifndef AFFT_PP_NT_ONLY   // development switch: build only the plain NT instantiation (compile time)
  if (x3 == 3) {   // fp16 + fp8 lo pass: nn.Linear forward GEMMs (NT) only
    if (!a_ks && !b_ks) return pp2x3_takes(g) ? launch_pp2<false, false, 3>(g, stream) : launch_pp<false, false, 3>(g, stream);
    afft_set_error("afft_gemm: the fp16 + fp8 mode (split3 = 3) is built for the NT layout only");
    return 1;
  }
  if (x3 == 2) {   // fp16x2: forward GEMMs only (NT, and NN for [in, out] weights)
    if (!a_ks && !b_ks) return pp2planes_takes<false, false, 2>(g) ? launch_pp2<false, false, 2>(g, stream) : launch_pp<false, false, 2>(g, stream);
    if (!a_ks && b_ks) return pp2planes_takes<false, true, 2>(g) ? launch_pp2<false, true, 2>(g, stream) : launch_pp<false, true, 2>(g, stream);
    afft_set_error("afft_gemm: the fp16 two-pass mode is built for the forward layouts only");
    return 1;
  }
  if (x3) {        // bf16x3 operand planes
    if (!a_ks && !b_ks) return pp2planes_takes<false, false, 1>(g) ? launch_pp2<false, false, 1>(g, stream) : launch_pp<false, false, 1>(g, stream);
    if (!a_ks && b_ks) return pp2planes_takes<false, true, 1>(g) ? launch_pp2<false, true, 1>(g, stream) : launch_pp<false, true, 1>(g, stream);
    if (a_ks && b_ks) return pp2planes_takes<true, true, 1>(g) ? launch_pp2<true, true, 1>(g, stream) : launch_pp<true, true, 1>(g, stream);
  }
#endif
  if (!a_ks && !b_ks) return pp2_takes<false, false>(g) ? launch_pp2<false, false>(g, stream) : launch_pp<false, false>(g, stream);
#ifndef AFFT_PP_NT_ONLY
  if (!a_ks && b_ks) return pp2_takes<false, true>(g) ? launch_pp2<false, true>(g, stream) : launch_pp<false, true>(g, stream);
  if (a_ks && b_ks) return pp2_takes<true, true>(g) ? launch_pp2<true, true>(g, stream) : launch_pp<true, true>(g, stream);
#endif
  afft_set_error("afft_gemm: layout (A k-strided, B k-contiguous) is not built");
  return 1;
}
