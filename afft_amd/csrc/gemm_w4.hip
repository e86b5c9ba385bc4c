// 256x256x64 bf16 MFMA GEMM, four waves of 128x128 each ("w4"): see the comment above the kernel.
// EXPERIMENTAL (afft_set_gemm_variant(5) = LDS-DMA staging, 6 = register staging; never picked automatically).  Correct on every layout (tests/test_kernels_gpu.py,
// *_w4 cases).  First build: slower than the 8-wave ping-pong kernel (8192^3 1.09 ms vs 0.89 ms, 5120x6144x2048 175 vs 129 us;
// profiles/r01_experiments_late.txt); with the branch-free K-loop (the LDS-DMA stream re-reads the last K-tile past the end of
// K, so there is no issue guard and ONE constant counted wait) it equals it: 8192^3 0.88-0.90 ms, TN shapes 4-7 % faster, NN
// +-3 %, NT 6-13 % slower; inside the training step TN on this kernel changes nothing (3784 vs 3783 clips/s).  With one wave per SIMD nothing hides what a wave's own LDS-DMA instructions cost
// at issue (60+ cycles each under back-pressure from the L2->LDS fill path, 4 per 512-cycle phase) nor the barrier at the end
// of every phase: MFMA + barriers alone 0.77 ms, + fragment reads 0.82, + LDS-DMA 0.96 / 1.09 with both.  The register-staged
// variant (second kernel below) reaches 0.98 ms: its ds_writes cost what the LDS-DMA issue stalls did.
#include "gemm_tiles.h"

using namespace afft_gemm_detail;

#ifndef AFFT_W4_LEAD
#define AFFT_W4_LEAD 7        // half-tiles of look-ahead of the LDS-DMA stream (3..9)
#endif
#ifndef AFFT_W4_DIAG
#define AFFT_W4_DIAG 0        // diagnostic builds only (wrong results): 1 = no fragment reads, 2 = no LDS-DMA in the loop, 4 = no MFMA,
                              // 8 = no barrier, 16 = vmcnt(0) instead of the counted wait chain, 32 = no ds_write of staged operands (w4r)
#endif

namespace {

// The 64 accumulator tiles (256 registers per lane) live in a[0:255] as state the compiler is not told about: every
// instruction that touches them is inline asm naming the registers literally.  (Declared as C++ values -- builtin or
// "+a" operands -- the loop-carried accumulators get VGPR-class virtual registers: v_accvgpr_write/read around every
// MFMA group and 400+ bytes of scratch.)  AFFT_CLOBBER_AGPRS at the phase boundaries makes the register allocator
// count all 256 AGPRs as used and keeps everything that lives across a phase out of them; tools/w4_check_isa.py
// asserts that no compiler-generated instruction of the kernel names an AGPR.
#define AFFT_A8(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
#define AFFT_CLOBBER_AGPRS()                                                                                             \
  asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", AFFT_A8(1), AFFT_A8(2), AFFT_A8(3),     \
               AFFT_A8(4), AFFT_A8(5), AFFT_A8(6), AFFT_A8(7), AFFT_A8(8), AFFT_A8(9), AFFT_A8(10), AFFT_A8(11),          \
               AFFT_A8(12), AFFT_A8(13), AFFT_A8(14), AFFT_A8(15), AFFT_A8(16), AFFT_A8(17), AFFT_A8(18), AFFT_A8(19),    \
               AFFT_A8(20), AFFT_A8(21), AFFT_A8(22), AFFT_A8(23), AFFT_A8(24), "a250", "a251", "a252", "a253", "a254",   \
               "a255")
template <int T>   // accumulator tile T (0..63) += x * y
__device__ __forceinline__ void mfma_tile(const bf16x8& x, const bf16x8& y) {
  asm volatile("v_mfma_f32_16x16x32_bf16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(x), "v"(y), "n"(4 * T), "n"(4 * T + 3));
}
template <int T>
__device__ __forceinline__ void zero_tile() {
  asm volatile("v_accvgpr_write_b32 a[%0], 0\n\tv_accvgpr_write_b32 a[%1], 0\n\tv_accvgpr_write_b32 a[%2], 0\n\t"
               "v_accvgpr_write_b32 a[%3], 0" ::"n"(4 * T), "n"(4 * T + 1), "n"(4 * T + 2), "n"(4 * T + 3));
}
template <int T>
__device__ __forceinline__ f32x4 read_tile() {
  f32x4 r;
  asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\t"
               "v_accvgpr_read_b32 %3, a[%7]"
               : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3])
               : "n"(4 * T), "n"(4 * T + 1), "n"(4 * T + 2), "n"(4 * T + 3));
  return r;
}

// ---------------------------------------------------------------------------------------------
// The idea: the waves of the 8-wave ping-pong kernel (gemm_pp.hip) own 128x64 of the output, so a K-tile costs
// 8 x 24 KiB of fragment reads; a wave that owns 128x128 reads 32 KiB per K-tile, four of them 128 KiB -- a third less
// LDS traffic and half the barriers' participants.  The price is 256 accumulator registers per lane: one wave per
// SIMD (__launch_bounds__(256) -> 512 registers, the accumulators live in AGPRs), so there is no second wave to hide
// a wave's own LDS / LDS-DMA / barrier latency: the fragment reads of the NEXT phase are interleaved with the MFMAs
// of the current one (software pipelining inside one wave).  (The fragment reads turned out not to be the bound: the
// diagnostic builds at the top of this file.)
//
// The operand stream, the LDS images and the staging are the ping-pong kernel's: half-tiles of 128 rows x 64 k
// (16 KiB), index m = 4*kt + q (q: 0 = A rows 0-127, 1 = B rows 0-127, 2 = B rows 128-255, 3 = A rows 128-255),
// ring of 8 slots, LEAD half-tiles in flight.  Wave (wr, wc) of the 2x2 wave grid owns rows ih*128 + wr*64 + 16i and
// columns jh*128 + wc*64 + 16j (ih, jh in {0,1}; i, j in 0..3): phase (ih, jh) of a K-tile = 32 MFMAs on the
// fragments of A half ih and B half jh, in the order (0,0) (0,1) (1,1) (1,0) so that each phase needs ONE new set of
// fragments, which is read during the phase before into the registers of the half that has just died:
//   phase n = 4kt + 0: MFMA(A0, B0)   reads B1(kt)     -> the B slot that held B1(kt-1)
//   phase n = 4kt + 1: MFMA(A0, B1)   reads A1(kt)     -> A slot 1
//   phase n = 4kt + 2: MFMA(A1, B1)   reads A0(kt+1)   -> A slot 0
//   phase n = 4kt + 3: MFMA(A1, B0)   reads B0(kt+1)   -> the slot of B1(kt)   (the two B slots swap roles per K-tile)
// In every phase the half-tile read is m = n + 2.  Rules (one s_waitcnt vmcnt + s_barrier at the end of every phase):
//   RAW: at the end of phase n every wave waits until ITS pieces of half-tile n + 3 have landed (LEAD - 3 half-tiles stay in
//        flight), then the barrier;
//   WAR: phase n issues half-tile n + LEAD into the slot of n + LEAD - 8 <= n + 1, which every wave finished reading
//        (into registers) before the barrier that ended phase n - 1.
template <bool A_KS, bool B_KS>
__global__ __launch_bounds__(256) void gemm_bf16_w4_kernel(const GemmFast g) {
  constexpr int HB = 128 * BK * 2;   // half-tile bytes (16 KiB)
  constexpr int LEAD = AFFT_W4_LEAD;
  static_assert(LEAD >= 3 && LEAD <= 9, "look-ahead");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  int tm, tn;
  tile_coords(g.tiles_m, g.tiles_n, blockIdx.x, tm, tn);
  const int m0 = tm * 256, n0 = tn * 256;
  const int M = g.e.M, N = g.e.N;
  const int nk = g.K / BK;

  // accumulator tile (ih, jh, i, j) = a[4*T : 4*T+3], T = ((ih*2 + jh)*4 + i)*4 + j
  static_for<0, 64>([&](auto tc) { zero_tile<decltype(tc)::value>(); });
  AFFT_CLOBBER_AGPRS();
  bf16x8 aF[2][4][2], bF[2][4][2];   // [slot][16-row block][k half]
  if (AFFT_W4_DIAG & 1) {
    for (int a = 0; a < 2; ++a) for (int i = 0; i < 4; ++i) for (int s = 0; s < 2; ++s) for (int e = 0; e < 8; ++e) {
      aF[a][i][s][e] = (short)(lane * 37 + i); bF[a][i][s][e] = (short)(lane * 11 + i);
    }
  }

  const unsigned lds0 = lds_addr(smem);
  const unsigned lda2 = (unsigned)(g.lda * 2), ldb2 = (unsigned)(g.ldb * 2);
  const LaneOffsets lo = lane_offsets(wave, lane);
  const unsigned voffA = A_KS ? lo.ks_row * lda2 + lo.ks_c16 : lo.kc_row * lda2 + lo.kc_chunk16;
  const unsigned voffB = B_KS ? lo.ks_row * ldb2 + lo.ks_c16 : lo.kc_row * ldb2 + lo.kc_chunk16;
  bool in_loop = false; (void)in_loop;
  // piece jj (0..3) of this wave's share of half-tile m; q = m & 3 is compile-time at every call site
  auto issue_piece = [&](int m, int q, int jj) {
    if ((AFFT_W4_DIAG & 2) && in_loop) return;
    // past the end of K the stream re-reads the last K-tile into the (dead) ring slot: no guard, and the number of
    // LDS-DMA instructions in flight is the same in every phase -> one constant counted wait, no scalar branch chain
    const int kt = min(m >> 2, nk - 1);
    const unsigned dst = (((m >> 2) & 1) * 4 + q) * HB, lds_wave = lds0 + wave * 1024;
    if (q == 0 || q == 3) {
      const int r0 = m0 + (q == 3 ? 128 : 0);
      if constexpr (A_KS) stage_ks_piece<4>(g.A, g.lda, lda2, voffA, lo, r0, kt * BK, dst, wave, jj, lds_wave);
      else stage_kc_piece<4>(g.A, g.lda, lda2, voffA, lo, r0, M, kt * BK, dst, wave, jj, lds_wave);
    } else {
      const int c0 = n0 + (q == 2 ? 128 : 0);
      if constexpr (B_KS) stage_ks_piece<4>(g.B, g.ldb, ldb2, voffB, lo, c0, kt * BK, dst, wave, jj, lds_wave);
      else stage_kc_piece<4>(g.B, g.ldb, ldb2, voffB, lo, c0, N, kt * BK, dst, wave, jj, lds_wave);
    }
  };
  auto issue = [&](int m, int q) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) issue_piece(m, q, jj);
  };
  auto load_a = [&](int kt, int ih, auto slotc) {
    constexpr int slot = decltype(slotc)::value;
    if (AFFT_W4_DIAG & 1) return;
    const char* base = smem + ((kt & 1) * 4 + (ih ? 3 : 0)) * HB;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (A_KS) aF[slot][i][s] = frag_ks<128>(base, 32 * s, wr * 4 + i, lane);
        else aF[slot][i][s] = frag_kc(base, wr * 64 + i * 16 + (lane & 15), s * 4 + (lane >> 4));
      }
  };
  auto load_b = [&](int kt, int jh, auto slotc) {
    constexpr int slot = decltype(slotc)::value;
    if (AFFT_W4_DIAG & 1) return;
    const char* base = smem + ((kt & 1) * 4 + 1 + jh) * HB;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (B_KS) bF[slot][j][s] = frag_ks<128>(base, 32 * s, wc * 4 + j, lane);
        else bF[slot][j][s] = frag_kc(base, wc * 64 + j * 16 + (lane & 15), s * 4 + (lane >> 4));
      }
  };
  // end of phase n: this wave's pieces of every half-tile <= n + 3 have landed; then everybody's
  auto end_phase = [&](int) {
    wait_vmcnt_only<4 * (LEAD - 3)>();
    if (!(AFFT_W4_DIAG & 8)) __builtin_amdgcn_s_barrier();
    AFFT_CLOBBER_AGPRS();
  };
  // One phase = 8 groups of { the k-th fragment read of the NEXT phase's new half ; 4 MFMAs }
  auto mfma4 = [&](auto ihc, auto jhc, auto sac, auto sbc, auto kc) {
    constexpr int ih = decltype(ihc)::value, jh = decltype(jhc)::value, sa = decltype(sac)::value, sb = decltype(sbc)::value;
    constexpr int k = decltype(kc)::value, s = k >> 2, i = k & 3;
    if (AFFT_W4_DIAG & 4) return;
    static_for<0, 4>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      mfma_tile<((ih * 2 + jh) * 4 + i) * 4 + j>(bF[sb][j][s], aF[sa][i][s]);
    });
  };
  auto read_a = [&](const char* base, auto slotc, auto kc) {   // fragment k = (s, i) of an A half
    constexpr int slot = decltype(slotc)::value, k = decltype(kc)::value, s = k >> 2, i = k & 3;
    if (AFFT_W4_DIAG & 1) return;
    if constexpr (A_KS) aF[slot][i][s] = frag_ks<128>(base, 32 * s, wr * 4 + i, lane);
    else aF[slot][i][s] = frag_kc(base, wr * 64 + i * 16 + (lane & 15), s * 4 + (lane >> 4));
  };
  auto read_b = [&](const char* base, auto slotc, auto kc) {
    constexpr int slot = decltype(slotc)::value, k = decltype(kc)::value, s = k >> 2, j = k & 3;
    if (AFFT_W4_DIAG & 1) return;
    if constexpr (B_KS) bF[slot][j][s] = frag_ks<128>(base, 32 * s, wc * 4 + j, lane);
    else bF[slot][j][s] = frag_kc(base, wc * 64 + j * 16 + (lane & 15), s * 4 + (lane >> 4));
  };
  // phase: MFMAs on (A slot sa, B slot sb) -> acc[ih][jh]; meanwhile the fragments of half-tile (next_kt, nq) -> slot ns,
  // and the 4 LDS-DMA instructions of half-tile m_issue behind MFMA groups AFFT_W4_DMA_AT.. (the matrix pipe is busy by then)
#ifndef AFFT_W4_DMA_AT
#define AFFT_W4_DMA_AT 1
#endif
  auto phase = [&](auto ihc, auto jhc, auto sac, auto sbc, auto nqc, auto nsc, int next_kt, int m_issue, auto iqc) {
    constexpr int nq = decltype(nqc)::value;   // 0 = A half 0, 1 = B half 0, 2 = B half 1, 3 = A half 1
    constexpr int iq = decltype(iqc)::value;
    const char* base = smem + ((next_kt & 1) * 4 + nq) * HB;
    static_for<0, 8>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if constexpr (k >= AFFT_W4_DMA_AT && k < AFFT_W4_DMA_AT + 4) issue_piece(m_issue, iq, k - AFFT_W4_DMA_AT);
      if constexpr (nq == 0 || nq == 3) read_a(base, nsc, kc); else read_b(base, nsc, kc);
      __builtin_amdgcn_sched_barrier(0);
      mfma4(ihc, jhc, sac, sbc, kc);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // prologue: LEAD half-tiles in flight, half-tiles 0..2 landed, first fragments in registers
  static_for<0, LEAD>([&](auto mc) { issue(decltype(mc)::value, decltype(mc)::value & 3); });
  end_phase(-1);
  load_a(0, 0, I0{});
  load_b(0, 0, I0{});
  in_loop = true;

  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  auto ktile = [&](auto Pc, int kt) {
    using SP = std::integral_constant<int, decltype(Pc)::value>;        // slot of B half 0 in this K-tile
    using SQ = std::integral_constant<int, 1 - decltype(Pc)::value>;    // slot of B half 1
    const int n = 4 * kt;
    using Q0 = std::integral_constant<int, (0 + LEAD) & 3>;
    using Q1 = std::integral_constant<int, (1 + LEAD) & 3>;
    using Q2 = std::integral_constant<int, (2 + LEAD) & 3>;
    using Q3 = std::integral_constant<int, (3 + LEAD) & 3>;
    phase(I0{}, I0{}, I0{}, SP{}, I2{}, SQ{}, kt, n + 0 + LEAD, Q0{});     end_phase(n + 0);
    phase(I0{}, I1{}, I0{}, SQ{}, I3{}, I1{}, kt, n + 1 + LEAD, Q1{});     end_phase(n + 1);
    phase(I1{}, I1{}, I1{}, SQ{}, I0{}, I0{}, kt + 1, n + 2 + LEAD, Q2{}); end_phase(n + 2);
    phase(I1{}, I0{}, I1{}, SP{}, I1{}, SQ{}, kt + 1, n + 3 + LEAD, Q3{}); end_phase(n + 3);
  };
  for (int kt = 0; kt < nk; kt += 2) {
    ktile(I0{}, kt);
    if (kt + 1 < nk) ktile(I1{}, kt + 1);
  }

  // Epilogue through LDS (the ring is free: the last end_phase waited for vmcnt(0) and every wave's last fragment
  // reads were consumed by its MFMAs): two passes of 128 rows, fp32 [128][256] image with a 1040-byte row pitch, then
  // every wave walks whole rows -- 16-byte LDS reads, fully coalesced global accesses (gemm_pp.hip's epilogue).
  constexpr int ESTRIDE = 1040;
  const DropParams dp = with_salt(g.e.drop);
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)" ::: "memory");   // the MFMAs are opaque to the hazard recognizer: let the
                                                                              // last ones retire; the overshoot LDS-DMA has landed
  static_for<0, 2>([&](auto ihc) {
    constexpr int ih = decltype(ihc)::value;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // pass 0: every wave is done with the ring; pass 1: pass 0 has been read back
    static_for<0, 32>([&](auto idx) {
      constexpr int v = decltype(idx)::value;
      constexpr int jh = v >> 4, i = (v >> 2) & 3, j = v & 3;
      const int row = wr * 64 + i * 16 + (lane & 15);
      const int col = jh * 128 + wc * 64 + j * 16 + 4 * (lane >> 4);
      *(f32x4*)(smem + row * ESTRIDE + col * 4) = read_tile<((ih * 2 + jh) * 4 + i) * 4 + j>();
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int rr = 0; rr < 16; ++rr) {     // two rows per step: a lane owns 8 consecutive columns (16-byte bf16 stores)
      const int row = wave * 32 + rr * 2 + (lane >> 5);
      const int c8 = lane & 31;
      const f32x4 t0 = *(const f32x4*)(smem + row * ESTRIDE + c8 * 32);
      const f32x4 t1 = *(const f32x4*)(smem + row * ESTRIDE + c8 * 32 + 16);
      float o[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
      epilogue8(g.e, dp, m0 + ih * 128 + row, n0 + 8 * c8, o);
    }
  });
}


// ---------------------------------------------------------------------------------------------
// Register-staged variant ("w4r", afft_set_gemm_variant(6)): same wave layout, phases, LDS images and epilogue, but the
// operands go global -> VGPR -> ds_write_b128 instead of LDS-DMA (what the vendor library's 256x256 kernels do): a
// global_load_dwordx4 issues in a few cycles where an LDS-DMA instruction holds its wave -- and, at one wave per SIMD, the
// matrix pipe -- for the 60+ cycles the fill path takes to accept it.  All memory operations are compiler-visible, so
// the s_waitcnt vmcnt before each ds_write is the compiler's (exact, in order), and there is no run-time wait selection.
//   half-tile h: loaded (4 x 16 B per lane, register set h & 3) in phase h - 6, written to LDS slot h & 7 in phase h - 3,
//   visible after the barrier that ends phase h - 3, read into fragments in phase h - 2, consumed from phase h - 1 on.
//   Loads past the end of K re-read the last K-tile (never consumed): the loop body has no guards, no tail copy.
template <bool A_KS, bool B_KS>
__global__ __launch_bounds__(256) void gemm_bf16_w4r_kernel(const GemmFast g) {
  constexpr int HB = 128 * BK * 2;   // half-tile bytes (16 KiB)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  int tm, tn;
  tile_coords(g.tiles_m, g.tiles_n, blockIdx.x, tm, tn);
  const int m0 = tm * 256, n0 = tn * 256;
  const int M = g.e.M, N = g.e.N;
  const int nk = g.K / BK;

  static_for<0, 64>([&](auto tc) { zero_tile<decltype(tc)::value>(); });
  AFFT_CLOBBER_AGPRS();
  bf16x8 aF[2][4][2], bF[2][4][2];   // [slot][16-row block][k half]
  bf16x8 st[4][4];                   // staging registers [set = half-tile & 3][piece]
  bool in_loop = false; (void)in_loop;

  const unsigned lda2 = (unsigned)(g.lda * 2), ldb2 = (unsigned)(g.ldb * 2);
  const LaneOffsets lo = lane_offsets(wave, lane);
  const unsigned voffA = A_KS ? lo.ks_row * lda2 + lo.ks_c16 : lo.kc_row * lda2 + lo.kc_chunk16;
  const unsigned voffB = B_KS ? lo.ks_row * ldb2 + lo.ks_c16 : lo.kc_row * ldb2 + lo.kc_chunk16;
  // piece jj (0..3) of this wave's share of half-tile m -> staging set; q = m & 3 and the set are compile-time
  auto gload = [&](int m, auto qc, auto setc, auto jjc) {
    constexpr int q = decltype(qc)::value, set = decltype(setc)::value, jj = decltype(jjc)::value;
    const int kt = min(m >> 2, nk - 1);
    const char* sbase; unsigned voff;
    if constexpr (q == 0 || q == 3) {
      const int r0 = m0 + (q == 3 ? 128 : 0);
      if constexpr (A_KS) src_ks_piece<4>(g.A, g.lda, lda2, voffA, lo, r0, kt * BK, wave, jj, sbase, voff);
      else src_kc_piece<4>(g.A, g.lda, lda2, voffA, lo, r0, M, kt * BK, wave, jj, sbase, voff);
    } else {
      const int c0 = n0 + (q == 2 ? 128 : 0);
      if constexpr (B_KS) src_ks_piece<4>(g.B, g.ldb, ldb2, voffB, lo, c0, kt * BK, wave, jj, sbase, voff);
      else src_kc_piece<4>(g.B, g.ldb, ldb2, voffB, lo, c0, N, kt * BK, wave, jj, sbase, voff);
    }
    if ((AFFT_W4_DIAG & 2) && in_loop) return;
    st[set][jj] = *(const bf16x8*)(sbase + voff);
  };
  auto lwrite = [&](int m, auto qc, auto setc, auto jjc) {   // staging set -> LDS slot of half-tile m (lane-linear 1-KiB piece)
    constexpr int q = decltype(qc)::value, set = decltype(setc)::value, jj = decltype(jjc)::value;
    if ((AFFT_W4_DIAG & 32) && in_loop) return;
    char* dst = smem + (((m >> 2) & 1) * 4 + q) * HB + (wave + jj * 4) * 1024 + lane * 16;
    *(bf16x8*)dst = st[set][jj];
  };
  auto mfma4 = [&](auto ihc, auto jhc, auto sac, auto sbc, auto kc) {
    constexpr int ih = decltype(ihc)::value, jh = decltype(jhc)::value, sa = decltype(sac)::value, sb = decltype(sbc)::value;
    constexpr int k = decltype(kc)::value, s = k >> 2, i = k & 3;
    static_for<0, 4>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      mfma_tile<((ih * 2 + jh) * 4 + i) * 4 + j>(bF[sb][j][s], aF[sa][i][s]);
    });
  };
  auto read_a = [&](const char* base, auto slotc, auto kc) {   // fragment k = (s, i) of an A half
    constexpr int slot = decltype(slotc)::value, k = decltype(kc)::value, s = k >> 2, i = k & 3;
    if constexpr (A_KS) aF[slot][i][s] = frag_ks<128>(base, 32 * s, wr * 4 + i, lane);
    else aF[slot][i][s] = frag_kc(base, wr * 64 + i * 16 + (lane & 15), s * 4 + (lane >> 4));
  };
  auto read_b = [&](const char* base, auto slotc, auto kc) {
    constexpr int slot = decltype(slotc)::value, k = decltype(kc)::value, s = k >> 2, j = k & 3;
    if constexpr (B_KS) bF[slot][j][s] = frag_ks<128>(base, 32 * s, wc * 4 + j, lane);
    else bF[slot][j][s] = frag_kc(base, wc * 64 + j * 16 + (lane & 15), s * 4 + (lane >> 4));
  };
  // phase p (0..3) of K-tile kt, n = 4kt + p: 32 MFMAs on (A slot sa, B slot sb) in 8 groups of 4; beside them
  //   groups 0-3: two fragment reads of half-tile n + 2 (-> slot ns) and the ds_write of one piece of half-tile n + 3,
  //   groups 4-7: the global load of one piece of half-tile n + 6;  then lgkmcnt(0) (the writes have landed) + barrier.
  auto phase = [&](auto pc, auto ihc, auto jhc, auto sac, auto sbc, auto nqc, auto nsc, int next_kt, int kt) {
    constexpr int p = decltype(pc)::value, nq = decltype(nqc)::value;
    const int n = 4 * kt + p;
    const char* base = smem + ((next_kt & 1) * 4 + nq) * HB;
    static_for<0, 8>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
#ifndef AFFT_W4R_ORDER
#define AFFT_W4R_ORDER 1      // 0 = ds_writes in groups 0-3, global loads in 4-7;  1 = loads in groups 0-3, ds_writes in 4-7 (-2 %);  2 = alternating (no gain)
#endif
      if constexpr (AFFT_W4R_ORDER == 2) {   // one fragment read per group; ds_writes in the even groups, global loads in the odd ones
        if (!(AFFT_W4_DIAG & 1)) { if constexpr (nq == 0 || nq == 3) read_a(base, nsc, kc); else read_b(base, nsc, kc); }
        if constexpr ((k & 1) == 0)
          lwrite(n + 3, std::integral_constant<int, (p + 3) & 3>{}, std::integral_constant<int, (p + 3) & 3>{},
                 std::integral_constant<int, (k >> 1)>{});
        else
          gload(n + 6, std::integral_constant<int, (p + 6) & 3>{}, std::integral_constant<int, (p + 6) & 3>{},
                std::integral_constant<int, (k >> 1)>{});
      } else {
      if constexpr (k < 4) {
        if (!(AFFT_W4_DIAG & 1))
        static_for<0, 2>([&](auto rc) {
          using FK = std::integral_constant<int, 2 * k + decltype(rc)::value>;
          if constexpr (nq == 0 || nq == 3) read_a(base, nsc, FK{}); else read_b(base, nsc, FK{});
        });
      }
      if constexpr ((k < 4) == (AFFT_W4R_ORDER == 0)) {
        lwrite(n + 3, std::integral_constant<int, (p + 3) & 3>{}, std::integral_constant<int, (p + 3) & 3>{},
               std::integral_constant<int, (k & 3)>{});
      } else {
        gload(n + 6, std::integral_constant<int, (p + 6) & 3>{}, std::integral_constant<int, (p + 6) & 3>{},
              std::integral_constant<int, (k & 3)>{});
      }
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma4(ihc, jhc, sac, sbc, kc);
      __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(AFFT_W4_DIAG & 8)) __builtin_amdgcn_s_barrier();
    AFFT_CLOBBER_AGPRS();
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;

  // prologue: half-tiles 0..2 in LDS, 3..5 in flight, K-tile 0's first fragments in registers
  static_for<0, 3>([&](auto hc) { static_for<0, 4>([&](auto jc) { gload(decltype(hc)::value, hc, hc, jc); }); });
  static_for<0, 3>([&](auto hc) { static_for<0, 4>([&](auto jc) { lwrite(decltype(hc)::value, hc, hc, jc); }); });
  static_for<3, 6>([&](auto hc) {
    constexpr int h = decltype(hc)::value;
    static_for<0, 4>([&](auto jc) { gload(h, std::integral_constant<int, h & 3>{}, std::integral_constant<int, h & 3>{}, jc); });
  });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  static_for<0, 8>([&](auto kc) { read_a(smem + 0 * HB, I0{}, kc); });
  static_for<0, 8>([&](auto kc) { read_b(smem + 1 * HB, I0{}, kc); });

  in_loop = true;
  auto ktile = [&](auto Pc, int kt) {
    using SP = std::integral_constant<int, decltype(Pc)::value>;        // slot of B half 0 in this K-tile
    using SQ = std::integral_constant<int, 1 - decltype(Pc)::value>;    // slot of B half 1
    phase(I0{}, I0{}, I0{}, I0{}, SP{}, I2{}, SQ{}, kt, kt);
    phase(I1{}, I0{}, I1{}, I0{}, SQ{}, I3{}, I1{}, kt, kt);
    phase(I2{}, I1{}, I1{}, I1{}, SQ{}, I0{}, I0{}, kt + 1, kt);
    phase(I3{}, I1{}, I0{}, I1{}, SP{}, I1{}, SQ{}, kt + 1, kt);
  };
  for (int kt = 0; kt < nk; kt += 2) {
    ktile(I0{}, kt);
    if (kt + 1 < nk) ktile(I1{}, kt + 1);
  }

  constexpr int ESTRIDE = 1040;
  const DropParams dp = with_salt(g.e.drop);
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)" ::: "memory");   // last MFMAs retired; the overshoot loads are back
  static_for<0, 2>([&](auto ihc) {
    constexpr int ih = decltype(ihc)::value;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    static_for<0, 32>([&](auto idx) {
      constexpr int v = decltype(idx)::value;
      constexpr int jh = v >> 4, i = (v >> 2) & 3, j = v & 3;
      const int row = wr * 64 + i * 16 + (lane & 15);
      const int col = jh * 128 + wc * 64 + j * 16 + 4 * (lane >> 4);
      *(f32x4*)(smem + row * ESTRIDE + col * 4) = read_tile<((ih * 2 + jh) * 4 + i) * 4 + j>();
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int rr = 0; rr < 16; ++rr) {
      const int row = wave * 32 + rr * 2 + (lane >> 5);
      const int c8 = lane & 31;
      const f32x4 t0 = *(const f32x4*)(smem + row * ESTRIDE + c8 * 32);
      const f32x4 t1 = *(const f32x4*)(smem + row * ESTRIDE + c8 * 32 + 16);
      float o[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
      epilogue8(g.e, dp, m0 + ih * 128 + row, n0 + 8 * c8, o);
    }
  });
}

template <bool A_KS, bool B_KS, bool RS>
int launch_w4(GemmFast& g, hipStream_t stream) {
  constexpr size_t lds = 128 * 1040;          // ring: 8 half-tiles x 16 KiB = 128 KiB; epilogue image: 130 KiB
  g.tiles_m = (g.e.M + 255) / 256;
  g.tiles_n = (g.e.N + 255) / 256;
  auto kern = RS ? gemm_bf16_w4r_kernel<A_KS, B_KS> : gemm_bf16_w4_kernel<A_KS, B_KS>;
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = afft_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, &attr_done)) return rc;
  hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n), dim3(256), lds, stream, g);
  AFFT_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int afft_gemm_launch_w4(int a_ks, int b_ks, int reg_staged, afft_gemm_detail::GemmFast& g, hipStream_t stream) {
  if (reg_staged) {
    if (!a_ks && !b_ks) return launch_w4<false, false, true>(g, stream);
#ifndef AFFT_W4_NT_ONLY
    if (!a_ks && b_ks) return launch_w4<false, true, true>(g, stream);
    if (a_ks && b_ks) return launch_w4<true, true, true>(g, stream);
#endif
  } else {
    if (!a_ks && !b_ks) return launch_w4<false, false, false>(g, stream);
#ifndef AFFT_W4_NT_ONLY   // development switch: build only the NT instantiation (compile time)
    if (!a_ks && b_ks) return launch_w4<false, true, false>(g, stream);
    if (a_ks && b_ks) return launch_w4<true, true, false>(g, stream);
#endif
  }
  afft_set_error("afft_gemm: layout (A k-strided, B k-contiguous) is not built");
  return 1;
}
