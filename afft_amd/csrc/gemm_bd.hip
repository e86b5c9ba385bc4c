// "B direct" bf16 MFMA GEMM for the k-contiguous (NT) layout: tiles of ROWS x 256 x 64 with ROWS = 160 or 256, four waves
// (one per SIMD), wave w owns all ROWS rows x columns 64w..64w+63.  See the comment above the kernel.
#include "gemm_tiles.h"

using namespace afft_gemm_detail;

#ifndef AFFT_BD_DIAG
#define AFFT_BD_DIAG 0        // diagnostic builds only (wrong results): 1 = no A fragment reads, 2 = no LDS-DMA in the loop, 4 = no MFMA,
                              // 8 = no B loads in the loop
#endif

namespace {

// ---- accumulators (and, for ROWS = 160, the B operand ring) live in AGPRs as state the compiler is not told about: every
// instruction that touches them is inline asm naming the registers literally (the technique of gemm_w4.hip; declared as C++
// values the loop-carried accumulators are given VGPR-class virtual registers and shuffled through v_accvgpr_* / scratch).
// AFFT_BD_CLOBBER_AGPRS at the K-step boundaries makes the register allocator count all 256 AGPRs as used and keeps its own
// values out of them; tools/bd_check_isa.py asserts that no compiler-generated instruction of the kernel names an AGPR.
#define AFFT_A8(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
#define AFFT_BD_CLOBBER_AGPRS()                                                                                          \
  asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", AFFT_A8(1), AFFT_A8(2), AFFT_A8(3),     \
               AFFT_A8(4), AFFT_A8(5), AFFT_A8(6), AFFT_A8(7), AFFT_A8(8), AFFT_A8(9), AFFT_A8(10), AFFT_A8(11),          \
               AFFT_A8(12), AFFT_A8(13), AFFT_A8(14), AFFT_A8(15), AFFT_A8(16), AFFT_A8(17), AFFT_A8(18), AFFT_A8(19),    \
               AFFT_A8(20), AFFT_A8(21), AFFT_A8(22), AFFT_A8(23), AFFT_A8(24), "a250", "a251", "a252", "a253", "a254",   \
               "a255")

template <int T>   // accumulator tile T += x * y, both operands in VGPRs
__device__ __forceinline__ void mfma_vv(const bf16x8& x, const bf16x8& y) {
  asm volatile("v_mfma_f32_16x16x32_bf16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(x), "v"(y), "n"(4 * T), "n"(4 * T + 3));
}
template <int T, int BR>   // accumulator tile T += a[BR:BR+3] * y: the B fragment is read from the accumulator file
__device__ __forceinline__ void mfma_av(const bf16x8& y) {
  asm volatile("v_mfma_f32_16x16x32_bf16 a[%1:%2], a[%3:%4], %0, a[%1:%2]" ::"v"(y), "n"(4 * T), "n"(4 * T + 3), "n"(BR), "n"(BR + 3));
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
// B fragment loads: 16 bytes per lane straight from global memory in MFMA operand layout (lane l: row l & 15 of the 16-column
// block, k = 8 (l >> 4) .. + 7 of the 32-deep K-step) -- wave-uniform base pointer + one loop-invariant lane offset + immediate
template <int BR, int OFF>
__device__ __forceinline__ void bload_a(const char* sbase, unsigned voff) {
  asm volatile("global_load_dwordx4 a[%2:%3], %0, %1 offset:%4" ::"v"(voff), "s"(sbase), "n"(BR), "n"(BR + 3), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void bload_v(bf16x8& dst, const char* sbase, unsigned voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}

// counted wait (+ optional lgkmcnt(0) and workgroup barrier); wait_pin4 also pins four VGPR fragments about to be consumed behind
// it (compiler-visible "+v": no use of them is scheduled above the wait)
template <int CNT, bool BAR>
__device__ __forceinline__ void wait_plain() {
  if constexpr (BAR) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(CNT) : "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
}
template <int CNT, bool BAR>
__device__ __forceinline__ void wait_pin4(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& d) {
  if constexpr (BAR) asm volatile("s_waitcnt vmcnt(%4) lgkmcnt(0)\n\ts_barrier" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(CNT) : "memory");
  else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(CNT) : "memory");
}

// ---------------------------------------------------------------------------------------------
// Why (profiles/r03_experiments.txt section 2): in the 256x256 kernels both operands go L2 -> LDS (LDS-DMA)
// -> registers (ds_read), 256 KiB of LDS traffic per 64-deep K-tile, and the diagnostics of the four-wave kernel showed that
// fragment reads (+102 us at 8192^3) and LDS-DMA writes (+135 us) cost +306 us together: the two legs collide in the LDS
// array.  Here the B operand never touches LDS: every wave owns 64 columns of the tile EXCLUSIVELY (four waves side by side,
// each ROWS x 64), so its B fragments are its own and are loaded global -> register in MFMA operand layout, 8 x 16 bytes per
// lane and K-tile, PB K-tiles ahead; only A (ROWS x 64 k) is staged through LDS (LDS-DMA, PA K-tiles ahead, ring of PA + 1
// slots) and read by all four waves.  LDS traffic per K-tile: ROWS = 256: 32 KiB written + 128 KiB read (256 KiB before);
// ROWS = 160: 20 + 80 KiB.  ROWS = 160 exists for the path's M = B*T*S = 5120 rows: 5120 = 32 x 160, so N = 2048 / 6144 / 8192
// give 256 / 768 / 1024 workgroups -- whole rounds on 256 CUs where 256-row tiles give 0.625 / 1.875 / 2.5.
//
// One wave per SIMD: 4 NI accumulator tiles (NI = ROWS / 16: 160 or 256 registers) in AGPRs; ROWS = 160 also keeps the B ring
// (3 K-tiles x 32 registers) in a[160:255], so nothing the compiler allocates is ever the destination of a load in flight.
// K-step u = 2 kt + s (32 deep): NI groups of { one A fragment read for K-step u + 1 ; 4 MFMAs }, the 4 B loads of K-step
// u + 2 PB and half of the A pieces of K-tile kt + PA spread over the first groups.  Waits (vmcnt counts in issue order):
//   top of (kt, 0): vmcnt(CB)            -> B fragments of (kt, 0) are in registers
//   top of (kt, 1): lgkmcnt(0), vmcnt(min(CA, CB)), s_barrier -> B of (kt, 1); A of K-tile kt + 1 has landed for every wave (RAW);
//                   every wave's reads of K-tile kt - 1's... kt's slot for K-step (kt, 1) are back (WAR: that slot is refilled
//                   from K-step (kt + 1, 0) on: ring = PA + 1 slots)
// Past the end of K the streams re-read the last K-tile (never consumed): no guards, constant counts.
template <int NI, int PA, int PB, bool PACKED = false>
__global__ __launch_bounds__(256) void gemm_bf16_bd_kernel(const GemmFast g) {
  constexpr int ROWS = NI * 16, NP = ROWS / 32;          // NP: 1-KiB LDS-DMA pieces of an A K-tile per wave
  constexpr int H0 = (NP + 1) / 2, H1 = NP - H0;         // ... issued in K-step 0 / 1 of a K-tile
  constexpr int RING = PA + 1, SLOT = ROWS * 128, NB = PB + 1;
  constexpr bool B_AGPR = NI * 16 + NB * 32 <= 256;      // room for the B ring behind the accumulators
  constexpr int BR0 = NI * 16;                           // first AGPR of the B ring
  static_assert(PA >= 2 && PB >= 1 && PB <= PA, "look-ahead");
  constexpr int CA = 4 * (2 * PA - 3) + (PA - 1) * H0 + (PA - 2) * H1;
  constexpr int CB = 4 * (2 * PB - 1) + PB * NP;
  constexpr int C1 = CA < CB ? CA : CB;
  static_assert(CA < 64 && CB < 64, "vmcnt immediate");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tm, tn;
  tile_coords(g.tiles_m, g.tiles_n, blockIdx.x, tm, tn);
  const int m0 = tm * ROWS, n0 = tn * 256;
  const int M = g.e.M, N = g.e.N;
  const int nk = g.K / BK;

  static_for<0, NI * 4>([&](auto tc) { zero_tile<decltype(tc)::value>(); });
  AFFT_BD_CLOBBER_AGPRS();
  bf16x8 aF[2][NI];                       // A fragments of K-step s live in aF[s]
  bf16x8 bF[B_AGPR ? 1 : NB][4][2];       // B ring in VGPRs (ROWS = 256 only)
#pragma unroll
  for (int a = 0; a < (B_AGPR ? 1 : NB); ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) { bF[a][j][0] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; bF[a][j][1] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
  if (AFFT_BD_DIAG & 1)
    for (int s = 0; s < 2; ++s) for (int i = 0; i < NI; ++i) for (int e = 0; e < 8; ++e) aF[s][i][e] = (short)(lane * 37 + i);

  // A: LDS-DMA staging of gemm_tiles.h, pieces dealt to the 4 waves round-robin (piece j = rows 8j..8j+7)
  const unsigned lds_wave = lds_addr(smem) + wave * 1024;
  const unsigned lda2 = (unsigned)(g.lda * 2);
  const LaneOffsets lo = lane_offsets(wave, lane);
  const unsigned voffA = lo.kc_row * lda2 + lo.kc_chunk16;
  bool in_loop = false; (void)in_loop;
  auto stage_a = [&](int kt, int jj) {     // piece jj (0..NP-1) of this wave's share of A K-tile kt
    if ((AFFT_BD_DIAG & 2) && in_loop) return;
    const int ks = min(kt, nk - 1);
    stage_kc_piece<4>(g.A, g.lda, lda2, voffA, lo, m0, M, ks * BK, (unsigned)((kt % RING) * SLOT), wave, jj, lds_wave);
  };
  // B: this wave's 64 columns, 4 blocks of 16; a block that starts past N - 16 re-reads the last block (the epilogue drops it).
  // PACKED: B is a FRAGMENT-MAJOR image -- for every 16-row block nb and 32-deep K-step ks the 64 lanes' 16-byte fragments
  // sit side by side in lane order, [nb][ks][lane = (n & 15) + 16 ((k >> 3) & 3)][8 k] -- so a wave's load instruction reads
  // 1 KiB contiguously (row-major: 16 rows x 64 B, four 128-byte lines per quad of lanes, ~4x the address-processing time)
  const unsigned voffB = PACKED ? (unsigned)(lane * 16) : (unsigned)(((lane & 15) * g.ldb + 8 * (lane >> 4)) * 2);
  const bf16_t* Bj[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int nb = min(n0 + wave * 64 + j * 16, max(N - 16, 0));
    Bj[j] = PACKED ? g.B + (int64_t)(nb >> 4) * (g.ldb >> 5) * 512 : g.B + (int64_t)nb * g.ldb;     // packed: ldb = K of the image
  }
  auto load_b = [&](auto slotc, auto jc, auto sc, int kt) {   // fragment (j, s) of K-tile kt -> ring slot
    constexpr int slot = decltype(slotc)::value, j = decltype(jc)::value, s = decltype(sc)::value;
    if ((AFFT_BD_DIAG & 8) && in_loop) return;
    const char* p = PACKED ? (const char*)(Bj[j] + (int64_t)min(kt, nk - 1) * 1024) : (const char*)(Bj[j] + (int64_t)min(kt, nk - 1) * BK);
    constexpr int off = PACKED ? s * 1024 : s * 64;
    if constexpr (B_AGPR) bload_a<BR0 + ((slot * 4 + j) * 2 + s) * 4, off>(p, voffB);
    else bload_v<off>(bF[slot][j][s], p, voffB);
  };
  auto read_a = [&](int kt, auto sc, auto ic) {    // A fragment i of K-step (kt, s) -> aF[s][i]
    constexpr int s = decltype(sc)::value, i = decltype(ic)::value;
    if (AFFT_BD_DIAG & 1) return;
    aF[s][i] = frag_kc(smem + (kt % RING) * SLOT, i * 16 + (lane & 15), s * 4 + (lane >> 4));
  };
  auto mfma4 = [&](auto sbc, auto sc, auto ic) {
    constexpr int sb = decltype(sbc)::value, s = decltype(sc)::value, i = decltype(ic)::value;
    if (AFFT_BD_DIAG & 4) return;
    static_for<0, 4>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (B_AGPR) mfma_av<i * 4 + j, BR0 + ((sb * 4 + j) * 2 + s) * 4>(aF[s][i]);
      else mfma_vv<i * 4 + j>(bF[sb][j][s], aF[s][i]);
    });
  };
  auto wait_b = [&](auto sbc, auto sc, auto cntc, auto barc) {
    constexpr int sb = decltype(sbc)::value, s = decltype(sc)::value, cnt = decltype(cntc)::value;
    constexpr bool bar = decltype(barc)::value;
    if constexpr (B_AGPR) wait_plain<cnt, bar>();
    else wait_pin4<cnt, bar>(bF[sb][0][s], bF[sb][1][s], bF[sb][2][s], bF[sb][3][s]);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // prologue: A K-tiles 0..PA-1 and B K-tiles 0..PB-1 issued and landed (vmcnt(0): from here on every counted wait below sees
  // at least the operations its count assumes), K-step (0, 0)'s A fragments in registers
  static_for<0, PA>([&](auto kc) { static_for<0, NP>([&](auto jc) { stage_a(decltype(kc)::value, decltype(jc)::value); }); });
  static_for<0, PB>([&](auto kc) {
    static_for<0, 4>([&](auto jc) { load_b(kc, jc, I0{}, decltype(kc)::value); });
    static_for<0, 4>([&](auto jc) { load_b(kc, jc, I1{}, decltype(kc)::value); });
  });
  if constexpr (!B_AGPR) {
    static_for<0, PB>([&](auto kc) {
      wait_b(kc, I0{}, I0{}, std::false_type{});
      wait_b(kc, I1{}, I0{}, std::false_type{});
    });
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  static_for<0, NI>([&](auto ic) { read_a(0, I0{}, ic); });
  in_loop = true;

  // one K-step: SB = ring slot of this K-tile's B fragments (static: the loop is unrolled NB K-tiles deep)
  auto kstep = [&](auto sbc, auto sc, int kt) {
    constexpr int sb = decltype(sbc)::value, s = decltype(sc)::value;
    using SN = std::integral_constant<int, (sb + PB) % NB>;       // slot of the K-tile whose B loads are issued now
    using S1 = std::integral_constant<int, 1 - s>;
    if constexpr (s == 0) wait_b(sbc, sc, std::integral_constant<int, CB>{}, std::false_type{});
    else wait_b(sbc, sc, std::integral_constant<int, C1>{}, std::true_type{});
    AFFT_BD_CLOBBER_AGPRS();
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, NI>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i < 4) load_b(SN{}, ic, sc, kt + PB);
      else if constexpr (i - 4 < (s == 0 ? H0 : H1)) stage_a(kt + PA, (s == 0 ? 0 : H0) + (i - 4));
      __builtin_amdgcn_sched_barrier(0);
      mfma4(sbc, sc, ic);
      __builtin_amdgcn_sched_barrier(0);
      // fragment i of the NEXT K-step, (kt, 1) or (kt + 1, 0), BEHIND the MFMAs that read fragment i of this one: the compiler's
      // own lgkmcnt wait in front of the first MFMA group of a K-step then only covers reads issued a whole K-step earlier
      // (in front of the group it waited for the read issued a moment before as well: one exposed LDS latency per K-step)
      read_a(s == 0 ? kt : kt + 1, S1{}, ic);
    });
  };
  for (int kt = 0; kt < nk; kt += NB) {
    static_for<0, NB>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      if (kt + b < nk) {
        kstep(bc, I0{}, kt + b);
        kstep(bc, I1{}, kt + b);
      }
    });
  }

  // Epilogue through LDS (gemm_pp.hip's scheme): two passes of ROWS / 2 rows; accumulators -> fp32 [ROWS / 2][256] image with a
  // 1040-byte row pitch, then every wave walks whole rows (16-byte LDS reads, fully coalesced global accesses).
  constexpr int ESTRIDE = 1040, HI = NI / 2, HR = ROWS / 2;
  const DropParams dp = with_salt(g.e.drop);
  // VGPR ring: the overshoot loads of the last K-tiles are still in flight and nothing consumes them -- to the compiler their
  // destination registers are dead from the loop exit on and free for the epilogue's address arithmetic, which a load landing
  // late would then overwrite (seen as a memory access fault).  Naming every ring register in a vmcnt(0) wait keeps them
  // allocated until the loads are back.
  if constexpr (!B_AGPR)
    static_for<0, NB>([&](auto kc) {
      wait_b(kc, I0{}, I0{}, std::false_type{});
      wait_b(kc, I1{}, I0{}, std::false_type{});
    });
  // the MFMAs are opaque to the hazard recognizer: let the last ones retire; every overshoot load / LDS-DMA has landed
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  AFFT_BD_CLOBBER_AGPRS();
  static_for<0, 2>([&](auto pc) {
    constexpr int p = decltype(pc)::value;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS reads of the ring have RETURNED (s_barrier alone does not say so: gemm.hip, g2 kernel)
    __builtin_amdgcn_s_barrier();   // pass 0: every wave is done with the ring; pass 1: pass 0 has been read back
    static_for<0, HI * 4>([&](auto idx) {
      constexpr int v = decltype(idx)::value;
      constexpr int i = v >> 2, j = v & 3;
      const int row = i * 16 + (lane & 15);
      const int col = wave * 64 + j * 16 + 4 * (lane >> 4);
      *(f32x4*)(smem + row * ESTRIDE + col * 4) = read_tile<(p * HI + i) * 4 + j>();
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll 2
    for (int rr = 0; rr < HR / 8; ++rr) {     // two rows per step: a lane owns 8 consecutive columns (16-byte bf16 stores)
      const int row = wave * (HR / 4) + rr * 2 + (lane >> 5);
      const int c8 = lane & 31;
      const f32x4 t0 = *(const f32x4*)(smem + row * ESTRIDE + c8 * 32);
      const f32x4 t1 = *(const f32x4*)(smem + row * ESTRIDE + c8 * 32 + 16);
      float o[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
      epilogue8(g.e, dp, m0 + p * HR + row, n0 + 8 * c8, o);
    }
  });
}

template <int NI, int PA, int PB, bool PACKED = false>
int launch_bd(GemmFast& g, hipStream_t stream) {
  constexpr int ROWS = NI * 16;
  constexpr size_t ring = (size_t)(PA + 1) * ROWS * 128, epi = (size_t)(ROWS / 2) * 1040;
  constexpr size_t lds = ring > epi ? ring : epi;
  static_assert(lds <= 160 * 1024, "LDS");
  g.tiles_m = (g.e.M + ROWS - 1) / ROWS;
  g.tiles_n = (g.e.N + 255) / 256;
  auto kern = gemm_bf16_bd_kernel<NI, PA, PB, PACKED>;
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = afft_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, &attr_done)) return rc;
  hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n), dim3(256), lds, stream, g);
  AFFT_LAUNCH_CHECK();
  return 0;
}

}  // namespace

#ifndef AFFT_BD_PA
#define AFFT_BD_PA 3
#endif
#ifndef AFFT_BD_PB160
#define AFFT_BD_PB160 2
#endif

// rows160 != 0: 160 x 256 tiles, else 256 x 256.  NT layout only, N % 16 == 0 (afft_gemm checks).
int afft_gemm_launch_bd(int rows160, int packed, afft_gemm_detail::GemmFast& g, hipStream_t stream) {
  if (packed) {
    if (rows160) return launch_bd<10, AFFT_BD_PA, AFFT_BD_PB160, true>(g, stream);
    return launch_bd<16, AFFT_BD_PA, 1, true>(g, stream);
  }
  if (rows160) return launch_bd<10, AFFT_BD_PA, AFFT_BD_PB160>(g, stream);
  return launch_bd<16, AFFT_BD_PA, 1>(g, stream);
}
