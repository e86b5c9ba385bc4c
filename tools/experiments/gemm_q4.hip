// 256x256 bf16 MFMA GEMM, "four quadrants": 4 waves, one per SIMD, each owning a 128x128 quadrant of the output tile
// (the main loop judged on joules).
//
// Why this shape.  The ping-pong kernel (gemm_pp.hip) reads 192 KiB of fragments from LDS per 64-deep K-tile (8 waves, each
// 128x64: every A row is read by 4 waves, every B row by 2) and needs two workgroup barriers per 16 MFMAs to keep the two waves of
// a SIMD out of each other's way: ~75 % of the LDS read pipe, 74 % of the MFMA issue slots (profiles/r04_power.txt).  With ONE
// wave per SIMD and a 128x128 register tile per wave every A row is read by 2 waves and every B row by 2: 128 KiB per K-tile
// (50 % of the LDS pipe), 64 MFMAs between two barriers, and nobody to alternate with -- the wave overlaps its own LDS reads
// with its own MFMAs: the fragments of K-step t + 1 are read into a second register set WHILE the 64 MFMAs of step t issue.
//
//   tile 256 x 256, K-step 32 (one v_mfma_f32_16x16x32_bf16 deep), 64 accumulator tiles of 16x16 per wave (256 registers)
//   LDS: ring of 5 slots x [A image 256 rows x 64 B | B image 256 rows x 64 B] = 160 KiB, filled by LDS-DMA 4 steps ahead
//        (global_load_lds_dwordx4, 8 per wave and step), 16-byte chunk c of row r stored at chunk c ^ (-(r >> 2) & 3):
//        conflict-free ds_read_b128 fragment reads, LDS writes stay lane-linear (the swizzle is on the source address)
//   step t:  issue tile t + 4 -> read fragments of tile t + 1 (landed everywhere since the last barrier) -> 64 MFMAs on the
//            fragments of tile t -> wait: own DMA pieces of tile t + 2 landed, own LDS reads returned -> s_barrier
//   RAW: tile t + 1 is read during step t; every wave waited for its pieces of it before the barrier that ended step t - 1.
//   WAR: tile t + 4 refills the slot of tile t - 1, whose last reads (step t - 2) had returned before the barrier that ended
//        step t - 2 (lgkmcnt(0) precedes every barrier): two barriers of distance.
// k-contiguous operands only (NT: forward of nn.Linear, data gradient of HF Conv1D).
//
// RESULT (round 5, profiles/r05_gemm_q4.txt): correct (bitwise equal to the ping-pong kernel), but 0.78-0.80x its rate -- 8192^3:
// 1001-1140 TFLOP/s against 1255-1453 (ping-pong) and 1543-1626 (vendor); 0.82 TFLOP/s per W against 1.06 / 1.17.  Builds without
// the LDS-DMA, without the fragment reads and without the barrier inside the K loop run at the SAME rate: the loop is bound by the
// issue of the 64 literal-AGPR MFMAs themselves (~34 cycles each at the power limit against ~22 in the ping-pong kernel), not by
// its memory path.  It failed the kill criterion set for it (>= 1550 TFLOP/s and >= 1.15 TFLOP/s per W at 8192^3) and is NOT part
// of the product library: tools/experiments/build_q4.sh builds a library with it (variant 11) for whoever wants to continue.
#include "../../afft_amd/csrc/gemm_tiles.h"

using namespace afft_gemm_detail;

#ifndef AFFT_Q4_DIAG
#define AFFT_Q4_DIAG 0     // diagnostic builds only (wrong results): 1 = no LDS-DMA inside the K loop, 2 = no fragment reads inside it, 4 = no barrier
#endif

namespace {

constexpr int QK = 32;                       // K-step
constexpr int Q_IMG = 256 * 64;              // one operand image: 256 rows x 64 B
constexpr int Q_SLOT = 2 * Q_IMG;            // A | B
constexpr int Q_NSLOT = 5, Q_LOOK = 4;

// ---- the 64 accumulator tiles live in a[0:255] as state the compiler is not told about (the technique of gemm_bd.hip: declared as
// C++ values -- even with "+a" constraints -- 256 accumulator + 128 fragment registers end up shuttled through v_accvgpr_* and
// scratch: 1100 moves and 200 scratch accesses in the first builds of this kernel).  Q4_CLOBBER_AGPRS keeps the register
// allocator's own values out of the accumulator file.
#define Q4_A8(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
#define Q4_CLOBBER_AGPRS()                                                                                                  \
  asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", Q4_A8(1), Q4_A8(2), Q4_A8(3), Q4_A8(4), Q4_A8(5), \
               Q4_A8(6), Q4_A8(7), Q4_A8(8), Q4_A8(9), Q4_A8(10), Q4_A8(11), Q4_A8(12), Q4_A8(13), Q4_A8(14), Q4_A8(15), Q4_A8(16),   \
               Q4_A8(17), Q4_A8(18), Q4_A8(19), Q4_A8(20), Q4_A8(21), Q4_A8(22), Q4_A8(23), Q4_A8(24), "a250", "a251", "a252",       \
               "a253", "a254", "a255")
template <int T>   // accumulator tile T += x * y  (D[row = n][col = m] with x = the B fragment, y = the A fragment)
__device__ __forceinline__ void q4_mfma(const bf16x8& x, const bf16x8& y) {
  asm volatile("v_mfma_f32_16x16x32_bf16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(x), "v"(y), "n"(4 * T), "n"(4 * T + 3));
}
template <int T>
__device__ __forceinline__ void q4_zero() {
  asm volatile("v_accvgpr_write_b32 a[%0], 0\n\tv_accvgpr_write_b32 a[%1], 0\n\tv_accvgpr_write_b32 a[%2], 0\n\t"
               "v_accvgpr_write_b32 a[%3], 0" ::"n"(4 * T), "n"(4 * T + 1), "n"(4 * T + 2), "n"(4 * T + 3));
}
template <int T>
__device__ __forceinline__ f32x4 q4_read() {
  f32x4 r;
  asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\t"
               "v_accvgpr_read_b32 %3, a[%7]"
               : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3])
               : "n"(4 * T), "n"(4 * T + 1), "n"(4 * T + 2), "n"(4 * T + 3));
  return r;
}

__device__ __forceinline__ bf16x8 frag64(const char* img, int row, int g) {
  // the four 16-lane groups of ds_read_b128 are {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS):
  // with h(q) = -q & 3 the 16 lanes of a group fall on 16 distinct 16-byte slots of the 256-byte bank row (h(q) = q: 2-way conflicts)
  return *(const bf16x8*)(img + row * 64 + ((g ^ ((0 - (row >> 2)) & 3)) << 4));
}

__global__ __launch_bounds__(256) void gemm_bf16_q4_kernel(const GemmFast g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int c15 = lane & 15, gq = lane >> 4;
  int tm, tn;
  tile_coords(g.tiles_m, g.tiles_n, blockIdx.x, tm, tn);
  const int m0 = tm * 256, n0 = tn * 256;
  const int M = g.e.M, N = g.e.N;
  const int nk = g.K / QK;

  Q4_CLOBBER_AGPRS();
  static_for<0, 64>([&](auto idx) { q4_zero<decltype(idx)::value>(); });      // accumulator tile T = i * 8 + j

  // LDS-DMA: a piece = 16 rows x 64 B = one wave-instruction; lane -> row (lane >> 2), stored chunk (lane & 3), source chunk swizzled
  const unsigned lds_wave = lds_addr(smem) + wave * 1024;
  const unsigned lda2 = (unsigned)(g.lda * 2), ldb2 = (unsigned)(g.ldb * 2);
  const unsigned prow = lane >> 2;
  const unsigned sch = (unsigned)(((lane & 3) ^ ((0 - (lane >> 4)) & 3)) << 4);
  const unsigned voffA = prow * lda2 + sch, voffB = prow * ldb2 + sch;
  auto issue_piece = [&](int kt, auto idxc) {         // piece idx of this wave's 8 per tile: 0-3 of A, 4-7 of B
    constexpr int idx = decltype(idxc)::value, jj = idx & 3;
    const unsigned slot = (unsigned)(kt % Q_NSLOT) * Q_SLOT;
    const int k0 = kt * QK;
    const int p = wave + 4 * jj;                       // piece of the 256-row image
    if constexpr (idx < 4) {
      const int pb = min(m0 + p * 16, M - 1), lim = M - 1 - pb;       // tail pieces re-read valid rows; the epilogue drops them
      const char* sb = (const char*)(g.A + (int64_t)pb * g.lda + k0);
      const unsigned vo = lim < 15 ? min(prow, (unsigned)lim) * lda2 + sch : voffA;
      glds16(sb, vo, lds_wave, slot + jj * 4096);
    } else {
      const int pb = min(n0 + p * 16, N - 1), lim = N - 1 - pb;
      const char* sb = (const char*)(g.B + (int64_t)pb * g.ldb + k0);
      const unsigned vo = lim < 15 ? min(prow, (unsigned)lim) * ldb2 + sch : voffB;
      glds16(sb, vo, lds_wave, slot + Q_IMG + jj * 4096);
    }
  };
  auto issue = [&](int kt) { static_for<0, 8>([&](auto idxc) { issue_piece(kt, idxc); }); };
  bf16x8 af[2][8], bfr[2][8];
  auto load_frags = [&](int kt, auto setc) {
    constexpr int S = decltype(setc)::value;
    const char* a = smem + (kt % Q_NSLOT) * Q_SLOT;
    const char* b = a + Q_IMG;
#pragma unroll
    for (int i = 0; i < 8; ++i) af[S][i] = frag64(a, wr * 128 + i * 16 + c15, gq);
#pragma unroll
    for (int j = 0; j < 8; ++j) bfr[S][j] = frag64(b, wc * 128 + j * 16 + c15, gq);
  };
  // end of step kt: own pieces of tile kt + 2 landed (tiles kt + 3, kt + 4 may stay in flight), own LDS reads returned, rendezvous
  auto wait_then_barrier = [&](int kt) {
    const int rem = nk - 3 - kt;               // tiles beyond kt + 2 that have been issued: min(2, rem)
    if (rem >= 2) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    else if (rem == 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (!(AFFT_Q4_DIAG & 4) || kt < 0) __builtin_amdgcn_s_barrier();
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

#pragma unroll
  for (int t = 0; t < Q_LOOK; ++t)
    if (t < nk) issue(t);
  wait_then_barrier(-1);                       // tiles 0 and 1 landed everywhere
  load_frags(0, I0{});

  // One wave per SIMD: whatever is not an MFMA has to issue in the shadow of one.  The 64 MFMAs of a step go out in four groups
  // of 16; the LDS-DMA of tile kt + 4 and the fragment reads of tile kt + 1 are placed BETWEEN the groups (sched_barrier keeps the
  // compiler from pulling them together again), so the matrix pipe always has queued work while they issue.
  // the 64 MFMAs of a step go out in eight groups of 8; behind each group ONE LDS-DMA piece of tile kt + 4 (an LDS-DMA piece costs
  // the issuing wave ~60 cycles: eight in a row starve the matrix pipe of a SIMD that has no second wave) and two of the 16
  // fragment reads of tile kt + 1; sched_barrier keeps the compiler from pulling them together again
  auto step = [&](int kt, auto curc, auto nxtc) {
    constexpr int C = decltype(curc)::value, S = decltype(nxtc)::value;
    const char* na = smem + ((kt + 1) % Q_NSLOT) * Q_SLOT;
    const char* nb = na + Q_IMG;
    const bool more = kt + 1 < nk, dma = kt + Q_LOOK < nk;
    static_for<0, 8>([&](auto qc) {
      constexpr int Q = decltype(qc)::value;        // row block i = Q: tiles Q * 8 + j
      static_for<0, 8>([&](auto jc) { q4_mfma<Q * 8 + decltype(jc)::value>(bfr[C][decltype(jc)::value], af[C][Q]); });
      __builtin_amdgcn_sched_barrier(0);
      if (dma && !(AFFT_Q4_DIAG & 1)) issue_piece(kt + Q_LOOK, qc);
      if (more && !(AFFT_Q4_DIAG & 2)) {
        af[S][Q] = frag64(na, wr * 128 + Q * 16 + c15, gq);
        bfr[S][Q] = frag64(nb, wc * 128 + Q * 16 + c15, gq);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    wait_then_barrier(kt);
    Q4_CLOBBER_AGPRS();
  };
  for (int kt = 0; kt < nk; kt += 2) {
    step(kt, I0{}, I1{});
    if (kt + 1 < nk) step(kt + 1, I1{}, I0{});
  }

  // Epilogue through LDS (the ring is free: the last barrier has passed): two passes of 128 tile rows.  The two waves that hold
  // those rows scatter their accumulators into an fp32 [128][256] image (row stride 1040 B), then all four waves walk 32 whole
  // rows each with 16-byte LDS reads and fully coalesced global accesses (as gemm_pp.hip).
  constexpr int ESTRIDE = 1040;
  const DropParams dp = with_salt(g.e.drop);
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the last MFMAs have written their tiles before they are read
  Q4_CLOBBER_AGPRS();
  static_for<0, 2>([&](auto ihc) {
    constexpr int ih = decltype(ihc)::value;
    __builtin_amdgcn_s_barrier();
    if (wr == ih) {
      static_for<0, 64>([&](auto idx) {
        constexpr int i = decltype(idx)::value >> 3, j = decltype(idx)::value & 7;
        const int row = i * 16 + c15;
        const int col = wc * 128 + j * 16 + 4 * gq;
        *(f32x4*)(smem + row * ESTRIDE + col * 4) = q4_read<i * 8 + j>();
      });
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll 2
    for (int rr = 0; rr < 16; ++rr) {     // two rows per step: a lane owns 8 consecutive columns
      const int row = wave * 32 + rr * 2 + (lane >> 5);
      const int c8 = lane & 31;
      const f32x4 t0 = *(const f32x4*)(smem + row * ESTRIDE + c8 * 32);
      const f32x4 t1 = *(const f32x4*)(smem + row * ESTRIDE + c8 * 32 + 16);
      float o[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
      epilogue8(g.e, dp, m0 + ih * 128 + row, n0 + 8 * c8, o);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  });
}

}  // namespace

int afft_gemm_launch_q4(afft_gemm_detail::GemmFast& g, hipStream_t stream) {
  constexpr size_t lds = (size_t)Q_NSLOT * Q_SLOT;       // 160 KiB ring; the epilogue image (130 KiB) reuses it
  g.tiles_m = (g.e.M + 255) / 256;
  g.tiles_n = (g.e.N + 255) / 256;
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = afft_ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_bf16_q4_kernel), lds, &attr_done)) return rc;
  hipLaunchKernelGGL(gemm_bf16_q4_kernel, dim3(g.tiles_m * g.tiles_n), dim3(256), lds, stream, g);
  AFFT_LAUNCH_CHECK();
  return 0;
}
