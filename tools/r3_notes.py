"""Assembles profiles/r03_experiments.txt from the raw outputs of the round-3 experiment scripts in gpurun_out/."""
import os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = lambda f: open(os.path.join(R, "gpurun_out", f)).read()
T = """Round 3 experiments on the GEMM side and the step schedule (one MI355X per run, boxes differ by a few %; cfg2 = the bench workload, B = 64).
Every number below was measured with the tools named beside it; raw outputs in gpurun_out/ of the build container (scratch).

== 1. Stream-K of the 256x256 ping-pong kernel (tools/gemm_bench.py VARIANTS=30,33 ; tools/sk_sweep.py ; tools/sk_diag.sh) -> built, tested, OFF by default
Plan (csrc/gemm_tiles.h SkPlan): one workgroup per CU; whole rounds of tiles run plain; the tiles of the partial round are cut along K so that
every workgroup gets the same number of K-iterations: "mains" take the head [0, Km) of one leftover tile each (aligned in K, so the XCD's L2 still
serves shared panels), "helpers" share the tails; pieces are parked in the workspace with 16-byte write-through stores and the last piece of a
tile to arrive adds them in K order and runs the epilogue (nobody waits; bitwise reproducible; tests: tests/test_kernels_gpu.py *_pp_splitk3_*).
First form (tile-major iteration space cut into equal contiguous ranges): neighbouring workgroups sit at different K offsets, nothing is shared in
L2 any more -> 5120x2048x8192 236 us against 185 us plain.  Aligned plan: 200-207 us.  Over K (NT / TN / NN, us):
{sweep}
Reading: plain = 1.36 us per K-iteration with 160 CUs busy + 10 us fixed; stream-K = 1.82 us per K-iteration with 256 CUs busy + 59 us fixed.
The part is power-limited: +60 % busy CUs buy +20 % throughput (117 -> 141 tile-iterations per us).  Fixed cost by diagnostic builds
(AFFT_HANDOFF_DIAG, wrong results; 5120x2048xK: K = 1024 | 2048 | 8192, then 5120x8192xK: K = 512 | 2048):
{diag}
-> reading the parked pieces back costs ~31 us, parking them ~12 us (115-190 MB through HBM, everybody at the same time); even with NO exchange
the balanced launch gains only 10 % (165 vs 185 us).  Vendor library on the same shape: 137-140 us.  Conclusion: on this part the idle CUs of a
partial round are worth far less than their count suggests; stream-K stays a tested option (afft_set_gemm_splitk(2 / 4), AFFT_SK_MAX_EFF) and the
automatic mode never picks it.  Side results that were kept: (a) the split-K hand-over of BOTH kernels now uses 16-byte write-through buffer
stores / loads (round 2: 4-byte system-scope atomics, ~6x the time per byte); (b) the LDS-DMA destination is passed as ONE live scalar (ring
address + wave lane) plus an immediate -- round 2 kept 16-22 precomputed M0 values in SGPRs, which the looping instantiations (stream-K, capped
grid) spilled to VGPR lanes and read back between the DMA issues (22 v_readlane per two K-tiles; per K-iteration 1.98 -> 1.82 us).

== 2. LDS-DMA issued inside the MFMA segment instead of the L segment (AFFT_PP_DMA_IN_C, tools/pp_ab.sh) -> 10-18 % SLOWER, switch kept, default off
Idea: the L segment (2 DMA issues + 4-8 fragment reads + counted wait) is the long pole of a phase (stamps: 470 vs 316 cycles); a VMEM issue between
MFMAs is free if the matrix pipe has queued work.  Measured (NT, us; base = same sources, switch off; dmacN = issue after N MFMAs of the 16):
{ab}
-> the issuing wave stops feeding the matrix pipe for the ~100 cycles a global_load_lds takes to issue; in the L segment that time belongs to the
other wave of the SIMD.  The ping-pong split of work is right; a dedicated loader wave per SIMD is not possible (VGPR allocation is per kernel:
three waves of 248 registers do not fit a SIMD).

Re-sweep of the loop's knobs after the M0 change (tools/pp_ab.sh, NT; prio0 / prio2 = AFFT_PP_PRIO, lead5 / lead6 = AFFT_PP_LEAD, dmalast =
AFFT_PP_DMA_FIRST=0; then "stag": half of a group's waves issue their LDS-DMA first while the other half reads fragments, then swap):
{ab2}
{ab3}
-> everything within +-3 % of the defaults (no s_setprio reads 1-3 % faster alone on three shapes and nothing in the step: cfg2 15.12 / 15.51 /
15.63 vs 15.41 / 15.25 / 15.65 ms over three alternating repeats); the staggered order loses 1-4 %.  Defaults unchanged.

The three legs of the loop, one at a time and together (tools/fill_bench.hip with two modes added in round 3: the GEMM's 24 fragment reads per wave
and K-tile from a resident tile without any fill, alone and with its 64 MFMAs; "fill TB/s" of those two rows counts the 192 KiB of fragment reads):
{fill}
-> per 128 K-tiles on 256 CUs: MFMAs alone 0.126 ms (2180 TFLOP/s), fragment reads alone 0.076 ms (85 TB/s = ~145 B/clk per CU: ds_read_b128 is in
the 128-B/clk class, so 192 KiB of reads + 64 KiB of LDS-DMA writes book the LDS array ~80 % of the MFMA time), LDS-DMA alone 0.086 ms (the texture
path, ~48 B/clk per CU).  Reads + MFMAs without any fill already drop to 1731 TFLOP/s, and with the fill to 1379 (simple lock-step schedule) -- the
ping-pong kernel's loop is at the same place (1270-1280 with its epilogue).  Each leg fits under the MFMA time on its own; what is lost is their
serialisation inside a wave (a wave cannot prefetch fragments into the registers its own MFMAs are reading, and its partner on the SIMD needs
reads + DMA issue + their latency -- ~360 cycles -- inside the 256 cycles of a 16-MFMA segment).  That is the number a next attempt has to beat:
fewer fragment bytes per MFMA (a 128x128 wave tile) or fragment double-buffering (96 more VGPRs) -- neither fits two waves per SIMD.

The four-wave kernels (gemm_w4.hip, EXPERIMENTAL=1) re-measured after the M0 change, with their diagnostic builds (tools/w4_variant.sh, tools/w4_ab.sh; us per
launch, "w4" = variant 5, "pp" = the ping-pong kernel in the same library; d1 = no fragment reads, d2 = no LDS-DMA in the loop, d3 = neither, d11 = d3
without the phase barriers -- wrong results, timing only):
{w4diag}
and with the four LDS-DMA instructions of a phase behind every second MFMA group, even / odd waves on even / odd groups ("stag"), or moved to groups
0-3 / 4-7 ("at0" / "at4"):
{w4stag}
-> a single wave per SIMD does keep the matrix pipe full (fill_bench "MFMA only, 4 waves": the same time per MFMA as 8 waves), and the structure's
floor without operands is 593 us for 8192^3 (vendor: 739 with everything).  Fragment reads cost +102 us, LDS-DMA +135 us, both together +306 us --
more than their sum, and no placement of the DMA instructions changes it: the two legs collide in the LDS array, not at issue.  The four-wave kernel
stays at the ping-pong kernel's time (890-900 vs 878-887 us); it needs an operand path that does not write LDS while fragments are read from it at
this rate (or half the fragment traffic again), which neither staging form here provides.

== 3. Weight-gradient GEMMs with the optimizer in the epilogue, per tile shape (tools/wgrad_sgd_bench.py; every launch its own p / momentum buffers)
{wg}
-> the fused update costs +38-40 us per 16.8 M-element weight = 235 MB more HBM traffic at 6.2 TB/s: the optimizer's traffic runs at the HBM
roofline inside the epilogues (36 launches x ~38 us = 1.4 ms per step on the auxiliary stream); 128x128 tiles (two workgroups per CU, one's
epilogue under the other's main loop) do not beat 256x256 tiles on any of the large shapes; the K = 1024 weight gradients of the predictor are
HBM-bound by their epilogue (90 us for 27 us of main loop).

== 4. Schedule: CU cap of the weight-gradient GEMMs, serial weight gradients (tools/r3_cap.sh; clips/s, ms/step; cfg2)
{cap}
-> with the capped (persistent) instantiation no longer spilling in its main loop the caps lose less than in round 2 (cap 192: 15.97 vs 15.56 ms;
round 2: 19.5 ms at cap 128) but still lose; serial weight gradients 17.1 ms.  Default stays: uncapped, two streams.

== 5. Deferred weight gradients of the predictor (built, bitwise-equal in tests, then REMOVED: slower)
The predictor's sub-layers (B*T = 1024 rows) come first in the backward pass; their weight-gradient GEMMs (optimizer traffic in the epilogue) run
beside their own latency-bound data-gradient chain and slow it 2-3x (in-step 82 us for nt 1024x2048x2048 against 33 us alone).  Variant: the
composite backward of small sub-layers skips its weight-gradient half (a defer flag + afft_*_sublayer_wgrad entry points) and every later LARGE
sub-layer backward hands one queued half to the auxiliary stream, the rest at the end of the backward pass.  Parameters / momentum bitwise equal to
the immediate form (t3_m5, t2_flt, 4 Trainer steps).  Same box, clips/s / ms per step: cfg2 4147 / 15.43 immediate vs 4072 / 15.72 deferred; EK100
widths 6632 / 9.65 vs 6157 / 10.39.  The predictor chain does run alone, but the same work then lands beside the fuser's chain (the bulk of the
step) and the tail grows.  Not kept (two more C-ABI entry points for a loss).

== 6. Where the step stands (tools/r3_baseline.sh: rocprofv3 stream timelines; tools/gemm_insitu.py with AFFT_OVERLAP_WGRAD=0)
With every kernel alone (serial weight gradients) the GEMM launches of a step add up to 14.5 ms (256x256 launches 11.1 ms = 836 TFLOP/s, the
predictor's 128x128 launches 3.4 ms), the non-GEMM kernels to 2.7 ms, each within 10-20 % of its own roofline alone (LayerNorm backward 29 us =
5.8 TB/s, attention backward 42 us, optimizer epilogues 6.2 TB/s).  The two-stream step (15.4-15.6 ms) hides 1.6 ms of that sum.  What is left is
the main loop of the 256x256 kernel (65 % of the power-limited MFMA rate; LDS array and matrix pipe both ~100 % booked by construction: 192 KiB of
fragment reads + 64 KiB of LDS-DMA writes per 2048 MFMA cycles) -- see DESIGN.md section 4.

== 7. 128x128 tiles for every GEMM with K < 2048 (the predictor's K = 1024 weight gradients, so that they can share a CU with the chain's 128x128
workgroups; AFFT_PP_MIN_K experiment, removed again)
{mink}
-> noise (cfg2 +0.5 %, EK100 widths -0.6 %).

== 8. Weight-gradient GEMMs cut along K into 2 / 3 sequential launches (so that their workgroups retire two / three times as often and the chain's
kernels find CUs sooner; timing-only hack in sublayer.hip's wgrad(), removed) -- clips/s, ms/step
{ksplit}
-> slower: the extra epilogue pass over the fp32 gradient and the extra launch cost more than the shorter waits return.

== 9. No fused update for sub-layers of fewer than 2048 rows (the predictor's K = 1024 weight gradients become plain 52-us GEMMs, their update goes
back to the per-bucket kernel; AFFT_FUSE_MIN_ROWS experiment, removed; three alternating repeats on one box) -- clips/s, ms/step, loss after 25 steps
{fusemin}
-> fused everywhere stays ahead (cfg2 15.10-15.26 vs 15.58-15.61 ms; EK100 widths mixed inside the box's noise).  The loss is identical: the
step audit of round 3 (Trainer._audit_fused_step) moves the skipped weights back to the regular update on the first fused step.

== 10. Weight gradients of the LATE layers inside the NEXT forward pass (timing prototype, removed)
The forward pass is a serial chain with nothing beside it (q2 idle for 4.9 ms, 96 CUs idle in every N = 2048 GEMM) while the backward pass has both
streams busy.  A layer that runs late in the forward pass (the predictor, the last fuser blocks) runs early in the backward pass, so its updated weights
are not read again for most of the next forward pass: its composite backward skipped the weight-gradient half and the next forward pass handed the
queue to the auxiliary stream when it started (newest first = the order in which that forward needs the weights); bench.py flushed inside the timed
region.  Prototype WITHOUT the per-layer event waits a correct version needs (they could only add stalls; the losses below differ for that reason),
two alternating repeats on one box -- clips/s, ms/step, loss; lazy = the predictor's sub-layers, fuser_blocks = how many trailing fuser blocks too:
{lazy}
-> slower in every setting (cfg2 14.80-15.01 ms plain vs 15.03-15.75 ms).  The work slows the forward chain by as much as it relieves the backward
chain: the two-stream step is bound by what the kernels need in total (matrix pipe / LDS in the GEMMs, HBM in the optimizer epilogues and the
LayerNorm / attention kernels), not by where the weight gradients are placed.  Together with sections 4, 5, 7, 8 and 9 this closes the schedule as
a lever; what is left is the GEMM main loop itself (section 2 and DESIGN.md section 4).

== 11. 128x128 tiles for EVERY weight-gradient GEMM (1024 short-lived workgroups, two per CU, instead of 256 that hold every CU for ~150 us, so that
the chain's kernels find CUs sooner; AFFT_TN_VARIANT experiment, removed; two alternating repeats) -- clips/s, ms/step, loss
{tnvar}
-> slower on cfg2 (14.94 -> 15.73 ms: the small tile's 21 % longer weight gradients cost more than the shorter waits return), a wash at the EK100 widths.
"""
open(os.path.join(R, "profiles", "r03_experiments.txt"), "w").write(T.format(sweep=g("r3_sk_sweep2.txt"), diag=g("r3_sk_diag.txt"), ab=g("r3_pp_ab1.txt"), ab2=g("r3_pp_ab2.txt"), ab3=g("r3_pp_ab3.txt"), fill=g("r3_fill_bench.txt"), w4diag=g("r3_w4_diag.txt"), w4stag=g("r3_w4_stag.txt"),
                                                                          wg=g("r3_wgrad_sgd.txt"), cap=g("r3_cap_sweep.txt"), mink=g("r3_minK.txt"), ksplit=g("r3_ksplit.txt"), fusemin=g("r3_fusemin.txt"), lazy=g("r3_lazy_proto.txt"), tnvar=g("r3_tnvar.txt")))
