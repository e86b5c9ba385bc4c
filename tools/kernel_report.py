"""Per-kernel roofline table of one bench.py command from its rocprofv3 outputs: the --kernel-trace --stats summary (average
duration per kernel symbol) and two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as the MI355X guide prescribes (both
in KiB; FETCH_SIZE reads half of a wide coalesced stream -> doubled; fabric-side: Infinity-Cache hits are counted).  For every
kernel: launches, average us, counter bytes per launch, achieved GB/s against the 8 TB/s HBM peak; for the GEMM symbols also
TFLOP/s against the 2.5 PFLOP/s dense bf16 peak, from the bench line's own per-kernel FLOP accounting (`roofline.by_kernel`).
BASELINE.json configs[4] ("rocprof HBM-BW vs roofline report").
usage: kernel_report.py <stats_dir> <fetch_dir> <write_dir> <bench.json> <out.txt> [title]"""
import collections, csv, glob, json, sys

PEAK_GBS, PEAK_TF = 8000.0, 2500.0


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def stats(d):
    out = {}
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]))
    return out


def pmc(d, counter):
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = short(r["Kernel_Name"])
                acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    return {k: acc[k] / cnt[k] for k in acc}


st, fe, wr = stats(sys.argv[1]), pmc(sys.argv[2], "FETCH_SIZE"), pmc(sys.argv[3], "WRITE_SIZE")
bench = {}
try:
    for line in open(sys.argv[4]):
        if line.startswith("{"):
            bench = json.loads(line)
except Exception:  # noqa: BLE001
    pass
byk = (bench.get("roofline") or {}).get("by_kernel", {})
title = sys.argv[6] if len(sys.argv) > 6 else ""
lines = [title, f"bench line: {bench.get('value')} {bench.get('unit')}, {bench.get('ms_per_step')} ms/step, config: {(bench.get('config') or {}).get('workload')}",
         "peaks: HBM 8 TB/s, dense bf16 MFMA 2.5 PFLOP/s (MI355X_MICROARCH.md).  bytes/launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB from separate --pmc passes",
         "(fabric-side: includes Infinity-Cache hits); us = average over the profiled run (rocprofv3 --kernel-trace --stats); GEMM TFLOP/s from the",
         "bench line's in-step HIP-event timing of the same symbols.", "",
         f"{'kernel':92} {'calls':>6} {'avg us':>8} {'% time':>6} {'MB/launch':>10} {'GB/s':>7} {'of HBM':>7} {'TFLOP/s':>8} {'of MFMA':>8}"]
for k, (calls, us, pct) in sorted(st.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    if pct < 0.05:
        continue
    b = (2.0 * fe.get(k, 0.0) + wr.get(k, 0.0)) * 1024
    gbs = b / (us * 1e-6) / 1e9 if us > 0 and b > 0 else 0.0
    tf = None
    for kk, v in byk.items():
        if kk.replace(" ", "") == k.replace(" ", "") and v.get("tflops"):
            tf = v["tflops"]
    lines.append(f"{k[:92]:92} {calls:6d} {us:8.1f} {pct:6.2f} {b / 1e6:10.1f} {gbs:7.0f} {gbs / PEAK_GBS:7.3f} "
                 + (f"{tf:8.1f} {tf / PEAK_TF:8.3f}" if tf else f"{'':8} {'':8}"))
open(sys.argv[5], "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:40]))
