"""HBM traffic per launch of each GEMM kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.
gfx950 corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB; FETCH_SIZE reads
exactly half of a wide coalesced stream (128-B requests tallied at 64 B) -> doubled; WRITE_SIZE is exact for 16-B stores.
usage: traffic_summary.py <fetch_dir> <write_dir> <out.json> [config batch precision gpus]   (the workload the passes ran: bench.py
only quotes a profile for the workload it was taken on; default cfg2 64 bf16 1)"""
import collections, csv, glob, json, sys

def per_kernel(d, counter):
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    return {k: (acc[k] / cnt[k], cnt[k]) for k in acc}

fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
wl = sys.argv[4:8]
out = {"_workload": {"config": wl[0] if wl else "cfg2", "batch": int(wl[1]) if len(wl) > 1 else 64,
                     "precision": wl[2] if len(wl) > 2 else "bf16", "gpus": int(wl[3]) if len(wl) > 3 else 1}}
for k in sorted(fetch, key=lambda k: -fetch[k][0] * fetch[k][1]):
    if "gemm" not in k:
        continue
    f_kib, n = fetch[k]
    w_kib = write.get(k, (0.0, 0))[0]
    out[k] = {"launches_profiled": n, "fetch_size_kib_raw": round(f_kib, 1), "write_size_kib": round(w_kib, 1),
              "hbm_bytes_per_launch": int((2.0 * f_kib + w_kib) * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
