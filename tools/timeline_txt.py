"""Text summary of timeline JSONs (tools/timeline.py): python tools/timeline_txt.py <title> <file.json> [<title> <file.json> ...]"""
import json
import sys

args = sys.argv[1:]
for title, fn in zip(args[0::2], args[1::2]):
    d = json.load(open(fn))
    print(f"== {title}")
    print(f"   step {d['step_ms']:.2f} ms, union busy {d['union_busy_ms']:.2f}, idle {d['idle_ms']:.2f}, >=2 queues busy {d['overlapped_ms']:.2f}"
          f"; idle gaps > 10 us: {d['idle_gaps']['over_10us']} ({d['idle_gaps']['sum_over_10us_ms']:.2f} ms)")
    for i, (q, v) in enumerate(sorted(d["queues"].items(), key=lambda kv: -kv[1]["busy_ms"])):
        gemm = sum(f["ms"] for k, f in v["families"].items() if "gemm" in k)
        print(f"   q{i + 1} busy {v['busy_ms']:.2f} ms (GEMM {gemm:.2f}, other {v['busy_ms'] - gemm:.2f}), alone {v['alone_ms']:.2f} ms, {v['kernels']} kernels")
        for k, f in list(v["families"].items())[:14]:
            print(f"      {k[:62]:62} n={f['n']:3d} {f['ms']:6.3f} ms  avg {1e3 * f['ms'] / f['n']:6.1f} us")
