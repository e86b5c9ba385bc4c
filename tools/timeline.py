"""Stream timeline of one training step from a rocprofv3 --kernel-trace CSV.

    python tools/timeline.py <dir with *_kernel_trace.csv> [marker-substring] [out.json]

A step is the window between two consecutive launches of the once-per-step marker kernel (default: the MSE kernel).
Reports, for the median-length window among the last ones: per-queue busy time, the union, the time during which
exactly one queue is busy (who is exposed), the idle time, and the time per kernel family on each queue.
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def family(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.strip()[:70]


def main():
    d = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "mse"
    out = sys.argv[3] if len(sys.argv) > 3 else None
    rows = []
    fs = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))
    if fs:
        with open(fs[0]) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
    else:       # rocpd sqlite output (rocprofv3's default format)
        import sqlite3
        db = sqlite3.connect(sorted(glob.glob(d + "/**/*.db", recursive=True))[0])
        rows = [(int(a), int(b), int(q), n) for a, b, q, n in db.execute("select start, end, queue_id, name from kernels")]
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[3]]
    wins = [(marks[i], marks[i + 1]) for i in range(len(marks) - 1)][-12:]
    wins.sort(key=lambda w: rows[w[1]][0] - rows[w[0]][0])
    a, b = wins[len(wins) // 2]
    t0, t1 = rows[a][0], rows[b][0]
    ks = [r for r in rows[a:b]]
    queues = sorted({r[2] for r in ks})
    ev = []
    for s, e, q, _ in ks:
        e = min(e, t1)
        ev.append((s, 1, q))
        ev.append((e, -1, q))
    ev.sort()
    active = defaultdict(int)
    last = t0
    only = defaultdict(int)
    union = 0
    both = 0
    for t, dlt, q in ev:
        live = [k for k, v in active.items() if v > 0]
        if live:
            union += t - last
            if len(live) == 1:
                only[live[0]] += t - last
            else:
                both += t - last
        active[q] += dlt
        last = t
    # the largest idle intervals (no queue busy) with the kernels either side of them: where the step waits for the host or a join
    gaps = []
    end_so_far, last_name = t0, "(window start)"
    for s_, e_, q_, n_ in sorted(ks):
        if s_ > end_so_far:
            gaps.append((s_ - end_so_far, (end_so_far - t0) / 1e6, last_name, family(n_)))
        if e_ > end_so_far:
            end_so_far, last_name = e_, family(n_)
    gaps.sort(reverse=True)
    busy = defaultdict(int)
    fam = defaultdict(lambda: defaultdict(lambda: [0, 0]))
    for s, e, q, n in ks:
        busy[q] += e - s
        x = fam[q][family(n)]
        x[0] += 1
        x[1] += e - s
    res = {
        "step_ms": (t1 - t0) / 1e6, "union_busy_ms": union / 1e6, "idle_ms": (t1 - t0 - union) / 1e6,
        "overlapped_ms": both / 1e6,
        "idle_gaps": {"count": len(gaps), "over_10us": sum(1 for g in gaps if g[0] > 10000), "sum_over_10us_ms": sum(g[0] for g in gaps if g[0] > 10000) / 1e6,
                      "largest": [{"us": round(g[0] / 1e3, 1), "at_ms": round(g[1], 3), "after": g[2][:50], "before": g[3][:50]} for g in gaps[:12]]},
        "queues": {str(q): {"busy_ms": busy[q] / 1e6, "alone_ms": only[q] / 1e6, "kernels": sum(v[0] for v in fam[q].values()),
                            "families": {k: {"n": v[0], "ms": round(v[1] / 1e6, 3)}
                                         for k, v in sorted(fam[q].items(), key=lambda kv: -kv[1][1])[:25]}}
                   for q in queues},
    }
    txt = json.dumps(res, indent=1)
    print(txt)
    if out:
        open(out, "w").write(txt)


if __name__ == "__main__":
    main()
