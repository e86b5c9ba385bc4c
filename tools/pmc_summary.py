"""Averages rocprofv3 --pmc counter_collection.csv per kernel. usage: pmc_summary.py <dir>"""
import collections, csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in acc.items():
        print(k, "avg_ns", sum(dur[k]) / len(dur[k]))
        for c, x in v.items():
            print("    %-36s %.4g" % (c, x / max(1, cnt[(k, c)])))
