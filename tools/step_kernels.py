#!/usr/bin/env python3
"""Per-kernel totals of ONE training step from a rocprofv3 --kernel-trace run of bench.py (rocpd SQLite):
   python tools/step_kernels.py <dir with *_results.db> [--timeline]
   python tools/step_kernels.py <dir> --by_position <kernel name part>    (the i-th launch of that kernel in every traced step: min / avg / max)"""
import collections
import glob
import os
import sqlite3
import sys

csvs = glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True)
if csvs:                                            # (--output-format csv)
    import csv
    rows = sorted(((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                   for r in csv.DictReader(open(max(csvs, key=os.path.getmtime)))), key=lambda r: r[1])
else:
    db = glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True)[0]
    rows = list(sqlite3.connect(db).execute("select name, start, end, duration from kernels order by start"))
idx = [i for i, r in enumerate(rows) if r[0].startswith("gather_groups")]
if "--by_position" in sys.argv:
    pat = sys.argv[sys.argv.index("--by_position") + 1]
    per = collections.defaultdict(list)
    for a, b in zip(idx[:-1], idx[1:]):
        for i, r in enumerate([r for r in rows[a:b] if pat in r[0]]):
            per[i].append(r[3] / 1e3)
    print("%s: launch position within a step, over %d traced steps (us)" % (pat, len(idx) - 1))
    for i in sorted(per):
        v = per[i]
        print("  #%d: min %6.1f  avg %6.1f  max %6.1f   (%d samples)" % (i, min(v), sum(v) / len(v), max(v), len(v)))
    sys.exit(0)
a, b = idx[-2], idx[-1]
tot = collections.defaultdict(lambda: [0, 0.0])
for r in rows[a:b]:
    n = r[0].split("(")[0].replace("void ", "")[:46]
    tot[n][0] += 1
    tot[n][1] += r[3] / 1e3
    if "--timeline" in sys.argv:
        print("%9.1f %8.1f us  %s" % ((r[1] - rows[a][1]) / 1e3, r[3] / 1e3, r[0][:80]))
print("step span %.1f us, kernel time %.1f us, %d launches" % ((rows[b][1] - rows[a][1]) / 1e3, sum(v[1] for v in tot.values()), b - a))
for n, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-48s %3d  %8.1f us  avg %6.1f" % (n, v[0], v[1], v[1] / v[0]))
