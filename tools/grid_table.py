"""One line per kernel of a traced run: launches, workgroups, threads per workgroup, average duration -- which launches run on far more
workgroups than the chip has slots for (a thread that sees five rows behind a prologue of 32 loads: round 4's streaming-pass find).
usage (GPU box): rocprofv3 --kernel-trace -d gpurun_out/gt -- python3 bench.py --main_only --steps 3 --warmup 2 [...]; python tools/grid_table.py gpurun_out/gt"""
import csv
import glob
import sys
from collections import defaultdict

rows = defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0][-60:]
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // wg
        k = (name, grid, wg)
        rows[k][0] += 1
        rows[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for (name, grid, wg), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print("%-62s %5d x  %6d workgroups x %4d threads   avg %8.1f us" % (name, n, grid, wg, t / n))
