#!/usr/bin/env python3
"""Turn rocprofv3 output (gpurun_out/...) into the small summaries committed under profiles/.

  kernel stats :  python tools/parse_profile.py stats  <dir with *_kernel_stats.csv>  profiles/rNN_kernel_stats.csv
  HBM traffic  :  python tools/parse_profile.py traffic <fetch dir> <write dir> profiles/rNN_traffic.json
  MFMA busy    :  python tools/parse_profile.py mfma <sq counter dir> profiles/rNN_mfma.json

MFMA busy: a --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE pass (own run, --kernel-trace only).  Per the guide
SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles summed over the SIMDs (32 per v_mfma_f32_32x32x16_bf16) and
GRBM_GUI_ACTIVE is the sum of the 8 XCDs' active clocks, so for a launch
    mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) * 1024 SIMDs)
and SQ_VALU_MFMA_BUSY_CYCLES / 32 is the number of MFMA instructions it issued (checked against the algorithmic count).

Traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE come from
separate --pmc passes (they do not fit one pass), are in KiB, and on gfx950 FETCH_SIZE reports exactly half
the bytes of a wide coalesced read stream, so  bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def one(pattern, required=True):
    files = glob.glob(pattern, recursive=True)
    if not files:
        if required:
            raise SystemExit(f"no file matches {pattern}")
        return None
    return files[0]


def db_of(d):
    """rocprofv3 of ROCm 7.2 writes a rocpd SQLite file (*_results.db) unless --output-format csv is given."""
    import sqlite3
    return sqlite3.connect(one(os.path.join(d, "**", "*_results.db")))


def stats_from_db(d, out):
    import math
    con = db_of(d)
    agg = collections.defaultdict(list)
    for name, dur in con.execute("select name, duration from kernels"):
        agg[name].append(float(dur))
    total = sum(sum(v) for v in agg.values())
    rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
    with open(out, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for name, v in rows:
            mean = sum(v) / len(v)
            sd = math.sqrt(sum((x - mean) ** 2 for x in v) / len(v))
            w.writerow([name, len(v), int(sum(v)), round(mean, 3), round(100 * sum(v) / total, 4), int(min(v)), int(max(v)), round(sd, 3)])


def per_kernel_db(d, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for name, val in db_of(d).execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        agg[name][0] += float(val)
        agg[name][1] += 1
    return agg


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]][0] += float(r["Counter_Value"])
            agg[r["Kernel_Name"]][1] += 1
    return agg


def main():
    mode = sys.argv[1]
    if mode == "stats":
        src = one(os.path.join(sys.argv[2], "**", "*_kernel_stats.csv"), required=False)
        if src:
            shutil.copy(src, sys.argv[3])
        else:
            stats_from_db(sys.argv[2], sys.argv[3])
    elif mode == "traffic":
        fc = one(os.path.join(sys.argv[2], "**", "*_counter_collection.csv"), required=False)
        wc = one(os.path.join(sys.argv[3], "**", "*_counter_collection.csv"), required=False)
        f = per_kernel(fc, "FETCH_SIZE") if fc else per_kernel_db(sys.argv[2], "FETCH_SIZE")
        w = per_kernel(wc, "WRITE_SIZE") if wc else per_kernel_db(sys.argv[3], "WRITE_SIZE")
        out = {}
        for k in sorted(set(f) | set(w)):
            fb = 2 * f[k][0] / max(f[k][1], 1) * 1024 if k in f else 0.0
            wb = w[k][0] / max(w[k][1], 1) * 1024 if k in w else 0.0
            out[k] = dict(launches=int(max(f[k][1] if k in f else 0, w[k][1] if k in w else 0)),
                          read_bytes_per_launch=fb, write_bytes_per_launch=wb, hbm_bytes_per_launch=fb + wb)
        # the step's own kernels: bench.py's copy-rate measurement (__amd_rocclr_copyBuffer, 1 GiB copies after the timed region) and
        # torch's fills are not part of a training step; gather_groups_kernel runs once per step
        step_k = {k: v for k, v in out.items() if not k.startswith("__amd_rocclr") and "at::native" not in k}
        tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in step_k.values())
        steps = max([v["launches"] for k, v in out.items() if k.startswith("gather_groups_kernel")] or [0])
        json.dump(dict(note="2*FETCH_SIZE*1024 + WRITE_SIZE*1024 per launch (gfx950 correction, separate PMC passes)",
                       steps_profiled=steps, hbm_bytes_per_step=tot / steps if steps else None, kernels=out), open(sys.argv[4], "w"), indent=1)
        print("HBM bytes of the step's kernels over the profiled run: %.2f GB in %d steps = %.2f GB/step" % (tot / 1e9, steps, tot / 1e9 / max(steps, 1)))
    elif mode == "mfma":
        d = sys.argv[2]
        cc = one(os.path.join(d, "**", "*_counter_collection.csv"), required=False)
        get = (lambda c: per_kernel(cc, c)) if cc else (lambda c: per_kernel_db(d, c))
        busy, sq, gui = get("SQ_VALU_MFMA_BUSY_CYCLES"), get("SQ_BUSY_CYCLES"), get("GRBM_GUI_ACTIVE")
        out = {}
        for k in sorted(busy):
            n = max(busy[k][1], 1)
            b = busy[k][0] / n
            g = gui[k][0] / max(gui[k][1], 1) if k in gui else 0.0
            if b <= 0:
                continue
            # busy cycles per instruction: 16x16x32 bf16 = 16, 32x32x16 bf16 = 32, MX 16x16x128 e4m3 = 32, MX 32x32x64 (gemm_tn8) = 64
            per = 16.0 if ("gemm_ws16" in k or "gemm_wsd16" in k) else 64.0 if "gemm_tn8" in k else 32.0
            out[k] = dict(launches=int(n), mfma_busy_cycles_per_launch=b, mfma_instructions_per_launch=b / per,
                          gui_active_per_launch=g, sq_busy_cycles_per_launch=sq[k][0] / max(sq[k][1], 1) if k in sq else None,
                          mfma_busy_frac=(b / ((g / 8.0) * 1024.0)) if g > 0 else None)
        json.dump(dict(note="SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) * 1024 SIMDs) per launch; busy / 32 = MFMA instructions "
                            "(v_mfma_f32_32x32x16_bf16 and the MX 16x16x128 e4m3 form; busy / 16 for the v_mfma_f32_16x16x32_bf16 kernels gemm_ws16 / gemm_wsd16, busy / 64 for gemm_tn8's MX 32x32x64)", kernels=out), open(sys.argv[3], "w"), indent=1)
        for k, v in sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_launch"] * kv[1]["launches"])[:8]:
            print("%-60s busy frac %s  MFMAs/launch %.3g" % (k[:60], "%.3f" % v["mfma_busy_frac"] if v["mfma_busy_frac"] else "n/a",
                                                              v["mfma_instructions_per_launch"]))
    elif mode == "counters":
        # every counter of a --pmc pass, averaged per launch and kernel:  parse_profile.py counters <dir> <out.json> [name filter]
        d = sys.argv[2]
        flt = sys.argv[4] if len(sys.argv) > 4 else ""
        agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
        for name, cname, val in db_of(d).execute("select kernel_name, counter_name, value from counters_collection"):
            if flt and flt not in name:
                continue
            a = agg[name][cname]
            a[0] += float(val)
            a[1] += 1
        out = {k: {c: v[0] / max(v[1], 1) for c, v in cs.items()} | {"launches": max(v[1] for v in cs.values())} for k, cs in agg.items()}
        json.dump(dict(note="per-launch averages of one rocprofv3 --pmc pass (SQ_* wave/wait counters are in quad-cycles summed over waves, "
                            "SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over SIMDs, GRBM_GUI_ACTIVE summed over the 8 XCDs)", kernels=out),
                  open(sys.argv[3], "w"), indent=1)
        for k, cs in sorted(out.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0) * kv[1]["launches"])[:14]:
            print(k[:70])
            print("    " + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(cs.items())))
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
