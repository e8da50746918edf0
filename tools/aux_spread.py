"""step-time spread of bench.py's main region with and without the second stream: python tools/aux_spread.py [dtype]"""
import json, os, subprocess, sys
dt = sys.argv[1] if len(sys.argv) > 1 else "fp8"
for rep in range(4):
    for aux in ("1", "0"):
        out = subprocess.run([sys.executable, "bench.py", "--main_only", "--dtype", dt, "--steps", "40"], capture_output=True, text=True,
                             env=dict(os.environ, CPNATIVE_AUX_STREAM=aux)).stdout
        r = json.loads(out.strip().splitlines()[-1])
        s = r["steps_spread"]
        print(f"{dt} aux={aux}: mean {r['ms_per_step']:.4f} min {s['min_ms']:.4f} median {s['median_ms']:.4f} max {s['max_ms']:.4f}")
