"""The evaluation pass of bench.py's `eval` record alone (validate()/test()-shaped: gather, eval forward, head, vote, subset vote), for
rocprofv3:   rocprofv3 --kernel-trace --stats -d gpurun_out/r04_eval_bf16 -- python3 tools/eval_profile.py bf16"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
print(json.dumps(bench.eval_record(torch.device("cuda", 0), [dt], steps=10)))
