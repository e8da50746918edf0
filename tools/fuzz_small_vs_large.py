"""Small-batch kernels against the large-batch kernels (cp_config option no_small) over a sweep of ragged group counts: losses equal to
1e-6 (f32), every parameter gradient's largest deviation printed relative to the tensor's largest entry.
READ WITH CARE: the two paths sum in different orders, so a pre-activation within one f32 rounding step of zero can come out +1e-9 on one
path and -1e-9 on the other -- the forward output moves by 1e-9, the ReLU mask flips, and that element's whole gradient term (1 / rows of a
column sum: 1e-3 .. 6e-2 at these batch sizes) is there or not.  With 0.4-9 M pre-activations per step that happens in about every second
size of the sweep (rows of 1e-2 beside rows of 7e-6 below); it is why tests/test_gpu_small_batch.py bounds the two paths loosely and why the
oracle comparisons of tests/test_gpu_parity.py replay the DEVICE's ReLU masks.  A real indexing fault shows up in
the loss column or as O(1) distances at EVERY size.  usage: python tools/fuzz_small_vs_large.py [dtype=f32]"""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
from test_gpu_small_batch import _step
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
tol = 2e-4 if dtype == "f32" else 3e-2
worst = {}
for groups in (1, 2, 3, 5, 7, 11, 16, 17, 25, 31, 32, 40, 49, 63, 64):
    s = _step(dtype, groups, False)
    l = _step(dtype, groups, True)
    gs, gl = s[2], l[2]
    assert abs(float(s[1][0]) - float(l[1][0])) <= (1e-6 if dtype == "f32" else 2e-2) * abs(float(l[1][0])), (groups, float(s[1][0]), float(l[1][0]))
    bad = []
    for k in gl:
        a, b = gs[k].double().flatten(), gl[k].double().flatten()
        if float(b.norm()) == 0:
            assert float(a.norm()) == 0, (groups, k)
            continue
        rel = float((a - b).abs().max() / b.abs().max())
        worst[k] = max(worst.get(k, 0.0), rel)
        if not rel < tol:
            bad.append((k, rel))
    print(f"{dtype} {groups:3d} groups: loss {float(s[1][0]):.6f} / {float(l[1][0]):.6f}  worst gradient distance {max(worst.values()):.2e}  {('BAD %d tensors, worst here ' % len(bad) + max(bad, key=lambda t: t[1])[0] + ' %.1e' % max(bad, key=lambda t: t[1])[1]) if bad else 'ok'}", flush=True)
    pass
print("largest distance per tensor over the sweep:")
for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:8]:
    print(f"  {k:44s} {v:.2e}")
