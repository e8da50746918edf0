import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_gpu_small_batch import _step
for dtype in ("f32", "bf16"):
    for groups in (8, 33, 64):
        zs, outs, gs, rs = _step(dtype, groups, False)[:4]
        zl, outl, gl, rl = _step(dtype, groups, True)[:4]
        print(dtype, groups, "z", float((zs - zl).abs().max() / zl.abs().max()), "loss", float(outs[0]), float(outl[0]))
        for k in gl:
            a, b = gs[k].double().flatten(), gl[k].double().flatten()
            if float(b.norm()) == 0: print("   ", k, "zero", float(a.norm())); continue
            print("    %-40s rel %.2e cos %.6f  |b| %.3e" % (k, float((a - b).norm() / b.norm()), float(a @ b / (a.norm() * b.norm())), float(b.norm())))
        for k in rs:
            print("    run %-36s %.2e" % (k, float((rs[k].float() - rl[k].float()).abs().max())))
