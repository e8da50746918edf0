#!/usr/bin/env python3
"""Benchmark of the contrastive sEMG training step on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One step = one full optimisation step of code/train.py:95-108 on one batch of synthetic
Ninapro-shaped windows already resident in HBM: group gather -> EMG encoder forward ->
(N>1: RCCL all-gather of the z embeddings, read by the global-negatives column loss) -> class encoder +
41x41 logits + symmetric CE -> backward -> (N>1: RCCL all-reduce of the 8 MB flat gradient) -> L2 regulariser + 2 x Adam.
N = 1 is BASELINE.json configs[1] with the reference's per-group loss and no collective.  N > 1 is configs[2] ("global batch
... with RCCL z all-gather"): the gathered z matrix feeds the global-negatives extension of the class->EMG direction
(cp_global_negatives + cp_head_gneg, the same code path as train.py --global_negatives); --global_negatives off drops
the collective and the extension, on forces them at N = 1 (where the "gathered" matrix is the rank's own z).
Workload at every N: BASELINE.json configs[1] per GPU ("synthetic 12-ch sEMG, 41-class one-hot, batch
4096, bf16"): 4096 groups = 167,936 windows per GPU per step (weak scaling; configs[2] is N=8).

Prints ONE JSON line (rank 0).  `roofline` is measured live, inside the timed region, with HIP
events recorded by the library on the stream the kernels run on (cp_profile_*); `cpu_baseline` is
the CPU oracle (a port of the reference step, see oracle/ref_cpu.py) timed on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T = 41
BEST = dict(d_e=16, lr_emg=9.761e-4, reg_emg=7.103e-5, dp_emg=0.0635,
            lr_glove=2.653e-3, reg_glove=2.840e-6, dp_glove=0.3817)   # reference data/cross_val_keys.npy[54]
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3, "fp8": 5000.0}


# profiler kind -> the ONE kernel it times, as rocprofv3 names it ({dyn} = the tile schedule, include/cpnative.h
# cp_config.tile_schedule: "false" static, "true" dynamic).  The library has ONE kernel per kind since round 3 (the superseded
# variants moved to the tools-only build, csrc/variants.cuh).
GEMM_KERNELS = {
    "bf16": {
        "fc_fwd_ws": "gemm_ws16_kernel",                      # weight-stationary forward, fc2..fc7 (K = 512), static schedule
        "fc_fwd": "gemm_ws16n_kernel",                        # weight-stationary, 32 features x all 768 k per wave: fc1 (K = 768)
        "fc_dgrad": "gemm_nt256p_kernel<1, 4, {dyn}>",        # persistent, plain data gradient (only under CP_OPT_UNFUSED_BN_BWD)
        "fc_dgrad_bn": "gemm_wsd16_kernel<0>",                # + BN/ReLU backward of the layer below against the saved activation
        "fc_dgrad_stats": "gemm_wsd16_kernel<1>",             # behind a dropout: mask + BN-backward sums against the saved activation
        "fc_wgrad": "gemm_tn256_kernel",
    },
    "fp8": {                                                  # csrc/fp8.cuh: block-scaled MFMA, e4m3 activations / weights, e5m2 gradients
        "fc_fwd_ws": "gemm_ws8_kernel<512>",
        "fc_fwd": "gemm_ws8_kernel<768>",
        "fc_dgrad_bn": "gemm_wsd8_kernel<0, false>",
        "fc_dgrad_conv": "gemm_wsd8_kernel<0, true>",         # fc1's data gradient: 16-bit output for the conv kernels
        "fc_dgrad_stats": "gemm_wsd8_kernel<1, false>",
        "fc_wgrad": "gemm_tn8_kernel",
    },
}
GEMM_KERNELS["f32"] = {k: "gemm_nt_kernel / gemm_tn_kernel<float>" for k in GEMM_KERNELS["bf16"]}


def gemm_symbol(kind: str, dyn: str, dtype: str = "bf16") -> str:
    if dtype == "bf16" and (os.environ.get("CPNATIVE_TILE_SCHEDULE") == "dynamic" or dyn == "true"):
        if kind in ("fc_fwd_ws", "fc_fwd"):
            return "gemm_nt256p_kernel<0, 4, true>"
        if kind == "fc_dgrad_bn":
            return "gemm_nt256p_kernel<3, 4, true>"
        if kind == "fc_dgrad_stats":
            return "gemm_nt256p_kernel<4, 4, true>"
    return GEMM_KERNELS[dtype][kind].format(dyn=dyn)


def gemm_model(kind: str, n: int, dtype: str, dropout: bool, paired_wgrads: int = 2):
    """Algorithmic (bytes, flops) of ONE average launch of a GEMM kind over n windows (DESIGN.md
    'measurement'): fc layers are 768->512 then 6 x 512->512; every activation/gradient element is
    moved once per kernel that must touch it (BN-barrier model), at the element size of the dtype
    (fp8: 1 byte; fc1's data gradient writes 2-byte elements for the conv kernels).  Dropout sits on the inputs
    of fc5..fc7: their data-gradient launches (kind fc_dgrad_stats) also read the saved activation (fp8: the dropout
    output); the others apply BatchNorm + ReLU backward of the layer below against its saved activation."""
    ks = [768] + [512] * 6
    es = {"f32": 4, "bf16": 2, "fp8": 1}[dtype]
    ws = dtype != "f32" and os.environ.get("CPNATIVE_TILE_SCHEDULE") != "dynamic"
    out_es = lambda i: es
    if kind == "fc_fwd_ws":         # read input, write post-ReLU output: the K = 512 layers on the weight-stationary kernel
        layers, per = (range(1, 7) if ws else range(0)), lambda k: k + 512
    elif kind == "fc_fwd":          # fc1 (K = 768) on its own kernel, or every layer without the weight-stationary kernels
        layers, per = (range(0, 1) if ws else range(7)), lambda k: k + 512
    elif kind == "fc_dgrad":        # read g_y, write g_v (only with CPNATIVE_UNFUSED_BN_BWD: the plain persistent launch)
        layers, per = range(0), lambda k: 512 + k          # (only under CP_OPT_UNFUSED_BN_BWD: never in this bench)
    elif kind == "fc_dgrad_bn":     # read g_y and the saved activation of the layer below, write its dL/d(pre-activation)
        lo = 1 if dtype == "fp8" else 0                       # (fp8: fc1's launch is its own kind)
        layers, per = (range(lo, 4) if dropout else range(lo, 7)), lambda k: 512 + 2 * k
    elif kind == "fc_dgrad_conv":   # fp8, fc1: read g_y (1 B) and conv2's saved output (1 B), write 2-byte gradients
        layers, per = (range(0, 1) if dtype == "fp8" else range(0)), lambda k: 512 + k + 2 * k
    elif kind == "fc_dgrad_stats":  # behind a dropout: read g_y and the saved activation, write g_v
        layers, per = (range(4, 7) if dropout else range(0)), lambda k: 512 + 2 * k
    else:                           # fc_wgrad: read g_y and the layer input
        layers, per = range(7), lambda k: 512 + k
    layers = list(layers)
    if not layers:
        return 0.0, 0.0
    launches = len(layers)
    if kind == "fc_wgrad" and dropout and dtype != "f32":
        launches -= paired_wgrads   # behind a dropout: fc7+fc6 (and, on one stream, fc5+fc4) share one launch each (api.hip, defer_wgrad)
    byts = sum(n * es * per(ks[i]) for i in layers) / launches
    flops = sum(2.0 * n * 512 * ks[i] for i in layers) / launches
    return byts, flops


def measure_practical_peaks(dev, dtype):
    """What this box sustains, measured here in ~30 ms: HBM by a 1 GiB device-to-device copy (read + write bytes / time), the
    matrix pipe from the committed probe outputs (profiles/): a live MFMA loop would need its own kernel in the library."""
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize(dev)
    gbs = 5 * 2 * 4 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b
    mfma = {"bf16": dict(tflops=1200.0, source="DESIGN.md section 4 / profiles/r02_probes.txt: MFMA + LDS-read loop on random bf16 data, chip at its power cap"),
            "fp8": dict(tflops=3300.0, source="profiles/r03_fp8_probe.txt: v_mfma_scale_f32_16x16x128_f8f6f4 loop on random e4m3 data"),
            "f32": dict(tflops=155.0, source="MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 measured")}[dtype]
    return dict(hbm_gbs=gbs, hbm_source="1 GiB torch device copy timed in this run (read + write)", mfma_tflops=mfma["tflops"], mfma_source=mfma["source"])


def cpu_baseline(seconds: float, threads: int):
    """The reference step on host cores: oracle (pure torch CPU restatement: per-item gather, 3x3
    Conv2d, per-group CE loop, separate norms, 2 x Adam).  Headline = B=64 groups (its best CPU batch);
    `b8` = BASELINE.json configs[0]'s own batch size (8 groups), about a third of the sample time.  The port's step time is
    held to the imported reference's in the build container by tools/time_port_vs_reference.py
    (profiles/r02_port_vs_reference.json: within 10 % at both sizes)."""
    from oracle import ref_cpu as oc
    torch.set_num_threads(threads)
    EMG, GLOVE = oc.synthetic_resident(1234, glove_d=64)
    db = oc.OracleDB23(EMG, GLOVE)
    db.set_mode("train")
    torch.manual_seed(0)
    emg_rand = oc.make_rand_table(torch.rand(db.TASKS, db.D))
    glove_rand = oc.make_rand_table(torch.rand(db.TASKS, db.D_g))

    def run(B, secs):
        sd = oc.init_state_dict(0, 16, adabn=False)
        m = oc.OracleModel(sd, BEST, adabn=False, requires_grad=True)
        m.set_train()
        opts = m.make_optimizers()

        def one():
            idx = torch.randperm(db.D)[:B]
            e, g, lab = oc.collate(db, emg_rand, glove_rand, idx)
            m.train_step(e, g, lab.reshape(-1), opts)

        one()
        one()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < secs:
            one()
            n += 1
        dt = time.perf_counter() - t0
        return dict(value=n * B * T / dt, unit="windows/s",
                    sample=f"{n} steps of B={B} groups ({B * T} windows) in {dt:.1f} s, fp32, oracle/ref_cpu.py")

    big = run(64, seconds * 2.0 / 3.0)
    small = run(8, seconds / 3.0)
    return dict(value=big["value"], unit="windows/s", cores=threads, kind="port", sample=big["sample"],
                b8=dict(small, cores=threads, note="BASELINE.json configs[0] batch size (train.py --batch_size 8 --no_adabn)"))


def small_batch_record(dev, dtype: str, seconds: float = 0.6):
    """The same training step at the reference's own batch sizes (code/train.py:185 --batch_size 8; 32 for the HPO sweeps), the kernels of
    csrc/small.cuh: ms per step issued call by call, inputs resident, a fresh engine per size.  Reported beside cpu_baseline.b8; it is
    not `value`.  8-bit storage has no small-batch path (its scale state needs a history): bf16 stands in on an fp8 line."""
    from contrastiveprosthetics_amd.engine import Engine
    dt = "bf16" if dtype == "fp8" else dtype
    D = 1800
    g = torch.Generator().manual_seed(7)
    table = (torch.randn(T, 1, 12, generator=g) + torch.randn(T, D, 12, generator=g)).reshape(T * D, 12).to(dev)
    emg_rand = (torch.rand(T, D, generator=g).argsort(-1) + torch.arange(T).reshape(T, 1) * D).to(dev)
    out = dict(dtype=dt, note="one process, one stream, call by call (tools/graph_bench.py: the same step as one HIP graph)")
    for Bs in (8, 32):
        e = Engine(adabn=False, dtype=dt, dp_emg=BEST["dp_emg"], device=dev, seed=1)
        e.init_parameters(2)
        perms = [torch.randperm(D, generator=g)[:Bs].to(dev) for _ in range(32)]
        labels = torch.arange(T).repeat(Bs).to(dev)

        def one(p):
            x = e.gather(table, emg_rand, p, 1)
            z = e.encoder_forward(x, training=True)
            e.head(z, labels, 1, want_grad=True)
            e.encoder_backward(x)
            e.adam_step(BEST)
        for p in perms[:8]:
            one(p)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < seconds / 2:
            for p in perms:
                one(p)
            torch.cuda.synchronize(dev)
            n += len(perms)
        ms = (time.perf_counter() - t0) / n * 1e3
        out[f"b{Bs}"] = dict(groups=Bs, windows=Bs * T, ms_per_step=ms, windows_per_s=Bs * T / ms * 1e3, steps_timed=n)
    return out


def _side_failed(rec_key, rec, ex, dev):
    """A side record failed: it never costs the line its main measurement -- unless the device itself is in an error state (a sticky
    hipError would make every later number meaningless): then the run ends non-zero instead of printing a line."""
    rec[rec_key] = dict(error=f"{type(ex).__name__}: {ex}")
    try:
        torch.cuda.synchronize(dev)
    except Exception as ex2:
        print(f"bench.py: device error after the {rec_key} side record ({ex2}); aborting without a result line", file=sys.stderr, flush=True)
        sys.exit(3)


def _time_steps(fn, n_warm, n_timed, dev):
    """seconds per call of a SIDE record: the median of per-call HIP-event times (one allocator stall in a ten-call region -- seen once:
    60 ms in the 8-bit evaluation record -- would otherwise read as the record's figure; the main metric is the plain mean of its region)"""
    for i in range(n_warm):
        fn(i)
    torch.cuda.synchronize(dev)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_timed + 1)]
    marks[0].record()
    for i in range(n_timed):
        fn(n_warm + i)
        marks[i + 1].record()
    torch.cuda.synchronize(dev)
    ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(n_timed))
    return ms[len(ms) // 2] * 1e-3


def eval_record(dev, dtypes, groups=160, V=25, steps=10):
    """The product's use-case (/root/reference/code/train.py:27-63, code/models.py:138-163): a validate()/test()-shaped
    pass -- gather of `groups` x 41 x V windows (V = 25 consecutive samples per window group), encoder forward in eval mode (stock BN:
    running statistics), 41 x 41 logits + loss + argmax for the groups x V sample groups, the 25-sample majority vote (cp_vote) and the
    class-subset restriction (cp_subset_vote over 64 random subsets) -- inputs resident, timed call by call.  Forward-only byte model:
    every one of the 9 stored activations written once and read once by the next layer (2 x 5,120 elements per window; the eval
    forward runs the training kernels, i.e. does NOT fold the running statistics across layers), 4.25 MFLOP per window."""
    from contrastiveprosthetics_amd import engine as E
    from contrastiveprosthetics_amd.engine import Engine
    D = 2000                                                     # test-mode table: D rows of 25 consecutive samples per class
    g = torch.Generator().manual_seed(11)
    table = (torch.randn(T, 1, 1, 12, generator=g) + torch.randn(T, D, V, 12, generator=g)).reshape(T * D * V, 12).to(dev)
    emg_rand = (torch.rand(T, D, generator=g).argsort(-1) + torch.arange(T).reshape(T, 1) * D).to(dev)
    labels = torch.arange(T).repeat(groups).to(dev)
    masks = (torch.rand(64, T, generator=g) < 0.5).to(torch.uint8)
    masks[:, 0] = 1
    perms = [torch.randperm(D, generator=g)[:groups].to(dev) for _ in range(steps + 3)]
    n = groups * T * V
    out = dict(groups=groups, V=V, windows=n, note="gather + eval forward + head + cp_vote + cp_subset_vote (64 subsets), stock BN running statistics, "
               "call by call; forward-only model: 2 x 5,120 stored elements and 4.25 MFLOP per window")
    for dt in dtypes:
        e = Engine(adabn=False, dtype=dt, dp_emg=BEST["dp_emg"], device=dev, seed=3)
        e.init_parameters(2)
        xw = e.gather(table, emg_rand, perms[0], V)
        for _ in range(2):                                       # running statistics (and, in 8 bits, scales) from two training-mode passes
            e.encoder_forward(xw, training=True)
        parts = {}

        def one(i, parts=parts, e=e):
            x = e.gather(table, emg_rand, perms[i], V)
            z = e.encoder_forward(x, training=False)
            o, pred, logits = e.head(z, labels, V, want_grad=False, want_logits=True)
            curve, y_pred = e.vote(pred, labels, groups, V)
            E.subset_vote(logits, labels[:T].contiguous(), groups, V, masks)
            parts["o"] = o
        sec = _time_steps(one, 3, steps, dev)

        def fwd_only(i, e=e, xw=xw):
            e.encoder_forward(xw, training=False)
        sec_f = _time_steps(fwd_only, 2, steps, dev)
        es = {"f32": 4, "bf16": 2, "fp8": 1}[dt]
        byts = n * (2 * 5120 * es + 48 + 64)
        out[dt] = dict(ms=sec * 1e3, windows_per_s=n / sec, forward_ms=sec_f * 1e3, loss=float(parts["o"][0]),
                       roofline=dict(bound="hbm" if dt != "f32" else "mfma", algorithmic_bytes=byts, achieved=byts / sec_f / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                                     frac=byts / sec_f / 1e9 / HBM_PEAK_GBS, mfma_tflops=4.249e6 * n / sec_f / 1e12,
                                     mfma_frac=4.249e6 * n / sec_f / 1e12 / MFMA_PEAK_TFLOPS[dt], scope="encoder forward alone (forward_ms)"))
        del e
    return out


def glove_record(dev, dtype, B, table, emg_rand, perms, labels, steps):
    """BASELINE.json configs[3] (0-based, as everywhere in this repo; SURVEY 8 counts the same list from 1: "cfg4"): the same training step with the glove-angle (20-dim) class encoder instead of the one-hot
    table (cp_glove_forward / cp_head_glove / cp_glove_backward), timed in the run of the default line.  Its own full line:
    python bench.py --class_encoder glove."""
    from contrastiveprosthetics_amd.engine import Engine
    g = torch.Generator().manual_seed(77)
    gm = torch.randn(T, 20, generator=g)
    rows = (gm[None] + 0.3 * torch.randn(B, T, 20, generator=g)).to(dev)
    e = Engine(adabn=False, dtype=dtype, dp_emg=BEST["dp_emg"], device=dev, seed=1000, class_encoder="glove")
    e.init_parameters(seed=42)
    e.workspace(B * T)
    res = {}

    def one(i):
        x = e.gather(table, emg_rand, perms[i % len(perms)], 1)
        z = e.encoder_forward(x, training=True)
        zg = e.glove_forward(rows, training=True)
        o, _, _ = e.head_glove(z, zg, labels, 1, want_grad=True)
        e.glove_backward()
        e.encoder_backward(x)
        e.adam_step(BEST)
        res["o"] = o
    sec = _time_steps(one, 4, steps, dev)
    # the class encoder's own launches, timed alone on the same engine
    def glove_only(i):
        zg = e.glove_forward(rows, training=True)
        e.glove_backward()
    z = e.encoder_forward(e.gather(table, emg_rand, perms[0], 1), training=True)
    e.head_glove(z, e.glove_forward(rows, training=True), labels, 1, want_grad=True)
    sec_g = _time_steps(glove_only, 2, steps, dev)
    return dict(ms_per_step=sec * 1e3, value=B * T / sec, unit="windows/s", steps=steps, loss=float(res["o"][0]), dtype=dtype,
                glove_forward_backward_ms=sec_g * 1e3,
                note="configs[3]: glove-angle class encoder (Linear(20,256) -> BN -> ReLU -> Linear(256,16), code/models.py:386-391,425-428,460-461); "
                     "parity unpinned (the reference keeps these layers as comments)")


def _timed_mode(mode, k_other, step, barrier, args, use_dist, dist, dev, world, N):
    """ms per step of the training step under another loss workload (bench.py `other_steps`)."""
    for i in range(2):
        step(i, mode)
    barrier()
    t1 = time.perf_counter()
    for i in range(k_other):
        step(args.warmup + (i % args.steps), mode)
    barrier()
    el = time.perf_counter() - t1
    if use_dist:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    return dict(ms_per_step=1e3 * el / k_other, value=world * N * k_other / el, steps=k_other,
                loss={"off": "reference per-group loss", "gather": "global negatives, z all-gather",
                      "reduce": "global negatives, partial sums + two 64-float all-reduces"}[mode])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch_size", type=int, default=4096, help="groups of 41 windows per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "fp8"])
    ap.add_argument("--adabn", action="store_true", help="AdaBN instead of stock BN (--no_adabn is what BASELINE.json configs[0] runs)")
    ap.add_argument("--dp_emg", type=float, default=BEST["dp_emg"])
    ap.add_argument("--cpu_seconds", type=float, default=15.0)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--main_only", action="store_true",
                    help="profiling runs (tools/profile_round.sh): the timed region alone -- no CPU baseline, no other_steps, no small_batch")
    ap.add_argument("--breakdown", action="store_true", help="extra untimed pass: per-kernel-kind times to stderr")
    ap.add_argument("--profile_every", type=int, default=20,
                    help="bracket the fc GEMM launches with HIP events (the live roofline numbers) on every n-th timed step")
    ap.add_argument("--global_negatives", default="auto", choices=["auto", "on", "gather", "reduce", "off"],
                    help="auto: gather when N > 1 (BASELINE.json configs[2] names the z all-gather; it must have a reader), off at N = 1 (configs[1]: "
                         "the reference's per-group loss).  on = gather.  reduce: the same {G, H} table from per-rank partial sums and two "
                         "64-float all-reduces (cp_global_negatives_g / _h): no z moves.  Whatever is chosen, the line also carries the "
                         "other workloads' step times from a second, shorter timed region (`other_steps`), so that the N = 1 and N > 1 "
                         "lines of a scaling run hold one workload between them")
    ap.add_argument("--sync_bn", action="store_true", help="BatchNorm statistics over the global batch (18 small all-reduces per step)")
    ap.add_argument("--class_encoder", default="onehot", choices=["onehot", "glove"],
                    help="glove = BASELINE.json configs[3] (glove-angle class encoder); the default line is configs[1] (one-hot)")
    args = ap.parse_args()
    if args.main_only:
        args.no_cpu_baseline = True

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}")
    # CP_BENCH_REHEARSE=gloo: rehearsal of the multi-rank control flow on a ONE-GPU box (all ranks share cuda:0,
    # collectives over gloo); timings of such a run mean nothing and the JSON line says so.
    rehearse = os.environ.get("CP_BENCH_REHEARSE", "")
    if rehearse:
        local_rank = min(local_rank, torch.cuda.device_count() - 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # CP_BENCH_FORCE_DIST=1 runs the collective code path at world size 1 too (used to rehearse the RCCL
    # calls on a single-GPU box under torch.distributed.run --nproc-per-node 1)
    use_dist = world > 1 or os.environ.get("CP_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group(rehearse)
        else:
            dist.init_process_group("nccl", device_id=dev)

    from contrastiveprosthetics_amd.engine import Engine

    B = args.batch_size
    N = B * T
    params = dict(BEST, dp_emg=args.dp_emg)
    eng = Engine(adabn=args.adabn, dtype=args.dtype, dp_emg=args.dp_emg, device=dev, seed=1000 + rank,
                 class_encoder=args.class_encoder)
    for o in filter(None, os.environ.get("CP_BENCH_OPTIONS", "").split(",")):      # A/B runs: cp_config.options of the main engine (e.g. finalize_launches)
        eng.options[o] = 1
    eng.init_parameters(seed=42)                      # identical replicas, as DDP broadcasts them
    eng.workspace(N)

    # synthetic resident table shaped like DB23.EMG_use in --db2 train mode: 41 classes x D rows x 12 ch
    D = max(20000, B)
    g = torch.Generator().manual_seed(1234 + rank)
    mu = torch.randn(T, 1, 12, generator=g)
    table = (mu + torch.randn(T, D, 12, generator=g)).reshape(T * D, 12).to(dev)
    emg_rand = (torch.rand(T, D, generator=g).argsort(-1) + torch.arange(T).reshape(T, 1) * D).to(dev)
    labels = torch.arange(T).repeat(B).to(dev)
    total = args.warmup + args.steps + 2
    perms = [torch.randperm(D, generator=g)[:B].to(dev) for _ in range(total)]
    gn = {"on": "gather", "auto": "gather" if use_dist else "off"}.get(args.global_negatives, args.global_negatives)
    if args.class_encoder != "onehot":
        gn = "off"
    gneg = gn != "off"
    z_all = torch.empty(world * N, 16, device=dev) if (use_dist and args.class_encoder == "onehot") else None
    if args.sync_bn and use_dist:
        eng.set_sync_bn(lambda t: dist.all_reduce(t), world)
    state = {}
    glove_rows = None
    if args.class_encoder == "glove":          # per-(group, class) glove-angle rows: class mean + within-class spread
        gm = torch.randn(T, 20, generator=g)
        glove_rows = (gm[None] + 0.3 * torch.randn(B, T, 20, generator=g)).to(dev)

    reduce_grads = None
    if use_dist:
        from contrastiveprosthetics_amd.dist import GradAllReduce
        reduce_grads = GradAllReduce(eng, force=True)

    def step(i, mode=None):
        mode = mode or gn
        x = eng.gather(table, emg_rand, perms[i], 1)
        z = eng.encoder_forward(x, training=True)
        gh = None
        if mode == "gather":
            # global-batch z matrix over xGMI (north_star): every rank's rows, rank-major; its reader is the column
            # direction of the loss (all windows of other classes in the global batch are negatives)
            if use_dist:
                dist.all_gather_into_tensor(z_all, z)
            gh = eng.global_negatives(z_all if use_dist else z, labels)
        elif mode == "reduce":
            # the same table from per-rank partial sums: two 256-byte all-reduces, no z moves
            gh = eng.global_negatives(z, labels, all_reduce=(lambda t: dist.all_reduce(t)) if use_dist else (lambda t: t))
        if glove_rows is not None:
            zg = eng.glove_forward(glove_rows, training=True)
            out, pred, _ = eng.head_glove(z, zg, labels, 1, want_grad=True)
            eng.glove_backward()             # first: its gradients sit in the bucket that encoder_backward's event releases
            eng.encoder_backward(x)
        else:
            out, pred, _ = eng.head(z, labels, 1, want_grad=True, gneg=gh)
            eng.encoder_backward(x)
        if use_dist:
            # the 8 MB gradient sum in two buckets: everything but the conv stack's 0.15 MB starts behind an event the
            # backward call records before its conv kernels (~0.5 ms of them); averaged by grad_scale inside Adam
            reduce_grads()
        eng.adam_step(params, grad_scale=1.0 / world)
        state["out"] = out

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i)
    gemm_kinds = ["fc_fwd_ws", "fc_fwd", "fc_dgrad", "fc_dgrad_bn", "fc_dgrad_conv", "fc_dgrad_stats", "fc_wgrad"]
    eng.profile_enable(gemm_kinds, max_records=64 * (args.steps + 1))
    # one event per step boundary on the launch stream: min / median / max step time inside the timed region (the same
    # launch moves by +-10 % with the clock state of the box; the spread says how steady this run was)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    aux_default = eng.aux_stream_enabled
    barrier()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        # the live per-kernel timing brackets each fc GEMM launch with two HIP events, and every event record idles
        # the queue for ~5 us (58 of them per step = 6 % of it): sample every --profile_every-th step
        # The profiled step runs on ONE stream: with the second stream (cp_config.aux_stream) the floating weight gradients share the
        # chip with the critical path's launches, and an event pair around a launch then times the contention too, not the kernel.
        if i % args.profile_every == 0:
            eng.profile_resume()
            eng.aux_stream_enabled = False
        else:
            eng.profile_disable()
            eng.aux_stream_enabled = aux_default
        step(args.warmup + i)
        marks[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    profiled_steps = len(range(0, args.steps, args.profile_every))
    eng.profile_disable()
    eng.aux_stream_enabled = aux_default
    prof = eng.profile_summary()
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss, correct = [float(v) for v in state["out"].tolist()]

    # the other loss workloads on the same engine, shorter: one scaling run's N = 1 and N > 1 lines then hold a common step
    other_steps = {}
    if args.class_encoder == "onehot" and not args.main_only:
        k_other = max(4, args.steps // 2)
        for mode in ("off", "gather", "reduce"):
            if mode == gn or "error" in other_steps:
                continue
            ok = 1
            try:
                other_steps[mode] = _timed_mode(mode, k_other, step, barrier, args, use_dist, dist, dev, world, N)
            except Exception as ex:                           # the side records never cost the line its main measurement
                other_steps["error"] = f"{mode}: {type(ex).__name__}: {ex}"
                ok = 0
            if use_dist:
                # every rank must leave the side records together: a rank that went on alone would wait in the next mode's
                # collectives for ranks that have stopped.  A rank that failed INSIDE a collective cannot be rescued
                # here -- the process group's timeout ends such a run -- but a failure before or after one is agreed on.
                flag = torch.tensor([ok], device=dev, dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    other_steps.setdefault("error", f"{mode}: failed on another rank")
            if "error" in other_steps:
                try:
                    torch.cuda.synchronize(dev)
                except Exception as ex2:
                    print(f"bench.py: device error after other_steps ({ex2}); aborting", file=sys.stderr, flush=True)
                    sys.exit(3)

    if args.breakdown and rank == 0:
        eng.profile_enable(None, max_records=4096)
        for i in range(2):
            step(args.warmup + args.steps + i)
        eng.profile_disable()
        bd = eng.profile_summary()
        tot = sum(v[0] for v in bd.values())
        for k, (ms, n) in sorted(bd.items(), key=lambda kv: -kv[1][0]):
            print(f"  {k:14s} {ms / 2:8.3f} ms/step  ({n // 2} launch groups/step, {100 * ms / tot:5.1f} %)", file=sys.stderr)
        print(f"  sum            {tot / 2:8.3f} ms/step", file=sys.stderr)

    if rank == 0:
        dt = args.dtype
        prefix = {"bf16": "r04", "fp8": "r04_fp8", "f32": "r04_f32"}[dt]          # the newest committed set of this command (else round 3's)
        if not os.path.exists(os.path.join(ROOT, "profiles", prefix + "_traffic.json")):
            prefix = prefix.replace("r04", "r03")
        dom = max(gemm_kinds, key=lambda k: prof.get(k, (0.0, 0))[0])
        ms, launches = prof[dom]
        dyn = "true" if eng.tile_schedule else "false"
        kname = gemm_symbol(dom, dyn, dt)
        # counter evidence from the COMMITTED profiles of this same command on the builder's box (tools/profile_round.sh): not
        # measured in this run, and labelled so
        mfma_busy = traffic = None
        mpath = os.path.join(ROOT, "profiles", prefix + "_mfma.json")
        tpath = os.path.join(ROOT, "profiles", prefix + "_traffic.json")
        if B == 4096 and os.path.exists(mpath):
            for k, v in json.load(open(mpath))["kernels"].items():
                if kname in k:
                    mfma_busy = dict(value=v["mfma_busy_frac"], source="profiles/" + os.path.basename(mpath) + " (SQ counters, builder's box)")
        if B == 4096 and os.path.exists(tpath):
            for k, v in json.load(open(tpath))["kernels"].items():
                if kname in k:
                    traffic = dict(value=v["hbm_bytes_per_launch"], source="profiles/" + os.path.basename(tpath) + " (PMC FETCH_SIZE / WRITE_SIZE passes, builder's box)")
        avg_s = ms / launches / 1e3
        paired = 2                  # (the profiled steps run on one stream: fc7+fc6 and fc5+fc4 share a launch each)
        byts, flops = gemm_model(dom, N, dt, args.dp_emg > 0, paired)
        gbs = byts / avg_s / 1e9
        tfl = flops / avg_s / 1e12
        mfma_peak = MFMA_PEAK_TFLOPS[dt]
        practical = measure_practical_peaks(dev, dt)
        per_kernel = {}
        for k in gemm_kinds:
            if k in prof and prof[k][1] > 0:
                kb, kf = gemm_model(k, N, dt, args.dp_emg > 0, paired)
                ks_ = prof[k][0] / prof[k][1] / 1e3
                per_kernel[k] = dict(symbol=gemm_symbol(k, dyn, dt), launches=prof[k][1], avg_us=ks_ * 1e6, algorithmic_bytes=kb,
                                     gbs=kb / ks_ / 1e9, hbm_frac=kb / ks_ / 1e9 / HBM_PEAK_GBS, tflops=kf / ks_ / 1e12)
        bound = "hbm" if byts / (HBM_PEAK_GBS * 1e9) >= flops / (mfma_peak * 1e12) else "mfma"
        # the same question against what the chip sustains: a launch whose FLOP/byte lies above practical MFMA / practical HBM is
        # bounded by the matrix pipe's power budget, and its HBM fraction cannot reach 1
        bound_practical = "hbm" if byts / (practical["hbm_gbs"] * 1e9) >= flops / (practical["mfma_tflops"] * 1e12) else "mfma"
        frac_practical = max(gbs / practical["hbm_gbs"], tfl / practical["mfma_tflops"])
        roof = dict(bound=bound, kernel=dom, kernel_symbol=kname, launches=launches, avg_us=avg_s * 1e6,
                    achieved=gbs if bound == "hbm" else tfl, peak=HBM_PEAK_GBS if bound == "hbm" else mfma_peak,
                    unit="GB/s" if bound == "hbm" else "TFLOP/s",
                    frac=(gbs / HBM_PEAK_GBS) if bound == "hbm" else (tfl / mfma_peak),
                    traffic=traffic["value"] if traffic else None, traffic_source=traffic["source"] if traffic else None,
                    algorithmic_bytes=byts,
                    practical_peak=practical, bound_practical=bound_practical, frac_practical=frac_practical,
                    mfma_tflops=tfl, mfma_frac=tfl / mfma_peak,
                    mfma_busy_frac=mfma_busy["value"] if mfma_busy else None, mfma_busy_source=mfma_busy["source"] if mfma_busy else None,
                    hbm_gbs=gbs, hbm_frac=gbs / HBM_PEAK_GBS,
                    gemm_ms_per_step={k: prof[k][0] / profiled_steps for k in gemm_kinds if k in prof},
                    second_stream=bool(aux_default and eng._aux is not None),
                    note="per-kernel durations: HIP events around each launch on the profiled steps, which run on ONE stream, as the timed steps do "
                         "by default (second_stream false; with CPNATIVE_AUX_STREAM=1 the timed steps float the weight gradients behind a dropout on "
                         "cp_config.aux_stream and a launch's wall time includes what it shares the chip with)",
                    profiled_steps=profiled_steps,
                    per_kernel=per_kernel)
        # the step as a whole against SURVEY 8d's byte model (51.3 KB per window in 16-bit storage, 102.5 in f32, 25.6 in 8-bit)
        kb_per_window = {"bf16": 51.3, "f32": 102.5, "fp8": 25.6}[dt]
        step_s = elapsed / args.steps
        roof["step"] = dict(model_kb_per_window=kb_per_window, hbm_frac=kb_per_window * 1e3 * N / step_s / (HBM_PEAK_GBS * 1e9),
                            mfma_frac=12.74e6 * N / step_s / (mfma_peak * 1e12))
        rec = dict(metric="sEMG windows/sec contrastive step, 12-ch Ninapro, 1/2/4/8 MI355X", value=world * N * args.steps / elapsed,
                   unit="windows/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=1e3 * elapsed / args.steps, higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype=args.dtype, data="synthetic",
                   config=dict(workload=f"synthetic 12-ch sEMG, 41-class {'glove-angle (20-dim) class encoder' if glove_rows is not None else 'one-hot'}, batch {B} groups/GPU "
                                        f"({N} windows/GPU/step), {'AdaBN' if args.adabn else 'stock BN (--no_adabn)'}, "
                                        f"dp_emg={args.dp_emg}, d_e=16, random-init weights",
                               global_batch_groups=world * B, windows_per_step=world * N,
                               parallelism=f"dp{world}" + ((" + z all-gather" if gn == "gather" else " + {G,H} all-reduce" if gn == "reduce" else "") + " + flat-gradient all-reduce (RCCL)"
                                                           + (" + synchronised BatchNorm" if args.sync_bn else "") if world > 1 else ""),
                               loss={"gather": "global negatives (class->EMG direction over the gathered z)",
                                     "reduce": "global negatives (partial sums + two 64-float all-reduces, no z all-gather)",
                                     "off": "reference per-group loss"}[gn],
                               tile_schedule="dynamic" if eng.tile_schedule else "static"),
                   loss=loss, train_acc=correct / N, roofline=roof,
                   steps_spread=dict(min_ms=step_ms[0], median_ms=step_ms[len(step_ms) // 2], max_ms=step_ms[-1],
                                     note="per-step HIP events on the launch stream, rank 0"))
        if other_steps:
            rec["other_steps"] = other_steps
        if rehearse:
            rec["rehearsal"] = f"ranks share one GPU over {rehearse}: control-flow check only, not a measurement"
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(args.cpu_seconds, threads=min(16, os.cpu_count() or 1))
            rec["gpu_over_cpu"] = rec["value"] / rec["cpu_baseline"]["value"]
        if world == 1 and not rehearse and args.class_encoder == "onehot" and not args.main_only and args.dtype == "bf16":
            try:
                # BASELINE.json configs[4]'s storage on the same workload, same run (its own line: python bench.py --dtype fp8): 8-bit activations,
                # weights and gradients on the block-scaled MFMA.  Parity unpinned by construction; it is not `value`.
                e8 = Engine(adabn=args.adabn, dtype="fp8", dp_emg=args.dp_emg, device=dev, seed=1000)
                e8.init_parameters(seed=42)
                e8.workspace(N)

                def step8(i):
                    x = e8.gather(table, emg_rand, perms[i], 1)
                    z = e8.encoder_forward(x, training=True)
                    o, _, _ = e8.head(z, labels, 1, want_grad=True)
                    e8.encoder_backward(x)
                    e8.adam_step(params)
                    return o
                for i in range(4):
                    step8(i)
                torch.cuda.synchronize(dev)
                k8 = max(5, args.steps // 2)
                t8 = time.perf_counter()
                for i in range(k8):
                    o8 = step8(args.warmup + (i % args.steps))
                torch.cuda.synchronize(dev)
                el8 = time.perf_counter() - t8
                rec["config4_fp8"] = dict(ms_per_step=1e3 * el8 / k8, value=N * k8 / el8, unit="windows/s", steps=k8, loss=float(o8[0]),
                                          note="BASELINE.json configs[4] (0-based) per GPU at this batch: the same workload with 8-bit storage + MX MFMA (CP_FP8), "
                                               "parity unpinned by construction; full line: bench.py --dtype fp8")
                del e8
            except Exception as ex:                       # a side record: never at the cost of the line
                _side_failed("config4_fp8", rec, ex, dev)
        if world == 1 and not rehearse and args.class_encoder == "onehot" and not args.main_only:
            try:
                rec["small_batch"] = small_batch_record(dev, args.dtype)
                if "cpu_baseline" in rec and "b8" in rec["cpu_baseline"]:
                    rec["small_batch"]["b8"]["over_cpu_b8"] = rec["small_batch"]["b8"]["windows_per_s"] / rec["cpu_baseline"]["b8"]["value"]
            except Exception as ex:
                _side_failed("small_batch", rec, ex, dev)
        if world == 1 and not rehearse and args.class_encoder == "onehot" and not args.main_only and args.dtype == "bf16" and B == 4096:
            try:
                rec["config3_glove"] = glove_record(dev, args.dtype, B, table, emg_rand, perms, labels, max(5, args.steps // 2))
                rec["config3_glove"]["over_onehot_ms"] = rec["config3_glove"]["ms_per_step"] - rec["ms_per_step"]
            except Exception as ex:
                _side_failed("config3_glove", rec, ex, dev)
        if world == 1 and not rehearse and not args.main_only:
            try:
                rec["eval"] = eval_record(dev, ["f32", "bf16", "fp8"] if args.dtype == "bf16" else [args.dtype])
            except Exception as ex:
                _side_failed("eval", rec, ex, dev)
        print(json.dumps(rec), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
