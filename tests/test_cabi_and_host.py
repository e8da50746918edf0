"""CPU-only checks: the C-ABI library loads and exports every symbol include/cpnative.h declares
(no compute call is made -- there is no GPU here), and the host-side logic (constants, parameter
table, CLI surface, sharding) matches the oracle / the reference's published layout."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import ref_cpu as oc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cpnative.h")
LIB = os.path.join(ROOT, "contrastiveprosthetics_amd", "libcpnative.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        subprocess.run(["make", "-C", os.path.join(ROOT, "contrastiveprosthetics_amd", "csrc")], check=True)
    from contrastiveprosthetics_amd import _lib
    return _lib.load()


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cp_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound(lib):
    from contrastiveprosthetics_amd import _lib
    names = declared_functions()
    assert len(names) >= 15
    raw = ctypes.CDLL(LIB)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in cpnative.h but not exported"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes prototype in _lib.SYMBOLS"
    assert set(_lib.SYMBOLS) == set(names)
    assert lib.cp_version() == 110


def test_struct_layouts_match_header():
    from contrastiveprosthetics_amd import _lib
    assert ctypes.sizeof(_lib.cp_params) == 8 * (4 + 7 + 7 + 9 + 9 + 3)
    assert ctypes.sizeof(_lib.cp_bn_buffers) == 8 * 18
    # 56 bytes of round 1-3 fields + options, tile_schedule, the sync-BN hook (fn, user, world, pad) and the gradient tap (ptr, bytes)
    assert ctypes.sizeof(_lib.cp_config) == (8 + 4 * 4 + 4 * 4 + 8 + 8) + 4 + 4 + 8 + 8 + 4 + 4 + 8 + 8 + 3 * 8
    hdr = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    body = hdr[hdr.index("typedef struct cp_config {"):hdr.index("} cp_config;")]
    fields = re.findall(r"\b(\w+)\s*;", body)
    assert fields == [f[0] for f in _lib.cp_config._fields_], fields
    for name, bit in _lib.OPTIONS.items():
        assert re.search(r"#define CP_OPT_%s %du\b" % (name.upper(), bit), open(HEADER).read()), name
    assert ctypes.sizeof(_lib.cp_adam_hyper) == 32


def test_host_only_entry_points(lib):
    # pure host functions: workspace sizing and argument validation (return before touching a device)
    small = lib.cp_workspace_bytes(41 * 8, 0, 0.0)
    big = lib.cp_workspace_bytes(41 * 4096, 1, 0.0635)
    assert 0 < small < big < 8 * 2 ** 30
    assert lib.cp_workspace_bytes(0, 0, 0.0) == 0
    assert lib.cp_workspace_bytes(41 * 4096, 1, 0.0635) > lib.cp_workspace_bytes(41 * 4096, 1, 0.0)
    rc = lib.cp_gather_groups(None, 0, None, 0, None, 0, 1, None, None)
    assert rc == 10001 and b"cp_gather_groups" in lib.cp_last_error()
    numel = (ctypes.c_int64 * 3)(5000, 64, 1)
    assert lib.cp_optimizer_scratch_floats(numel, 3) >= 3 + 1 + 1


def test_missing_library_is_an_error(monkeypatch):
    from contrastiveprosthetics_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libcpnative.so")
    with pytest.raises(_lib.CpNativeError):
        _lib.load()


def test_engine_refuses_cpu():
    from contrastiveprosthetics_amd import _lib
    from contrastiveprosthetics_amd.engine import Engine
    with pytest.raises(_lib.CpNativeError):
        Engine(adabn=True, dtype="f32", device="cpu")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "contrastiveprosthetics_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), f"{f} mentions the oracle"


def test_constants_match_oracle_and_golden(golden_dir):
    from contrastiveprosthetics_amd import constants as c
    o = oc.split_constants()
    assert np.array_equal(c.d2_idxs, o["d2_idxs"]) and np.array_equal(c.d3_idxs, o["d3_idxs"])
    assert np.array_equal(c.TASKS, o["tasks"])
    g = np.load(os.path.join(golden_dir, "db23_sampler.npz"))
    assert np.array_equal(np.concatenate((c.TASKS, [0])), g["tasks_mask"])        # reference values
    assert np.array_equal(c.d3_idxs + 40, g["people_mask"])
    assert (c.MAX_TASKS, c.EMG_DIM, c.GLOVE_DIM, c.PREDICTION_WINDOW_SIZE, c.AMT_PREDICTION_WINDOWS) == (41, 12, 20, 25, 4)


@pytest.mark.parametrize("adabn", [False, True])
def test_param_table_matches_reference_state_dict(adabn):
    from contrastiveprosthetics_amd.engine import l2_member, param_specs
    specs = param_specs(adabn)
    ref = oc.init_state_dict(0, 16, adabn)
    ref_params = [k for k, v in ref.items() if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var"))
                  and k != "logit_scale"]
    assert list(specs) == ref_params
    for k in specs:
        assert tuple(ref[k].shape) == tuple(specs[k])
    m = oc.OracleModel(ref, dict(reg_emg=1, reg_glove=1), adabn=adabn)
    emg, glove = m.l2_keys()
    assert sorted(k for k in specs if l2_member(k)) == sorted(emg + glove)
    assert sum(int(np.prod(s)) for s in specs.values()) == 2027616


def test_cli_surface_matches_reference():
    from contrastiveprosthetics_amd.train import build_parser
    a = build_parser().parse_args([])
    assert (a.crossval_size, a.crossval_epochs, a.batch_size, a.final_epochs) == (10, 1, 32, 10)
    assert (a.glove, a.db2, a.load_model, a.crossval_load, a.prediction, a.test) == (False,) * 6
    assert (a.no_adabn, a.no_checkpoint, a.no_verbose) == (True, True, True)          # store_false polarity
    b = build_parser().parse_args(["--no_adabn", "--no_checkpoint", "--no_verbose", "--test", "--batch_size=8"])
    assert (b.no_adabn, b.no_checkpoint, b.no_verbose, b.test, b.batch_size) == (False, False, False, True, 8)


def test_shard_range_partitions():
    from contrastiveprosthetics_amd.dist import shard_range
    for n in (0, 1, 7, 8, 4096, 1801):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("final_epochs", [8, 12])
def test_lr_closed_forms_equal_the_reference_schedulers(final_epochs):
    """train.lr_scales against torch.optim.lr_scheduler objects built exactly as code/train.py:75-80 builds them and
    stepped once per epoch as at :112-113: CosineAnnealingLR(T_max=final_epochs, eta_min=0) on both optimisers when
    annealing, else two StepLR(step_size=5, gamma=.2) that BOTH wrap optimizer_glove (the reference's quirk: the emg
    rate never moves, the glove rate falls by .2**2 every 5 epochs)."""
    import torch.optim as optim
    from contrastiveprosthetics_amd.train import lr_scales
    lr_e, lr_g = 9.761e-4, 2.653e-3
    for annealing in (True, False):
        pe, pg = torch.nn.Parameter(torch.zeros(1)), torch.nn.Parameter(torch.zeros(1))
        oe = optim.Adam([pe], lr=lr_e, weight_decay=0)
        og = optim.Adam([pg], lr=lr_g, weight_decay=0)
        if annealing:
            se = optim.lr_scheduler.CosineAnnealingLR(oe, T_max=final_epochs, eta_min=0)
            sg = optim.lr_scheduler.CosineAnnealingLR(og, T_max=final_epochs, eta_min=0)
        else:
            se = optim.lr_scheduler.StepLR(og, step_size=5, gamma=.2)
            sg = optim.lr_scheduler.StepLR(og, step_size=5, gamma=.2)
        for epoch in range(12):
            s = lr_scales(epoch, annealing, final_epochs)
            assert oe.param_groups[0]["lr"] == pytest.approx(lr_e * s[0], rel=1e-6, abs=1e-12), (annealing, epoch)
            assert og.param_groups[0]["lr"] == pytest.approx(lr_g * s[1], rel=1e-6, abs=1e-12), (annealing, epoch)
            oe.step(); og.step()
            se.step(); sg.step()


def test_register_stationary_gemm_kernel_has_no_spills():
    """The weight-stationary GEMM kernels (csrc/gemm_ws.cuh: gemm_ws16_kernel, gemm_ws16n_kernel, gemm_wsd16_kernel<0|1>; csrc/fp8.cuh:
    gemm_ws8_kernel<512|768>, gemm_wsd8_kernel<0|1>) keep 128-256 weight registers per wave and run at the edge of the 512-register file.  A spill costs them their
    speed (scratch traffic inside the k loop), and in round 2 every build that spilled also returned wrong values -- that turned out to be
    the store-data hazard that csrc/gemm_ws.cuh::store_b128_settled now closes, not the spills themselves; the guard stays for the speed:
    hipcc's resource remarks are checked at build time."""
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "kernel_resources.sh"), "gemm_wsd?(16n?|8)?_kernel"], capture_output=True, text=True,
                         check=True).stdout
    lines = [l for l in out.splitlines() if "gemm_ws" in l]
    assert len(lines) >= 8, out          # (ws16, ws16n, wsd16<0|1>, ws8<512|768>, wsd8<0|1>)
    for l in lines:
        assert re.search(r"spill\s+0\s+scratch\s+0\b", l), l


def test_no_buffer_store_is_followed_by_a_write_of_its_data_registers():
    """gfx950 store-data hazard (csrc/gemm_ws.cuh, store_b128_settled; DESIGN.md section 4, "The lanes 12-15 fault"): hipcc places a
    VALU write of a buffer_store_dwordx4's data registers directly behind the store when the store's soffset is an SGPR, and the
    store then sends the new values for some lanes.  tools/store_hazard_scan.py reads the device assembly of the whole library: no
    96/128-bit buffer store may have such a write within the next two instructions."""
    for extra in ([], ["--variants"]):       # the product library, and the tools-only build whose timings DESIGN.md quotes
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "store_hazard_scan.py")] + extra, capture_output=True, text=True)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        assert out.stdout.strip().startswith("0 unprotected"), out.stdout[-500:]


def test_pinned_mfmas_keep_their_wait_states():
    """The 8-bit weight-stationary kernels (csrc/fp8.cuh: gemm_ws8_kernel, gemm_wsd8_kernel) issue their MFMAs as volatile asm statements so
    that the previous tile's epilogue can be paced between them; hipcc's hazard tables do not see inside such a statement.
    tools/pinned_mfma_audit.py reads the device assembly: no vector write of an operand within two states in front of a pinned MFMA, no
    vector write of a C operand that is not the destination within seven states behind it, no scratch in those kernels."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pinned_mfma_audit.py")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    first = out.stdout.strip().split("\n")[0]
    assert first.endswith("0 finding(s)") and int(first.split()[0]) >= 6, out.stdout[-500:]


def test_no_getenv_on_a_launch_path():
    """the product library never reads the environment (a call's settings travel in its cp_config); the tools-only build reads it
    once, when it is loaded, to seed the variant switches (csrc/api.hip, seed_variants_from_env, under #ifdef CP_VARIANTS)"""
    import glob
    hits = []
    for f in glob.glob(os.path.join(ROOT, "contrastiveprosthetics_amd", "csrc", "*")):
        for i, line in enumerate(open(f, errors="replace")):
            if "getenv(" in line and not line.lstrip().startswith("//"):
                hits.append((os.path.basename(f), i + 1, line.strip()))
    assert len(hits) == 1 and hits[0][0] == "api.hip" and "getenv(env)" in hits[0][2], hits
    src = open(os.path.join(ROOT, "contrastiveprosthetics_amd", "csrc", "api.hip")).read()
    block = src[src.index("static int seed_variants_from_env()"):]
    assert src[:src.index("static int seed_variants_from_env()")].rstrip().split("#")[-1].startswith("ifdef CP_VARIANTS") or \
        "#ifdef CP_VARIANTS" in src[src.index("extern \"C\" int cp_has_variants"):src.index("static int seed_variants_from_env()")]
    # and nothing process-wide is left for the training path to consult
    assert not re.search(r"^static [^(]*\bg_(opt|sync_fn|sync_user|sync_world|grad_tap|tile_schedule)\b", src, flags=re.M)
    out = subprocess.run(["nm", "-D", "--defined-only", LIB], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in subprocess.run(["nm", "-D", "--undefined-only", LIB], capture_output=True, text=True, check=True).stdout


def test_dropout_hash_keeps_every_index_bit_live():
    """the round-3 hash kept 24 bits of state, so a 167,936 x 512 tensor re-used whole rows of masks.  The numpy emulation
    of csrc/common.cuh::dropout_quad (tools/dropout_hash_check.py) at the bench's tensor size: no two rows share a mask, the drop
    rate and the correlations (neighbours, rows, the 2^22 / 2^24-quad lags, key bits) are those of independent draws; the emulation of
    the superseded form must FAIL the same structural test (so the test is known to see the defect)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dropout_hash_check as dh
    cu = open(os.path.join(ROOT, "contrastiveprosthetics_amd", "csrc", "common.cuh")).read()
    for const in ("0xB5297Bu", "0x9E3779u", "0x8DA6B5u", "0x3C6EF3u", "x >> 24", "x >> 15", "x >> 13", "x >> 11"):
        assert const in cu[cu.index("uint2 dropout_quad("):cu.index("dropout_pair(")], const      # the emulation follows the kernel source
    s = dh.structural(dh.quad, 167936)
    assert s["duplicate_row_masks"] == 0 and s["duplicate_quad_frac"] < 0.01, s
    assert dh.structural(dh.quad_r3, 167936)["duplicate_row_masks"] > 30000
    st = dh.statistics(dh.quad, n=1 << 20)
    assert all(abs(r - 0.0635) < 1.5e-3 for r in st["rates"]), st
    assert st["cross_draw_worst"] < 6e-3 and st["key_bit_worst"] < 8e-3 and max(st["lag_worst"].values()) < 6e-3, st
