"""SURVEY.md 8f row f4 on the GPU: the packed random search (code/train.py:140-166,175-194 spread over ranks)
gives, for every configuration, the same numbers whichever rank trains it -- 2 ranks sharing the MI355X against
1 rank, compared bit for bit -- and leaves the files the reference's search leaves."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(nproc, out, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "contrastiveprosthetics_amd.train",
           "--hpo_pack", "--crossval_size", "3", "--crossval_epochs", "1", "--final_epochs", "1", "--batch_size", "64",
           "--synthetic", "--no_adabn", "--data_dir", str(out), "--checkpoint_dir", str(out)]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    return p.stdout


@pytest.mark.timeout(900)
def test_hpo_pack_two_ranks_equal_one_rank(tmp_path):
    a, b = tmp_path / "one", tmp_path / "two"
    out1 = run(1, a, 29631)
    out2 = run(2, b, 29632)
    v1, k1 = np.load(a / "cross_val_values.npy"), np.load(a / "cross_val_keys.npy")
    v2, k2 = np.load(b / "cross_val_values.npy"), np.load(b / "cross_val_keys.npy")
    assert v1.shape == (3, 2) and k1.shape == (3, 7)              # (loss, acc) per configuration; 7-tuple keys
    assert np.array_equal(k1, k2)
    assert np.array_equal(v1, v2), (v1, v2)                       # same numbers whichever rank trained them
    assert np.all(np.isfinite(v1)) and np.all(v1[:, 1] >= 0) and np.all(v1[:, 1] <= 1)
    assert "Best combination" in out1 and "Best combination" in out2
    assert (a / "contrastive.pt").exists() and (b / "contrastive.pt").exists()
