"""oracle/eval_cpu.py against the reference: its own Model code (golden fixture) and its published
output files (tests/golden/reference_results/*.npy, copied data from the reference's data/ directory)."""
import os

import numpy as np

from oracle import eval_cpu as ev

T = 41


def test_reference_output_files_relations(golden_dir):
    d = os.path.join(golden_dir, "reference_results")
    y_pred, y_true = np.load(os.path.join(d, "y_pred.npy")), np.load(os.path.join(d, "y_true.npy"))
    cm, voting = np.load(os.path.join(d, "confusion_matrix.npy")), np.load(os.path.join(d, "voting.npy"))
    B = y_pred.size // T
    assert (B, voting.shape) == (48, (48, 24))
    assert np.array_equal(y_true, np.tile(np.arange(T), B))
    counts = ev.confusion_counts(y_true, y_pred)
    assert counts.sum() == y_pred.size and np.array_equal(counts.sum(1), np.full(T, B))
    np.testing.assert_allclose(counts / B, cm, atol=1e-12)            # the file holds the row-normalised matrix
    per_group = (y_pred == y_true).reshape(B, T).mean(1)
    np.testing.assert_allclose(voting[:, -1], per_group, atol=1e-12)  # last vote column == accuracy of y_pred


def test_vote_matches_reference_model_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "eval_vote_B2_adabn.npz"))
    logits, labels = g["eval_logits"], np.arange(T)
    B, V = 2, 25
    assert logits.shape == (B * V, T, T)
    correct, y_pred = ev.subset_vote(logits, labels, B, V, np.ones(T, dtype=np.uint8))
    assert np.array_equal(y_pred, g["y_pred"])
    # the fixture's vote is per group (B, 24): prefixes 1..24; compare the group-summed curve
    np.testing.assert_allclose(correct[:24] / T, g["vote"].sum(0), atol=1e-9)
    curve = ev.voting_curve(correct, B, T)
    assert curve.shape == (249,) and np.all(curve[24:] == curve[24])
    assert abs(curve[-1] - float(g["acc"])) < 1e-7                     # the fixture holds the float32 of Model.correct()


def test_subset_semantics():
    rng = np.random.default_rng(0)
    logits = rng.standard_normal((6, T, T)).astype(np.float32)
    mask = np.zeros(T, dtype=np.uint8)
    mask[[3, 7, 20]] = 1
    p = ev.subset_predict(logits, mask)
    assert set(np.unique(p[:, [3, 7, 20]])) <= {3, 7, 20}
    assert np.all(p[:, np.flatnonzero(mask == 0)] == -1)
    assert np.array_equal(ev.subset_predict(logits, np.ones(T))[:, :], logits.argmax(-1))
    assert np.array_equal(ev.prefix_mode(np.array([5, 2, 2, 5, 7])), [5, 2, 2, 2, 2])   # ties -> smallest id
    m = ev.random_subsets(range(2, 5), 3, 1)
    assert m.shape == (9, T) and list(m.sum(1)) == [2, 2, 2, 3, 3, 3, 4, 4, 4]
