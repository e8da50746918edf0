"""CPU oracle for the evaluation post-processing -- TEST INFRASTRUCTURE ONLY.

numpy restatement of what the reference does with eval logits: per-sample argmax, the prefix majority
vote and its accuracy curve (code/models.py:146-163), the saved predictions and the confusion matrix
(code/results.py:24-64), plus the class-subset prediction the README describes (README.md:11-19: "the
user can choose which classes to classify", i.e. rows and columns of the 41 x 41 logits restricted to
the chosen set).  Only tests/ may import it; the product package never does.

Parity status:
  * vote curve / y_pred / accuracy with ALL 41 classes: PINNED -- tests/golden/eval_vote_B2_adabn.npz was
    produced by running the reference's own Model code (tools/make_golden.py), and the reference's
    published output files data/{y_pred,y_true,voting,confusion_matrix}.npy (copied as data fixtures to
    tests/golden/reference_results/) pin the relations voting[:, -1] == per-group accuracy of y_pred and
    confusion_matrix == row-normalised counts of (y_true, y_pred);
  * the subset restriction itself: "parity unpinned" -- the reference holds no code for it (the curve in
    its README was computed offline from logs.npy); the restatement follows the README's description
    and is anchored on the all-classes case above.
"""
from __future__ import annotations

import numpy as np

N_TASKS = 41


def subset_predict(logits: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """logits (G,41,41) -> pred (G,41) int64; argmax over the columns inside `mask` (first maximum wins, as
    torch.argmax after the monotone softmax of code/models.py:147); rows outside the subset get -1."""
    mask = np.asarray(mask, dtype=bool)
    cols = np.flatnonzero(mask)
    sub = logits[:, :, cols]
    pred = cols[np.argmax(sub, axis=-1)]
    pred = np.where(mask[None, :], pred, -1)
    return pred.astype(np.int64)


def prefix_mode(pred: np.ndarray) -> np.ndarray:
    """pred (V,) of class ids -> (V,) mode of pred[:w+1]; ties -> smallest class id (torch.mode,
    code/models.py:154)."""
    out = np.empty(len(pred), dtype=np.int64)
    cnt = np.zeros(N_TASKS, dtype=np.int64)
    for w, p in enumerate(pred):
        cnt[p] += 1
        out[w] = int(np.argmax(cnt))            # first maximum = smallest id among the most frequent
    return out


def subset_vote(logits: np.ndarray, labels: np.ndarray, B: int, V: int, mask: np.ndarray):
    """The eval branch of contrastive_loopy_loss (code/models.py:138-163) restricted to a class subset.
    logits (B*V,41,41) with group g = b*V + v; labels (41,) = labels[:tasks].
    Returns correct (V,) int64 = number of (b, t in subset) whose mode over the first w+1 samples equals
    labels[t], and y_pred (B,41) int64 = mode over all V samples (-1 outside the subset)."""
    mask = np.asarray(mask, dtype=bool)
    pred = subset_predict(logits, mask).reshape(B, V, N_TASKS)
    correct = np.zeros(V, dtype=np.int64)
    y_pred = np.full((B, N_TASKS), -1, dtype=np.int64)
    for b in range(B):
        for t in np.flatnonzero(mask):
            m = prefix_mode(pred[b, :, t])
            correct += (m == labels[t]).astype(np.int64)
            y_pred[b, t] = m[-1]
    return correct, y_pred


def voting_curve(correct: np.ndarray, B: int, k: int, prediction_window: int = 250) -> np.ndarray:
    """Accuracy per window length as Model.voting_raw lays it out (code/models.py:151-156: win = 1 ..
    PREDICTION_WINDOW-1, a slice longer than V is the whole group): (prediction_window-1,) floats."""
    V = len(correct)
    acc = correct.astype(np.float64) / (B * k)
    idx = np.minimum(np.arange(1, prediction_window), V) - 1
    return acc[idx]


def confusion_counts(y_true: np.ndarray, y_pred: np.ndarray) -> np.ndarray:
    """sklearn.metrics.confusion_matrix(y_true, y_pred) over the 41 classes (code/results.py:58), as counts;
    entries with y_pred < 0 (rows outside a subset) are skipped."""
    c = np.zeros((N_TASKS, N_TASKS), dtype=np.int64)
    ok = y_pred >= 0
    np.add.at(c, (y_true[ok], y_pred[ok]), 1)
    return c


def random_subsets(sizes, trials: int, seed: int) -> np.ndarray:
    """(len(sizes)*trials, 41) uint8 masks: `trials` uniformly random subsets of every size (the README's
    "average accuracy per prediction set size on 144 trials")."""
    rng = np.random.default_rng(seed)
    out = []
    for k in sizes:
        for _ in range(trials):
            m = np.zeros(N_TASKS, dtype=np.uint8)
            m[rng.choice(N_TASKS, size=k, replace=False)] = 1
            out.append(m)
    return np.stack(out)
