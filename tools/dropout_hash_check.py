"""Statistical and structural check of the dropout hash of csrc/common.cuh (dropout_quad), emulated in numpy bit for bit.

    python tools/dropout_hash_check.py            # everything, at the bench's tensor sizes (167,936 x 512 and 335,872 x 512)

Round 4: the round-3 form fed a 32-bit quad index through `x ^= x >> 15` into a 24-bit multiply, i.e. kept 24 bits of
state per key: tensors of more than 2^24 quads (131,072 rows x 512) re-used whole rows of masks (row r and row (r ^ 4) + 131,072 in
all 512 columns).  `quad_r3` below is that form, kept so that the structural test can be seen to catch it; `quad` is the shipped one:
the index's top byte enters through its own multiply, and the two output words come from different 24-bit windows of the state, so all
32 index bits stay live.  Checks: (1) no two ROWS of a tensor share a mask (exhaustive over the real sizes); (2) drop rates of the four
draws; (3) correlations between draws, neighbours, rows, the 2^22 / 2^24 / 2^25-quad lags and one-bit key changes.
tests/test_cabi_and_host.py runs `structural()` and `statistics()` at the bench size.
"""
import sys

import numpy as np

M32 = np.uint64(0xFFFFFFFF)
U = np.uint64


def mul24(a, c):
    return ((a & U(0xFFFFFF)) * U(c)) & M32


def rot16(k):
    return ((k >> U(16)) | (k << U(16))) & M32


def quad_r3(idx, key):
    """round 3's dropout_quad (24 bits of state: superseded)"""
    x = (idx ^ key) & M32
    x ^= x >> U(15)
    x = mul24(x, 0xB5297B)
    x ^= rot16(key)
    x ^= x >> U(13)
    a, b = mul24(x, 0x8DA6B5), mul24(x, 0x3C6EF3)
    return a ^ (a >> U(16)), b ^ (b >> U(16))


def quad(idx, key):
    """csrc/common.cuh dropout_quad: idx = (row * ld + col) >> 2 < 2^30"""
    x = (idx ^ key) & M32
    hi = x >> U(24)
    x ^= x >> U(15)
    x = mul24(x, 0xB5297B) ^ mul24(hi, 0x9E3779)
    x ^= rot16(key)
    x ^= x >> U(13)
    a, b = mul24(x, 0x8DA6B5), mul24(x ^ (x >> U(11)), 0x3C6EF3)
    return a ^ (a >> U(16)), b ^ (b >> U(16))


def draws(a, b):
    return [(a & U(0xFFFF)).astype(np.int64), (a >> U(16)).astype(np.int64), (b & U(0xFFFF)).astype(np.int64), (b >> U(16)).astype(np.int64)]


def structural(h, rows, ld=512, key=0x9d2c5681, p=0.0635, chunk_rows=1 << 15):
    """Rows of a [rows x ld] tensor whose whole drop mask equals another row's (must be 0), and quads whose four draws equal another
    quad's (birthday collisions of a 32-bit state: ~ n / 2^33 of n quads)."""
    t = int(p * 65536 + 0.5)
    qpr = ld // 4
    sig = np.empty(rows, dtype=np.uint64)
    words = []
    w = np.random.default_rng(5).integers(1, 1 << 63, size=qpr, dtype=np.uint64) | U(1)         # random per-column weights of the row signature
    for r0 in range(0, rows, chunk_rows):
        r1 = min(rows, r0 + chunk_rows)
        idx = (np.arange(r0, r1, dtype=np.uint64)[:, None] * U(qpr) + np.arange(qpr, dtype=np.uint64)[None, :])
        a, b = h(idx, U(key))
        d = draws(a, b)
        bits = sum(((d[i] < t).astype(np.uint64) << U(i)) for i in range(4))              # the 4 drop bits of each quad
        with np.errstate(over="ignore"):
            sig[r0:r1] = ((bits + U(1)) * w[None, :]).sum(axis=1, dtype=np.uint64)
        words.append(((a << U(32)) | b).reshape(-1))
    dup_rows = rows - np.unique(sig).size
    allw = np.concatenate(words)
    dup_quads = allw.size - np.unique(allw).size
    return dict(rows=rows, quads=int(allw.size), duplicate_row_masks=int(dup_rows), duplicate_quads=int(dup_quads),
                duplicate_quad_frac=dup_quads / allw.size)


def corr(a, b):
    return float(np.corrcoef(a, b)[0, 1])


def statistics(h, n=1 << 22, base=777, key=0x9d2c5681, p=0.0635, lags=(1, 128, 1 << 22, 1 << 24, 1 << 25)):
    t = int(p * 65536 + 0.5)
    idx = np.arange(n, dtype=np.uint64) + U(base)
    d = [(x < t).astype(np.float64) for x in draws(*h(idx, U(key)))]
    res = dict(rates=[round(float(x.mean()), 5) for x in d])
    res["cross_draw_worst"] = max(abs(corr(d[i], d[j])) for i in range(4) for j in range(i + 1, 4))
    worst = {}
    for lag in lags:
        e = [(x < t).astype(np.float64) for x in draws(*h(idx + U(lag), U(key)))]
        worst[lag] = max(abs(corr(d[i], e[j])) for i in range(4) for j in range(4))
        res.setdefault("equal_mask_frac", {})[lag] = float(np.mean(np.all([d[i] == e[i] for i in range(4)], axis=0)))
    res["lag_worst"] = worst
    kb = 0.0
    for bit in range(32):
        e = [(x < t).astype(np.float64) for x in draws(*h(idx, U(key ^ (1 << bit))))]
        kb = max(kb, max(abs(corr(d[i], e[i])) for i in range(4)))
    res["key_bit_worst"] = kb
    v = draws(*h(idx, U(key)))
    res["chi2_256"] = [round(float(((np.bincount(x >> 8, minlength=256) - n / 256) ** 2 / (n / 256)).sum()), 1) for x in v]
    res["rates_by_p"] = {pp: [round(float((x < int(pp * 65536 + 0.5)).mean()), 5) for x in v] for pp in (0.01, 0.1, 0.3817, 0.5)}
    return res


if __name__ == "__main__":
    for name, h in (("round 3 (superseded)", quad_r3), ("shipped", quad)):
        print("====", name)
        for rows in (167936, 335872):
            print(" structural", structural(h, rows))
        print(" statistics", statistics(h))
    # expected fraction of quads whose four drop bits all equal those of an unrelated quad at p = 0.0635: (1 - 2 p (1 - p))^4
    print("independent-mask equality rate at p = 0.0635:", round((1 - 2 * 0.0635 * (1 - 0.0635)) ** 4, 4))
    sys.exit(0)
