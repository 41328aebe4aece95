#!/usr/bin/env python
"""How does the device assignment (bts_lapjv) behave on cost matrices WITH exact ties, next to the dense JV restatement
of lap.lapjv (oracle/lapjv_oracle.c)?  Prints, per family of tie cases, how many cases give identical (x, y) and how many
give the same objective.  (Ties are "parity unpinned" in the reference: lap's source is absent; this measures the gap.)"""
import importlib, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
pkg = importlib.import_module("bot-sort-onnx-tensorrt_amd")
from oracle import bts_oracle as bo


def objective(c, x, y, limit):
    rows = np.nonzero(x >= 0)[0]
    return float(c[rows, x[rows]].sum()) + limit / 2 * ((x < 0).sum() + (y < 0).sum())


def run(name, cases, limit):
    same = same_obj = 0
    first = None
    for c in cases:
        c = np.ascontiguousarray(c, dtype=np.float64)
        _, xd, yd = pkg.lapjv(c, extend_cost=True, cost_limit=limit)
        _, xo, yo = bo.lapjv(c, extend_cost=True, cost_limit=limit)
        xo, yo = np.asarray(xo), np.asarray(yo)
        eq = np.array_equal(xd, xo) and np.array_equal(yd, yo)
        same += eq
        same_obj += abs(objective(c, xd, yd, limit) - objective(c, xo, yo, limit)) < 1e-9
        if not eq and first is None:
            first = (c, xd, xo)
    print("%-34s cases %4d  identical %4d  same objective %4d" % (name, len(cases), same, same_obj))
    if first is not None and first[0].size <= 36:
        print("   first difference: cost=\n%s\n   device x=%s  dense-JV x=%s" % (first[0], first[1], first[2]))


def main():
    rng = np.random.default_rng(0)
    L = 0.8
    run("1 row x 2 equal columns", [np.full((1, 2), v) for v in (0.1, 0.3, 0.5, 0.79)], L)
    run("2 rows x 1 equal column", [np.full((2, 1), v) for v in (0.1, 0.3, 0.5, 0.79)], L)
    run("k x k all-equal blocks", [np.full((k, k), 0.25) for k in range(2, 10)], L)
    run("n x m all-equal blocks", [np.full((n, m), 0.25) for n in range(1, 6) for m in range(1, 6) if n != m], L)
    run("cost exactly at the limit", [np.full((1, 1), L), np.array([[L, 0.2], [0.3, L]]), np.array([[L, L], [L, L]])], L)
    run("cost exactly at half the limit", [np.full((1, 1), L / 2), np.array([[L / 2, L / 2]]), np.array([[L / 2], [L / 2]])], L)
    cases = []
    for _ in range(200):
        n, m = rng.integers(2, 9, 2)
        cases.append(rng.integers(1, 5, (n, m)) / 5.0)       # values 0.2 .. 0.8: many ties
    run("small integer-grid costs", cases, L)
    cases = []
    for _ in range(100):
        n, m = rng.integers(3, 12, 2)
        c = rng.uniform(0.05, 0.75, (n, m))
        c[rng.random((n, m)) < 0.5] = 1.0                    # infeasible entries
        j = rng.integers(0, m)
        c = np.concatenate([c, c[:, j:j + 1]], axis=1)       # a duplicated detection column
        cases.append(c)
    run("random + one duplicated column", cases, L)
    cases = []
    for _ in range(100):
        n, m = rng.integers(3, 12, 2)
        c = rng.uniform(0.05, 0.75, (n, m))
        c[rng.random((n, m)) < 0.5] = 1.0
        i = rng.integers(0, n)
        c = np.concatenate([c, c[i:i + 1]], axis=0)          # a duplicated track row
        cases.append(c)
    run("random + one duplicated row", cases, L)


if __name__ == "__main__":
    main()
