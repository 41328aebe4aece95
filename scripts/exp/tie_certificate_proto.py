"""Prototype (CPU, NumPy) of the uniqueness certificate for lap.lapjv(extend_cost=True, cost_limit=L) problems:
given an optimal visible assignment x, y on the sparse feasible edges, decide whether ANY other optimal (or within tol)
visible assignment exists.  unique => every exact solver, the dense JV included, returns this x."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import bts_oracle as bo
from scipy.optimize import linear_sum_assignment

TOL = 1e-9


def certificate(c, L, x, y):
    """True = unique (no alternating cycle / Z-path in the near-tight graph)."""
    n, m = c.shape
    rows, cols = np.nonzero(c <= L)
    cost = c[rows, cols]
    E = len(rows)
    INF = 1e300
    p = np.where(y >= 0, INF, 0.0)
    cm = np.zeros(n)
    for e in range(E):
        i, j = rows[e], cols[e]
        if x[i] == j:
            cm[i] = cost[e]; p[j] = L - cost[e]
    for rnd in range(n + m + 2):
        changed = False
        for e in range(E):
            i, j = rows[e], cols[e]
            if x[i] >= 0 and x[i] != j:
                cand = (cost[e] - cm[i]) + p[j]
                if cand < p[x[i]]:
                    p[x[i]] = cand; changed = True
        if not changed:
            break
    else:
        return False
    u = np.where(x >= 0, cm + p[np.maximum(x, 0)], L)
    # arcs among matched rows + Z (index n)
    Z = n
    arcs = []
    for e in range(E):
        i, j = rows[e], cols[e]
        if x[i] == j:
            continue
        slack = cost[e] + p[j] - u[i]
        if slack > TOL:
            continue
        src = i if x[i] >= 0 else Z
        dst = y[j] if y[j] >= 0 else Z
        if src == Z and dst == Z:
            return False
        arcs.append((src, dst))
    for i in range(n):
        if x[i] >= 0:
            if u[i] >= L - TOL:
                arcs.append((i, Z))
            if p[x[i]] <= TOL:
                arcs.append((Z, i))
    alive = np.zeros(n + 1, bool)
    alive[:n] = x >= 0
    alive[Z] = True
    while True:
        has = np.zeros(n + 1, bool)
        for s, d in arcs:
            if alive[s] and alive[d]:
                has[s] = True
        new = alive & has
        if (new == alive).all():
            break
        alive = new
    return not alive.any()


def scipy_visible(c, L):
    n, m = c.shape
    ext = np.full((n + m, n + m), L / 2.0)
    ext[n:, m:] = 0
    ext[:n, :m] = c
    r, q = linear_sum_assignment(ext)
    x = np.full(n, -1); y = np.full(m, -1)
    for i, j in zip(r, q):
        if i < n and j < m:
            x[i] = j; y[j] = i
    return x, y


def families(rng, count):
    L = 0.8
    for _ in range(count):
        kind = rng.integers(0, 6)
        n, m = rng.integers(1, 10, 2)
        if kind == 0:
            c = rng.integers(1, 6, (n, m)) / 5.0
        elif kind == 1:
            c = rng.uniform(0.05, 0.95, (n, m)); c[rng.random((n, m)) < 0.5] = 1.0
        elif kind == 2:
            c = rng.uniform(0.05, 0.75, (n, m)); c[rng.random((n, m)) < 0.5] = 1.0
            j = rng.integers(0, m); c = np.concatenate([c, c[:, j:j + 1]], 1)
        elif kind == 3:
            c = rng.uniform(0.05, 0.75, (n, m)); c[rng.random((n, m)) < 0.5] = 1.0
            i = rng.integers(0, n); c = np.concatenate([c, c[i:i + 1]], 0)
        elif kind == 4:
            c = np.full((n, m), rng.choice([0.25, 0.4, 0.8])); c[rng.random((n, m)) < 0.3] = 1.0
        else:
            c = rng.choice([0.1, 0.3, 0.4, 0.5, 0.7, 0.8, 1.0], (n, m))
        yield np.ascontiguousarray(c, dtype=np.float64), L


def main():
    rng = np.random.default_rng(0)
    stats = {"unique": 0, "tied": 0, "unique_but_differs": 0, "tied_and_same": 0}
    for c, L in families(rng, 6000):
        _, xo, yo = bo.lapjv(c, extend_cost=True, cost_limit=L)
        xo, yo = np.asarray(xo), np.asarray(yo)
        xs, ys = scipy_visible(c, L)
        uniq = certificate(c, L, xs, ys)
        same = np.array_equal(xs, xo) and np.array_equal(ys, yo)
        if uniq:
            stats["unique"] += 1
            if not same:
                stats["unique_but_differs"] += 1
                print("SOUNDNESS VIOLATION\n", c, xs, xo)
            # the certificate must say the same thing from the JV's own solution
            assert certificate(c, L, xo, yo)
        else:
            stats["tied"] += 1
            stats["tied_and_same"] += same
    print(stats)


if __name__ == "__main__":
    main()


def tie_free_check():
    rng = np.random.default_rng(1)
    fp = 0
    for t in range(300):
        n, m = rng.integers(5, 60, 2)
        c = rng.uniform(0.05, 0.95, (n, m)); c[rng.random((n, m)) < 0.8] = 1.0
        xs, ys = scipy_visible(c, 0.8)
        fp += not certificate(c, 0.8, xs, ys)
    print("false positives on tie-free float costs:", fp, "of 300")


if __name__ == "__main__" and "--tiefree" in sys.argv:
    tie_free_check()
