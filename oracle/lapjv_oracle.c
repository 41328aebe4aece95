/*
 * oracle/lapjv_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the dense Jonker-Volgenant linear assignment solver that the
 * reference reaches through `lap.lapjv(cost, extend_cost=True, cost_limit=thresh)`
 * (reference call site: demo_bottrack_onnx_tflite.py:1686, wrapper 1682-1693).
 *
 * The algorithm itself lives in the third-party package `lap==0.4.0`
 * (pinned in the reference's Dockerfile:26 and demo_bottrack_onnx_tflite.py:7); its
 * source is NOT under /root/reference and the package is not installable here, so
 * this file restates the *published* algorithm:
 *   R. Jonker, A. Volgenant, "A shortest augmenting path algorithm for dense and
 *   sparse linear assignment problems", Computing 38 (1987) 325-340
 * in the phase structure lap 0.4.0 is documented to use:
 *   (1) column reduction, columns visited high -> low, first minimal row wins,
 *   (2) reduction transfer for rows that own exactly one column,
 *   (3) two sweeps of augmenting row reduction,
 *   (4) shortest augmenting path (Dijkstra-like, "ready/scan/todo" column sets)
 *       for every row still free.
 * and the `extend_cost/cost_limit` wrapper semantics: the n x m cost is embedded
 * in an (n+m) x (n+m) matrix filled with cost_limit/2, whose lower-right m x n
 * block is 0; assignments into the padding are reported as -1.
 *
 * PARITY STATUS: "parity unpinned" for tie-breaking -- there is no lap binary or golden vector in the reference to pin
 * against; the phase structure, scan orders and strictness of every comparison below follow lap 0.4.0's lapjv.cpp
 * (_ccrrt_dense, _carr_dense, _find_dense, _scan_dense, find_path_dense, _ca_dense) as published.  What IS pinned
 * (tests/test_oracle_vs_golden.py): optimal objective and, on tie-free inputs, the visible matching, against
 * scipy.optimize.linear_sum_assignment on the same extended matrix.  On exact ties this file is the DEFINITION the product's
 * exact-tie mode is tested against, index for index (tests/test_lap_exact_host.py on the CPU, tests/test_zz_gpu_ties.py on
 * the GPU).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Build: see oracle/Makefile  (gcc -O2 -shared -fPIC).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define BIG DBL_MAX

typedef struct {
    int n;              /* square size */
    const double *c;    /* row-major n*n */
    int *x;             /* row -> col */
    int *y;             /* col -> row */
    double *v;          /* column duals */
} jv_t;

#define C(p, i, j) ((p)->c[(size_t)(i) * (size_t)(p)->n + (size_t)(j)])

/* Phase 1+2: column reduction and reduction transfer. Returns number of free rows. */
static int jv_column_reduction(jv_t *p, int *free_rows)
{
    const int n = p->n;
    int i, j, nfree = 0;
    unsigned char *owns_one = (unsigned char *)malloc((size_t)n);

    for (i = 0; i < n; i++) { p->x[i] = -1; p->v[i] = BIG; p->y[i] = 0; }
    /* per-column minimum; rows are swept in ascending order and only a strictly
       smaller entry replaces the incumbent, so the first minimal row wins */
    for (i = 0; i < n; i++) {
        const double *row = &C(p, i, 0);
        for (j = 0; j < n; j++) {
            if (row[j] < p->v[j]) { p->v[j] = row[j]; p->y[j] = i; }
        }
    }
    memset(owns_one, 1, (size_t)n);
    for (j = n - 1; j >= 0; j--) {          /* columns high -> low */
        i = p->y[j];
        if (p->x[i] < 0) {
            p->x[i] = j;
        } else {                             /* row already took a higher column */
            owns_one[i] = 0;
            p->y[j] = -1;
        }
    }
    for (i = 0; i < n; i++) {
        if (p->x[i] < 0) {
            free_rows[nfree++] = i;
        } else if (owns_one[i]) {            /* reduction transfer */
            const int j1 = p->x[i];
            double mn = BIG;
            const double *row = &C(p, i, 0);
            for (j = 0; j < n; j++) {
                if (j == j1) continue;
                const double r = row[j] - p->v[j];
                if (r < mn) mn = r;
            }
            p->v[j1] -= mn;
        }
    }
    free(owns_one);
    return nfree;
}

/* Phase 3: one sweep of augmenting row reduction. Returns new number of free rows. */
static int jv_row_reduction(jv_t *p, int nfree, int *free_rows)
{
    const int n = p->n;
    int cur = 0, nnew = 0;
    long long sweeps = 0;

    while (cur < nfree) {
        const int fi = free_rows[cur++];
        const double *row = &C(p, fi, 0);
        int j, j1 = 0, j2 = -1, i0;
        double r1 = row[0] - p->v[0], r2 = BIG, v1new;
        int lowers;
        sweeps++;
        for (j = 1; j < n; j++) {
            const double r = row[j] - p->v[j];
            if (r < r2) {
                if (r >= r1) { r2 = r; j2 = j; }
                else { r2 = r1; r1 = r; j2 = j1; j1 = j; }
            }
        }
        i0 = p->y[j1];
        v1new = p->v[j1] - (r2 - r1);
        lowers = v1new < p->v[j1];
        if (sweeps < (long long)cur * (long long)n) {
            if (lowers) {
                p->v[j1] = v1new;
            } else if (i0 >= 0 && j2 >= 0) {
                j1 = j2;
                i0 = p->y[j2];
            }
            if (i0 >= 0) {
                if (lowers) free_rows[--cur] = i0;   /* re-examine the evicted row now */
                else        free_rows[nnew++] = i0;  /* ... or in the next sweep */
            }
        } else if (i0 >= 0) {
            free_rows[nnew++] = i0;
        }
        p->x[fi] = j1;
        p->y[j1] = fi;
    }
    return nnew;
}

/* Collect, among todo columns cols[lo..n), those with minimal d; they are moved to
   cols[lo..hi).  Returns hi. */
static int jv_collect_min(int n, int lo, const double *d, int *cols)
{
    int hi = lo + 1, k;
    double mind = d[cols[lo]];
    for (k = hi; k < n; k++) {
        const int j = cols[k];
        if (d[j] <= mind) {
            if (d[j] < mind) { hi = lo; mind = d[j]; }
            cols[k] = cols[hi];
            cols[hi++] = j;
        }
    }
    return hi;
}

/* Scan the columns in cols[lo..hi) (all at distance mind); relax the todo columns.
   Returns a free column reached at distance mind, or -1. */
static int jv_scan(jv_t *p, int *plo, int *phi, double *d, int *cols, int *pred)
{
    const int n = p->n;
    int lo = *plo, hi = *phi, k;
    while (lo != hi) {
        int j = cols[lo++];
        const int i = p->y[j];
        const double mind = d[j];
        const double h = C(p, i, j) - p->v[j] - mind;
        const double *row = &C(p, i, 0);
        for (k = hi; k < n; k++) {
            double red;
            j = cols[k];
            red = row[j] - p->v[j] - h;
            if (red < d[j]) {
                d[j] = red;
                pred[j] = i;
                if (red == mind) {
                    if (p->y[j] < 0) return j;
                    cols[k] = cols[hi];
                    cols[hi++] = j;
                }
            }
        }
    }
    *plo = lo; *phi = hi;
    return -1;
}

/* Phase 4: shortest augmenting path from every remaining free row. */
static void jv_augment(jv_t *p, int nfree, const int *free_rows)
{
    const int n = p->n;
    int *pred = (int *)malloc(sizeof(int) * (size_t)n);
    int *cols = (int *)malloc(sizeof(int) * (size_t)n);
    double *d = (double *)malloc(sizeof(double) * (size_t)n);
    int f;
    for (f = 0; f < nfree; f++) {
        const int start = free_rows[f];
        const double *row = &C(p, start, 0);
        int lo = 0, hi = 0, nready = 0, endj = -1, k, i, j;
        for (j = 0; j < n; j++) { cols[j] = j; pred[j] = start; d[j] = row[j] - p->v[j]; }
        while (endj < 0) {
            if (lo == hi) {
                nready = lo;
                hi = jv_collect_min(n, lo, d, cols);
                for (k = lo; k < hi; k++) {
                    if (p->y[cols[k]] < 0) endj = cols[k];
                }
            }
            if (endj < 0) endj = jv_scan(p, &lo, &hi, d, cols, pred);
        }
        {
            const double mind = d[cols[lo]];
            for (k = 0; k < nready; k++) { j = cols[k]; p->v[j] += d[j] - mind; }
        }
        /* flip the path */
        j = endj; i = -1;
        while (i != start) {
            int t;
            i = pred[j];
            p->y[j] = i;
            t = p->x[i]; p->x[i] = j; j = t;
        }
    }
    free(pred); free(cols); free(d);
}

/* Square dense solve. x[n], y[n] out. Returns 0. */
int bts_oracle_lapjv_square(int n, const double *cost, int *x, int *y)
{
    jv_t p;
    int *free_rows, nfree, sweep = 0;
    double *v;
    if (n <= 0) return 0;
    free_rows = (int *)malloc(sizeof(int) * (size_t)n);
    v = (double *)malloc(sizeof(double) * (size_t)n);
    p.n = n; p.c = cost; p.x = x; p.y = y; p.v = v;
    nfree = jv_column_reduction(&p, free_rows);
    while (nfree > 0 && sweep < 2) { nfree = jv_row_reduction(&p, nfree, free_rows); sweep++; }
    if (nfree > 0) jv_augment(&p, nfree, free_rows);
    free(free_rows); free(v);
    return 0;
}

/*
 * lap.lapjv(cost[n_rows, n_cols], extend_cost=True, cost_limit=limit) semantics.
 * x[n_rows], y[n_cols] out (-1 = unmatched).  *opt receives the sum of the matched
 * real costs.  Returns 0, or -1 on allocation failure.
 */
int bts_oracle_lapjv_extended(int n_rows, int n_cols, const double *cost, double limit,
                              int *x, int *y, double *opt)
{
    const int n = n_rows + n_cols;
    size_t i, j;
    double *ext;
    int *xx, *yy;
    double total = 0.0;
    if (n == 0) { if (opt) *opt = 0.0; return 0; }
    ext = (double *)malloc(sizeof(double) * (size_t)n * (size_t)n);
    xx = (int *)malloc(sizeof(int) * (size_t)n);
    yy = (int *)malloc(sizeof(int) * (size_t)n);
    if (!ext || !xx || !yy) { free(ext); free(xx); free(yy); return -1; }
    for (i = 0; i < (size_t)n * (size_t)n; i++) ext[i] = limit / 2.0;
    for (i = (size_t)n_rows; i < (size_t)n; i++)
        for (j = (size_t)n_cols; j < (size_t)n; j++) ext[i * (size_t)n + j] = 0.0;
    for (i = 0; i < (size_t)n_rows; i++)
        memcpy(&ext[i * (size_t)n], &cost[i * (size_t)n_cols], sizeof(double) * (size_t)n_cols);
    bts_oracle_lapjv_square(n, ext, xx, yy);
    for (i = 0; i < (size_t)n_rows; i++) {
        x[i] = (xx[i] >= n_cols) ? -1 : xx[i];
        if (x[i] >= 0) total += cost[i * (size_t)n_cols + (size_t)x[i]];
    }
    for (j = 0; j < (size_t)n_cols; j++) y[j] = (yy[j] >= n_rows) ? -1 : yy[j];
    if (opt) *opt = total;
    free(ext); free(xx); free(yy);
    return 0;
}
