/*
 * oracle/lsap_oracle.c -- TEST INFRASTRUCTURE ONLY (never imported by the product path).
 *
 * CPU restatement of the rectangular linear-sum-assignment solver the reference calls:
 *   scipy.optimize.linear_sum_assignment  (third-party, NOT vendored under /root/reference and not
 *   version-pinned by it; call sites thirdparty/mmdetection/mmdet/core/bbox/assigners/
 *   hungarian_assigner.py:136 and detr_ssod/models/dino_detr_ssod.py:279).  Pinned here against
 *   scipy 1.15.3 (the version in the build image).
 *
 * Published algorithm restated: D. F. Crouse, "On implementing 2D rectangular assignment
 * algorithms", IEEE T-AES 52(4), 2016 -- shortest augmenting path with dual variables (u, v); the
 * problem is transposed when rows > cols; the unscanned-column list is kept as an array initialised
 * in DESCENDING column order and shrunk by swap-with-last; on equal reduced cost an UNASSIGNED
 * column replaces the incumbent (this tie rule + the list order are what make indices bit-exact).
 *
 * Return: 0 ok, -1 infeasible (scipy: ValueError "cost matrix is infeasible"),
 *         -2 invalid entry NaN / -inf (scipy: ValueError "matrix contains invalid numeric entries").
 * Output: a[k], b[k] for k < min(nr, nc): row indices ascending and their columns (int64).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static int64_t augment_path(int64_t nc, const double *cost, const double *u, const double *v,
                            int64_t *path, const int64_t *row4col, double *spc, int64_t i,
                            char *SR, char *SC, int64_t *remaining, double *p_min)
{
    double min_val = 0;
    int64_t num_remaining = nc;
    for (int64_t it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
    for (int64_t j = 0; j < nc; ++j) { SC[j] = 0; spc[j] = INFINITY; }
    int64_t sink = -1;
    while (sink == -1) {
        int64_t index = -1;
        double lowest = INFINITY;
        SR[i] = 1;
        for (int64_t it = 0; it < num_remaining; ++it) {
            const int64_t j = remaining[it];
            const double r = min_val + cost[i * nc + j] - u[i] - v[j];
            if (r < spc[j]) { path[j] = i; spc[j] = r; }
            if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) {
                lowest = spc[j];
                index = it;
            }
        }
        min_val = lowest;
        if (min_val == INFINITY) return -1;
        const int64_t j = remaining[index];
        if (row4col[j] == -1) sink = j; else i = row4col[j];
        SC[j] = 1;
        remaining[index] = remaining[--num_remaining];
    }
    *p_min = min_val;
    return sink;
}

static int cmp_pair(const void *x, const void *y)
{
    const int64_t *a = (const int64_t *)x, *b = (const int64_t *)y;
    return (a[0] > b[0]) - (a[0] < b[0]);
}

/* cost: row-major (nr x nc) double. */
int lsap_oracle_solve(int64_t nr, int64_t nc, const double *cost_in, int64_t *a, int64_t *b)
{
    if (nr == 0 || nc == 0) return 0;
    const int transpose = nc < nr;
    double *cost = (double *)malloc(sizeof(double) * nr * nc);
    if (transpose) {
        for (int64_t i = 0; i < nr; ++i)
            for (int64_t j = 0; j < nc; ++j) cost[j * nr + i] = cost_in[i * nc + j];
        int64_t t = nr; nr = nc; nc = t;
    } else {
        for (int64_t k = 0; k < nr * nc; ++k) cost[k] = cost_in[k];
    }
    for (int64_t k = 0; k < nr * nc; ++k)
        if (cost[k] != cost[k] || cost[k] == -INFINITY) { free(cost); return -2; }

    double *u = (double *)calloc(nr, sizeof(double)), *v = (double *)calloc(nc, sizeof(double));
    double *spc = (double *)malloc(sizeof(double) * nc);
    int64_t *path = (int64_t *)malloc(sizeof(int64_t) * nc);
    int64_t *col4row = (int64_t *)malloc(sizeof(int64_t) * nr);
    int64_t *row4col = (int64_t *)malloc(sizeof(int64_t) * nc);
    int64_t *remaining = (int64_t *)malloc(sizeof(int64_t) * nc);
    char *SR = (char *)malloc(nr), *SC = (char *)malloc(nc);
    for (int64_t j = 0; j < nc; ++j) { path[j] = -1; row4col[j] = -1; }
    for (int64_t i = 0; i < nr; ++i) col4row[i] = -1;
    int rc = 0;
    for (int64_t cur = 0; cur < nr; ++cur) {
        double min_val;
        for (int64_t i = 0; i < nr; ++i) SR[i] = 0;
        const int64_t sink = augment_path(nc, cost, u, v, path, row4col, spc, cur, SR, SC,
                                          remaining, &min_val);
        if (sink < 0) { rc = -1; break; }
        u[cur] += min_val;
        for (int64_t i = 0; i < nr; ++i)
            if (SR[i] && i != cur) u[i] += min_val - spc[col4row[i]];
        for (int64_t j = 0; j < nc; ++j)
            if (SC[j]) v[j] -= min_val - spc[j];
        int64_t j = sink;
        for (;;) {
            const int64_t i = path[j];
            row4col[j] = i;
            const int64_t t = col4row[i]; col4row[i] = j; j = t;
            if (i == cur) break;
        }
    }
    if (rc == 0) {
        if (transpose) {
            int64_t *pairs = (int64_t *)malloc(sizeof(int64_t) * 2 * nr);
            for (int64_t i = 0; i < nr; ++i) { pairs[2 * i] = col4row[i]; pairs[2 * i + 1] = i; }
            qsort(pairs, nr, 2 * sizeof(int64_t), cmp_pair);
            for (int64_t i = 0; i < nr; ++i) { a[i] = pairs[2 * i]; b[i] = pairs[2 * i + 1]; }
            free(pairs);
        } else {
            for (int64_t i = 0; i < nr; ++i) { a[i] = i; b[i] = col4row[i]; }
        }
    }
    free(cost); free(u); free(v); free(spc); free(path); free(col4row); free(row4col);
    free(remaining); free(SR); free(SC);
    return rc;
}
