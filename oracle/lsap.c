/* Rectangular linear sum assignment - plain C restatement, TEST ORACLE ONLY.
 *
 * Restates the algorithm of SciPy's `scipy.optimize.linear_sum_assignment`
 * (scipy 1.15.x, scipy/optimize/rectangular_lsap/rectangular_lsap.cpp; D. F. Crouse,
 * "On implementing 2D rectangular assignment algorithms", IEEE TAES 52(4), 2016) which the
 * reference calls at src/d_fine/matcher.py:243,264.  SciPy's source is not vendored in
 * /root/reference nor present in the container (only the compiled _lsap module), so this file
 * is written from the published algorithm and pinned by differential tests against the
 * container's scipy (tests/test_lsap_oracle.py) and the committed golden vectors.
 *
 * Conventions reproduced: float64 arithmetic; a tall matrix (rows > cols) is transposed first;
 * rows are inserted in index order, each by one shortest augmenting path with dual updates;
 * the unscanned-column list is initialised in REVERSE order (so an all-equal matrix yields the
 * identity) and shrunk by swap-with-last; among equal minimum reduced costs a column that is
 * still unassigned wins; output rows ascending.
 *
 * int oracle_lsap(nr, nc, cost[nr*nc] row-major, rows_out[min], cols_out[min])
 *   returns min(nr,nc) on success, -1 infeasible, -2 invalid (NaN / -inf) entry.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int64_t augment(int64_t nc, const double *cost, const double *u, const double *v,
                       int64_t *path, const int64_t *row4col, double *spc, int64_t i,
                       char *SR, char *SC, int64_t *remaining, double *p_min)
{
    double min_val = 0.0;
    int64_t n_rem = nc;
    for (int64_t it = 0; it < nc; it++) remaining[it] = nc - it - 1;
    memset(SR, 0, (size_t)nc > 0 ? (size_t)nc : 1); /* SR sized >= nr by caller (nr <= nc) */
    memset(SC, 0, (size_t)nc);
    for (int64_t j = 0; j < nc; j++) spc[j] = INFINITY;

    int64_t sink = -1;
    while (sink == -1) {
        int64_t index = -1;
        double lowest = INFINITY;
        SR[i] = 1;
        for (int64_t it = 0; it < n_rem; it++) {
            int64_t j = remaining[it];
            double r = min_val + cost[i * nc + j] - u[i] - v[j];
            if (r < spc[j]) { path[j] = i; spc[j] = r; }
            if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) {
                lowest = spc[j];
                index = it;
            }
        }
        min_val = lowest;
        if (min_val == INFINITY) return -1;
        int64_t j = remaining[index];
        if (row4col[j] == -1) sink = j; else i = row4col[j];
        SC[j] = 1;
        remaining[index] = remaining[--n_rem];
    }
    *p_min = min_val;
    return sink;
}

int oracle_lsap(int64_t nr, int64_t nc, const double *cost_in, int64_t *rows_out, int64_t *cols_out)
{
    if (nr == 0 || nc == 0) return 0;
    int transpose = nc < nr;
    double *tmp = NULL;
    const double *cost = cost_in;
    if (transpose) {
        tmp = (double *)malloc(sizeof(double) * (size_t)(nr * nc));
        for (int64_t i = 0; i < nr; i++)
            for (int64_t j = 0; j < nc; j++) tmp[j * nr + i] = cost_in[i * nc + j];
        int64_t t = nr; nr = nc; nc = t;
        cost = tmp;
    }
    for (int64_t k = 0; k < nr * nc; k++)
        if (cost[k] != cost[k] || cost[k] == -INFINITY) { free(tmp); return -2; }

    double *u = (double *)calloc((size_t)nr, sizeof(double));
    double *v = (double *)calloc((size_t)nc, sizeof(double));
    double *spc = (double *)malloc(sizeof(double) * (size_t)nc);
    int64_t *path = (int64_t *)malloc(sizeof(int64_t) * (size_t)nc);
    int64_t *col4row = (int64_t *)malloc(sizeof(int64_t) * (size_t)nr);
    int64_t *row4col = (int64_t *)malloc(sizeof(int64_t) * (size_t)nc);
    int64_t *remaining = (int64_t *)malloc(sizeof(int64_t) * (size_t)nc);
    char *SR = (char *)malloc((size_t)nc), *SC = (char *)malloc((size_t)nc);
    for (int64_t j = 0; j < nc; j++) { path[j] = -1; row4col[j] = -1; }
    for (int64_t i = 0; i < nr; i++) col4row[i] = -1;

    int status = 0;
    for (int64_t cur = 0; cur < nr; cur++) {
        double min_val;
        int64_t sink = augment(nc, cost, u, v, path, row4col, spc, cur, SR, SC, remaining, &min_val);
        if (sink < 0) { status = -1; break; }
        u[cur] += min_val;
        for (int64_t i = 0; i < nr; i++)
            if (SR[i] && i != cur) u[i] += min_val - spc[col4row[i]];
        for (int64_t j = 0; j < nc; j++)
            if (SC[j]) v[j] -= min_val - spc[j];
        int64_t j = sink;
        for (;;) {
            int64_t i = path[j];
            row4col[j] = i;
            int64_t t = col4row[i]; col4row[i] = j; j = t;
            if (i == cur) break;
        }
    }
    int n = (int)nr;
    if (status == 0) {
        if (transpose) {
            /* original rows = our columns: emit (col4row[v], v) sorted by col4row[v] */
            int64_t k = 0;
            for (int64_t c = 0; c < nc; c++)
                if (row4col[c] != -1) { rows_out[k] = c; cols_out[k] = row4col[c]; k++; }
        } else {
            for (int64_t i = 0; i < nr; i++) { rows_out[i] = i; cols_out[i] = col4row[i]; }
        }
    }
    free(tmp); free(u); free(v); free(spc); free(path); free(col4row); free(row4col);
    free(remaining); free(SR); free(SC);
    return status == 0 ? n : status;
}
