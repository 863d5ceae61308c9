/*
 * ezpz_oracle_solve.c -- TEST INFRASTRUCTURE ONLY (see ezpz_oracle.h).
 *
 * Part 2: restates
 *   ezpz/src/solver.rs:142-189   validate_variables
 *   ezpz/src/solver.rs:192-300   Model::new (J sparsity, symbolic Cholesky)
 *   ezpz/src/solver.rs:318-356   Model::residual
 *   ezpz/src/solver.rs:359-440   Model::refresh_jacobian
 *   ezpz/src/solver/newton.rs:29-145, :232-236   solve_levenberg_marquardt, eval
 *   ezpz/src/lib.rs:148-370      solve_with_priority_inner, solve_inner, is_satisfied
 *   ezpz/src/warnings.rs:34-60   lint
 * faer 0.24.0 (Cargo.lock:601-604; not under /root/reference) is replaced by
 *   - a dense Cholesky (ORC_LINSOLVE_DENSE), and
 *   - an up-looking sparse Cholesky with elimination tree, natural ordering
 *     (ORC_LINSOLVE_SPARSE; all symbolic work redone per solve like Model::new does).
 * Both fail exactly when a pivot is not > 0, which is what LltError::Numeric reports.
 */
#define _POSIX_C_SOURCE 200809L
#include "ezpz_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EPSILON 1e-4
#define LM_LAMBDA_INCR 10.0 /* newton.rs:15 */
#define LM_LAMBDA_DECR 0.1  /* newton.rs:16 */

double orc_angle_to_degrees(uint8_t tag, double val);

void orc_default_config(OrcConfig* cfg) {
    /* solver.rs:72-81 */
    cfg->max_iterations = 35;
    cfg->residual_tolerance = 1e-8;
    cfg->step_tolerance = 1e-12;
    cfg->initial_lambda = 1e-9;
}

/* ---- warnings ---------------------------------------------------------------------------------- */
typedef struct {
    OrcWarning* buf;
    size_t cap;
    uint64_t count;
} WarnSink;
static void warn_push(WarnSink* w, int32_t about, int32_t content) {
    if (w->buf && w->count < w->cap) {
        w->buf[w->count].about_constraint = about;
        w->buf[w->count].content = content;
    }
    w->count++;
}

static int nearly_eq(double a, double b) { return fabs(a - b) < EPSILON; } /* warnings.rs:85-87 */

/* warnings.rs:34-60 */
static void lint(const OrcConstraint* cs, const uint64_t* orig_ids, size_t n, WarnSink* w) {
    for (size_t i = 0; i < n; ++i) {
        const OrcConstraint* c = &cs[i];
        if (c->kind != ORC_LINES_AT_ANGLE) continue;
        if (c->tag != ORC_ANGLE_OTHER_DEG && c->tag != ORC_ANGLE_OTHER_RAD) continue;
        double deg = orc_angle_to_degrees(c->tag, c->param);
        int32_t id = (int32_t)(orig_ids ? orig_ids[i] : i);
        if (nearly_eq(deg, 0.0) || nearly_eq(deg, 360.0) || nearly_eq(deg, 180.0)) {
            warn_push(w, id, ORC_WARN_SHOULD_BE_PARALLEL);
        } else if (nearly_eq(deg, 90.0) || nearly_eq(deg, -90.0)) {
            warn_push(w, id, ORC_WARN_SHOULD_BE_PERPENDICULAR);
        }
    }
}

/* ---- Model ---------------------------------------------------------------------------------------- */
typedef struct {
    size_t m, n, n_cs;
    const OrcConstraint* cs;
    /* J in CSC (faer SymbolicSparseColMat): col_ptr[n+1], row_idx[nnz] sorted, deduplicated */
    size_t* col_ptr;
    size_t* row_idx;
    double* vals;
    size_t nnz;
    WarnSink* warnings;
    int linsolve;
    /* dense workspace */
    double* A; /* n*n */
    /* sparse workspace */
    size_t* jt_ptr; /* CSR of J: row -> (col, slot) */
    size_t* jt_col;
    size_t* jt_slot;
    size_t* a_ptr; /* upper-triangular A in CSC: col k holds rows i<=k, sorted, diagonal last */
    size_t* a_row;
    double* a_val;
    size_t* parent; /* etree */
    size_t* l_ptr;  /* L in CSC, diagonal first in each column */
    size_t* l_row;
    double* l_val;
    size_t* l_fill; /* next free slot per column during numeric */
    size_t* stack;
    size_t* flag;
    double* work;
} Model;

static void model_free(Model* md) {
    free(md->col_ptr);
    free(md->row_idx);
    free(md->vals);
    free(md->A);
    free(md->jt_ptr);
    free(md->jt_col);
    free(md->jt_slot);
    free(md->a_ptr);
    free(md->a_row);
    free(md->a_val);
    free(md->parent);
    free(md->l_ptr);
    free(md->l_row);
    free(md->l_val);
    free(md->l_fill);
    free(md->stack);
    free(md->flag);
    free(md->work);
}

typedef struct {
    size_t row, col;
} Pair;
static int pair_cmp(const void* a, const void* b) {
    const Pair* p = (const Pair*)a;
    const Pair* q = (const Pair*)b;
    if (p->col != q->col) return p->col < q->col ? -1 : 1;
    if (p->row != q->row) return p->row < q->row ? -1 : 1;
    return 0;
}

#define NONE ((size_t)-1)

/* ereach: nonzero pattern of row k of L = nodes reachable in the etree from the entries of A(0:k-1,k).
 * Returns `top`; pattern is stack[top..n-1] in topological order.  flag[] must hold values != mark. */
static size_t ereach(const Model* md, size_t k, size_t mark, size_t* stack, size_t* flag) {
    size_t n = md->n, top = n;
    flag[k] = mark;
    for (size_t p = md->a_ptr[k]; p < md->a_ptr[k + 1]; ++p) {
        size_t i = md->a_row[p];
        if (i >= k) continue;
        size_t len = 0;
        for (; flag[i] != mark; i = md->parent[i]) {
            stack[len++] = i;
            flag[i] = mark;
        }
        while (len > 0) stack[--top] = stack[--len];
    }
    return top;
}

/* Symbolic phase for the sparse path: CSR of J, pattern of upper(A)=JtJ+I, etree, pattern of L. */
static int model_symbolic_sparse(Model* md) {
    size_t m = md->m, n = md->n, nnz = md->nnz;
    md->jt_ptr = (size_t*)calloc(m + 2, sizeof(size_t));
    md->jt_col = (size_t*)malloc((nnz + 1) * sizeof(size_t));
    md->jt_slot = (size_t*)malloc((nnz + 1) * sizeof(size_t));
    if (!md->jt_ptr || !md->jt_col || !md->jt_slot) return ORC_ERR_INTERNAL;
    for (size_t p = 0; p < nnz; ++p) md->jt_ptr[md->row_idx[p] + 1]++;
    for (size_t r = 0; r < m; ++r) md->jt_ptr[r + 1] += md->jt_ptr[r];
    {
        size_t* next = (size_t*)malloc((m + 1) * sizeof(size_t));
        if (!next) return ORC_ERR_INTERNAL;
        memcpy(next, md->jt_ptr, (m + 1) * sizeof(size_t));
        for (size_t c = 0; c < n; ++c) {
            for (size_t p = md->col_ptr[c]; p < md->col_ptr[c + 1]; ++p) {
                size_t r = md->row_idx[p];
                md->jt_col[next[r]] = c;
                md->jt_slot[next[r]] = p;
                next[r]++;
            }
        }
        free(next);
    }
    /* upper(A) pattern, column k: all i<=k sharing a row with k, plus the diagonal (lambda I). */
    md->flag = (size_t*)malloc((n + 1) * sizeof(size_t));
    md->a_ptr = (size_t*)calloc(n + 2, sizeof(size_t));
    if (!md->flag || !md->a_ptr) return ORC_ERR_INTERNAL;
    for (size_t i = 0; i < n; ++i) md->flag[i] = NONE;
    size_t cap = nnz * 4 + n + 16, cnt = 0;
    md->a_row = (size_t*)malloc(cap * sizeof(size_t));
    if (!md->a_row) return ORC_ERR_INTERNAL;
    for (size_t k = 0; k < n; ++k) {
        size_t start = cnt;
        md->flag[k] = k;
        for (size_t p = md->col_ptr[k]; p < md->col_ptr[k + 1]; ++p) {
            size_t r = md->row_idx[p];
            for (size_t q = md->jt_ptr[r]; q < md->jt_ptr[r + 1]; ++q) {
                size_t i = md->jt_col[q];
                if (i < k && md->flag[i] != k) {
                    md->flag[i] = k;
                    if (cnt + 2 >= cap) {
                        cap *= 2;
                        size_t* t = (size_t*)realloc(md->a_row, cap * sizeof(size_t));
                        if (!t) return ORC_ERR_INTERNAL;
                        md->a_row = t;
                    }
                    md->a_row[cnt++] = i;
                }
            }
        }
        /* sort the strictly-upper rows (insertion sort; columns are short) */
        for (size_t a = start + 1; a < cnt; ++a) {
            size_t v = md->a_row[a], b = a;
            while (b > start && md->a_row[b - 1] > v) {
                md->a_row[b] = md->a_row[b - 1];
                --b;
            }
            md->a_row[b] = v;
        }
        md->a_row[cnt++] = k; /* diagonal last */
        md->a_ptr[k + 1] = cnt;
    }
    md->a_val = (double*)malloc((cnt + 1) * sizeof(double));
    /* etree (Liu) */
    md->parent = (size_t*)malloc((n + 1) * sizeof(size_t));
    size_t* ancestor = (size_t*)malloc((n + 1) * sizeof(size_t));
    if (!md->a_val || !md->parent || !ancestor) return ORC_ERR_INTERNAL;
    for (size_t k = 0; k < n; ++k) {
        md->parent[k] = NONE;
        ancestor[k] = NONE;
        for (size_t p = md->a_ptr[k]; p < md->a_ptr[k + 1]; ++p) {
            size_t i = md->a_row[p];
            while (i != NONE && i < k) {
                size_t inext = ancestor[i];
                ancestor[i] = k;
                if (inext == NONE) md->parent[i] = k;
                i = inext;
            }
        }
    }
    free(ancestor);
    /* column counts of L through row patterns */
    md->stack = (size_t*)malloc((n + 1) * sizeof(size_t));
    md->l_ptr = (size_t*)calloc(n + 2, sizeof(size_t));
    md->l_fill = (size_t*)malloc((n + 1) * sizeof(size_t));
    md->work = (double*)calloc(n + 1, sizeof(double));
    if (!md->stack || !md->l_ptr || !md->l_fill || !md->work) return ORC_ERR_INTERNAL;
    for (size_t i = 0; i < n; ++i) md->flag[i] = NONE;
    size_t* counts = (size_t*)calloc(n + 1, sizeof(size_t));
    if (!counts) return ORC_ERR_INTERNAL;
    for (size_t k = 0; k < n; ++k) {
        size_t top = ereach(md, k, k, md->stack, md->flag);
        for (size_t t = top; t < n; ++t) counts[md->stack[t]]++;
        counts[k]++; /* diagonal */
    }
    for (size_t k = 0; k < n; ++k) md->l_ptr[k + 1] = md->l_ptr[k] + counts[k];
    free(counts);
    size_t lnz = md->l_ptr[n];
    md->l_row = (size_t*)malloc((lnz + 1) * sizeof(size_t));
    md->l_val = (double*)malloc((lnz + 1) * sizeof(double));
    if (!md->l_row || !md->l_val) return ORC_ERR_INTERNAL;
    return ORC_OK;
}

/* solver.rs:192-284 Model::new (+ validate_variables :142-189) */
static int model_new(Model* md, const OrcConstraint* cs, const uint64_t* orig_ids, size_t n_cs,
                     const uint32_t* all_variables, size_t n_vars, size_t n_initial_values, int linsolve,
                     WarnSink* warnings, OrcOutcome* out) {
    memset(md, 0, sizeof(*md));
    md->cs = cs;
    md->n_cs = n_cs;
    md->n = n_vars;
    md->warnings = warnings;
    md->linsolve = linsolve;
    if (n_vars != n_initial_values) {
        out->error = ORC_ERR_WRONG_NUMBER_GUESSES;
        return out->error;
    }
    /* validate_variables: `all_variables.contains(v)`.  An O(1) membership table gives the same answer. */
    uint32_t max_id = 0;
    for (size_t i = 0; i < n_vars; ++i)
        if (all_variables[i] > max_id) max_id = all_variables[i];
    uint8_t* present = (uint8_t*)calloc((size_t)max_id + 2, 1);
    if (!present) return ORC_ERR_INTERNAL;
    for (size_t i = 0; i < n_vars; ++i) present[all_variables[i]] = 1;
    size_t m = 0;
    for (size_t i = 0; i < n_cs; ++i) {
        uint32_t r0[8], r1[8];
        int n0, n1;
        orc_nonzeroes(&cs[i], r0, &n0, r1, &n1);
        for (int k = 0; k < n0 + n1; ++k) {
            uint32_t v = (k < n0) ? r0[k] : r1[k - n0];
            if (n_vars == 0 || v > max_id || !present[v]) {
                out->error = ORC_ERR_MISSING_GUESS;
                out->err_constraint_id = (int32_t)(orig_ids ? orig_ids[i] : i);
                out->err_variable = v;
                free(present);
                return out->error;
            }
        }
        m += (size_t)orc_residual_dim(&cs[i]);
    }
    free(present);
    md->m = m;
    /* (row, col) pairs -> sorted, deduplicated CSC.  Layout::index_of(var) = var (solver.rs:107-109). */
    Pair* pairs = (Pair*)malloc((8 * m + 1) * sizeof(Pair));
    if (!pairs) return ORC_ERR_INTERNAL;
    size_t np = 0, row_num = 0;
    for (size_t i = 0; i < n_cs; ++i) {
        uint32_t r0[8], r1[8];
        int n0, n1;
        int dim = orc_nonzeroes(&cs[i], r0, &n0, r1, &n1);
        for (int k = 0; k < n0; ++k) {
            pairs[np].row = row_num;
            pairs[np++].col = r0[k];
        }
        row_num++;
        if (dim > 1) {
            for (int k = 0; k < n1; ++k) {
                pairs[np].row = row_num;
                pairs[np++].col = r1[k];
            }
            row_num++;
        }
    }
    /* A column index >= n_vars would be rejected by faer's try_new_from_indices (FaerMatrix error). */
    for (size_t p = 0; p < np; ++p) {
        if (pairs[p].col >= n_vars) {
            free(pairs);
            out->error = ORC_ERR_INTERNAL;
            return out->error;
        }
    }
    qsort(pairs, np, sizeof(Pair), pair_cmp);
    size_t nnz = 0;
    for (size_t p = 0; p < np; ++p) {
        if (p == 0 || pair_cmp(&pairs[p], &pairs[p - 1]) != 0) pairs[nnz++] = pairs[p];
    }
    md->nnz = nnz;
    md->col_ptr = (size_t*)calloc(n_vars + 2, sizeof(size_t));
    md->row_idx = (size_t*)malloc((nnz + 1) * sizeof(size_t));
    md->vals = (double*)calloc(nnz + 1, sizeof(double));
    if (!md->col_ptr || !md->row_idx || !md->vals) {
        free(pairs);
        return ORC_ERR_INTERNAL;
    }
    for (size_t p = 0; p < nnz; ++p) {
        md->col_ptr[pairs[p].col + 1]++;
        md->row_idx[p] = pairs[p].row;
    }
    for (size_t c = 0; c < n_vars; ++c) md->col_ptr[c + 1] += md->col_ptr[c];
    free(pairs);
    /* precompute_symbolic_cholesky, solver.rs:289-300 */
    if (linsolve == ORC_LINSOLVE_DENSE) {
        md->A = (double*)malloc((n_vars * n_vars + 1) * sizeof(double));
        md->work = (double*)calloc(n_vars + 1, sizeof(double));
        if (!md->A || !md->work) return ORC_ERR_INTERNAL;
        return ORC_OK;
    }
    return model_symbolic_sparse(md);
}

/* solver.rs:318-356 */
static void model_residual(const Model* md, const double* x, double* out) {
    size_t row_num = 0;
    for (size_t i = 0; i < md->n_cs; ++i) {
        int degenerate = 0;
        double r[3] = {0.0, 0.0, 0.0};
        orc_residual(&md->cs[i], x, r, &degenerate);
        if (degenerate) warn_push(md->warnings, (int32_t)i, ORC_WARN_DEGENERATE);
        int dim = orc_residual_dim(&md->cs[i]);
        for (int k = 0; k < dim; ++k) out[row_num++] = md->cs[i].weight * r[k];
    }
}

/* solver.rs:359-440 */
static void model_refresh_jacobian(Model* md, const double* x) {
    memset(md->vals, 0, md->nnz * sizeof(double));
    size_t row_num = 0;
    for (size_t i = 0; i < md->n_cs; ++i) {
        int degenerate = 0;
        uint32_t ids0[8], ids1[8];
        double pd0[8], pd1[8];
        int n0, n1;
        orc_jacobian_rows(&md->cs[i], x, ids0, pd0, &n0, ids1, pd1, &n1, &degenerate);
        if (degenerate) warn_push(md->warnings, (int32_t)i, ORC_WARN_DEGENERATE);
        int dim = orc_residual_dim(&md->cs[i]);
        for (int rr = 0; rr < dim; ++rr) {
            size_t this_row = row_num++;
            const uint32_t* ids = rr == 0 ? ids0 : ids1;
            const double* pd = rr == 0 ? pd0 : pd1;
            int cnt = rr == 0 ? n0 : n1;
            for (int k = 0; k < cnt; ++k) {
                double weighted_partial = md->cs[i].weight * pd[k];
                size_t col = ids[k];
                /* linear slot search, solver.rs:412-418 */
                for (size_t p = md->col_ptr[col]; p < md->col_ptr[col + 1]; ++p) {
                    if (md->row_idx[p] == this_row) {
                        md->vals[p] += weighted_partial;
                        break;
                    }
                }
            }
        }
    }
}

/* (JtJ + lambda I) d = -Jt r, dense.  Returns 0 ok, 1 numeric failure (non-positive pivot). */
static int linsolve_dense(Model* md, const double* r, double lambda, double* d) {
    size_t n = md->n;
    double* A = md->A;
    memset(A, 0, n * n * sizeof(double));
    /* A(i,j) = sum_rows J(row,i) J(row,j); walk pairs of entries sharing a row via a row-bucket pass */
    /* gather by row: use jt arrays lazily built */
    if (!md->jt_ptr) {
        size_t m = md->m, nnz = md->nnz;
        md->jt_ptr = (size_t*)calloc(m + 2, sizeof(size_t));
        md->jt_col = (size_t*)malloc((nnz + 1) * sizeof(size_t));
        md->jt_slot = (size_t*)malloc((nnz + 1) * sizeof(size_t));
        for (size_t p = 0; p < nnz; ++p) md->jt_ptr[md->row_idx[p] + 1]++;
        for (size_t rr = 0; rr < m; ++rr) md->jt_ptr[rr + 1] += md->jt_ptr[rr];
        size_t* next = (size_t*)malloc((m + 1) * sizeof(size_t));
        memcpy(next, md->jt_ptr, (m + 1) * sizeof(size_t));
        for (size_t c = 0; c < n; ++c)
            for (size_t p = md->col_ptr[c]; p < md->col_ptr[c + 1]; ++p) {
                size_t rr = md->row_idx[p];
                md->jt_col[next[rr]] = c;
                md->jt_slot[next[rr]] = p;
                next[rr]++;
            }
        free(next);
    }
    for (size_t i = 0; i < n; ++i) d[i] = 0.0;
    for (size_t row = 0; row < md->m; ++row) {
        for (size_t p = md->jt_ptr[row]; p < md->jt_ptr[row + 1]; ++p) {
            size_t ci = md->jt_col[p];
            double vi = md->vals[md->jt_slot[p]];
            d[ci] += vi * -r[row]; /* b = Jt * (-r), newton.rs:84 */
            for (size_t q = md->jt_ptr[row]; q < md->jt_ptr[row + 1]; ++q) {
                size_t cj = md->jt_col[q];
                if (cj <= ci) A[ci * n + cj] += vi * md->vals[md->jt_slot[q]];
            }
        }
    }
    for (size_t i = 0; i < n; ++i) A[i * n + i] += lambda;
    /* Cholesky, lower, row-major in place */
    for (size_t j = 0; j < n; ++j) {
        double s = A[j * n + j];
        for (size_t k = 0; k < j; ++k) s -= A[j * n + k] * A[j * n + k];
        if (!(s > 0.0)) return 1;
        double ljj = sqrt(s);
        A[j * n + j] = ljj;
        for (size_t i = j + 1; i < n; ++i) {
            double t = A[i * n + j];
            for (size_t k = 0; k < j; ++k) t -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = t / ljj;
        }
    }
    /* forward */
    for (size_t i = 0; i < n; ++i) {
        double t = d[i];
        for (size_t k = 0; k < i; ++k) t -= A[i * n + k] * d[k];
        d[i] = t / A[i * n + i];
    }
    /* backward */
    for (size_t ii = n; ii-- > 0;) {
        double t = d[ii];
        for (size_t k = ii + 1; k < n; ++k) t -= A[k * n + ii] * d[k];
        d[ii] = t / A[ii * n + ii];
    }
    return 0;
}

/* sparse: numeric A, up-looking Cholesky, two triangular solves. */
static int linsolve_sparse(Model* md, const double* r, double lambda, double* d) {
    size_t n = md->n;
    /* numeric upper(A): for column k, scatter into work[] */
    double* w = md->work;
    for (size_t k = 0; k < n; ++k) {
        for (size_t p = md->a_ptr[k]; p < md->a_ptr[k + 1]; ++p) w[md->a_row[p]] = 0.0;
        double bk = 0.0;
        for (size_t p = md->col_ptr[k]; p < md->col_ptr[k + 1]; ++p) {
            size_t row = md->row_idx[p];
            double vk = md->vals[p];
            bk += vk * -r[row];
            for (size_t q = md->jt_ptr[row]; q < md->jt_ptr[row + 1]; ++q) {
                size_t i = md->jt_col[q];
                if (i <= k) w[i] += md->vals[md->jt_slot[q]] * vk;
            }
        }
        d[k] = bk;
        for (size_t p = md->a_ptr[k]; p < md->a_ptr[k + 1]; ++p) md->a_val[p] = w[md->a_row[p]];
        md->a_val[md->a_ptr[k + 1] - 1] += lambda; /* diagonal is last */
    }
    for (size_t i = 0; i < n; ++i) {
        w[i] = 0.0;
        md->flag[i] = NONE;
        md->l_fill[i] = md->l_ptr[i];
    }
    /* up-looking factorization: row k of L solves L(0:k-1,0:k-1) y = A(0:k-1,k) */
    for (size_t k = 0; k < n; ++k) {
        size_t top = ereach(md, k, k, md->stack, md->flag);
        double dk = 0.0;
        for (size_t p = md->a_ptr[k]; p < md->a_ptr[k + 1]; ++p) {
            size_t i = md->a_row[p];
            if (i < k)
                w[i] = md->a_val[p];
            else
                dk = md->a_val[p];
        }
        for (size_t t = top; t < n; ++t) {
            size_t i = md->stack[t];
            double lki = w[i] / md->l_val[md->l_ptr[i]];
            w[i] = 0.0;
            for (size_t p = md->l_ptr[i] + 1; p < md->l_fill[i]; ++p) w[md->l_row[p]] -= md->l_val[p] * lki;
            dk -= lki * lki;
            size_t p = md->l_fill[i]++;
            md->l_row[p] = k;
            md->l_val[p] = lki;
        }
        if (!(dk > 0.0)) return 1;
        size_t p = md->l_fill[k]++;
        md->l_row[p] = k;
        md->l_val[p] = sqrt(dk);
    }
    /* forward L y = b */
    for (size_t j = 0; j < n; ++j) {
        d[j] /= md->l_val[md->l_ptr[j]];
        for (size_t p = md->l_ptr[j] + 1; p < md->l_ptr[j + 1]; ++p) d[md->l_row[p]] -= md->l_val[p] * d[j];
    }
    /* backward Lt x = y */
    for (size_t jj = n; jj-- > 0;) {
        for (size_t p = md->l_ptr[jj] + 1; p < md->l_ptr[jj + 1]; ++p) d[jj] -= md->l_val[p] * d[md->l_row[p]];
        d[jj] /= md->l_val[md->l_ptr[jj]];
    }
    return 0;
}

/* newton.rs:29-145.  Returns ORC_OK / ORC_ERR_EMPTY_SYSTEM. */
static int solve_levenberg_marquardt(Model* md, double* x, const OrcConfig* cfg, uint64_t* iterations,
                                     int* converged, double* lambda_out, double* resid_inf_out) {
    size_t m = md->m, n = md->n;
    double* global_residual = (double*)calloc(m + 1, sizeof(double));
    double* next_residual = (double*)calloc(m + 1, sizeof(double));
    double* d = (double*)calloc(n + 1, sizeof(double));
    if (!global_residual || !next_residual || !d) return ORC_ERR_INTERNAL;
    double lambda = cfg->initial_lambda;
    /* eval, newton.rs:232-236 */
    model_residual(md, x, global_residual);
    model_refresh_jacobian(md, x);
    double residual_sq = 0.0;
    for (size_t i = 0; i < m; ++i) residual_sq += global_residual[i] * global_residual[i];
    int rc = ORC_OK;
    *iterations = cfg->max_iterations;
    *converged = 0;
    for (uint64_t this_iteration = 0; this_iteration < cfg->max_iterations; ++this_iteration) {
        if (m == 0) {
            rc = ORC_ERR_EMPTY_SYSTEM; /* newton.rs:54 */
            break;
        }
        double largest = fabs(global_residual[0]);
        for (size_t i = 1; i < m; ++i) largest = fmax(largest, fabs(global_residual[i]));
        if (largest <= cfg->residual_tolerance) {
            *iterations = this_iteration;
            *converged = 1;
            break;
        }
        int failed = (md->linsolve == ORC_LINSOLVE_DENSE) ? linsolve_dense(md, global_residual, lambda, d)
                                                          : linsolve_sparse(md, global_residual, lambda, d);
        if (failed) { /* newton.rs:96-99 */
            lambda *= LM_LAMBDA_INCR;
            continue;
        }
        double step_inf_norm = 0.0;
        if (n > 0) {
            step_inf_norm = fabs(d[0]);
            for (size_t i = 1; i < n; ++i) step_inf_norm = fmax(step_inf_norm, fabs(d[i]));
        }
        for (size_t i = 0; i < n; ++i) x[i] += d[i];
        model_residual(md, x, next_residual);
        double next_residual_sq = 0.0;
        for (size_t i = 0; i < m; ++i) next_residual_sq += next_residual[i] * next_residual[i];
        if (next_residual_sq < residual_sq) {
            double* t = global_residual;
            global_residual = next_residual;
            next_residual = t;
            model_refresh_jacobian(md, x);
            residual_sq = next_residual_sq;
            lambda *= LM_LAMBDA_DECR;
        } else {
            for (size_t i = 0; i < n; ++i) x[i] -= d[i];
            lambda *= LM_LAMBDA_INCR;
        }
        if (step_inf_norm <= cfg->step_tolerance) {
            *iterations = this_iteration;
            *converged = 1;
            break;
        }
    }
    *lambda_out = lambda;
    double largest = 0.0;
    if (m > 0) {
        largest = fabs(global_residual[0]);
        for (size_t i = 1; i < m; ++i) largest = fmax(largest, fabs(global_residual[i]));
    }
    *resid_inf_out = largest;
    free(global_residual);
    free(next_residual);
    free(d);
    return rc;
}


/* ---- FreedomAnalysis: solver/find_dof.rs ---------------------------------------------------------------------------
 * Column-pivoted Householder QR of the dense weighted Jacobian, rank from the diagonal of R (find_dof.rs:36-49),
 * null-space basis by back substitution (:51-70), un-permute (:72-77), orthonormalise (thin Q, :79), variables whose
 * squared row norm in that basis exceeds (1e-3 * max)^2 are underconstrained (:82-103).  faer 0.24.0's ColPivQr / Qr
 * are replaced by textbook Householder QR; the row norms are the diagonal of the orthogonal projector onto null(J)
 * and so do not depend on the basis either implementation picks. */
#define FREEDOM_TOLERANCE_BASE 1E-8

/* Householder vector for x[0..len): on return x[0] = beta (the R entry), v = (1, x[1..]) scaled, returns tau. */
static double householder(double* x, size_t len) {
    double norm = 0.0;
    for (size_t i = 0; i < len; ++i) norm = hypot(norm, x[i]);
    if (norm == 0.0) return 0.0;
    double alpha = x[0];
    double beta = alpha >= 0.0 ? -norm : norm;
    double denom = alpha - beta;
    for (size_t i = 1; i < len; ++i) x[i] /= denom;
    x[0] = beta;
    return (beta - alpha) / beta;
}

/* a[0..len) -= tau * v * (v . a), v = (1, vtail) */
static void apply_reflector(const double* vtail, double tau, double* a, size_t len) {
    if (tau == 0.0) return;
    double dot = a[0];
    for (size_t i = 1; i < len; ++i) dot += vtail[i] * a[i];
    dot *= tau;
    a[0] -= dot;
    for (size_t i = 1; i < len; ++i) a[i] -= dot * vtail[i];
}

int orc_freedom_analysis_dense(const double* jac_colmajor, size_t m, size_t n, uint32_t* under_out,
                               uint64_t* n_under_out, double* participation_out) {
    *n_under_out = 0;
    if (participation_out)
        for (size_t i = 0; i < n; ++i) participation_out[i] = 0.0;
    const size_t ndiag = m < n ? m : n;
    if (ndiag == 0) return ORC_ERR_EMPTY_SYSTEM; /* find_dof.rs:43-44: reduce over an empty diagonal */
    double* a = (double*)malloc(m * n * sizeof(double));
    size_t* perm = (size_t*)malloc(n * sizeof(size_t));
    double* tmp = (double*)malloc((m + n) * sizeof(double));
    memcpy(a, jac_colmajor, m * n * sizeof(double));
    for (size_t j = 0; j < n; ++j) perm[j] = j;
    for (size_t k = 0; k < ndiag; ++k) {
        /* pivot: remaining column of largest norm (first wins ties) */
        size_t best = k;
        double best_norm = -1.0;
        for (size_t j = k; j < n; ++j) {
            double nj = 0.0;
            for (size_t i = k; i < m; ++i) nj = hypot(nj, a[j * m + i]);
            if (nj > best_norm) {
                best_norm = nj;
                best = j;
            }
        }
        if (best != k) {
            memcpy(tmp, a + k * m, m * sizeof(double));
            memcpy(a + k * m, a + best * m, m * sizeof(double));
            memcpy(a + best * m, tmp, m * sizeof(double));
            size_t t = perm[k];
            perm[k] = perm[best];
            perm[best] = t;
        }
        double tau = householder(a + k * m + k, m - k);
        for (size_t j = k + 1; j < n; ++j) apply_reflector(a + k * m + k, tau, a + j * m + k, m - k);
    }
#define R_(i, j) a[(j) * m + (i)]
    double largest_diagonal = fabs(R_(0, 0));
    for (size_t i = 1; i < ndiag; ++i) largest_diagonal = fmax(largest_diagonal, fabs(R_(i, i)));
    const double tolerance = FREEDOM_TOLERANCE_BASE * largest_diagonal;
    size_t rank = 0;
    while (rank < ndiag && fabs(R_(rank, rank)) > tolerance) rank++;
    const size_t nullity = n - rank;
    free(tmp);
    if (nullity == 0) {
        free(a);
        free(perm);
        return ORC_OK;
    }
    /* permuted null-space basis, then rows back to variable order */
    double* ns = (double*)calloc(n * nullity, sizeof(double)); /* column-major n x nullity */
    double* z = (double*)calloc(n, sizeof(double));
    for (size_t fc = 0; fc < nullity; ++fc) {
        const size_t free_var = rank + fc;
        memset(z, 0, n * sizeof(double));
        z[free_var] = 1.0;
        for (size_t i = rank; i-- > 0;) {
            double rhs = free_var < n && i < m ? R_(i, free_var) : 0.0;
            for (size_t j = i + 1; j < rank; ++j) rhs += R_(i, j) * z[j];
            z[i] = -rhs / R_(i, i);
        }
        for (size_t i = 0; i < n; ++i) ns[fc * n + perm[i]] = z[i];
    }
#undef R_
    /* thin Q of ns: Householder QR, then Q = H_0 .. H_{k-1} [I; 0] */
    double* taus = (double*)calloc(nullity, sizeof(double));
    for (size_t k = 0; k < nullity; ++k) {
        taus[k] = householder(ns + k * n + k, n - k);
        for (size_t j = k + 1; j < nullity; ++j) apply_reflector(ns + k * n + k, taus[k], ns + j * n + k, n - k);
    }
    double* q = (double*)calloc(n * nullity, sizeof(double));
    for (size_t j = 0; j < nullity; ++j) {
        double* col = q + j * n;
        col[j] = 1.0;
        for (size_t k = nullity; k-- > 0;) apply_reflector(ns + k * n + k, taus[k], col + k, n - k);
    }
    double* participation = z;
    double max_participation = 0.0;
    for (size_t i = 0; i < n; ++i) {
        double sq = 0.0;
        for (size_t j = 0; j < nullity; ++j) sq += q[j * n + i] * q[j * n + i];
        participation[i] = sq;
        if (participation_out) participation_out[i] = sq;
        max_participation = fmax(max_participation, sq);
    }
    const double var_tol = 1e-3 * max_participation;
    const double squared_tol = var_tol * var_tol;
    uint64_t cnt = 0;
    for (size_t j = 0; j < n; ++j)
        if (participation[j] > squared_tol) under_out[cnt++] = (uint32_t)j;
    *n_under_out = cnt;
    free(a);
    free(perm);
    free(ns);
    free(z);
    free(taus);
    free(q);
    return ORC_OK;
}

/* Model::freedom_analysis, find_dof.rs:14-29: the Jacobian cache as the LM loop left it, densified. */
static int model_freedom_analysis(const Model* md, uint32_t* under_out, uint64_t* n_under_out) {
    const size_t m = md->m, n = md->n;
    double* dense = (double*)calloc(m * n + 1, sizeof(double));
    if (!dense) return ORC_ERR_INTERNAL;
    for (size_t c = 0; c < n; ++c)
        for (size_t p = md->col_ptr[c]; p < md->col_ptr[c + 1]; ++p) dense[c * m + md->row_idx[p]] = md->vals[p];
    int rc = orc_freedom_analysis_dense(dense, m, n, under_out, n_under_out, NULL);
    free(dense);
    return rc;
}

/* lib.rs:358-370 */
static int is_satisfied(int residual_dim, const double r[3]) {
    int sat0 = fabs(r[0]) < EPSILON;
    int sat1 = fabs(r[1]) < EPSILON;
    int sat2 = fabs(r[2]) < EPSILON;
    switch (residual_dim) {
    case 1:
        return sat0;
    case 2:
        return sat0 && sat1;
    default:
        return sat0 && sat1 && sat2;
    }
}

/* lib.rs:265-356 */
static int solve_inner_impl(const OrcConstraint* cs, const uint64_t* orig_ids, size_t n_cs, const uint32_t* var_ids,
                            const double* guesses, size_t n_guesses, const OrcConfig* cfg, int linsolve, double* x_out,
                            uint64_t* unsat_ids, OrcWarning* warn_buf, size_t warn_cap, OrcOutcome* out,
                            uint32_t* under_out, uint64_t* n_under_out) {
    memset(out, 0, sizeof(*out));
    out->num_vars = n_guesses;
    size_t num_eqs = 0;
    for (size_t i = 0; i < n_cs; ++i) num_eqs += (size_t)orc_residual_dim(&cs[i]);
    out->num_eqs = num_eqs;
    WarnSink sink = {warn_buf, warn_cap, 0};
    lint(cs, orig_ids, n_cs, &sink);
    Model md;
    int rc = model_new(&md, cs, orig_ids, n_cs, var_ids, n_guesses, n_guesses, linsolve, &sink, out);
    if (rc != ORC_OK) {
        out->error = rc;
        out->n_warnings = sink.count;
        model_free(&md);
        return rc;
    }
    double* values = (double*)malloc((n_guesses + 1) * sizeof(double));
    memcpy(values, guesses, n_guesses * sizeof(double));
    uint64_t iterations = 0;
    int converged = 0;
    rc = solve_levenberg_marquardt(&md, values, cfg, &iterations, &converged, &out->final_lambda,
                                   &out->final_residual_inf);
    out->n_warnings = sink.count;
    if (rc != ORC_OK) {
        out->error = rc;
        free(values);
        model_free(&md);
        return rc;
    }
    /* unsatisfied list, lib.rs:305-327 (unweighted residuals; degenerate flag discarded) */
    uint64_t n_unsat = 0;
    for (size_t i = 0; i < n_cs; ++i) {
        double r[3] = {0.0, 0.0, 0.0};
        int degenerate = 0;
        orc_residual(&cs[i], values, r, &degenerate);
        if (!is_satisfied(orc_residual_dim(&cs[i]), r)) {
            if (unsat_ids) unsat_ids[n_unsat] = orig_ids ? orig_ids[i] : i;
            n_unsat++;
        }
    }
    out->n_unsatisfied = n_unsat;
    if (under_out) { /* lib.rs:328-338: A::analyze(model), an error fails the tier */
        rc = model_freedom_analysis(&md, under_out, n_under_out);
        if (rc != ORC_OK) {
            out->error = rc;
            free(values);
            model_free(&md);
            return rc;
        }
    }
    uint32_t lowest_priority = 0; /* lib.rs:340-344: max priority in the subset */
    for (size_t i = 0; i < n_cs; ++i)
        if (cs[i].priority > lowest_priority) lowest_priority = cs[i].priority;
    out->priority_solved = lowest_priority;
    out->iterations = iterations;
    out->converged = converged;
    if (x_out) memcpy(x_out, values, n_guesses * sizeof(double));
    free(values);
    model_free(&md);
    return ORC_OK;
}

int orc_solve_inner(const OrcConstraint* cs, const uint64_t* orig_ids, size_t n_cs, const uint32_t* var_ids,
                    const double* guesses, size_t n_guesses, const OrcConfig* cfg, int linsolve, double* x_out,
                    uint64_t* unsat_ids, OrcWarning* warn_buf, size_t warn_cap, OrcOutcome* out) {
    return solve_inner_impl(cs, orig_ids, n_cs, var_ids, guesses, n_guesses, cfg, linsolve, x_out, unsat_ids,
                            warn_buf, warn_cap, out, NULL, NULL);
}

static int u32_cmp(const void* a, const void* b) {
    uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* lib.rs:148-263 */
static int solve_impl(const OrcConstraint* reqs_in, size_t n_reqs, const uint32_t* var_ids, const double* guesses,
                      size_t n_guesses, const OrcConfig* cfg, int linsolve, double* x_out, uint64_t* unsat_ids,
                      OrcWarning* warn_buf, size_t warn_cap, OrcOutcome* out, uint32_t* under_out,
                      uint64_t* n_under_out) {
    memset(out, 0, sizeof(*out));
    if (n_under_out) *n_under_out = 0; /* A::no_constraints(), lib.rs:157,250 */
    if (n_reqs == 0) { /* lib.rs:155-170 */
        if (x_out) memcpy(x_out, guesses, n_guesses * sizeof(double));
        out->converged = 1;
        out->num_vars = n_guesses;
        return ORC_OK;
    }
    /* initial_values[id] = guess, lib.rs:172-180 */
    size_t max_id = 0;
    for (size_t i = 0; i < n_guesses; ++i)
        if (var_ids[i] > max_id) max_id = var_ids[i];
    double* initial_values = (double*)calloc(max_id + 2, sizeof(double));
    for (size_t i = 0; i < n_guesses; ++i) initial_values[var_ids[i]] = guesses[i];
    OrcConstraint* reqs = (OrcConstraint*)malloc(n_reqs * sizeof(OrcConstraint));
    memcpy(reqs, reqs_in, n_reqs * sizeof(OrcConstraint));
    /* Reference indexes initial_values[id] unchecked (would panic on out-of-range ids); guard here. */
    for (size_t i = 0; i < n_reqs; ++i) {
        int ok = 1;
        if ((reqs[i].kind == ORC_LINE_TANGENT_TO_CIRCLE || reqs[i].kind == ORC_CIRCLE_TANGENT_TO_CIRCLE) &&
            reqs[i].tag == ORC_SIDE_UNDEFINED) {
            int cnt = reqs[i].kind == ORC_LINE_TANGENT_TO_CIRCLE ? 7 : 6;
            for (int k = 0; k < cnt; ++k)
                if (reqs[i].ids[k] > max_id || n_guesses == 0) ok = 0;
            if (ok) orc_set_from_initial_values(&reqs[i], initial_values);
        }
    }
    free(initial_values);
    /* distinct priorities ascending, lib.rs:199-203 */
    uint32_t* prios = (uint32_t*)malloc(n_reqs * sizeof(uint32_t));
    for (size_t i = 0; i < n_reqs; ++i) prios[i] = reqs[i].priority;
    qsort(prios, n_reqs, sizeof(uint32_t), u32_cmp);
    size_t n_prios = 0;
    for (size_t i = 0; i < n_reqs; ++i)
        if (i == 0 || prios[i] != prios[i - 1]) prios[n_prios++] = prios[i];

    OrcConstraint* subset = (OrcConstraint*)malloc(n_reqs * sizeof(OrcConstraint));
    uint64_t* subset_ids = (uint64_t*)malloc(n_reqs * sizeof(uint64_t));
    double* x_try = (double*)malloc((n_guesses + 1) * sizeof(double));
    uint64_t* unsat_try = (uint64_t*)malloc((n_reqs + 1) * sizeof(uint64_t));
    OrcWarning* warn_try = warn_cap ? (OrcWarning*)malloc(warn_cap * sizeof(OrcWarning)) : NULL;
    uint32_t* under_try = under_out ? (uint32_t*)malloc((n_guesses + 1) * sizeof(uint32_t)) : NULL;
    uint64_t n_under_try = 0;
    int have_res = 0;
    int rc = ORC_OK;
    for (size_t pi = 0; pi < n_prios; ++pi) {
        uint32_t curr_max_priority = prios[pi];
        size_t ns = 0;
        for (size_t i = 0; i < n_reqs; ++i) {
            if (reqs[i].priority <= curr_max_priority) {
                subset[ns] = reqs[i];
                subset_ids[ns] = i;
                ns++;
            }
        }
        OrcOutcome o;
        int r = solve_inner_impl(subset, subset_ids, ns, var_ids, guesses, n_guesses, cfg, linsolve, x_try, unsat_try,
                                 warn_try, warn_cap, &o, under_try, &n_under_try);
        if (r == ORC_OK) {
            if (o.n_unsatisfied > 0 && have_res) break; /* lib.rs:232-234: return previous res */
            /* adopt this outcome */
            *out = o;
            if (x_out) memcpy(x_out, x_try, n_guesses * sizeof(double));
            if (unsat_ids) memcpy(unsat_ids, unsat_try, o.n_unsatisfied * sizeof(uint64_t));
            if (under_out) {
                memcpy(under_out, under_try, n_under_try * sizeof(uint32_t));
                *n_under_out = n_under_try;
            }
            if (warn_buf && warn_cap) {
                size_t nw = o.n_warnings < warn_cap ? (size_t)o.n_warnings : warn_cap;
                memcpy(warn_buf, warn_try, nw * sizeof(OrcWarning));
            }
            have_res = 1;
            if (o.n_unsatisfied > 0) break;
        } else {
            if (!have_res) { /* lib.rs:239-244 */
                *out = o;
                if (warn_buf && warn_cap) {
                    size_t nw = o.n_warnings < warn_cap ? (size_t)o.n_warnings : warn_cap;
                    memcpy(warn_buf, warn_try, nw * sizeof(OrcWarning));
                }
                rc = r;
            }
            break;
        }
    }
    free(reqs);
    free(prios);
    free(subset);
    free(subset_ids);
    free(x_try);
    free(unsat_try);
    free(warn_try);
    free(under_try);
    return rc;
}

int orc_solve(const OrcConstraint* reqs, size_t n_reqs, const uint32_t* var_ids, const double* guesses,
              size_t n_guesses, const OrcConfig* cfg, int linsolve, double* x_out, uint64_t* unsat_ids,
              OrcWarning* warn_buf, size_t warn_cap, OrcOutcome* out) {
    return solve_impl(reqs, n_reqs, var_ids, guesses, n_guesses, cfg, linsolve, x_out, unsat_ids, warn_buf, warn_cap,
                      out, NULL, NULL);
}

/* lib.rs:134-146 solve_analysis (A = FreedomAnalysis) */
int orc_solve_analysis(const OrcConstraint* reqs, size_t n_reqs, const uint32_t* var_ids, const double* guesses,
                       size_t n_guesses, const OrcConfig* cfg, int linsolve, double* x_out, uint64_t* unsat_ids,
                       OrcWarning* warn_buf, size_t warn_cap, OrcOutcome* out, uint32_t* under_out,
                       uint64_t* n_under_out) {
    return solve_impl(reqs, n_reqs, var_ids, guesses, n_guesses, cfg, linsolve, x_out, unsat_ids, warn_buf, warn_cap,
                      out, under_out, n_under_out);
}

static double now_seconds(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double orc_time_solves(const OrcConstraint* reqs, size_t n_reqs, const uint32_t* var_ids, const double* guesses,
                       size_t n_guesses, const OrcConfig* cfg, int linsolve, int repeats, uint64_t* iterations_out) {
    double* x = (double*)malloc((n_guesses + 1) * sizeof(double));
    uint64_t* unsat = (uint64_t*)malloc((n_reqs + 1) * sizeof(uint64_t));
    OrcOutcome o;
    double t0 = now_seconds();
    for (int i = 0; i < repeats; ++i) {
        orc_solve(reqs, n_reqs, var_ids, guesses, n_guesses, cfg, linsolve, x, unsat, NULL, 0, &o);
    }
    double t1 = now_seconds();
    if (iterations_out) *iterations_out = o.iterations;
    free(x);
    free(unsat);
    return t1 - t0;
}

/* The same loop over solve_analysis (lib.rs:134-146): what the reference's `*_analysis` benchmarks time
 * (ezpz/benches/solver_bench.rs:27-41). */
double orc_time_solves_analysis(const OrcConstraint* reqs, size_t n_reqs, const uint32_t* var_ids, const double* guesses,
                                size_t n_guesses, const OrcConfig* cfg, int linsolve, int repeats, uint64_t* iterations_out) {
    double* x = (double*)malloc((n_guesses + 1) * sizeof(double));
    uint64_t* unsat = (uint64_t*)malloc((n_reqs + 1) * sizeof(uint64_t));
    uint32_t* under = (uint32_t*)malloc((n_guesses + 1) * sizeof(uint32_t));
    uint64_t n_under = 0;
    OrcOutcome o;
    double t0 = now_seconds();
    for (int i = 0; i < repeats; ++i) {
        orc_solve_analysis(reqs, n_reqs, var_ids, guesses, n_guesses, cfg, linsolve, x, unsat, NULL, 0, &o, under, &n_under);
    }
    double t1 = now_seconds();
    if (iterations_out) *iterations_out = o.iterations;
    free(x);
    free(unsat);
    free(under);
    return t1 - t0;
}

int orc_solve_batch(const OrcConstraint* reqs, size_t n_reqs, size_t n_vars, const double* guesses, size_t batch,
                    const OrcConfig* cfg, int linsolve, int nthreads, double* x_out, uint32_t* iterations,
                    uint8_t* converged, uint32_t* n_unsatisfied) {
    uint32_t* var_ids = (uint32_t*)malloc((n_vars + 1) * sizeof(uint32_t));
    for (size_t i = 0; i < n_vars; ++i) var_ids[i] = (uint32_t)i;
    int bad = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 64)
    for (long long s = 0; s < (long long)batch; ++s) {
        OrcOutcome o;
        uint64_t* unsat = (uint64_t*)malloc((n_reqs + 1) * sizeof(uint64_t));
        int rc = orc_solve(reqs, n_reqs, var_ids, guesses + (size_t)s * n_vars, n_vars, cfg, linsolve,
                           x_out + (size_t)s * n_vars, unsat, NULL, 0, &o);
        if (rc != ORC_OK) {
#pragma omp atomic write
            bad = 1;
        }
        if (iterations) iterations[s] = (uint32_t)o.iterations;
        if (converged) converged[s] = (uint8_t)o.converged;
        if (n_unsatisfied) n_unsatisfied[s] = (uint32_t)o.n_unsatisfied;
        free(unsat);
    }
    free(var_ids);
    return bad ? ORC_ERR_INTERNAL : ORC_OK;
}
