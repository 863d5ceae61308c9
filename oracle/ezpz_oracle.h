/*
 * ezpz_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the KittyCAD/ezpz Newton / Levenberg-Marquardt
 * constraint-solve path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (ezpz_amd/) never does.
 *
 * Parity status: PINNED against the reference's own known-answer tests
 * (tests/cases.py, transcribed from ezpz/src/tests.rs with file:line and run
 * on this library by tests/test_oracle_pins.py) -- the reference itself is Rust and cannot be built in this image
 * (no cargo/rustc; faer 0.24.0 / libm 0.2.16 are un-vendored crates.io deps).
 * faer's sparse LLT is restated here as a textbook up-looking sparse Cholesky
 * (and a dense Cholesky); equality with faer is up to rounding only.
 *
 * All file:line citations are relative to /root/reference/.
 */
#ifndef EZPZ_ORACLE_H
#define EZPZ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Constraint kind tags, numbered by the enum order of ezpz/src/constraints.rs:37-93. */
enum {
    ORC_LINE_TANGENT_TO_CIRCLE = 0,
    ORC_CIRCLE_TANGENT_TO_CIRCLE = 1,
    ORC_DISTANCE = 2,
    ORC_DISTANCE_VAR = 3,
    ORC_VERTICAL_DISTANCE = 4,
    ORC_HORIZONTAL_DISTANCE = 5,
    ORC_VERTICAL = 6,
    ORC_HORIZONTAL = 7,
    ORC_LINES_AT_ANGLE = 8,
    ORC_FIXED = 9,
    ORC_SCALAR_EQUAL = 10,
    ORC_POINTS_COINCIDENT = 11,
    ORC_CIRCLE_RADIUS = 12,
    ORC_LINES_EQUAL_LENGTH = 13,
    ORC_ARC_RADIUS = 14,
    ORC_ARC = 15,
    ORC_MIDPOINT = 16,
    ORC_POINT_LINE_DISTANCE = 17,
    ORC_VERTICAL_POINT_LINE_DISTANCE = 18,
    ORC_HORIZONTAL_POINT_LINE_DISTANCE = 19,
    ORC_SYMMETRIC = 20,
    ORC_POINT_ARC_COINCIDENT = 21,
    ORC_ARC_LENGTH = 22,
    ORC_ARC_ANGLE = 23,
    ORC_POINTS_AT_ANGLE = 24,
    ORC_NUM_KINDS = 25
};

/* tag values. LineSide constraints.rs:109-116, CircleSide :122-129, AngleKind datatypes.rs:9-16 */
enum { ORC_SIDE_UNDEFINED = 0, ORC_LINE_LEFT = 1, ORC_LINE_RIGHT = 2 };
enum { ORC_CIRCLE_EXTERIOR = 1, ORC_CIRCLE_INTERIOR = 2 };
enum { ORC_ANGLE_PARALLEL = 0, ORC_ANGLE_PERPENDICULAR = 1, ORC_ANGLE_OTHER_DEG = 2, ORC_ANGLE_OTHER_RAD = 3 };

/*
 * Flat POD form of ezpz::ConstraintRequest (constraint_request.rs) /
 * Constraint (constraints.rs:37-93).  56 bytes.  `ids` hold the variable ids of
 * the datum fields in struct declaration order:
 *   point            -> x, y
 *   line segment     -> p0.x, p0.y, p1.x, p1.y
 *   circle           -> center.x, center.y, radius
 *   circular arc     -> center.x, center.y, start.x, start.y, end.x, end.y
 * followed by further datums in the order of the variant's fields.
 */
typedef struct {
    uint16_t kind;
    uint8_t tag;
    uint8_t flags;
    uint32_t priority;
    uint32_t ids[8];
    double param;
    double weight;
} OrcConstraint;

/* ezpz::Config, solver.rs:31-81 */
typedef struct {
    uint64_t max_iterations;
    double residual_tolerance;
    double step_tolerance;
    double initial_lambda;
} OrcConfig;

/* error codes mirroring NonLinearSystemError, error.rs:35-86 */
enum {
    ORC_OK = 0,
    ORC_ERR_WRONG_NUMBER_GUESSES = -2,
    ORC_ERR_MISSING_GUESS = -3,
    ORC_ERR_EMPTY_SYSTEM = -8,
    ORC_ERR_INTERNAL = -100
};

/* warning content, warnings.rs:22-32 */
enum { ORC_WARN_DEGENERATE = 0, ORC_WARN_SHOULD_BE_PARALLEL = 1, ORC_WARN_SHOULD_BE_PERPENDICULAR = 2 };

typedef struct {
    int32_t about_constraint;
    int32_t content;
} OrcWarning;

typedef struct {
    int32_t error;             /* ORC_OK or negative */
    int32_t err_constraint_id; /* MissingGuess.constraint_id */
    int64_t err_variable;      /* MissingGuess.variable */
    uint64_t iterations;
    int32_t converged;
    uint32_t priority_solved;
    uint64_t n_unsatisfied;
    uint64_t n_warnings; /* total produced (may exceed capacity) */
    uint64_t num_vars;
    uint64_t num_eqs;
    double final_lambda;
    double final_residual_inf; /* max |weighted r| at exit (extra, for tests) */
} OrcOutcome;

enum { ORC_LINSOLVE_DENSE = 0, ORC_LINSOLVE_SPARSE = 1 };

int orc_residual_dim(const OrcConstraint* c);
/* ids of each Jacobian row as `Constraint::nonzeroes` emits them. Returns residual_dim. */
int orc_nonzeroes(const OrcConstraint* c, uint32_t row0[8], int* n0, uint32_t row1[8], int* n1);
/* Constraint::residual: unweighted residuals. */
void orc_residual(const OrcConstraint* c, const double* x, double r[3], int* degenerate);
/* Constraint::jacobian_rows */
void orc_jacobian_rows(const OrcConstraint* c, const double* x, uint32_t ids0[8], double pd0[8], int* n0,
                       uint32_t ids1[8], double pd1[8], int* n1, int* degenerate);
/* Constraint::set_from_initial_values */
void orc_set_from_initial_values(OrcConstraint* c, const double* initial_values);

/*
 * lib.rs:265-356 solve_inner: one priority tier.  `cs` already side-resolved.
 * `orig_ids[i]` = ConstraintEntry.id (index into the original request list), may be NULL (= i).
 * unsat_ids receives up to n_cs original ids.  warn_buf receives up to warn_cap warnings.
 */
int orc_solve_inner(const OrcConstraint* cs, const uint64_t* orig_ids, size_t n_cs, const uint32_t* var_ids,
                    const double* guesses, size_t n_guesses, const OrcConfig* cfg, int linsolve, double* x_out,
                    uint64_t* unsat_ids, OrcWarning* warn_buf, size_t warn_cap, OrcOutcome* out);

/* lib.rs:148-263 solve_with_priority_inner (A = NoAnalysis). */
int orc_solve(const OrcConstraint* reqs, size_t n_reqs, const uint32_t* var_ids, const double* guesses,
              size_t n_guesses, const OrcConfig* cfg, int linsolve, double* x_out, uint64_t* unsat_ids,
              OrcWarning* warn_buf, size_t warn_cap, OrcOutcome* out);

/* lib.rs:134-146 solve_analysis: orc_solve plus FreedomAnalysis (solver/find_dof.rs) of the tier that is returned.
 * under_out receives up to n_guesses variable indices (ascending). */
int orc_solve_analysis(const OrcConstraint* reqs, size_t n_reqs, const uint32_t* var_ids, const double* guesses,
                       size_t n_guesses, const OrcConfig* cfg, int linsolve, double* x_out, uint64_t* unsat_ids,
                       OrcWarning* warn_buf, size_t warn_cap, OrcOutcome* out, uint32_t* under_out,
                       uint64_t* n_under_out);
/* find_dof.rs:31-103 on a dense column-major m x n Jacobian; participation_out (optional, n) = squared row norms of
 * the orthonormal null-space basis. */
int orc_freedom_analysis_dense(const double* jac_colmajor, size_t m, size_t n, uint32_t* under_out,
                               uint64_t* n_under_out, double* participation_out);

/* CLI timing protocol (ezpz-cli/src/main.rs:86-100): `repeats` back-to-back full solves; returns seconds. */
double orc_time_solves(const OrcConstraint* reqs, size_t n_reqs, const uint32_t* var_ids, const double* guesses,
                       size_t n_guesses, const OrcConfig* cfg, int linsolve, int repeats, uint64_t* iterations_out);
double orc_time_solves_analysis(const OrcConstraint* reqs, size_t n_reqs, const uint32_t* var_ids, const double* guesses,
                       size_t n_guesses, const OrcConfig* cfg, int linsolve, int repeats, uint64_t* iterations_out);

/* Batch of independent systems sharing one request list; guesses AoS [batch][n]. OpenMP over systems. */
int orc_solve_batch(const OrcConstraint* reqs, size_t n_reqs, size_t n_vars, const double* guesses, size_t batch,
                    const OrcConfig* cfg, int linsolve, int nthreads, double* x_out, uint32_t* iterations,
                    uint8_t* converged, uint32_t* n_unsatisfied);

void orc_default_config(OrcConfig* cfg);

#ifdef __cplusplus
}
#endif
#endif
