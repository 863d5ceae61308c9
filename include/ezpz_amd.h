/*
 * ezpz_amd.h -- C ABI of the MI355X-native ezpz constraint-solve path.
 *
 * This is the drop-in boundary for ONE path of KittyCAD/ezpz: the Newton / Levenberg-Marquardt
 * constraint-solve loop.  The reference has no FFI seam of its own (it is a pure-Rust crate); the
 * seam is the public function `ezpz::solve` and, inside it, `solve_inner` -> `Model::new` +
 * `Model::solve_levenberg_marquardt`.  Each entry point below names the reference interface it
 * replaces (paths relative to the reference checkout).  A Rust shim binding these symbols is
 * shown in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes, caller owns every buffer, nothing throws, return value is
 * 0 (EZPZ_OK) or a negative EzpzError that mirrors `NonLinearSystemError` (ezpz/src/error.rs:35-86).
 * All arithmetic is IEEE fp64.  The library needs a HIP device (gfx950); without one every solve
 * entry point returns EZPZ_ERR_NO_DEVICE -- there is no CPU fallback.
 */
#ifndef EZPZ_AMD_H
#define EZPZ_AMD_H

#ifndef __HIPCC_RTC__
#include <stddef.h>
#include <stdint.h>
#else /* run-time compilation of the device kernels (hiprtc has no system headers) */
typedef unsigned char uint8_t;
typedef unsigned short uint16_t;
typedef unsigned int uint32_t;
typedef unsigned long long uint64_t;
typedef int int32_t;
typedef long long int64_t;
typedef unsigned long long uintptr_t;
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Constraint, ezpz/src/constraints.rs:37-93 (kind = enum declaration order) ------------------ */
enum EzpzKind {
    EZPZ_LINE_TANGENT_TO_CIRCLE = 0,          /* ids: line p0.x p0.y p1.x p1.y, circle c.x c.y radius; tag = EzpzLineSide */
    EZPZ_CIRCLE_TANGENT_TO_CIRCLE = 1,        /* ids: a.cx a.cy a.r b.cx b.cy b.r; tag = EzpzCircleSide */
    EZPZ_DISTANCE = 2,                        /* ids: p0 p1; param = distance */
    EZPZ_DISTANCE_VAR = 3,                    /* ids: p q D */
    EZPZ_VERTICAL_DISTANCE = 4,               /* ids: p0 p1; param */
    EZPZ_HORIZONTAL_DISTANCE = 5,             /* ids: p0 p1; param */
    EZPZ_VERTICAL = 6,                        /* ids: line */
    EZPZ_HORIZONTAL = 7,                      /* ids: line */
    EZPZ_LINES_AT_ANGLE = 8,                  /* ids: line0 line1; tag = EzpzAngleKind; param = angle value */
    EZPZ_FIXED = 9,                           /* ids: var; param = value */
    EZPZ_SCALAR_EQUAL = 10,                   /* ids: a b */
    EZPZ_POINTS_COINCIDENT = 11,              /* ids: p0 p1 */
    EZPZ_CIRCLE_RADIUS = 12,                  /* ids: c.x c.y radius; param */
    EZPZ_LINES_EQUAL_LENGTH = 13,             /* ids: line0 line1 */
    EZPZ_ARC_RADIUS = 14,                     /* ids: arc; param */
    EZPZ_ARC = 15,                            /* ids: arc */
    EZPZ_MIDPOINT = 16,                       /* ids: line, point */
    EZPZ_POINT_LINE_DISTANCE = 17,            /* ids: point, line; param */
    EZPZ_VERTICAL_POINT_LINE_DISTANCE = 18,   /* ids: point, line; param */
    EZPZ_HORIZONTAL_POINT_LINE_DISTANCE = 19, /* ids: point, line; param */
    EZPZ_SYMMETRIC = 20,                      /* ids: line, a, b */
    EZPZ_POINT_ARC_COINCIDENT = 21,           /* ids: arc, point */
    EZPZ_ARC_LENGTH = 22,                     /* ids: arc; param */
    EZPZ_ARC_ANGLE = 23,                      /* ids: arc; tag = EZPZ_ANGLE_OTHER_DEG/RAD; param */
    EZPZ_POINTS_AT_ANGLE = 24,                /* ids: p0 p1 p2; tag = EzpzAngleKind; param */
    EZPZ_NUM_KINDS = 25
};
/* datum id order: point = x,y; line = p0,p1; circle = center,radius; arc = center,start,end
 * (struct field order of ezpz/src/datatypes/inputs.rs). */

enum EzpzLineSide { EZPZ_SIDE_UNDEFINED = 0, EZPZ_LINE_LEFT = 1, EZPZ_LINE_RIGHT = 2 };   /* constraints.rs:109-116 */
enum EzpzCircleSide { EZPZ_CIRCLE_EXTERIOR = 1, EZPZ_CIRCLE_INTERIOR = 2 };             /* constraints.rs:122-129 */
enum EzpzAngleKind {                                                                     /* datatypes.rs:9-29 */
    EZPZ_ANGLE_PARALLEL = 0,
    EZPZ_ANGLE_PERPENDICULAR = 1,
    EZPZ_ANGLE_OTHER_DEG = 2,
    EZPZ_ANGLE_OTHER_RAD = 3
};

/* ConstraintRequest {constraint, priority, weight}, ezpz/src/constraint_request.rs.  56 bytes. */
typedef struct EzpzConstraint {
    uint16_t kind; /* EzpzKind */
    uint8_t tag;
    uint8_t flags; /* reserved, 0 */
    uint32_t priority;
    uint32_t ids[8];
    double param;
    double weight;
} EzpzConstraint;

/* Config, ezpz/src/solver.rs:31-81 */
typedef struct EzpzConfig {
    uint64_t max_iterations;   /* default 35 */
    double residual_tolerance; /* default 1e-8 */
    double step_tolerance;     /* default 1e-12 */
    double initial_lambda;     /* default 1e-9 */
} EzpzConfig;

/* NonLinearSystemError, ezpz/src/error.rs:35-86 (+ library conditions <= -100) */
enum EzpzError {
    EZPZ_OK = 0,
    EZPZ_ERR_NOT_FOUND = -1,
    EZPZ_ERR_WRONG_NUMBER_GUESSES = -2,
    EZPZ_ERR_MISSING_GUESS = -3,
    EZPZ_ERR_MATRIX = -4, /* FaerMatrix: a variable id outside [0, n_vars) */
    EZPZ_ERR_EMPTY_SYSTEM = -8,
    EZPZ_ERR_NO_DEVICE = -100,
    EZPZ_ERR_HIP = -101,
    EZPZ_ERR_TOO_LARGE = -102,
    EZPZ_ERR_INVALID_ARGUMENT = -103,
    EZPZ_ERR_KERNEL_BUDGET = -104,   /* ezpz_system_specialize: the process already holds its 512 specialised kernels */
    EZPZ_ERR_PARSE = -110,           /* textual front end: winnow parse failure */
    EZPZ_ERR_TEXT_MISSING_GUESS = -111,  /* TextualError, error.rs:10-33 */
    EZPZ_ERR_TEXT_UNUSED_GUESSES = -112,
    EZPZ_ERR_TEXT_UNDEFINED_POINT = -113
};

/* Warning / WarningContent, ezpz/src/warnings.rs:8-32 */
enum EzpzWarningContent { EZPZ_WARN_DEGENERATE = 0, EZPZ_WARN_SHOULD_BE_PARALLEL = 1, EZPZ_WARN_SHOULD_BE_PERPENDICULAR = 2 };
typedef struct EzpzWarning {
    int32_t about_constraint;
    int32_t content;
} EzpzWarning;

/* Per-system result of the device LM loop: SuccessfulSolve (solver/newton.rs:18-24) plus what
 * solve_inner derives (lib.rs:305-327).  32 bytes. */
#define EZPZ_ITERATIONS_TEAM_TIMEOUT 0xFFFFFFFFu /* EzpzStatus.iterations: see ezpz_system_solve_batch_device */
typedef struct EzpzStatus {
    uint32_t iterations;
    uint32_t converged;
    uint32_t n_unsatisfied;
    uint32_t n_warnings;       /* Degenerate warnings the model produced (every evaluation, no dedup) */
    double final_residual_inf; /* max |weighted r| at exit */
    double final_lambda;
} EzpzStatus;

/* SolveOutcome (solve_outcome.rs:12-26) / FailureOutcome (:126-136) in flat form. */
typedef struct EzpzOutcome {
    int32_t error; /* EzpzError; != 0 => FailureOutcome */
    int32_t err_constraint_id;
    int64_t err_variable;
    uint64_t iterations;
    int32_t converged;
    uint32_t priority_solved;
    uint64_t n_unsatisfied;
    uint64_t n_warnings;
    uint64_t num_vars;
    uint64_t num_eqs;
    double final_lambda;
    double final_residual_inf;
} EzpzOutcome;

/* Sizes of one analysed topology (host symbolic phase), for roofline accounting. */
typedef struct EzpzSystemInfo {
    uint64_t n_constraints, n_vars, n_rows;
    uint64_t nnz_j;  /* zJ */
    uint64_t nnz_a;  /* zA: nnz(lower(JtJ + lambda I)) */
    uint64_t nnz_l;  /* zL: nnz(L) incl. diagonal */
    uint64_t n_levels;      /* elimination-tree height */
    uint64_t n_components;  /* connected components of the variable graph */
    uint64_t program_bytes; /* device-resident topology program */
    uint64_t workspace_bytes; /* per-system LDS / global workspace */
    uint32_t team_size;     /* lanes cooperating on one system */
    uint32_t workspace_in_lds;
    uint32_t team_mode;     /* 0 sub-wavefront teams, 1 wavefront-partitioned workgroup, 2 barrier workgroup,
                             * 3 component-resident (one lane per connected component; n_partitions = chunks of <= 64
                             * components of one isomorphism class), 4 barrier workgroup whose linear solve is a record
                             * walk (one connected sketch, or a system of up to 127 components: the automatic shapes),
                             * 5 frontal: a tree of dense fronts, one wavefront per front (grid_workgroups = workgroups that
                             * share one system, n_partitions = fronts, n_levels = levels of the tree per workgroup) */
    uint32_t n_partitions;  /* partitions (balanced unions of components), one per wavefront in mode 1 */
    uint32_t program_in_lds;
    uint32_t grid_workgroups; /* workgroups that share one system (grid team: one large system on many CUs), else 1 */
    /* A system created for batches (team_size 0) of one connected sketch also carries the frontal plan (team_mode 5's) and
     * takes it for calls of at most front_max_batch systems -- calls too small to fill the device with one workgroup per
     * system; the fields above then describe the shape of its larger calls.  front_workgroups = workgroups that share one
     * system on that plan (0: no such plan); 0xFFFFFFFF in front_max_batch: every call (the latency shapes). */
    uint32_t front_workgroups, front_max_batch;
} EzpzSystemInfo;

typedef struct EzpzSystem EzpzSystem; /* opaque: one analysed topology, resident on one device */

void ezpz_default_config(EzpzConfig* cfg); /* Config::default(), solver.rs:72-81 */
int ezpz_device_count(void);
/* The calling thread's current HIP device, or -1 without one.  ezpz_solve* keep their cached systems on it. */
int ezpz_current_device(void);
const char* ezpz_error_string(int err);

/* ---- symbolic phase ------------------------------------------------------------------------------
 * Replaces Model::new (ezpz/src/solver.rs:192-300): validate_variables (:142-189), row numbering,
 * Jacobian sparsity, and the symbolic Cholesky of JtJ + lambda I that faer's SymbolicLlt does there.
 * `cs` is one priority tier, already side-resolved, in request order; ids index the value vector
 * directly (Layout::index_of, solver.rs:107-109).  On MissingGuess, err_constraint/err_variable
 * receive the offending position in `cs` and the id.  `team_size` 0 = choose automatically for batch
 * throughput; EZPZ_TEAM_AUTO_LATENCY = choose automatically for the latency of one solve (what ezpz_solve does:
 * a connected sketch of a few hundred variables then runs on a 256-512 lane workgroup with its lists staged in LDS
 * instead of one wavefront / a lean 128-lane workgroup: ~35 % sooner per solve at less than half the batch rate). */
#define EZPZ_TEAM_AUTO_LATENCY 0xFFFFFFFFu
/* automatic choice among the list-walk shapes only (team_mode 0-2, dense phases included): never the component-resident
 * shape, one lane per system or the record walk (A/B runs and tests) */
#define EZPZ_TEAM_AUTO_LISTS 0xFFFFFFFEu
/* automatic, and a connected sketch of more than 20 variables runs one lane per system (lanes across the batch) at every
 * batch size instead of from 64 x 2 x CUs systems per call (A/B runs and tests) */
#define EZPZ_TEAM_BATCH_LANES 0xFFFFFFFDu
/* EZPZ_TEAM_AUTO_LATENCY without the record walk (team_mode 4): one connected sketch then ends its elimination with
 * dense phases on the barrier workgroup (team_mode 2), as every latency shape did before round 3 (A/B runs and tests) */
#define EZPZ_TEAM_LATENCY_PHASES 0xFFFFFFFCu
/* EZPZ_TEAM_AUTO_LATENCY, and a small system (<= 20 variables) runs one solve on one WAVEFRONT per system -- constraint
 * sweeps and the assembly of the normal equations across its lanes, the lane kernel's arithmetic operation for operation
 * -- whenever that form exists; EZPZ_TEAM_AUTO_LATENCY itself takes it only where it pays (>= 8 variables and >= 3
 * constraints of the non-linear kinds: `square` 47 -> 40 us of kernel, `two rectangles dependent` 63 -> 53; a 4-variable
 * linear system loses 12 -> 16) (A/B runs and tests) */
#define EZPZ_TEAM_LATENCY_WAVE 0xFFFFFFFBu
/* the FRONTAL shape (team_mode 5: multifrontal supernodal Cholesky on dense fronts, csrc/fronts.cpp) whatever the size of the
 * system, with the automatic latency shape behind it where the shape does not apply (a front of more than 63 rows ...);
 * EZPZ_TEAM_AUTO_LATENCY takes it from a size on its own (EzpzLaunchPolicy.front_min_vars_one_solve) */
#define EZPZ_TEAM_FRONTS 0xFFFFFFFAu
/* EZPZ_TEAM_AUTO_LATENCY without the frontal shape: one connected sketch then walks records (team_mode 4), as every latency
 * shape did before round 5 (A/B runs and tests) */
#define EZPZ_TEAM_LATENCY_RECORDS 0xFFFFFFF9u
int ezpz_system_create(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, int device, uint32_t team_size,
                       EzpzSystem** out, int32_t* err_constraint, int64_t* err_variable);
void ezpz_system_destroy(EzpzSystem* sys);
int ezpz_system_info(const EzpzSystem* sys, EzpzSystemInfo* info);

/* Host-only analysis (no device needed): the same symbolic phase, sizes only. */
int ezpz_analyze(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, EzpzSystemInfo* info, int32_t* err_constraint,
                 int64_t* err_variable);

/* ---- evaluation only (kernel K1 of the design: residual + Jacobian sweep) -----------------------------
 * Replaces Model::residual + Model::refresh_jacobian (ezpz/src/solver.rs:318-440) for `batch` value
 * vectors: r_out [batch][n_rows] weighted residuals, jv_out [batch][nnz_j] weighted Jacobian values in
 * slot order; ezpz_system_jacobian_pattern gives the (row, col) of every slot.  Host pointers. */
int ezpz_system_eval_batch(EzpzSystem* sys, const double* x, size_t batch, double* r_out, double* jv_out,
                           uint32_t* degenerate_count_out);
int ezpz_system_jacobian_pattern(const EzpzSystem* sys, uint32_t* rows, uint32_t* cols);

/* ---- numeric phase -------------------------------------------------------------------------------
 * Replaces Model::solve_levenberg_marquardt (ezpz/src/solver/newton.rs:29-145) followed by the
 * unsatisfied check of solve_inner (ezpz/src/lib.rs:305-327), for `batch` independent systems that
 * share the analysed topology.  x0 / x_out are AoS [batch][n_vars] and may alias -- but a block system of linear constraints
 * with unit weights (massive_parallel_system) is solved ~20 % faster when they do not overlap: its kernel stores a system's
 * values before the verdicts of the LM control are in and re-solves the few systems whose verdicts are not the expected ones
 * from x0 (csrc/jit_kernel.hip.hpp: solve_kernel_fast); with overlapping buffers the loop kernel serves, same results.
 * unsat_mask is optional ([batch][n_cs] bytes, 1 = unsatisfied, indexed by position in `cs`); warn_log is optional
 * ([batch][warn_cap] entries (pass << 32 | position), chronological once sorted).
 * Which kernel serves a call may depend on the call's size (EzpzSystemInfo.front_max_batch: small calls of a connected
 * sketch take the frontal elimination order, larger ones the record walk or the lanes): integer and flag outputs are the
 * reference's either way, coordinates of connected sketches agree to rounding (1e-6 relative on determined coordinates), not
 * bit for bit across call sizes -- a caller who chunks a batch itself and needs identical bits per chunk size creates the system
 * with one shape (team_size: EZPZ_TEAM_*).  Block systems are bitwise the same on every path.
 * The _device form takes device pointers and only enqueues on `stream` (a hipStream_t; NULL = default);
 * the host form copies in, runs, copies out and synchronises.  Launches on one EzpzSystem must not overlap in time
 * when the system uses per-system device scratch (a global workspace or a grid team, see EzpzSystemInfo): enqueue
 * them on one stream, or create one EzpzSystem per stream.
 * Thread safety: the _device form may be called on one EzpzSystem from several threads at once (each enqueueing on its
 * own stream; what a launch creates on first use is created under a lock); the host form serialises its callers per
 * EzpzSystem; ezpz_solve* may be called from any number of threads.
 * Grid teams (EzpzSystemInfo.grid_workgroups > 1) need all their workgroups resident at once; launches are sized to
 * the device's CU count, and should a rendezvous still time out (~1 s: another process occupying the device) the
 * affected systems report iterations == EZPZ_ITERATIONS_TEAM_TIMEOUT, converged == 0 instead of hanging; the host
 * form returns EZPZ_ERR_HIP in that case and the EzpzSystem must be destroyed. */
/* Host buffers a caller reuses across batch calls can be page-locked once (hipHostRegister underneath): when both x0
 * and x_out of an ezpz_system_solve_batch call lie inside registered ranges (and no mask / warning log is asked for) the
 * call streams the batch through three device slots, so that the PCIe copy in, the kernels and the copy out overlap.
 * The caller must unregister a range before freeing it. */
int ezpz_host_register(void* p, size_t bytes);
int ezpz_host_unregister(void* p);

int ezpz_system_solve_batch_device(EzpzSystem* sys, const double* x0_dev, size_t batch, const EzpzConfig* cfg,
                                   double* x_out_dev, EzpzStatus* status_dev, uint8_t* unsat_mask_dev,
                                   uint64_t* warn_log_dev, uint32_t warn_cap, void* stream);
int ezpz_system_solve_batch(EzpzSystem* sys, const double* x0, size_t batch, const EzpzConfig* cfg, double* x_out,
                            EzpzStatus* status, uint8_t* unsat_mask, uint64_t* warn_log, uint32_t warn_cap);

/* ---- solve_inner, ezpz/src/lib.rs:265-356: one tier, one system -----------------------------------
 * lint (warnings.rs:34-60) + Model::new + LM + unsatisfied list.  orig_ids (may be NULL) are the
 * ConstraintEntry.id values reported in `unsat_ids` and lint warnings. */
int ezpz_solve_inner(const EzpzConstraint* cs, const uint64_t* orig_ids, size_t n_cs, const uint32_t* var_ids,
                     const double* guesses, size_t n_guesses, const EzpzConfig* cfg, double* x_out,
                     uint64_t* unsat_ids, EzpzWarning* warn_buf, size_t warn_cap, EzpzOutcome* out);

/* ---- ezpz::solve, ezpz/src/lib.rs:80-87 -> solve_with_priority_inner (:148-263) ---------------------
 * Side inference from the guesses (constraints.rs:146-193), priority tiers, returns the last fully
 * satisfied tier.  guesses are (id, value) pairs exactly as the reference takes them.  x_out has
 * n_guesses entries, unsat_ids up to n_reqs, warn_buf up to warn_cap. */
int ezpz_solve(const EzpzConstraint* reqs, size_t n_reqs, const uint32_t* var_ids, const double* guesses,
               size_t n_guesses, const EzpzConfig* cfg, double* x_out, uint64_t* unsat_ids, EzpzWarning* warn_buf,
               size_t warn_cap, EzpzOutcome* out);

/* ---- FreedomAnalysis (reference ezpz/src/solver/find_dof.rs:14-103, analysis.rs:24-77) ----------
 * ezpz_solve_analysis replaces ezpz::solve_analysis (lib.rs:134-146): ezpz_solve plus, for the tier that is returned,
 * the indices of the underconstrained variables (ascending; under_out has room for n_guesses entries).
 * ezpz_system_freedom_batch[_device] analyse `batch` value vectors of one topology (normally the final values of a
 * batch solve): under_mask [batch][n_vars] gets 1 where the variable is underconstrained, participation
 * [batch][n_vars] (optional) the squared row norms of the orthonormal null-space basis (find_dof.rs:90-95).  The
 * Jacobian is re-evaluated at the values given (the reference reuses the LM loop's last refresh, which is the
 * Jacobian at the final values).  Where the system has a frontal plan (EzpzSystemInfo.front_workgroups) the host entries do not
 * factorise J at all: the projector onto its null space is applied to a few probe vectors by the frontal factorisation and the
 * null vectors are refined by subspace iteration -- the same sets, participation equal to rounding; more
 * than four degrees of freedom, a singular value within 1e-7 ... 3e-6 of J's largest entry, or more than 64 systems per call go
 * to the pivoted QR, like every call of the _device entry.  A large component's QR runs on many workgroups that wait for each other's chunks; should one
 * of them give up waiting (workgroups that never became resident beside another process's or stream's kernels), the host entries
 * run the analysis once more as a chain of launches that wait for nobody (EZPZ_ERR_HIP only if that fails too), and the
 * asynchronous _device entry marks the system: 0xFF in every byte of its mask, 0xFFFFFFFF as its count -- as a timed-out grid
 * team's solve reports EZPZ_ITERATIONS_TEAM_TIMEOUT. */
int ezpz_solve_analysis(const EzpzConstraint* reqs, size_t n_reqs, const uint32_t* var_ids, const double* guesses,
                        size_t n_guesses, const EzpzConfig* cfg, double* x_out, uint64_t* unsat_ids,
                        EzpzWarning* warn_buf, size_t warn_cap, EzpzOutcome* out, uint32_t* under_out,
                        uint64_t* n_under_out);
int ezpz_system_freedom_batch(EzpzSystem* sys, const double* x, size_t batch, uint8_t* under_mask,
                              double* participation);
int ezpz_system_freedom_batch_device(EzpzSystem* sys, const double* x_dev, size_t batch, uint8_t* under_mask_dev,
                                     double* participation_dev, uint32_t* n_under_dev, void* stream);

/* ---- ezpz::solve for a batch (new; SURVEY.md 8f #3: priority tiers + weights end-to-end on device batches) --------
 * `batch` systems share the request list and differ in their guesses (x0 AoS [batch][n_vars], id == index).  Per
 * system exactly the semantics of ezpz_solve: LineSide / CircleSide inferred from that system's own guesses
 * (constraints.rs:146-193; systems are grouped by the inferred sides), cumulative priority tiers each re-solved from
 * the original guesses (lib.rs:215-246), the last fully satisfied tier is returned.  priority_solved [batch] and
 * unsat_mask [batch][n_reqs] (indexed by position in `reqs`) are optional.  Topology-level errors (missing guess ...)
 * in the first tier fail the call; in a later tier every system keeps its previous tier, like the reference. */
int ezpz_solve_batch(const EzpzConstraint* reqs, size_t n_reqs, size_t n_vars, const double* x0, size_t batch,
                     const EzpzConfig* cfg, double* x_out, EzpzStatus* status, uint32_t* priority_solved,
                     uint8_t* unsat_mask, int32_t* err_constraint, int64_t* err_variable);

/* ---- one batch over several devices of a node (new; SURVEY.md 8b last row, 8e) ------------------------------------------
 * The reference's caller is a host loop over independent solves (ezpz-cli/src/main.rs:96-98); independent systems have
 * no exchange step, so a batch shards contiguously over the devices of `device_mask` (bit d = HIP device d; 0 = every
 * device of the node): device index g of G gets systems [g * ceil(batch / G), ...), trailing devices may be short or
 * idle.  One host worker thread and one analysed topology (an EzpzSystem) per device; every shard moves over its own
 * device's host link, straight between the caller's buffers and that device -- no peer copies, no collective -- through
 * the single-device path of ezpz_system_solve_batch (registered caller buffers, ezpz_host_register, are pipelined on
 * every device at once).  Results are those of ezpz_system_solve_batch on each shard: a kernel is chosen by the size of the
 * call it serves (lanes across the batch from 64 x 4 x CUs systems, run-time compilation from 1024), so a shard may run on
 * another kernel of its topology than the whole batch would on one device -- bit-identical for component-resident block
 * systems, equal to rounding (and in iteration counts) for connected sketches.
 * `cs` is one side-resolved tier as for ezpz_system_create.  A handle serves one batch call at a time (calls from
 * several threads queue); ezpz_multi_specialize is ezpz_system_specialize on every device (the smallest state is returned).
 * ezpz_multi_shard tells which systems device index `index` takes of a batch.
 * ezpz_system_solve_batch_multi is the same in one call: the handles live in a small cache keyed by the request bytes
 * and the mask (dropped by ezpz_cache_clear). */
typedef struct EzpzMultiSystem EzpzMultiSystem; /* opaque: one analysed topology resident on several devices */
int ezpz_multi_create(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, uint64_t device_mask, uint32_t team_size,
                      EzpzMultiSystem** out, int32_t* err_constraint, int64_t* err_variable);
void ezpz_multi_destroy(EzpzMultiSystem* multi);
int ezpz_multi_device_count(const EzpzMultiSystem* multi);        /* G */
int ezpz_multi_device(const EzpzMultiSystem* multi, int index);  /* HIP device of device index `index`, -1 out of range */
void ezpz_multi_shard(const EzpzMultiSystem* multi, size_t batch, int index, size_t* first, size_t* count);
int ezpz_multi_specialize(EzpzMultiSystem* multi, int wait);
int ezpz_multi_solve_batch(EzpzMultiSystem* multi, const double* x0, size_t batch, const EzpzConfig* cfg, double* x_out,
                           EzpzStatus* status, uint8_t* unsat_mask);
int ezpz_system_solve_batch_multi(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, uint64_t device_mask,
                                  const double* x0, size_t batch, const EzpzConfig* cfg, double* x_out, EzpzStatus* status);

/* ---- one batch of systems of DIFFERENT topologies (new; SURVEY.md 8b last row) -------------------------------------------
 * The reference's callers loop over arbitrary systems, one solve() after the other (ezpz-cli/src/main.rs:96-98,
 * ezpz-wasm/src/lib.rs:96); this is that loop as one call.  System b of the batch has the topology
 * handles[topology_of_system[b]] (side-resolved tiers, all on the calling thread's current device); the batch is RAGGED:
 * x0 / x_out hold the systems' rows one after the other in batch order, system b's n_vars(b) values at
 * x_offset[b] = sum of n_vars over the systems before it (ezpz_mixed_offsets gives the batch + 1 offsets,
 * ezpz_mixed_total_values their last entry).  Inside, the batch is regrouped by topology -- each topology's rows gathered
 * into one block, solved by that topology's kernels on a stream of its own, scattered back -- and every result is the one
 * ezpz_system_solve_batch gives for that system's topology, bit for bit, in caller order.
 * An EzpzMixedBatch is the grouping of one (handles, topology_of_system) pair, reusable for any number of solves (one at
 * a time; the handles must outlive it); the _device form takes device pointers and only enqueues on `stream`;
 * ezpz_system_solve_batch_mixed is create + solve + destroy in one call.  ezpz_multi_solve_batch_mixed shards the
 * batch contiguously (by systems) over the devices the EzpzMultiSystems share -- multis[t] is topology t on every device
 * of one mask -- each device running its shard through the single-device path over its own host link. */
typedef struct EzpzMixedBatch EzpzMixedBatch;
int ezpz_mixed_create(EzpzSystem* const* handles, size_t n_handles, const uint32_t* topology_of_system, size_t batch,
                      EzpzMixedBatch** out);
void ezpz_mixed_destroy(EzpzMixedBatch* mixed);
size_t ezpz_mixed_total_values(const EzpzMixedBatch* mixed);
void ezpz_mixed_offsets(const EzpzMixedBatch* mixed, uint64_t* x_offset /* batch + 1 */);
int ezpz_mixed_solve_device(EzpzMixedBatch* mixed, const double* x0_dev, const EzpzConfig* cfg, double* x_out_dev,
                            EzpzStatus* status_dev, void* stream);
int ezpz_mixed_solve(EzpzMixedBatch* mixed, const double* x0, const EzpzConfig* cfg, double* x_out, EzpzStatus* status);
int ezpz_system_solve_batch_mixed(EzpzSystem* const* handles, size_t n_handles, const uint32_t* topology_of_system,
                                  const double* x0, size_t batch, const EzpzConfig* cfg, double* x_out, EzpzStatus* status);
int ezpz_multi_solve_batch_mixed(EzpzMultiSystem* const* multis, size_t n_multis, const uint32_t* topology_of_system,
                                 const double* x0, size_t batch, const EzpzConfig* cfg, double* x_out, EzpzStatus* status);

/* Constraint::set_from_initial_values (ezpz/src/constraints.rs:146-193) over a request list, in place: every
 * LineTangentToCircle / CircleTangentToCircle whose side is EZPZ_SIDE_UNDEFINED gets the side the values imply
 * (values by id, id == index).  ezpz_solve / ezpz_solve_batch do this themselves; callers of the handle API
 * (ezpz_system_create takes side-resolved tiers) use this first. */
int ezpz_resolve_sides(EzpzConstraint* cs, size_t n_cs, const double* values, size_t n_vars);

/* ---- class-specialised kernels ------------------------------------------------------------------------------------
 * Systems that run component-resident (EzpzSystemInfo.team_mode 3) can have their kernel compiled at run time
 * (hiprtc) into straight-line code for exactly their classes of components: same operations in the same order,
 * state in registers instead of LDS.  Batch calls of >= 1024 systems start that compilation on a background thread
 * (so do topologies solved more than 256 times, one call after the other: an interactive sketch) and switch to the
 * specialised kernel once it is ready (results are bit-identical either way for component-resident systems; the
 * lane-per-system form sums residuals in request order, i.e. agrees to rounding); EZPZ_JIT=0 in the
 * environment turns it off.  ezpz_system_specialize starts it explicitly and, with wait != 0, returns when it is
 * done: 2 = ready, 1 = still compiling, 0 = this system has no specialised form, negative = compilation failed.
 * Compiled code objects are kept on disk ($EZPZ_JIT_CACHE_DIR, else $XDG_CACHE_HOME/ezpz_amd, else
 * $HOME/.cache/ezpz_amd; EZPZ_JIT_CACHE=0 turns the cache off), keyed by the generated source, the embedded device
 * headers, the compiler options and the hiprtc version: a process that finds its kernel there has it from its first few
 * solves (a system's first launch asks the cache in the background) instead of after 256 solves and a compilation.
 * Budget: a loaded code object is never unloaded, so a process holds at most 512 specialised kernels (distinct
 * topologies; systems with the same generated source share one).  Beyond that ezpz_system_specialize returns
 * EZPZ_ERR_KERNEL_BUDGET and batch calls keep running on the interpreting kernels (same results, lower rate).
 * ezpz_specialized_source (no device needed) writes the generated source of a request into buf (NUL-terminated,
 * truncated to cap) and returns its length, 0 when the request gets no component plan; with compile != 0 it also
 * compiles it for gfx950 (compile == 2: through the on-disk cache, like the solve entry points) and returns a negative
 * error with the compiler's log in buf on failure. */
int ezpz_system_specialize(EzpzSystem* sys, int wait);
long ezpz_specialized_source(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, int compile, char* buf, size_t cap);

/* ezpz_solve / ezpz_solve_inner keep a small cache of analysed topologies keyed by the request bytes, so that
 * repeated solves of one problem (ezpz-cli's 100-run loop, main.rs:96-98) skip the symbolic phase.  This drops
 * it (used to time cold solves). */
void ezpz_cache_clear(void);

/* The launch-shape thresholds (csrc/policy.hpp has each number's measurement): what decides which kernel serves a call,
 * for a device of `compute_units` CUs.  "systems per call that fill the device" scale with the CU count; what belongs to
 * one workgroup or one CU's LDS does not.  ezpz_launch_policy(0, ...) = the full 256-CU MI355X. */
typedef struct EzpzLaunchPolicy {
    uint32_t compute_units;
    /* scaled by the CU count */
    uint64_t lanes_min_systems_small;  /* systems per call from which a connected sketch runs one lane per system (64 x 4 x CUs) */
    uint64_t lanes_min_systems_large;  /* ... a sketch of >= lanes_large_from_vars variables (64 x 2 x CUs) */
    uint32_t lanes_large_from_vars;
    uint64_t jit_lane_min_batch;       /* a call this large starts the run-time compilation at once: one lane per system (16 x CUs) */
    uint64_t jit_comp_min_batch;       /* ... component-resident block systems (4 x CUs) */
    uint64_t jit_comp_min_values;      /* ... or this many variables in the call (8192 x CUs) */
    /* not scaled */
    uint32_t jit_after_launches;       /* smaller calls earn the compilation by repetition */
    uint32_t lane_max_vars, lane_max_constraints;
    uint32_t comp_min_components, comp_max_component_vars, comp_max_component_constraints, comp_max_classes;
    uint32_t rec_min_vars_one_solve, rec_min_vars_batch, rec_one_wavefront_max_vars, rec_max_components;
    uint32_t rec_wide_one_solve_max_vars;
    uint32_t sub_team_max_width, dense8_max_vars;
    uint64_t zero_copy_max_bytes, h2h_piece_min_bytes, h2h_piece_max_bytes;
    uint32_t h2h_pieces_per_call;
    uint32_t one_call_host_mask_max_constraints, one_call_host_log_max_entries;
    /* the frontal shape (team_mode 5): one solve of a connected sketch takes it from this many variables; a system created for
     * batches (team_size 0) carries the plan from that many (0 = never) and takes it for calls of up to
     * EzpzSystemInfo.front_max_batch systems (= the systems the device holds at once, compute units / workgroups per system, times
     * max(1, workgroups per system / front_small_call_wgs_per_round) rounds); one workgroup per system up to
     * front_vars_per_workgroup x 2 variables, then one more per that many */
    uint32_t front_min_vars_one_solve, front_min_vars_batch, front_vars_per_workgroup, front_max_workgroups;
    uint32_t front_small_call_wgs_per_round;
} EzpzLaunchPolicy;
int ezpz_launch_policy(int compute_units, EzpzLaunchPolicy* out);

/* Diagnostic: per-stage time stamps of the calling thread's ezpz_solve* calls.  While `buf` is set, every stage boundary of
 * the one-call path appends an (id, CLOCK_MONOTONIC nanoseconds) pair (ids: csrc/call_trace.hpp, CallStage) up to `cap`
 * words; buf == NULL turns it off.  Returns the number of words written since the previous call
 * (tools/solve_call_breakdown.py -> profiles/r04_solve_call_breakdown.txt). */
size_t ezpz_debug_call_trace(uint64_t* buf, size_t cap);

/* Diagnostic: a device buffer for in-kernel time stamps, NULL to stop (the -DEZPZ_STAMPS builds of the list-walk and frontal
 * kernels: tools/front_stamps.py; the run-time compiled kernel of a system on several workgroups when the process runs with
 * EZPZ_JIT_STAMPS=1: tools/ladder_stamps.py).  Not for production callers. */
void ezpz_debug_set_stamps(unsigned long long* dev_buf);

/* Diagnostic: how this process's FreedomAnalysis calls by null-space probes ended (csrc/freedom.hip: freedom_by_probes): out8[0]
 * systems decided fully constrained by the first eight probes, [1] decided with null vectors found, [2] calls that took the
 * second opinion at lambda / 1000; calls handed to the pivoted QR because [3] an answer was not finite, [4] five or more
 * candidate directions, [5] a direction undecided at both lambdas, [6] an unsettled direction, [7] probes not applicable. */
void ezpz_debug_freedom_exits(unsigned long long* out8);

/* Diagnostic: run-time compilations (hiprtc) this process has performed so far -- a kernel found in the on-disk cache of code
 * objects does not count (tests/test_abi_cpu.py::test_code_object_cache_on_disk). */
unsigned long long ezpz_debug_jit_compilations(void);

/* Diagnostic (host only, no device needed): the symbolic phase of the FRONTAL launch shape (team_mode 5, csrc/fronts.cpp:
 * the supernodal counterpart of faer's SymbolicLlt, solver.rs:289-300) for `wgs` workgroups per system (0 = automatic) on
 * workgroups of `lds_bytes` of LDS.  Copies the plan's device blob into buf (up to cap bytes) and fills info[0..15] =
 * workgroups, scratch chunks, first pivot-flag chunk, verdict chunk, LDS bytes, fronts, levels, largest front's rows /
 * pivots, lanes per workgroup, modelled cycles, panel doubles, update doubles, 0...; returns the blob's size, 0 when the
 * shape does not apply to the system, or a negative EZPZ_ERR_*.  tests/front_ref.py executes the blob in numpy. */
long ezpz_debug_front_plan(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, uint32_t wgs, uint32_t max_wgs,
                           uint64_t lds_bytes, unsigned char* buf, size_t cap, uint64_t* info);

/* ---- textual front end, ezpz/src/textual.rs:43-49 (Problem: FromStr) + executor.rs:40-445 ------------ */
typedef struct EzpzProblem EzpzProblem; /* opaque: parsed + lowered problem text */
int ezpz_problem_parse(const char* text, size_t len, EzpzProblem** out, char* errbuf, size_t errcap);
void ezpz_problem_destroy(EzpzProblem* p);
size_t ezpz_problem_num_constraints(const EzpzProblem* p);
size_t ezpz_problem_num_vars(const EzpzProblem* p);
const EzpzConstraint* ezpz_problem_constraints(const EzpzProblem* p);
const double* ezpz_problem_guesses(const EzpzProblem* p); /* values by id (id == index) */
/* label tables, executor.rs:521-566: kind 0 = points, 1 = circles, 2 = arcs */
size_t ezpz_problem_num_labels(const EzpzProblem* p, int kind);
const char* ezpz_problem_label(const EzpzProblem* p, int kind, size_t index);

#ifdef __cplusplus
}
#endif
#endif /* EZPZ_AMD_H */
