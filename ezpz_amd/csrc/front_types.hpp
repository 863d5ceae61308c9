// Plain-data types shared by the host planner (fronts.cpp) and the kernel (front_kernel.hip.hpp) of the FRONTAL launch shape:
// the linear solve of one connected sketch as a multifrontal supernodal Cholesky factorisation.
//
// Where the reference hands every pattern to faer's symbolic + numeric LLT -- simplicial or supernodal as the pattern
// demands (ezpz/src/solver.rs:289-300, solver/newton.rs:87-102) -- the list / record walks of lm_kernel.hip.hpp eliminate
// one COLUMN per synchronisation.  Here the unit is a FRONT: a supernode of K <= 16 consecutive columns of the elimination
// order with the S - K later rows they touch, held as a dense panel; a wavefront owns a front (lane = row), children hand
// their Schur complements to the parent (extend-add), independent subtrees run side by side on the wavefronts of a
// workgroup -- and, for large systems, on different workgroups that meet at the top of the tree.
#pragma once
#include <cstdint>

#include "dev_types.hpp"

namespace ezpz {

constexpr uint32_t kFrontMaxRows = 63;    // S: rows of a front (lane r = row r; lane S = the right-hand side's row)
constexpr uint32_t kFrontMaxPivots = 16;  // K: columns eliminated in one front
constexpr uint32_t kFrontMaxWgs = 64;     // workgroups that share one system
constexpr uint32_t kFrontRedValues = 4;   // values per grid reduction

// One front.  Panel: column-major (S + 1) x K doubles at `panel` (row S = right-hand side: b, then y); update matrix: the
// packed lower triangle of the (R + 1) x (R + 1) trailing block (R = S - K; entry (a, b), a >= b, at a (a + 1) / 2 + b;
// row R = right-hand side) at `upd`.  Offsets count doubles from the workgroup's workspace.
struct alignas(16) FrontDesc {
    uint16_t K, S;
    uint16_t n_child;      // children in OTHER workgroups (their update matrices arrive as chunks; this workgroup's own children
                           // are sources of the front's source stream)
    uint16_t flags;        // FRONT_*
    uint32_t panel;        // workspace offset of the panel
    uint32_t upd;          // workspace offset of the update matrix (unused when R == 0)
    uint32_t rows;         // index into the rows table: S local variable indices (pivots first)
    uint32_t child0;       // index of the first FrontChild
    uint32_t src_off;      // word offset of the front's source stream in the staged tables
    uint16_t src_n;        // entries in it
    uint16_t n_kids_local; // children in THIS workgroup: the front starts when that many have signed in (front_kernel.hip.hpp)
    uint8_t src_v[4];      // source words per entry in trip 0, 1, 2 and >= 3 of 64 entries (4-byte aligned: read as one word)
    uint32_t up_chunk;     // FRONT_REMOTE_PARENT: first chunk of the update matrix in the system's scratch
    uint32_t exp0;         // FRONT_EXPORTS: index into the export table: K entries (chunk of pivot k's step, or ~0)
    uint32_t parent_local; // the parent's index among this workgroup's fronts, or ~0 (a root of the workgroup's share)
};
static_assert(sizeof(FrontDesc) == 48, "FrontDesc layout");
constexpr uint16_t FRONT_REMOTE_PARENT = 1, FRONT_EXPORTS = 2;

// One child of a front: where its update matrix lies, and for each of its R_c + 1 rows the parent's row (the right-hand
// side's row maps to the parent's row S).
struct alignas(16) FrontChild {
    uint32_t upd;      // workspace offset (doubles), or the first chunk in the system's scratch (FRONT_CHILD_REMOTE)
    uint16_t rows;     // R_c + 1
    uint16_t flags;
    uint32_t map;      // byte offset into the map table
    uint32_t pad;
};
static_assert(sizeof(FrontChild) == 16, "FrontChild layout");
constexpr uint16_t FRONT_CHILD_REMOTE = 1;

// A ghost variable of a workgroup: a later (ancestor) variable that one of its fronts or constraints touches and another
// workgroup eliminates; its step arrives as a chunk.
struct FrontGhost {
    uint32_t local;  // local variable index (>= n_own)
    uint32_t chunk;  // chunk of the system's scratch that carries its step
};

// One workgroup's share of a system.  Byte offsets count from the plan blob's start, `l_*` doubles from the workspace's.
struct alignas(16) FrontWg {
    uint32_t n_loc, n_own, n_ghost;   // local variables: own pivots first, ghosts behind
    uint32_t n_cons, n_rows, zj;      // constraints evaluated here, their residual rows, their Jacobian slots
    uint32_t n_fronts, n_levels;
    uint32_t o_var_glob;              // uint32[n_loc]: caller's variable id
    uint32_t o_cons;                  // DevCon[n_cons] (ids = local variables, row0 / jbase local)
    uint32_t o_tables;                // start of the tables staged into LDS once per workgroup: tab_bytes of
    uint32_t tab_bytes;               //   FrontDesc[n_fronts] | level_ptr (the planner's levels: tests/front_ref.py walks them) | FrontChild[] | rows | exports | maps | streams | schedules
    uint32_t t_level_ptr, t_children, t_rows, t_exports, t_maps;  // byte offsets inside the staged tables
    uint32_t t_stream;                // ... of the streams (uint32 words): the fronts' source streams, then the assembly stream
    uint32_t asm_word0, asm_trips;    // the assembly stream: its first word in the streams, its trips of 64 entries
    uint32_t t_cons;                  // byte offset of a copy of the constraint table inside the staged tables, or ~0: read it at o_cons
    uint32_t o_ghosts;                // FrontGhost[n_ghost]
    uint32_t l_x, l_d, l_r, l_rn, l_jv, l_panels, l_upool;  // workspace carve-up (doubles)
    uint32_t ws_doubles;              // workspace doubles (state), tables excluded
    uint32_t n_remote_children;       // fronts of this workgroup that wait for chunks of other workgroups
    // The wavefronts' SCHEDULES (byte offset inside the staged tables; uint16 words): [threads / 64 + 1] starts of the forward lists,
    // [threads / 64 + 1] starts of the backward lists (in words from the table's start), then the lists -- the fronts a wavefront
    // factorises, in order (a front starts when its local children have signed in), and the fronts it substitutes back, in order
    // (a front starts when its local parent is through).  Made by list scheduling on the cost model (fronts.cpp): no barrier
    // between the levels of the tree, a wavefront waits for exactly what its next front needs.
    uint32_t t_sched;
    // Which entry of the Jacobian a slot is (byte offset from the plan blob's start; uint16 per slot of this workgroup: the
    // local variable of its column | the row inside its constraint << 15): what J times a vector needs (the null-space probes of
    // FreedomAnalysis, front_kernel.hip.hpp); read from global memory, not staged.
    uint32_t o_slotmap;
    uint32_t pad[3];
};
static_assert(sizeof(FrontWg) % 16 == 0, "FrontWg layout");

// The ASSEMBLY stream of a workgroup (once per linear solve, all lanes, no order among entries): per trip of 64 entries, 64
// header words, then w x 64 operand words, w = header >> 24 (the same in all 64 headers of a trip; trips are sorted by it).  An
// entry is one element of one front -- of its panel or of its update matrix -- that the workgroup's constraints contribute to:
// header = destination (doubles from l_panels) | flags | w << 24; operand word = a | b << 16: Jacobian slots (element of JtJ:
// += jv[a] * jv[b]), or with FASM_RHS slot a and residual row b (element of -Jt r: -= jv[a] * r[b]); FASM_DIAG adds lambda.
// Padding operands are (zj, zj) / (zj, n_rows): a Jacobian slot and a residual row that hold zero.  Elements no entry names stay
// zero (the fill of merged supernodes, elements that only children contribute to): the panels and update matrices are zeroed first.
// The SOURCE stream of a front (before its factorisation, by its wavefront): per trip of 64 entries, 64 header words (the
// destination, doubles from l_panels), then v x 64 source words (v = src_v[trip]): s0 | s1 << 16, two elements of update matrices of
// this workgroup's own children (doubles from l_panels) that the extend-add sends to the destination; padding sources are 0: the
// first double at l_panels stays zero.
constexpr uint32_t FASM_DIAG = 1u << 17, FASM_RHS = 1u << 18, FASM_NOP = 1u << 19;

// The scratch of one system in flight on several workgroups (all values travel as self-validating 16-byte chunks, see
// lm_kernel.hip.hpp: grid_store / grid_wait).  chunks[]: update matrices of fronts whose parent lives in another workgroup,
// then the steps of exported variables.
struct FrontScratchHead {
    int nwarn[2];
    int dead;
    int pad0;
    unsigned int hop[kFrontMaxWgs];  // per workgroup: the last hop / reduction sequence number of its previous launch on this slot
    unsigned int red[kFrontMaxWgs];
    unsigned int pad1[60];
};
static_assert(sizeof(FrontScratchHead) == 768, "FrontScratchHead layout");
// behind the head: reduction partials [2 parities][kFrontRedValues][kFrontMaxWgs] chunks, results [2][kFrontMaxWgs][kFrontRedValues]
// chunks (one 64-byte line per workgroup), then the plan's n_chunks chunks
constexpr uint32_t kFrontScratchRedBytes = 2u * kFrontRedValues * kFrontMaxWgs * 16u;
inline constexpr uint32_t front_scratch_bytes(uint32_t n_chunks) {
    return (uint32_t)sizeof(FrontScratchHead) + 2u * kFrontScratchRedBytes + ((n_chunks * 16u + 255u) & ~255u);
}

// The kernel's argument block (front.hip fills it from the launch's SolveArgs and the system's plan).
struct FrontArgs {
    const unsigned char* plan;  // FrontPlan::blob on the device
    uint32_t n_wgs;             // workgroups per system
    uint32_t n_vars, n_cons;    // of the whole system: row length of x0 / x_out, of the unsatisfied mask
    const double* x0;
    double* x_out;
    EzpzStatus* status;
    uint8_t* unsat_mask;
    uint64_t* warn_log;
    uint32_t warn_cap;
    uint32_t max_iterations;
    uint64_t batch;
    double residual_tolerance, step_tolerance, initial_lambda;
    uint32_t unit_weights;
    uint32_t tab_lds_bytes;     // LDS reserved for the staged tables (largest workgroup's, a multiple of 16)
    uint32_t ws_doubles;        // ... and for the workspace
    uint32_t n_chunks, bad_chunk0, verdict_chunk;
    unsigned char* scratch;     // n_wgs > 1: one FrontScratch per system in flight
    uint32_t scratch_stride;
    // NULL-SPACE PROBES (FreedomAnalysis of a system the fronts serve; freedom.hip): probe_m > 0 -- no LM loop; `batch` counts work items =
    // systems x probe_m, item i = probe i % probe_m of system i / probe_m (a system's probes side by side on as many workgroups); at x0 the kernel
    // evaluates J and, for j < probe_m, solves (JtJ + lambda_p I) d = -JtJ w_j for the pseudo-random vector w_j (uniform in [-1, 1): a hash of j and
    // the variable's id; or the caller's own vectors, probe_in) and writes y_j = w_j + d = lambda_p (JtJ + lambda_p I)^-1 w_j to probe_out[(system x probe_m + j) x n_vars +
    // variable]: the projection of w_j onto J's null space up to lambda_p / sigma^2 (lambda_p = 1e-11 x the largest squared entry
    // of J).  A failed pivot writes NaN.  x_out / status / masks are not written.
    uint32_t probe_m;
    double* probe_out;
    const double* probe_in;  // [batch][probe_m][n_vars] the vectors w_j themselves instead of the pseudo-random ones, or null
    double probe_scale;      // lambda_p = probe_scale x the largest squared entry of J (1e-11; 1e-14 for a second opinion)
    unsigned long long* stamps;  // diagnostic builds
    DoneWord done;
};

}  // namespace ezpz
