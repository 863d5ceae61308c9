// Component-resident solve: the host side.
//
// A constraint system made of many small independent pieces -- the 2000 x 2000 massive_parallel_system is 1500 of them
// (500 x {Vertical, Fixed} on two variables, 1000 x {Fixed} on one) -- needs no per-system index lists at all: its
// connected components fall into a handful of isomorphism CLASSES (same kinds, tags, weights and local variable
// pattern; only the caller's variable ids and the constraint parameters differ).  The component plan therefore holds
//   * one tiny program per class: its constraint records and the straight-line operation stream of its linear solve
//     (normal equations, Cholesky, substitutions), derived from the ordinary symbolic phase (build_program) run on
//     one representative component -- same elimination order, same operation order as the list-walk kernels;
//   * per class, structure-of-arrays tables over its instances: caller's variable ids, constraint parameters,
//     caller's constraint positions;
//   * chunks of up to 64 instances of one class (one per lane of a wavefront) dealt to the wavefronts of a workgroup.
// The kernel (comp_kernel.hip.hpp) runs one lane per component instance; every wavefront executes one class program
// at a time, so the program is read through the scalar unit and all control flow is uniform.
// Replaces, for such systems, the same reference code as lm_kernel.hip.hpp (ezpz/src/solver/newton.rs:29-145,
// solver.rs:192-440).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/ezpz_amd.h"
#include "dev_types.hpp"

namespace ezpz {

// Operation stream of a class: records of 8 words.
//   w0 = opcode | n_items << 8 | flags << 16     (flags: 1 = first record of the operation, 2 = last)
//   w1 = a | b << 16                              (operands, rows of the chunk's state)
//   w2.. items
// General build (Jacobian values in LDS): one word per item, two 16-bit row numbers; up to 6 items per record.
// (b, y and the step d share the rows at o_d; the diagonal and, once it is dead, the tentative x share scratch row 0..)
//   DIAG  a = variable v          items (Jacobian slot, residual row):  A_vv = sum j^2 + lambda, b_v = sum j * -r
//   OFF   a = L slot s            items (slot, slot):                   A_s = sum j_a * j_b
//   COL   a = variable v          items (L slot of (v,k), variable k):  d_v = sqrt(A_vv - sum l^2), y_v = (b_v - sum l y_k) / d_v
//   SLOT  a = L slot s, b = col j items (L slot (i,k), L slot (j,k)):   l_s = (A_s - sum l l) / d_j
//   BWD   a = variable v          items (L slot of (i,v), variable i):  dx_v = (y_v - sum l dx_i) / d_v
// Linear build (every kind linear, Jacobian entries are class constants): DIAG carries sum j^2 in w2:w3 and up to two
// items (row, j as f32) in w4..w7; OFF carries its constant in w2:w3.
// Fused streams only (one lane per system): a column's DIAG + COL, and a slot's OFF + SLOT, as ONE record when their
// items fit it -- every record is a round trip to memory for the lane, and the assembly records hold ~1 item of 6:
//   DIAGCOL a = variable v     items [0, n0): (Jacobian slot, residual row), [n0, n): (L slot of (v,k), variable k)
//   SLOTA   a = L slot s       items [0, n0): (Jacobian slot, slot),         [n0, n): (L slot (i,k), L slot (j,k));  n0 = w0 >> 24
// (same operations in the same order as the two records they replace)
enum CompOpcode : uint32_t { COMP_DIAG = 1, COMP_OFF = 2, COMP_COL = 3, COMP_SLOT = 4, COMP_BWD = 5, COMP_DIAGCOL = 6, COMP_SLOTA = 7 };
constexpr uint32_t kCompRecWords = 8;
constexpr uint32_t kCompItemsGen = 6, kCompItemsLin = 2;
constexpr uint32_t kCompFirst = 1u << 16, kCompLast = 2u << 16;
// fused streams (one lane per system, batch_kernel.hip.hpp): the entry of JtJ is assembled right before its column /
// slot is eliminated and stays in the accumulator -- kCompKeep on the last record of a DIAG / OFF: do not store;
// kCompCont on the first record of the COL / SLOT that follows: do not load
constexpr uint32_t kCompKeep = 4u << 16, kCompCont = 8u << 16;
// fused streams run column by column (a column, then its slots): kCompDivReg on a SLOT = its divisor d_j is the one the
// COL before it just computed, still in a register
constexpr uint32_t kCompDivReg = 16u << 16;

// Constraint record of a class: 16 words.
//   w0 kind | tag << 8 | nrows << 16 | nslots << 24      w1 row0 | jbase << 16
//   w2..w5 ids[8] (16-bit, class-internal variable numbers)   w6..w9 jloc[16]   w10:w11 weight   w12 index in the class
constexpr uint32_t kCompConWords = 16;

// Chunk descriptor: one wavefront's worth of instances (<= 64, one per lane) of one class, with everything the
// kernel needs to run the class program on them.  32 words, read through the scalar unit once per pass.
struct CompChunk {
    uint32_t count;                 // lanes in use
    uint32_t nv, m, ncons;          // variables, residual rows, constraints of the class
    uint32_t n_ops, ops_off;        // operation stream (records, word offset into the blob)
    uint32_t cons_off;              // constraint records (word offset)
    uint32_t row0;                  // first LDS row (of 64 doubles) of the chunk's persistent state: x at 0,
    uint32_t o_d, o_r0, o_r1, o_j, o_wm;  // ... step d (b -> y -> d during a solve), residuals (two copies), Jacobian, warning mask
    uint32_t s_l;                   // the wavefront's scratch rows: diagonal (later the tentative x) at 0, L here
    uint32_t stride;                // instances of the class, padded to a multiple of 64: row stride of its tables
    uint32_t ids_off;               // u32 [nv][stride]     caller's variable id      (word offsets, first instance
    uint32_t par_off;               // f64 [ncons][stride]  constraint parameter       of the chunk folded in)
    uint32_t pos_off;               // u32 [ncons][stride]  caller's constraint position
    uint32_t pad[14];
};
static_assert(sizeof(CompChunk) == 32 * 4, "CompChunk layout");

struct CompPlan {
    std::vector<uint32_t> blob;   // everything the kernel reads, as 32-bit words
    uint32_t o_waves = 0;         // [n_waves + 1] first chunk of every wavefront
    uint32_t o_chunks = 0;        // CompChunk[n_chunks], wavefront-major
    uint32_t n_waves = 0, n_chunks = 0, n_classes = 0, n_instances = 0;
    bool linear = false;          // every class linear with f32-exact Jacobian constants: the linear build of the kernel
    bool unit_weights = true;
    uint32_t rows_persistent = 0, scratch_rows = 0;  // LDS rows: all chunks' state; per wavefront scratch
    uint32_t lds_bytes = 0;
    uint32_t n_vars = 0, n_cons = 0, n_rows = 0;
    uint32_t max_levels = 0;
    uint64_t zj = 0, za = 0, zl = 0;  // totals over all components (EzpzSystemInfo)
    // The same plan as source text for the class-specialised kernel (jit_kernel.hip.hpp + jit.cpp): one struct per
    // class, the slot sequence of a wavefront and the kernel entry point; `jit_waves` wavefronts per system, each with
    // `jit_slots` slots; blob[o_jit_slots ...] = [wave][slot] {ids_off, par_off, pos_off, count}.  Empty = none.
    std::string jit_source;
    uint32_t jit_waves = 0, jit_slots = 0, o_jit_slots = 0;
    uint32_t o_jit_ranges = 0;  // blob[...] = [wave] {first variable, count} when every wavefront's variables are one contiguous run (else 0)
    uint32_t jit_wgs = 1;        // workgroups that share one system in the specialised kernel (grid reductions when > 1)
    bool interpretable = true;   // the state fits one CU's LDS: comp_solve_kernel can run the plan (else specialised only)
};

constexpr size_t kJitGridScratchBytes = 197184;  // sizeof(ezpz::jit::GridScratch), one per system in flight

struct CompLimits {
    size_t lds_bytes = 160 * 1024;
    uint32_t max_waves_linear = 8, max_waves_general = 8;
};

// True when the system is worth running component-resident (many small components in few classes, state fits the
// LDS); `plan` is then complete.  False leaves the caller with its other launch shapes.  Request errors (bad ids ...)
// are not diagnosed here: the caller has validated the request.
bool comp_plan_build(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, const CompLimits& lim, CompPlan& plan);

}  // namespace ezpz

namespace ezpz {

// One batch launch of the component-resident kernel (comp.hip).  Device pointers; `stream` is a hipStream_t.
struct CompLaunch {
    const double* x0;
    double* x_out;
    EzpzStatus* status;
    uint8_t* unsat_mask;
    uint64_t* warn_log;
    uint32_t warn_cap;
    uint64_t batch;
    uint32_t max_iterations;
    double residual_tolerance, step_tolerance, initial_lambda;
    DoneWord done;  // one-call launches: the completion word (dev_types.hpp), else null
    // lane-per-system kernel only: the systems of one topology inside a ragged batch, in place (jit_kernel.hip.hpp: LaneArgs)
    const uint64_t* row_offset = nullptr;
    const uint32_t* sys_of = nullptr;
    // the specialised kernels' work counters (jit_kernel.hip.hpp: JitArgs::ticket) and where each of the eight stands
    unsigned int* ticket = nullptr;
    const unsigned int* ticket_base = nullptr;
};
int comp_launch(const CompPlan& plan, const uint32_t* dev_blob, const CompLaunch& launch, int device, int cus,
                size_t lds_limit, void* stream);

}  // namespace ezpz

namespace ezpz {

// One lane per system (jit_kernel.hip.hpp, lane_kernel): batches of one small system (<= 20 variables) as run-time
// compiled straight-line code, every lane running the whole LM loop of its own system.
struct LanePlan {
    std::string jit_source;
    std::string wave_source;  // the same class on one wavefront per system (one solve()'s shape), or empty
    uint32_t n_vars = 0, n_cons = 0, n_rows = 0;
    bool unit_weights = true;
};
bool lane_plan_build(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, LanePlan& plan);

// Lanes across the batch (batch_kernel.hip.hpp): one lane per system for batches of one connected sketch -- the class
// program of the whole system (parameters in the constraint records, records in request order) + the workspace layout.
struct BatchPlan {
    std::vector<uint32_t> blob;
    uint32_t nv = 0, m = 0, zj = 0, zlo = 0, ncons = 0, n_ops = 0, ops_off = 0, cons_off = 0, var_off = 0, inv_off = 0;
    uint32_t o_d = 0, o_r = 0, o_rn = 0, o_j = 0, o_dg = 0, o_l = 0, rows = 0;  // workspace rows of 64 doubles per wavefront
    bool unit_weights = true;
};
bool batch_plan_build(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, BatchPlan& plan);
// wavefronts of the kernel the device holds at once (their workspaces are what the caller allocates)
uint64_t batch_launch_waves(int cus);
uint32_t batch_straggler_lanes();  // working lanes at which a wavefront hands its systems to the teams (0 = never)
// strag_list / strag_count / strag_cap: where wavefronts that are down to their last few lanes leave the systems they
// give up (batch_kernel.hip.hpp) and `strag_state` the LM state they had reached; null: every system is solved by its lane
int batch_launch(const BatchPlan& plan, const uint32_t* dev_blob, double* dev_ws, uint64_t ws_waves, uint32_t n_cons,
                 const CompLaunch& launch, void* stream, uint32_t* strag_list = nullptr, uint32_t* strag_count = nullptr,
                 uint32_t strag_cap = 0, LmResume* strag_state = nullptr);

// The class-specialised kernel of a plan (jit.cpp): run-time compiled (hiprtc) on a background thread.
// comp_jit_create returns nullptr when the plan carries no source or EZPZ_JIT=0.  comp_jit_request starts the
// compilation if it has not started (and waits for it if asked); returns 0 idle, 1 compiling, 2 ready, -1 failed,
// kJitBudgetExhausted.
struct CompJit;
CompJit* comp_jit_create(const CompPlan& plan);
CompJit* comp_jit_create_source(const std::string& source, const char* entry);
int wave_jit_launch(CompJit* jit, const LanePlan& plan, const CompLaunch& launch, int device, int cus, void* stream);
int lane_jit_launch(CompJit* jit, const LanePlan& plan, const CompLaunch& launch, int device, int cus, void* stream);
void comp_jit_destroy(CompJit* jit);
int comp_jit_request(CompJit* jit, bool wait);
int comp_jit_state(const CompJit* jit);
const char* comp_jit_log(const CompJit* jit);
// (a system on several workgroups: `grid_slots` systems in flight on `grid_scratch`; fast_slots / redo: the same for the kernel
// that does not wait for its verdicts, and the device list -- 1 + batch words, count zero -- of the systems it leaves to the loop;
// redo_next: the list of the system's next call, whose count this call zeroes; redo_seen_dev / _host: a word of mapped host memory
// the loop's launch leaves its count in, the hint for the width of the next one.  0 / null: the loop alone)
int comp_jit_launch(CompJit* jit, const CompPlan& plan, const uint32_t* dev_blob, const CompLaunch& launch, int device, int cus, void* stream,
                    void* grid_scratch = nullptr, uint32_t grid_slots = 0, uint32_t fast_slots = 0, unsigned int* redo = nullptr,
                    unsigned int* redo_next = nullptr, unsigned int* redo_seen_dev = nullptr, const unsigned int* redo_seen_host = nullptr);
// workgroups of the specialised kernel the device holds at once (loads the code object on first use); 0 on failure
uint64_t comp_jit_capacity(CompJit* jit, const CompPlan& plan, int device, int cus);
// ... of its `_fast` entry (0: the kernel has none)
uint64_t comp_jit_capacity_fast(CompJit* jit, const CompPlan& plan, int device, int cus);
// whether this launch may take the `_fast` entry (then: fast_slots, redo, redo_next to comp_jit_launch)
bool comp_jit_fast_ok(CompJit* jit, const CompPlan& plan, const CompLaunch& launch, int device, int cus);
// (through the on-disk cache of code objects; _uncached always compiles; comp_jit_cached: is it in the cache?)
int comp_jit_compile(const std::string& source, std::vector<char>& code, std::string& log);
int comp_jit_compile_uncached(const std::string& source, std::vector<char>& code, std::string& log);
bool comp_jit_cached(const std::string& source);
unsigned long long comp_jit_compilations();  // hiprtc compilations of this process so far (cache hits do not count)
// asks the on-disk cache for the kernel in the background, once (a system's first launch does this)
void comp_jit_probe(CompJit* jit);
// comp_jit_request when the process already holds its budget of resident specialised kernels (512: code objects are
// never unloaded, jit.cpp): the entry stays idle and the interpreters serve
constexpr int kJitBudgetExhausted = -2;

}  // namespace ezpz
