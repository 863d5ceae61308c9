// Host-side symbolic phase: turns one tier of constraints into the device "topology program".
//
// This is the MI355X-side counterpart of Model::new (reference ezpz/src/solver.rs:192-300): where the
// reference builds a CSC pattern and asks faer for a SymbolicLlt, we emit flat index lists that the LM
// kernel walks with one lane per list:
//   * constraint table (kind-sorted, with the Jacobian slot of every partial it emits),
//   * column view of J            -> diag(JtJ) and b = -Jt r,
//   * J-slot pair lists           -> strict lower part of JtJ, stored directly in L's slots,
//   * elimination-tree levels     -> level-scheduled sparse Cholesky (pair lists per L entry),
//   * row / column lists of L     -> forward / backward substitution.
// Nothing here depends on variable values, so it is computed once per topology and cached.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/ezpz_amd.h"
#include "dev_types.hpp"

namespace ezpz {

struct ProgramCounts {
    uint32_t n_cons = 0, n_vars = 0, n_rows = 0;
    uint32_t zj = 0;       // Jacobian slots (= nnz(J) of the reference's deduplicated CSC)
    uint32_t zlo = 0;      // strictly-lower entries of L
    uint32_t za = 0;       // nnz(lower(JtJ + lambda I)) incl. diagonal
    uint32_t n_levels = 0;  // elimination-tree height (max over partitions)
    uint32_t n_components = 0;
    uint32_t n_parts = 1;
    uint32_t dense = 0;  // dense factor layout (see build_program)
    uint64_t n_apairs = 0, n_lpairs = 0;
};

struct Program {
    ProgramCounts c;
    std::vector<DevCon> cons;                          // sorted by (partition, kind)
    std::vector<PartDesc> parts;
    std::vector<uint32_t> colj_ptr, colj_items;        // per internal var: (jslot,row)*
    std::vector<uint32_t> apair_ptr, apairs;           // per L offdiag slot: (ja,jb)*
    // Internal numbering (chosen so that consecutive lanes touch consecutive LDS words): variables are numbered
    // in elimination-schedule order (partition, level, column), rows and Jacobian slots in constraint-table order.
    std::vector<uint32_t> var_of;                      // internal variable -> caller's variable id
    std::vector<uint32_t> row_of;                      // internal row -> caller's row (request order)
    std::vector<uint32_t> slot_row, slot_col;          // internal Jacobian slot -> caller's (row, variable id)
    std::vector<uint32_t> lvl_cptr;                    // per (partition, level): first internal variable eliminated there
    std::vector<uint32_t> lvl_sptr;                    // per (partition, level): first offdiag slot
    std::vector<uint32_t> l_col;                       // per offdiag slot: var of its column
    std::vector<uint32_t> lpair_ptr, lpairs;           // per offdiag slot: (slot_ik, slot_jk)*
    std::vector<uint32_t> fwd_ptr, fwd_items;          // per var j: (slot(j,k), var k)*  -- row of L
    std::vector<uint32_t> bwd_ptr, bwd_items;          // per var j: (slot(i,j), var i)*  -- column of L
    // per (partition, level): lanes that share one list of the level's Cholesky walks (a power of two; 1 = one lane
    // per list).  Filled by the launch-shape code (records.cpp: choose_level_groups); empty means 1 everywhere.
    std::vector<uint32_t> lvl_grp;
    // Dense phases (records.cpp: make_dense_phases; one-partition programs of one connected component on a barrier
    // workgroup): the top of the elimination tree -- its last n_dense "levels", [dense_level0, nlev) -- is not walked
    // column by column.  Each of these levels is a PHASE: a few whole levels of the original schedule merged, whose columns
    // fall into <= 8 independent blocks (the connected pieces of the elimination tree inside the phase, <= 16 columns
    // each; the last phase is the root block).  A block is a dense panel: rows = its columns, then the later columns
    // that have entries in them, then b; the lists of its entries keep only the terms of columns before the phase.
    //   dense_col[j - lvl_cptr[dense_level0]]  = block | local column << 4                      (columns of all phases)
    //   dense_slot[s - lvl_sptr[dense_level0]] = block | local column << 4 | local row << 8     (their strictly-lower slots)
    //   dense_tab = [n_dense] [offset of phase p's record]... ; record = [blocks] then per block [columns K, rows R (incl.
    //   b), LDS offset (doubles, from the first block's), row stride, offset in dense_tab of the R - 1 row variables] ...
    uint32_t n_dense = 0, dense_level0 = 0, dense_lds_doubles = 0;
    std::vector<uint32_t> dense_col, dense_slot, dense_tab;
};

struct BuildError {
    int code = EZPZ_OK;
    int32_t constraint = -1;
    int64_t variable = -1;
    std::string message;
};

// residual_dim, constraints.rs:954-993
int residual_dim(uint16_t kind);
// number of ids a kind uses
int kind_num_ids(uint16_t kind);

// Builds the program.  `want_parts` > 1 asks for that many balanced partitions (one per wavefront of a
// workgroup team); if the components cannot be balanced the program comes back with a single partition.
// Returns false and fills `err` on MissingGuess / bad ids / size limits.
// `dense` allows the dense factor layout (granted when JtJ is at least 60 % full; see ProgramCounts::dense): tiny
// systems, one partition: L keeps every strictly-lower entry, column by column in elimination order
// (slot of column j: lvl_sptr[j] + (i - j - 1)), every column is its own level, and no Cholesky / substitution
// lists are emitted -- the kernel factorises with plain loops over i, j instead of walking lists level by level.
// `serial`: the factorisation will be walked by ONE lane (lanes across the batch): the elimination order is chosen for
// the fewest entries of L alone, not for few levels.
bool build_program(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, Program& out, BuildError& err,
                   uint32_t want_parts = 1, bool dense = false, bool serial = false);

// 64-bit topology hash (kinds, tags, ids; not params/weights) for the host-side program cache.
uint64_t topology_hash(const EzpzConstraint* cs, size_t n_cs, size_t n_vars);

}  // namespace ezpz
