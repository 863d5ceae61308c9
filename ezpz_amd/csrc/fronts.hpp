// Host side of the FRONTAL launch shape (front_types.hpp): the symbolic phase that turns one tier of constraints into a
// tree of dense fronts -- the counterpart of faer's SymbolicLlt when it goes supernodal (reference ezpz/src/solver.rs:289-300).
// Nothing here depends on values; computed once per topology.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/ezpz_amd.h"
#include "front_types.hpp"

namespace ezpz {

struct FrontOptions {
    uint32_t wgs = 1;               // workgroups per system; 0 = choose from the size of the system
    uint32_t max_wgs = 32;          // ... at most
    uint32_t vars_per_wg = 160;     // ... one workgroup up to twice this many variables, then one more per this many
    size_t lds_bytes = 160 * 1024;  // LDS one workgroup may take (tables + workspace)
    uint32_t threads = 512;         // lanes per workgroup
    // elimination order per connected component: 0 = nested dissection (balanced trees: parallel fronts, workgroups), 1 = minimum
    // degree (small fronts on bushy graphs whose breadth-first levels make wide separators); front_plan_build tries 0, then 1
    uint32_t ordering = 0;
};

struct FrontPlan {
    uint32_t n_vars = 0, n_cons = 0, n_rows = 0, zj = 0;
    uint32_t n_wgs = 1;
    bool unit_weights = true, linear_only = true;
    // device blob: FrontWg[n_wgs] at offset 0, then every workgroup's arrays (offsets inside FrontWg)
    std::vector<unsigned char> blob;
    uint32_t n_chunks = 0;        // 16-byte chunks of scratch per system in flight (n_wgs > 1): update matrices that cross
                                  // workgroups, then ...
    uint32_t bad_chunk0 = 0;      // ... one flag per workgroup (a pivot of its fronts was not positive), the steps of the exported
    uint32_t verdict_chunk = 0;   // variables, and workgroup 0's verdict on the whole factorisation
    uint32_t ws_doubles_max = 0;  // largest workspace of a workgroup
    uint32_t tab_bytes_max = 0;   // largest staged table block
    uint32_t threads = 512;
    size_t lds_bytes = 0;         // dynamic LDS of the launch: tables + workspace + reduction scratch
    // statistics (EzpzSystemInfo, tools)
    uint32_t n_fronts = 0, n_levels = 0, max_rows = 0, max_pivots = 0;
    uint32_t n_components = 0;  // connected components of the variable graph
    uint32_t ordering = 0;      // the elimination order the plan was made with (FrontOptions::ordering)
    uint64_t panel_doubles = 0, update_doubles = 0, fill_zeros = 0;
    double model_cycles = 0.0;  // the planner's own estimate of one factorisation + substitution on the critical path
};

// False when the shape does not apply: a front of more than kFrontMaxRows rows, a workgroup's share that does not fit its
// LDS or 16-bit indices, ...; `why` (optional) says which.
bool front_plan_build(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, const FrontOptions& opt, FrontPlan& out,
                      const char** why = nullptr);

}  // namespace ezpz
