// The launch-shape thresholds in ONE place (include/ezpz_amd.h: EzpzLaunchPolicy / ezpz_launch_policy).
//
// Which kernel serves a call is decided from the topology (shape.cpp: analyze_into) and from the size of the call
// (launch.hip: launch).  Every number below was measured on one 256-CU MI355X; the ones that are "how many systems fill the
// device" are stored per compute unit and scaled by the CU count of the device the system lives on (a partitioned MI355X
// -- CPX / DPX -- or a CU-masked process sees fewer), the others are properties of one workgroup / one CU's LDS and do not
// scale.  tests/test_abi_cpu.py pins the table (the 256-CU values and the scaling).
#pragma once
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <cstdint>

#include "../../include/ezpz_amd.h"

namespace ezpz {

// Diagnostics on stderr, one switch for all of them: EZPZ_DEBUG=<topic>[,<topic>...] or EZPZ_DEBUG=all -- topics: hip (failing
// runtime calls), comp (why a system got no component plan), jit (compilation logs, the wavefront kernel's elimination), lanes
// (the lanes-across-the-batch operation stream), front (the frontal plan), dense (dense phases), rec (the record walk), h2h (the
// host-to-host pipeline).  (Round 5 had eight environment switches for this.)
inline bool debug_topic(const char* topic) {
    const char* e = std::getenv("EZPZ_DEBUG");
    if (!e || !*e) return false;
    if (!std::strcmp(e, "all") || !std::strcmp(e, "1")) return true;
    const size_t n = std::strlen(topic);
    for (const char* p = e; (p = std::strstr(p, topic)) != nullptr; p += n)
        if ((p == e || p[-1] == ',') && (p[n] == 0 || p[n] == ',')) return true;
    return false;
}


inline EzpzLaunchPolicy launch_policy_for(int compute_units) {
    const uint64_t cus = (uint64_t)std::max(compute_units, 1);
    EzpzLaunchPolicy p{};
    p.compute_units = (uint32_t)cus;
    // ---- scaled by the CU count ------------------------------------------------------------------------------------------
    // lanes across the batch (batch_kernel.hip.hpp) pay once the batch gives every SIMD a wavefront or two: 64 lanes x 4 (2)
    // wavefronts x CUs systems per call; measured at 256 CUs: 32 768 systems of 300 variables 2.0 M solves/s against the
    // teams' 1.5 M (and 3.1 M since the teams walk records: hence 65 536 up to 600 variables), 500 variables 0.69 / 1.27 M at
    // 32 768 / 65 536 against the teams' 0.61-0.91 M
    p.lanes_min_systems_small = 64 * 4 * cus;
    p.lanes_min_systems_large = 64 * 2 * cus;
    p.lanes_large_from_vars = 601;
    // a call this large starts the run-time compilation of the topology's kernel at once (it amortises 0.5 s of hiprtc);
    // smaller calls earn it by repetition (jit_after_launches)
    p.jit_lane_min_batch = 16 * cus;   // one lane per system (<= 20 variables): 4096 at 256 CUs
    p.jit_comp_min_batch = 4 * cus;    // component-resident block systems: 1024 at 256 CUs
    p.jit_comp_min_values = 8192 * cus;  // ... or this many variables in the call (2^21 at 256 CUs: 1049 systems of 2000)
    // ---- per workgroup / per CU: not scaled -----------------------------------------------------------------------------
    p.jit_after_launches = 256;     // an interactive sketch solved again and again earns its kernel (ezpz-cli makes 101 solves)
    p.lane_max_vars = 20;           // one lane per system, state in registers (comp_program.cpp: lane_plan_build)
    p.lane_max_constraints = 40;
    p.comp_min_components = 128;    // fewer components than two wavefronts' lanes: the other shapes serve
    p.comp_max_component_vars = 24;
    p.comp_max_component_constraints = 48;
    p.comp_max_classes = 32;
    // one connected system walks records (records.cpp: build_records) instead of level lists from this many variables: one
    // solve of 32 / 50 / 64 variables 61 -> 54, 142 -> 112, 119 -> 82 us; batches of 64 variables 17.0 -> 23.1 M solves/s but
    // of 50 variables 19.2 -> 16.8 M
    p.rec_min_vars_one_solve = 25;
    p.rec_min_vars_batch = 57;
    p.rec_one_wavefront_max_vars = 160;  // batches: one wavefront per system up to here, then 128-512 lanes
    p.rec_max_components = 127;          // a few components of one system walk records as one partition
    p.rec_wide_one_solve_max_vars = 3500;  // one solve walks wide (global-memory) records up to here
    p.sub_team_max_width = 64;           // max(constraints, variables) up to which sub-wavefront teams are considered
    p.dense8_max_vars = 8;               // register-resident 8 x 8 solve on teams of four lanes
    // calls moving less than this go through mapped host memory (no DMA descriptor); more: pageable 16 MB chunks, or the
    // three-stage pipeline on registered buffers
    p.zero_copy_max_bytes = 1u << 20;
    p.h2h_piece_min_bytes = 4u << 20;
    p.h2h_piece_max_bytes = 16u << 20;
    p.h2h_pieces_per_call = 16;
    p.one_call_host_mask_max_constraints = 256;  // ezpz_solve: unsatisfied mask / warning log straight to mapped host memory up to here
    p.one_call_host_log_max_entries = 8192;
    // the frontal shape (fronts.cpp): one solve of a connected sketch from 48 variables; a system created for batches carries
    // the plan from the same size and takes it for SMALL calls: as many systems as the device holds at once (CUs / workgroups per
    // system), times one ROUND per front_small_call_wgs_per_round workgroups per system -- the more workgroups a system is spread
    // over, the further ahead of the record walk its solve is (one solve, fronts vs records: 6 workgroups x1.9, 14 x3.6, 22 x9.4;
    // profiles/r05_small_calls.txt); device-filling batches stay on the record walk / the lanes, which need half the
    // instructions per solve
    p.front_min_vars_one_solve = 48;
    p.front_min_vars_batch = 48;
    p.front_small_call_wgs_per_round = 4;
    p.front_vars_per_workgroup = 160;
    p.front_max_workgroups = 64;  // (20 000 variables need 48 to fit their shares into the LDS)
    return p;
}

}  // namespace ezpz
