// Plain-data types shared by the host launch code and the device kernels of the list-walk / record-walk teams
// (lm_kernel.hip.hpp) and of FreedomAnalysis (freedom.hip.hpp): the view of a topology program in device memory, the
// kernel's argument block, the scratch of a grid team.  No device code here: host translation units that never launch a
// kernel (shape.cpp, records.cpp, pipeline.cpp) include this without the kernels.
#pragma once
#include <hip/hip_runtime.h>  // (uint2 / uint4)

#include <cstdint>

#include "dev_types.hpp"

namespace ezpz {

enum TeamMode { MODE_SUB = 0, MODE_PART = 1, MODE_WGB = 2 };

// Byte offsets of every list inside the program blob (one device allocation, optionally copied to LDS).
struct ProgramView {
    const unsigned char* base;
    uint32_t o_cons, o_parts;
    uint32_t o_colj_ptr, o_colj_items;
    uint32_t o_apair_ptr, o_apairs;
    uint32_t o_lvl_cptr, o_var_of, o_lvl_sptr, o_l_col, o_lvl_grp;
    uint32_t o_lpair_ptr, o_lpairs;
    uint32_t o_fwd_ptr, o_fwd_items;
    uint32_t o_bwd_ptr, o_bwd_items;
    uint32_t o_dense_col, o_dense_slot, o_dense_tab;  // dense phases (Program::dense_col ...)
    uint32_t o_pos, o_weights, o_patterns;  // side arrays of a packed constraint table
    uint32_t packed;                        // constraint table holds 32-byte PackedCon records
    // Per-level stream of the Cholesky lists (32-bit programs of one partition; see pack_program): level lv is words
    // [lvl_off[lv], lvl_off[lv + 1]) of lvl_stream, a block that a team copies to LDS in one go.
    uint32_t o_lvl_off, o_lvl_stream, lvl_words_max;
    uint32_t o_lvl_boff, o_lvl_bstream;  // the same for the backward substitution: [n_items, 0] [bwd_ptr rel] [bwd_items]
    uint32_t blob_bytes;
    uint32_t stage_bytes;  // leading bytes of the blob to copy into LDS (index lists, or the whole blob); 0 = none
    uint32_t n_cons, n_vars, n_rows, zj, zlo, n_parts;
};

// ---- grid teams (lm_kernel.hip.hpp): one system on several workgroups --------------------------------------------------
constexpr int kGridMaxWgs = 256;
typedef unsigned int gridchunk_t __attribute__((ext_vector_type(4)));  // (value lo, value hi, seq, 0)
struct GridScratch {
    int nwarn[2];                           // Degenerate-warning counters, by parity of the system's index in this slot
    int dead;                               // a rendezvous timed out (some workgroup of the team never became resident)
    int pad[13];
    gridchunk_t arr[2][2][kGridMaxWgs];     // [parity of the sequence number][value][workgroup]: partials
    gridchunk_t out[2][kGridMaxWgs][4];     // [parity][workgroup]: (a, seq), (b, seq) in one 64-byte line
};

struct SolveArgs {
    ProgramView p;
    const double* x0;
    double* x_out;
    EzpzStatus* status;
    uint8_t* unsat_mask;   // optional
    uint64_t* warn_log;    // optional
    double* gws;           // global workspace (LDSWS=false), ws_doubles per workgroup
    uint64_t batch;
    uint32_t warn_cap;
    uint32_t ws_doubles;        // doubles per team workspace (incl. the small int area, rounded to 2 doubles)
    uint32_t prog_lds_doubles;  // LDS doubles reserved for the staged program (PLDS), 0 otherwise
    uint32_t max_iterations;
    uint32_t unit_weights;  // every constraint weight == 1.0 (the common case): weighted r == unweighted r
    double residual_tolerance, step_tolerance, initial_lambda;
    unsigned long long* stamps;  // diagnostic builds (-DEZPZ_STAMPS) only: (id, s_memtime) pairs of block 0, lane 0
    GridScratch* grid_scratch;   // grid teams: one per system in flight
    const ProgramView* grid_views;  // grid teams: the sub-program of each workgroup of a system
    uint32_t grid_wgs;           // workgroups per system (1 = every other team shape)
    // level staging (see the Cholesky loop): LDS offset (doubles) of the level tables, words reserved for the tables,
    // words of one level buffer (0 = off)
    uint32_t lvl_lds_off, lvl_tab_words, lvl_buf_words;
    // dense phases (Program::n_dense; barrier workgroups): how many, the first of their levels, LDS offset (doubles) of the
    // blocks' panels and the doubles they take together; n_dense == 0: none
    uint32_t n_dense, dense_level0, dense_lds_off, dense_lds_doubles;
    // an indirect batch (optional): the systems to solve are sys_list[0 .. min(*sys_count, batch)), indices into x0 / x_out /
    // status -- the stragglers a lanes-across-the-batch launch handed over (batch_kernel.hip.hpp); the count is only known
    // on the device, `batch` is the list's capacity
    const uint32_t* sys_list;
    const uint32_t* sys_count;
    // ... which may be RESUMED rather than solved from their guesses: entry q of `resume` carries the LM state the lanes
    // kernel had reached for system sys_list[q] (whose current values it left in x_out): eval() runs at those values, its
    // warnings are not logged again, and the loop goes on with that lambda, iteration count and pass number
    const LmResume* resume;  // (dev_types.hpp)
    // record walk (REC builds; records.cpp: build_records): the linear solve of one connected system as a sequence of ROUNDS.
    // rec_desc[round * wavefronts + wavefront] = (flags, first chunk): what that wavefront does in that round (copied to
    // LDS once per workgroup, rec_desc_off doubles in); rec_chunks[(chunk + c) * 64 + lane of the wavefront]: the lanes'
    // records of a working wavefront, one to three 16-byte chunks each -- chunk 0 = target | diagonal << 16, destination
    // | lane flags << 16 and two (a | b << 16) operand pairs, the others four pairs each; all addresses count doubles from
    // the start of the LDS.  rec_dd_delta: from an entry's diagonal A_jj to where 1 / d_j goes (the factor's diagonal has
    // its own n doubles behind the workspace proper, then one double that stays zero: the operand of padding pairs).
    const uint2* rec_desc;
    const uint4* rec_chunks;
    uint32_t rec_rounds, rec_dd_delta, rec_zero, rec_desc_off;
    // rec_jglobal (LDS form, batches): the Jacobian's values live in global memory (gws + workgroup x rec_jstride doubles), not in
    // the workspace -- they are written once per accepted step by the sweep and read once per iteration by the packed assembly,
    // and their zJ doubles are a fifth of a system's LDS: one more workgroup per CU
    uint32_t rec_jglobal, rec_jstride;
    // packed assembly (REC builds; 0 chunks = the lists are walked): the Jacobian-slot pairs of every column of A (a = J slot, b
    // = row of r) and of every entry of its strict lower part (a, b = J slots) as rec_asm_kc / rec_asm_ks 16-byte chunks of four
    // (a | b << 16) pairs per item, chunk k of item i at [k * items + i] -- one coalesced request per chunk instead of pointer,
    // items, values; padded with pairs of the zero
    const uint4* rec_asm_cols;
    const uint4* rec_asm_slots;
    uint32_t rec_asm_kc, rec_asm_ks;
    DoneWord done;  // one-call launches: the completion word (dev_types.hpp), else null
};

// a round's descriptor (per wavefront): chunks to load (0 = the wavefront has no item), log2 of the lanes per list, ...
constexpr uint32_t REC_NCH_MASK = 7u, REC_LG_SHIFT = 3u, REC_BARRIER = 1u << 7, REC_BWD = 1u << 8;
// ... and a lane's own flags (high half of the record's second word)
constexpr uint32_t REC_WRITER = 1u << 16, REC_ISCOL = 1u << 17;
constexpr int REC_MAX_CHUNKS = 3, REC_MAX_PAIRS = 10;  // per lane and round: 2 pairs beside the header, then 4 + 4
// The WIDE form (REC == 2: the workspace in global memory, 32-bit addresses counted from its start): chunk 0 = target, diagonal,
// destination, lane flags; up to four more chunks of two (a, b) pairs each.
constexpr int REC_WIDE_CHUNKS = 5, REC_WIDE_PAIRS = 8;


// ---- FreedomAnalysis (freedom.hip.hpp) ---------------------------------------------------------------------------------------
struct FreedomComp {
    uint32_t m, n;          // rows / variables of the component
    uint32_t item0, item1;  // its Jacobian slots: items[2i] = internal slot, items[2i+1] = lcol * m + lrow
    uint32_t var0;          // comp_vars[var0 + lcol] = caller's variable id
    uint32_t pad;
};

}  // namespace ezpz
