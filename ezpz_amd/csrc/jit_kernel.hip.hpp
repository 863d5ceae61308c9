// Class-specialised component-resident LM kernel: the skeleton.
//
// comp_kernel.hip.hpp interprets a class program record by record; for batches large enough to pay for a run-time
// compilation (hiprtc, jit.cpp) the same plan is compiled into straight-line code instead.  The host emits, per
// isomorphism class of components, a small struct (NV, M, ..., and three functions: residuals, jacobian, solve) whose
// bodies are the class program with every index a literal -- the same operations in the same order as the interpreter
// and the list-walk kernels, so results stay bit-identical -- and one line naming the sequence of classes a wavefront
// handles.  This header, compiled together with that text, is everything else: it is hand-written and fixed.
//
// What the specialisation buys on gfx950: a lane keeps the whole state of its components (x, d, r, r_next, Jacobian
// values, constraint parameters, the caller's variable ids) in VGPRs -- arrays indexed by literals only -- so there is
// no LDS traffic and no address arithmetic at all, the scalar unit no longer decodes records, and with no LDS
// footprint several workgroups share a CU.  LDS holds only the reduction scratch of the LM control
// (newton.rs:50-60,:96-139): one rendezvous per LM iteration, exactly as in comp_kernel.hip.hpp.
// Replaces the same reference code as comp_kernel.hip.hpp (ezpz/src/solver/newton.rs:29-145, solver.rs:318-440,
// lib.rs:305-327).
#pragma once
#include "constraint_eval.hip.hpp"
#include "wave_ops.hip.hpp"

namespace ezpz {
namespace jit {

struct JitArgs {
    const uint32_t* blob;  // the component plan's blob (per-class tables of ids / parameters / positions)
    uint32_t o_slots;      // word offset of the slot table: [wave][slot] {ids_off, par_off, pos_off, count}
    uint32_t n_row;        // values per system in x0 / x_out
    uint32_t n_cons;       // constraints per system (unsat mask row)
    uint32_t n_rows_total;
    const double* x0;
    double* x_out;
    EzpzStatus* status;
    uint8_t* unsat_mask;  // optional
    uint64_t* warn_log;   // optional
    uint32_t warn_cap;
    uint32_t max_iterations;
    uint64_t batch;
    double residual_tolerance, step_tolerance, initial_lambda;
    struct GridScratch* grid;  // systems too large for one workgroup: one scratch per system in flight (else null)
    uint32_t grid_wgs;         // workgroups that share a system (1 = the ordinary case)
    uint32_t o_ranges;         // (the `_fast` entries of a plan whose wavefronts own contiguous pieces of a row: word offset of [wave] {first variable, count}; else 0)
    DoneWord done;             // one-call launches: the completion word (dev_types.hpp)
    // Batches larger than the launch (one workgroup per system): after its first system (its own index) a workgroup DRAWS the
    // next one from this counter -- system = workgroups of the launch + (drawn value - ticket_base) -- instead of striding.  The
    // launch holds as many workgroups as the device has room for; when something else occupies one of those places (another
    // stream's kernel, a resident one-call kernel of this library) the workgroup that does not fit starts when the others END,
    // and with a fixed share of the batch it then ran alone for as long again: 100.9 -> 53.7 M solves/s beside one resident
    // one-call kernel (profiles/r05_resident_cost.txt).  Drawn, its share is one system.  Null: strides.
    // Eight counters, kTicketStride words apart, workgroup b on counter b % 8 (its XCD's, as workgroups are dealt out), value t of
    // counter c = system `workgroups + 8 t + c`: ONE counter answered 768 workgroups 84 M times a second and no faster -- 100.5 ->
    // 83.7 M solves/s.  (32 bits, modular: a launch draws fewer than 2^32 values.)
    unsigned int* ticket;
    unsigned int ticket_base[8];
    // diagnostic (tools/ladder_stamps.py; null otherwise): a solve_kernel_grid_fast compiled with EZPZ_JIT_STAMPS leaves wall-clock
    // stamps of thread 0 of every workgroup here, 16 per (system, workgroup)
    unsigned long long* stamps;
    // a system on several workgroups: the systems solve_kernel_grid_fast could not finish -- redo[0] of them, redo[1], ... -- which it
    // appends to and the launch of solve_kernel_grid behind it solves (null: that kernel solves the whole batch)
    unsigned int* redo;
    // ... and the list of the system's NEXT call (the host alternates between two), whose count the launch of solve_kernel_grid
    // zeroes: it was read for the last time by the call before this one (a memset per call was a 5 us kernel of its own).
    // redo_seen: a word of mapped host memory that launch leaves its count in -- the host's hint for how wide to launch it next time
    unsigned int* redo_next;
    unsigned int* redo_seen;
};
constexpr unsigned int kTicketStride = 1024;  // words between two counters (4 KB: another channel)
static_assert(sizeof(JitArgs) == 248, "JitArgs is restated on the host (jit.cpp: JitArgsHost)");

// One system on several workgroups ("grid team", as in lm_kernel.hip.hpp): every workgroup owns its wavefronts' slots;
// the reductions of the LM control cross workgroups through this per-system scratch.  Every workgroup publishes its
// partials; workgroup 0 alone polls them, folds them in a fixed tree and writes the results to one 64-byte line per
// workgroup; every other workgroup polls only its own line.  A value travels as a 16-byte (value, sequence number)
// chunk moved by one device-coherent (sc0 sc1) 128-bit access, so it validates itself: no atomics, no fences.  All
// workgroups of a launch must be resident at once (the host sizes the launch; the spin is bounded all the same).
constexpr int kGridMaxWgs = 256;
constexpr int kRedDoubles = 160;  // Red::buf: two turns of up to 5 x 16 partials; flag words and the resident word follow (smem[kRedDoubles + 16])
typedef unsigned int gridchunk_t __attribute__((ext_vector_type(4)));  // (value lo, value hi, seq, 0)
struct GridScratch {
    int nwarn[2];  // Degenerate-warning counters, by parity of the system's turn in this slot
    int dead;      // a rendezvous timed out
    int pad[13];
    gridchunk_t arr[2][4][kGridMaxWgs];  // [parity of the sequence number][value][workgroup]: partials
    gridchunk_t out[2][kGridMaxWgs][4];  // [parity][workgroup]: the four results in one 64-byte line
    // systems whose verdicts are not waited for (solve_kernel_grid): a ring of four systems in flight per slot
    gridchunk_t ring_p[4][kGridMaxWgs][8];  // [sequence number & 3][workgroup][value]: a workgroup's eight partials, one 128-byte line
    gridchunk_t ring_v[4][8];               // [sequence number & 3][0]: workgroup 0's verdict on that system (a line each)
};
static_assert(sizeof(GridScratch) == 197184, "GridScratch is sized on the host (comp_program.hpp: kJitGridScratchBytes)");

__device__ __forceinline__ void grid_store(gridchunk_t* p, double v, unsigned int seq) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    gridchunk_t c;
    c.x = (unsigned int)u;
    c.y = (unsigned int)(u >> 32);
    c.z = seq;
    c.w = 0;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(c) : "memory");
}
__device__ __forceinline__ gridchunk_t grid_peek(const gridchunk_t* p) {
    gridchunk_t c;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(c) : "v"(p) : "memory");
    return c;
}
// When a wait gives up: every 1024th look, after 2^21 looks or as soon as anybody else has given up (the caller then raises the flag
// itself, which makes the whole system's status EZPZ_ITERATIONS_TEAM_TIMEOUT).  A macro, and the spin loops stay written out where
// they are (they differ in how many chunks they request together): this kernel's schedule notices helper functions -- the same test
// as an inline function with the store inside, and the chunk's value through one, cost the ladder 1 % on one box (428.5-429.9 vs
// 423.7-426.1 k solves/s); a generalised form of the reductions 8 % (DESIGN_HISTORY.md).
#define EZPZ_GRID_TIMED_OUT(spins, dead) \
    (((spins) & 1023u) == 1023u && ((spins) >= (1u << 21) || __hip_atomic_load((dead), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
__device__ __forceinline__ double grid_wait(const gridchunk_t* p, unsigned int seq, int* dead) {
    gridchunk_t c;
    for (unsigned int spins = 0;; ++spins) {
        c = grid_peek(p);
        if (c.z == seq) break;
        if (EZPZ_GRID_TIMED_OUT(spins, dead)) {
            __hip_atomic_store(dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return __builtin_nan("");
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return __builtin_bit_cast(double, ((unsigned long long)c.y << 32) | c.x);
}
// One component per lane of one class: everything in registers (every index below is a literal after inlining).
template <class C>
struct Slot {
    double x[C::NV], d[C::NV];
    double xn[C::NV];                     // (solve_kernel_grid_fast: the NEXT system's guesses, on their way while this one is solved)
    double F[2][C::NF];                   // (solve_kernel_grid_fast: a linear class's factorisation at the two lambdas, wave-uniform)
    bool fbad[2], fok[2];
    double r[C::M > 0 ? C::M : 1], rn[C::M > 0 ? C::M : 1];
    double J[C::ZJS > 0 ? C::ZJS : 1];    // Jacobian values (classes with a non-linear member; constant otherwise)
    double par[C::NC > 0 ? C::NC : 1];    // constraint parameters of this lane's instance
    uint32_t ids[C::NV];                  // the caller's ids of its variables
    const uint32_t* pos;                  // the caller's positions of its constraints: pos[ci * STRIDE]
    unsigned long long wmask;             // degenerate evaluations of the speculative residual sweep
    bool active;
    bool exact;                           // this lane's solve of the current iteration must be repeated with plain divisions
};

template <class P>
struct class_of;
template <class C>
struct class_of<C*> {
    typedef C type;
};

// The sequence of slots a wavefront owns, as a compile-time list.
template <class... Cs>
struct Slots;
template <>
struct Slots<> {
    static constexpr int N = 0;
    static constexpr int NVS = 0;
    template <class F>
    __device__ __forceinline__ void each(F&&, int = 0) {}
};
template <class A, class B>
struct same_class {
    static constexpr bool value = false;
};
template <class A>
struct same_class<A, A> {
    static constexpr bool value = true;
};
template <class C, class... Rest>
struct Slots<C, Rest...> {
    static constexpr int N = 1 + sizeof...(Rest);
    static constexpr int NVS = C::NV + Slots<Rest...>::NVS;  // variables per lane over all the slots
    Slot<C> head;
    Slots<Rest...> tail;
    template <class X>
    __device__ __forceinline__ Slot<X>& first() {  // the first slot of class X (what a class's slots share lives there)
        if constexpr (same_class<X, C>::value)
            return head;
        else
            return tail.template first<X>();
    }
    template <class F>
    __device__ __forceinline__ void each(F&& f, int index = 0) {
        f(head, static_cast<C*>(nullptr), index);
        tail.each(f, index + 1);
    }
};

__device__ __forceinline__ double uniform(double v) {  // a value every lane holds -> scalar registers
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)u);
    const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Division by a column's diagonal, shared by every numerator of the column.  The compiler expands `n / D` into
//   div_scale x 2, y0 = rcp(D), two Newton steps on y, q0 = n * y, e = fma(-D, q0, n), q = div_fmas(e, y, q0), div_fixup
// -- correctly rounded, 11 instructions, one of them quarter rate.  Everything before q0 depends on D alone whenever
// div_scale leaves both operands as they are, and div_fmas / div_fixup are then a plain fma and the identity.  recip_of
// is that denominator part, once per column (and, for a class of linear constraints, once per workgroup: D is a
// function of lambda there, and the compiler hoists it out of the slots); div_by is the three-instruction tail.  `ok`
// stays true while every operand was in the range where div_scale does not scale and div_fixup has nothing to fix
// (V_DIV_SCALE_F64: denominator normal and its reciprocal normal, exponent(n) - exponent(D) in (-1022, 768),
// exponent(n) > 53); the caller repeats the solve with plain divisions for the lanes where it is false, so results are
// the correctly rounded quotients -- bit for bit what the interpreters and a CPU `n / D` compute -- in every case.
// Zero numerators stay on the short path: with e = fma(D, q0, -n) and q = fma(-e, y, q0) a zero keeps the sign of n
// (D > 0), as div_fixup would give it.
__device__ __forceinline__ double recip_of(double D, bool& ok) {
    ok = ok && (D >= 0x1p-40) && (D <= 0x1p+40);  // (false for NaN: a failed pivot)
    double y = __builtin_amdgcn_rcp(D);
    double e = __builtin_fma(-D, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-D, y, 1.0);
    y = __builtin_fma(y, e, y);
    return y;
}
__device__ __forceinline__ double div_by(double n, double D, double y, bool& ok) {
    const double a = __builtin_fabs(n);
    ok = ok && !(a > 0x1p+600) && (!(a < 0x1p-900) || n == 0.0);  // (NaN passes: the quotient is NaN either way)
    const double q0 = n * y;
    const double e = __builtin_fma(D, q0, -n);
    return __builtin_fma(-e, y, q0);
}

// A copy of `a` the optimiser cannot relate to the original: the fallback's inputs.  Without it the sums the two
// solves have in common (the normal equations) are formed once and kept in registers across the short solve for a
// fallback that almost never runs (`square`: 228 -> 332 VGPRs, one wavefront per SIMD instead of two).
template <int N>
__device__ __forceinline__ void opaque_copy(const double (&a)[N], double (&b)[N]) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double v = a[k];
        asm volatile("" : "+v"(v));
        b[k] = v;
    }
}

// Workgroup reductions with one barrier (the same scheme as CompRed in comp_kernel.hip.hpp).
struct Red {
    double* buf;  // 2 x 3 x 16 doubles
    int* flags;   // 3 words
    int flip, turn;
    GridScratch* grid;  // null: the workgroup owns its system alone
    uint32_t grid_wgs, grid_wg;
    unsigned int grid_seq;
    // step()'s three values (a sum, two maxima) over the workgroups of a system, its flag riding in the spare word of the first chunk:
    // every thread passes the workgroup's values, every thread gets the system's.  Three chunks per workgroup to publish, gather and
    // scatter (a grid reduction costs by the lines it touches)
    __device__ __forceinline__ bool across_workgroups3(double& s0, double& m1, double& m2, bool flag, int lane, uint32_t wave, uint32_t nwaves) {
        using namespace ezpz::dev;
        const int tid = threadIdx.x;
        const unsigned int seq = ++grid_seq;
        const unsigned int par = seq & 1u;
        auto put = [&](gridchunk_t* p, double v, unsigned int w) {
            const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
            gridchunk_t c;
            c.x = (unsigned int)u, c.y = (unsigned int)(u >> 32), c.z = seq, c.w = w;
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(c) : "memory");
        };
        if (tid < 3) put(&grid->arr[par][tid][grid_wg], tid == 0 ? s0 : tid == 1 ? m1 : m2, tid == 0 && flag ? 1u : 0u);
        double* b = buf + (flip ? 48 : 0);
        flip ^= 1;
        bool failed;
        if (grid_wg == 0) {
            double a0 = 0.0, a1 = __builtin_nan(""), a2 = a1, a3 = a1;
            for (uint32_t g = tid; g < grid_wgs; g += blockDim.x) {
                gridchunk_t c0, c1, c2;
                for (unsigned int spins = 0;; ++spins) {
                    asm volatile(
                        "global_load_dwordx4 %0, %3, off sc0 sc1\n\t"
                        "global_load_dwordx4 %1, %4, off sc0 sc1\n\t"
                        "global_load_dwordx4 %2, %5, off sc0 sc1\n\t"
                        "s_waitcnt vmcnt(0)"
                        : "=&v"(c0), "=&v"(c1), "=&v"(c2)
                        : "v"(&grid->arr[par][0][g]), "v"(&grid->arr[par][1][g]), "v"(&grid->arr[par][2][g])
                        : "memory");
                    if (c0.z == seq && c1.z == seq && c2.z == seq) break;
                    if (EZPZ_GRID_TIMED_OUT(spins, &grid->dead)) {
                        __hip_atomic_store(&grid->dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned long long nan = __builtin_bit_cast(unsigned long long, __builtin_nan(""));
                        c0.x = c1.x = c2.x = (unsigned int)nan, c0.y = c1.y = c2.y = (unsigned int)(nan >> 32);
                        c0.w = 0;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                a0 = a0 + __builtin_bit_cast(double, ((unsigned long long)c0.y << 32) | c0.x);
                a1 = fmax_nc(a1, __builtin_bit_cast(double, ((unsigned long long)c1.y << 32) | c1.x));
                a2 = fmax_nc(a2, __builtin_bit_cast(double, ((unsigned long long)c2.y << 32) | c2.x));
                if (c0.w) a3 = 1.0;
            }
            a0 = reduce_wave_to_last_lane(a0, OpSum());
            a1 = reduce_wave_to_last_lane(a1, OpMax());
            a2 = reduce_wave_to_last_lane(a2, OpMax());
            a3 = reduce_wave_to_last_lane(a3, OpMax());
            if (lane == 63) {
                b[wave] = a0;
                b[12 + wave] = a1;
                b[24 + wave] = a2;
                b[36 + wave] = a3;
            }
            __syncthreads();
            const bool in = (uint32_t)lane < nwaves;
            const int l = lane & 15;
            s0 = uniform(reduce_lanes<16>(in ? b[l] : 0.0, OpSum()));
            m1 = uniform(reduce_lanes<16>(in ? b[12 + l] : __builtin_nan(""), OpMax()));
            m2 = uniform(reduce_lanes<16>(in ? b[24 + l] : __builtin_nan(""), OpMax()));
            failed = uniform(reduce_lanes<16>(in ? b[36 + l] : __builtin_nan(""), OpMax())) > 0.0;
            for (uint32_t g = 1 + tid; g < grid_wgs; g += blockDim.x) {
                put(&grid->out[par][g][0], s0, failed ? 1u : 0u);
                put(&grid->out[par][g][1], m1, 0u);
                put(&grid->out[par][g][2], m2, 0u);
            }
        } else {
            if (tid < 3) {
                gridchunk_t c;
                for (unsigned int spins = 0;; ++spins) {
                    c = grid_peek(&grid->out[par][grid_wg][tid]);
                    if (c.z == seq) break;
                    if (EZPZ_GRID_TIMED_OUT(spins, &grid->dead)) {
                        __hip_atomic_store(&grid->dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned long long nan = __builtin_bit_cast(unsigned long long, __builtin_nan(""));
                        c.x = (unsigned int)nan, c.y = (unsigned int)(nan >> 32), c.w = 0;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                b[tid] = __builtin_bit_cast(double, ((unsigned long long)c.y << 32) | c.x);
                if (tid == 0) b[3] = c.w ? 1.0 : 0.0;
            }
            __syncthreads();
            s0 = uniform(b[0]);
            m1 = uniform(b[1]);
            m2 = uniform(b[2]);
            failed = uniform(b[3]) > 0.0;
        }
        return failed;
    }
    // ... and eval()'s two values alone (a sum, a maximum): two chunks per workgroup
    __device__ __forceinline__ void across_workgroups2(double& s0, double& m1, int lane, uint32_t wave, uint32_t nwaves) {
        using namespace ezpz::dev;
        const int tid = threadIdx.x;
        const unsigned int seq = ++grid_seq;
        const unsigned int par = seq & 1u;
        if (tid < 2) grid_store(&grid->arr[par][tid][grid_wg], tid == 0 ? s0 : m1, seq);
        double* b = buf + (flip ? 48 : 0);
        flip ^= 1;
        if (grid_wg == 0) {
            double a0 = 0.0, a1 = __builtin_nan("");
            for (uint32_t g = tid; g < grid_wgs; g += blockDim.x) {
                gridchunk_t c0, c1;
                for (unsigned int spins = 0;; ++spins) {
                    asm volatile(
                        "global_load_dwordx4 %0, %2, off sc0 sc1\n\t"
                        "global_load_dwordx4 %1, %3, off sc0 sc1\n\t"
                        "s_waitcnt vmcnt(0)"
                        : "=&v"(c0), "=&v"(c1)
                        : "v"(&grid->arr[par][0][g]), "v"(&grid->arr[par][1][g])
                        : "memory");
                    if (c0.z == seq && c1.z == seq) break;
                    if (EZPZ_GRID_TIMED_OUT(spins, &grid->dead)) {
                        __hip_atomic_store(&grid->dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned long long nan = __builtin_bit_cast(unsigned long long, __builtin_nan(""));
                        c0.x = c1.x = (unsigned int)nan, c0.y = c1.y = (unsigned int)(nan >> 32);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                a0 = a0 + __builtin_bit_cast(double, ((unsigned long long)c0.y << 32) | c0.x);
                a1 = fmax_nc(a1, __builtin_bit_cast(double, ((unsigned long long)c1.y << 32) | c1.x));
            }
            a0 = reduce_wave_to_last_lane(a0, OpSum());
            a1 = reduce_wave_to_last_lane(a1, OpMax());
            if (lane == 63) {
                b[wave] = a0;
                b[12 + wave] = a1;
            }
            __syncthreads();
            const bool in = (uint32_t)lane < nwaves;
            const int l = lane & 15;
            s0 = uniform(reduce_lanes<16>(in ? b[l] : 0.0, OpSum()));
            m1 = uniform(reduce_lanes<16>(in ? b[12 + l] : __builtin_nan(""), OpMax()));
            for (uint32_t g = 1 + tid; g < grid_wgs; g += blockDim.x) {
                grid_store(&grid->out[par][g][0], s0, seq);
                grid_store(&grid->out[par][g][1], m1, seq);
            }
        } else {
            if (tid < 2) b[tid] = grid_wait(&grid->out[par][grid_wg][tid], seq, &grid->dead);
            __syncthreads();
            s0 = uniform(b[0]);
            m1 = uniform(b[1]);
        }
    }
    // (W: the power of two >= the number of wavefronts, <= 16 -- the fold of the wavefronts' partials stops there; the
    // lanes beyond hold the identity, so the narrower tree gives the bits of the 16-lane one)
    template <int W>
    __device__ __forceinline__ void sum_max(double& s0, double& m1, int lane, uint32_t wave, uint32_t nwaves) {
        using namespace ezpz::dev;
        s0 = reduce_wave_to_last_lane(s0, OpSum());
        m1 = reduce_wave_to_last_lane(m1, OpMax());
        if (nwaves == 1) {  // one wavefront per system: no LDS, no barrier
            s0 = uniform(__shfl(s0, 63, 64));
            m1 = uniform(__shfl(m1, 63, 64));
            return;
        }
        double* b = buf + (flip ? 48 : 0);
        flip ^= 1;
        if (lane == 63) {
            b[wave] = s0;
            b[16 + wave] = m1;
        }
        __syncthreads();
        const bool in = (uint32_t)lane < nwaves;
        const int l = lane & 15;
        s0 = uniform(reduce_lanes<W>(in ? b[l] : 0.0, OpSum()));
        m1 = uniform(reduce_lanes<W>(in ? b[16 + l] : __builtin_nan(""), OpMax()));
        if (grid) across_workgroups2(s0, m1, lane, wave, nwaves);
    }
    // step() with eval()'s two wave totals (e_s: a sum, e_m: a maximum; uniform per wavefront) riding in the same exchange:
    // the system's first iteration then needs no rendezvous of its own for eval() (solve_kernel, FUSE).  One workgroup per
    // system only (no grid teams).  buf holds 2 x 5 x 16 doubles for this form (kRedDoubles).
    template <int W>
    __device__ __forceinline__ bool step5(double& s0, double& m1, double& m2, double& e_s, double& e_m, bool flag, int lane, uint32_t wave,
                                          uint32_t nwaves) {
        using namespace ezpz::dev;
        s0 = reduce_wave_to_last_lane(s0, OpSum());
        m1 = reduce_wave_to_last_lane(m1, OpMax());
        m2 = reduce_wave_to_last_lane(m2, OpMax());
        const bool any = __ballot(flag) != 0;
        if (nwaves == 1) {
            s0 = uniform(__shfl(s0, 63, 64));
            m1 = uniform(__shfl(m1, 63, 64));
            m2 = uniform(__shfl(m2, 63, 64));
            return any;
        }
        double* b = buf + (flip ? 80 : 0);
        flip ^= 1;
        int* f = flags + turn;
        const int next = turn == 2 ? 0 : turn + 1;
        if (lane == 63) {
            b[wave] = s0;
            b[16 + wave] = m1;
            b[32 + wave] = m2;
            b[48 + wave] = e_s;
            b[64 + wave] = e_m;
            if (any) atomicOr(f, 1);
            if (wave == 0) flags[next] = 0;
        }
        turn = next;
        __syncthreads();
        const bool in = (uint32_t)lane < nwaves;
        const int l = lane & 15;
        s0 = uniform(reduce_lanes<W>(in ? b[l] : 0.0, OpSum()));
        m1 = uniform(reduce_lanes<W>(in ? b[16 + l] : __builtin_nan(""), OpMax()));
        m2 = uniform(reduce_lanes<W>(in ? b[32 + l] : __builtin_nan(""), OpMax()));
        e_s = uniform(reduce_lanes<W>(in ? b[48 + l] : 0.0, OpSum()));
        e_m = uniform(reduce_lanes<W>(in ? b[64 + l] : __builtin_nan(""), OpMax()));
        return __builtin_amdgcn_readfirstlane(*f) != 0;
    }
    template <class Op>
    static __device__ __forceinline__ double wave_total(double v, Op op) {  // the wavefront's reduction as a scalar-register value
        v = ezpz::dev::reduce_wave_to_last_lane(v, op);
        const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
        const unsigned int lo = __builtin_amdgcn_readlane((unsigned int)u, 63), hi = __builtin_amdgcn_readlane((unsigned int)(u >> 32), 63);
        return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    }
    template <int W>
    __device__ __forceinline__ bool step(double& s0, double& m1, double& m2, bool flag, int lane, uint32_t wave, uint32_t nwaves) {
        using namespace ezpz::dev;
        s0 = reduce_wave_to_last_lane(s0, OpSum());
        m1 = reduce_wave_to_last_lane(m1, OpMax());
        m2 = reduce_wave_to_last_lane(m2, OpMax());
        const bool any = __ballot(flag) != 0;
        if (nwaves == 1) {
            s0 = uniform(__shfl(s0, 63, 64));
            m1 = uniform(__shfl(m1, 63, 64));
            m2 = uniform(__shfl(m2, 63, 64));
            return any;
        }
        double* b = buf + (flip ? 48 : 0);
        flip ^= 1;
        int* f = flags + turn;
        const int next = turn == 2 ? 0 : turn + 1;
        if (lane == 63) {
            b[wave] = s0;
            b[16 + wave] = m1;
            b[32 + wave] = m2;
            if (any) atomicOr(f, 1);
            if (wave == 0) flags[next] = 0;  // last read two reductions ago, next set after this barrier
        }
        turn = next;
        __syncthreads();
        const bool in = (uint32_t)lane < nwaves;
        const int l = lane & 15;
        s0 = uniform(reduce_lanes<W>(in ? b[l] : 0.0, OpSum()));
        m1 = uniform(reduce_lanes<W>(in ? b[16 + l] : __builtin_nan(""), OpMax()));
        m2 = uniform(reduce_lanes<W>(in ? b[32 + l] : __builtin_nan(""), OpMax()));
        bool failed = __builtin_amdgcn_readfirstlane(*f) != 0;
        if (grid) failed = across_workgroups3(s0, m1, m2, failed, lane, wave, nwaves);
        return failed;
    }
};

// A DevCon with literal fields for the evaluators (the compiler folds the kind switch and indexes x by constants).
__device__ __forceinline__ DevCon mkcon(uint32_t kind, uint32_t tag, uint32_t nrows, uint32_t i0, uint32_t i1, uint32_t i2, uint32_t i3,
                                        uint32_t i4, uint32_t i5, uint32_t i6, uint32_t i7, double weight, double param) {
    DevCon c;
    c.ids[0] = i0, c.ids[1] = i1, c.ids[2] = i2, c.ids[3] = i3, c.ids[4] = i4, c.ids[5] = i5, c.ids[6] = i6, c.ids[7] = i7;
    c.param = param;
    c.weight = weight;
    c.row0 = 0, c.jbase = 0, c.pos = 0;
    c.kind = (uint8_t)kind, c.tag = (uint8_t)tag, c.nrows = (uint8_t)nrows, c.nslots = 0;
    return c;
}

// SEQ: Slots<...> of one wavefront; NWAVES wavefronts share a system; UNIT_W: every weight is 1.  RESIDENT: the entry of
// one-call launches (`<entry>_one`), which publishes its completion word and waits for the calling thread's next request
// (wave_ops.hip.hpp: publish_done, resident_next).  FUSE: eval()'s sums ride in the first iteration's rendezvous.  GRID: the system
// may be spread over several workgroups (the reductions' grid stage is compiled in).  The batch entry is compiled without that loop: everything set up before it would stay
// live across it -- 2000 x 2000 at three wavefronts per SIMD: 39 -> 75 spilled scalar registers, 88 -> 78 M solves/s.
template <class SEQ, int NWAVES, bool ANY_NONLINEAR, bool UNIT_W, bool RESIDENT = false, bool FUSE = false, bool GRID = true>
__device__ __forceinline__ void solve_kernel(const JitArgs& a, double* smem) {
    using namespace ezpz::dev;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((uint32_t)tid >> 6);
    Red red;
    red.buf = smem;
    red.flags = reinterpret_cast<int*>(smem + kRedDoubles);
    red.flip = 0;
    red.turn = 0;
    int* nwarn2 = red.flags + 4;
    if (tid < 8) red.flags[tid] = 0;
    if (NWAVES > 1) __syncthreads();
    // a system on several workgroups (NWAVES > 1 then): workgroup g of the G that share it, slot = which system in flight
    // (GRID false -- what the generator says of a system on one workgroup -- folds every grid branch of the reductions away:
    // code the kernel never runs still costs it its schedule, 92.8 -> 90.4 M solves/s for 70 lines added to one of them)
    const uint32_t grid_wgs = GRID && a.grid ? a.grid_wgs : 1u;
    const uint32_t grid_wg = blockIdx.x % grid_wgs, grid_slot = blockIdx.x / grid_wgs, n_slots = gridDim.x / grid_wgs;
    red.grid = GRID && a.grid ? a.grid + grid_slot : nullptr;
    red.grid_wgs = grid_wgs;
    red.grid_wg = grid_wg;
    red.grid_seq = 0;
    if (red.grid) {  // continue the slot's sequence numbers where the previous launch left them (this workgroup's own)
        const gridchunk_t c0 = grid_peek(&red.grid->arr[0][0][grid_wg]), c1 = grid_peek(&red.grid->arr[1][0][grid_wg]);
        red.grid_seq = c0.z > c1.z ? c0.z : c1.z;
    }
    const uint32_t wave_global = grid_wg * NWAVES + wave;
    constexpr int W = NWAVES <= 2 ? 2 : NWAVES <= 4 ? 4 : NWAVES <= 8 ? 8 : 16;

    // ---- this wavefront's slots: what never changes from system to system lives in registers for the whole launch ----
    SEQ seq;
    seq.each([&](auto& s, auto* cls, int index) {
        using C = typename class_of<decltype(cls)>::type;
        const uint32_t* t = a.blob + a.o_slots + 4 * ((size_t)wave_global * SEQ::N + index);
        const uint32_t ids_off = t[0], par_off = t[1], pos_off = t[2], count = t[3];
        s.active = (uint32_t)lane < count;
#pragma unroll
        for (int k = 0; k < C::NV; ++k) s.ids[k] = a.blob[ids_off + (size_t)k * C::STRIDE + lane];
        const double* par = reinterpret_cast<const double*>(a.blob + par_off) + lane;
#pragma unroll
        for (int k = 0; k < C::NC; ++k) s.par[k] = par[(size_t)k * C::STRIDE];
        s.pos = a.blob + pos_off + lane;
        s.wmask = 0;
    });

    uint32_t parity = 0;
    // (a resident launch -- DoneWord::request, one workgroup -- serves one request after the other on the same buffers)
    const unsigned long long born = wall_clock64();
    DoneWord done = a.done;
    // (tickets: JitArgs::ticket.  Thread 0 asks the counter right ahead of its wavefront's loads of the guesses -- the answer takes
    // no longer than they do -- and leaves the drawn system in LDS when eval() is through, two words by parity: every system has
    // a rendezvous after that point, everybody reads the word after the system's last.)
    const bool tickets = !GRID && !RESIDENT && a.ticket != nullptr;
    unsigned int* const drawn_lds = reinterpret_cast<unsigned int*>(smem + kRedDoubles + 10);
    const uint32_t ticket_c = blockIdx.x & 7u;
    do {
    uint64_t sys_next = 0;
    for (uint64_t sys = grid_slot; sys < a.batch; sys = sys_next, parity ^= 1u) {
        unsigned int drawn = 0;
        const double* x0 = a.x0 + sys * a.n_row;
        // (Pulling the NEXT system's guesses towards L2 while this one is solved -- one 4-byte load per 64 bytes of its row
        // into a register nothing reads -- was measured and not kept: 2000 x 2000, 16 384 systems per launch, 75.0 -> 70.4 M
        // solves/s; the other workgroups of the CU already cover the first touch.)
        // (several workgroups: the counter is the scratch's, zeroed by workgroup 0 two systems ago)
        int* nwarn = red.grid ? &red.grid->nwarn[parity] : nwarn2 + parity;
        auto log_warning = [&](uint32_t pass, uint32_t pos) {  // Warning::Degenerate, every evaluation (solver.rs:340-346)
            const int idx = atomicAdd(nwarn, 1);
            if (a.warn_log && (uint32_t)idx < a.warn_cap) a.warn_log[sys * a.warn_cap + idx] = ((uint64_t)pass << 32) | pos;
        };
        auto log_mask = [&](auto& s, auto* cls, unsigned long long m, uint32_t pass) {
            using C = typename class_of<decltype(cls)>::type;
            if (!s.active) m = 0;
            while (m) {
                const int ci = __builtin_ctzll(m);
                m &= m - 1;
                log_warning(pass, s.pos[(size_t)ci * C::STRIDE]);
            }
        };

        // ---- load the initial values; eval() (newton.rs:45, :232-236) ------------------------------------------------------------
        double sq = 0.0, mx = __builtin_nan("");
        seq.each([&](auto& s, auto* cls, int index) {
            using C = typename class_of<decltype(cls)>::type;
            // (the answer is wanted after the slots' evaluation, and the COMPILER must know it is on its way: an instruction of its
            // own that reads the register before then -- a spill -- waits for it.  Round 5 issued the atomic in an asm statement and
            // waited in another: a register spilled between the two would have been saved before the answer arrived.  The kernels
            // are compiled without the atomic optimizer (jit.cpp), whose wavefront-wide form reads the answer on the spot.)
            if (index == 0 && tickets && tid == 0)
                drawn = __hip_atomic_fetch_add(a.ticket + ticket_c * kTicketStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < C::NV; ++k) s.x[k] = x0[s.ids[k]];
            unsigned long long wm = 0;
            double sq_s = sq, mx_s = mx;
            C::residuals(s.x, s.par, s.r, true, sq_s, mx_s, wm);
            if (s.active) {
                sq = sq_s;
                mx = mx_s;
            }
            if constexpr (!C::LINEAR) {
                log_mask(s, cls, wm, 0);
                wm = 0;
                C::jacobian(s.x, s.par, s.J, wm);
                log_mask(s, cls, wm, 1);
            }
        });
        if (tickets && tid == 0) {
            asm volatile("" : "+v"(drawn));  // (the first use, here and not earlier: the compiler's wait for the answer comes with it)
            drawn_lds[parity] = (drawn - a.ticket_base[ticket_c]) * 8u + ticket_c;
        }
        // FUSE: eval()'s sums do not get a rendezvous of their own -- the wavefront's totals wait in scalar registers and ride in
        // the first iteration's exchange (Red::step5; the first step is computed before anybody knows whether the system had
        // converged already: x only moves after the rendezvous).  Every iteration runs the same five-value exchange: ONE loop
        // body (a peeled first iteration cost the compiler its schedule).
        const bool fused = FUSE && a.max_iterations > 0;
        double eval_sq = 0.0, eval_mx = __builtin_nan("");
        if (fused) {
            eval_sq = Red::wave_total(sq, OpSum());
            eval_mx = Red::wave_total(mx, OpMax());
        } else {
            red.template sum_max<W>(sq, mx, lane, wave, NWAVES);
        }
        double residual_sq = sq, largest = mx;
        uint32_t pass = 2;
        double lambda = a.initial_lambda;
        uint32_t it = 0, iterations = a.max_iterations, converged = 0;
        bool r_is_at_x = true;
        bool eval_pending = fused;  // residual_sq / largest are not the system's yet

        // ---- the LM loop (newton.rs:47-139) ------------------------------------------------------------------------------------------
        for (;;) {
            if (it >= a.max_iterations) break;            // newton.rs:141-144
            if (!eval_pending && largest <= a.residual_tolerance) {  // newton.rs:50-60
                iterations = it;
                converged = 1;
                break;
            }
            bool lane_bad = false;
            double dmax = __builtin_nan("");
            sq = 0.0;
            mx = __builtin_nan("");
            // Pass 1 -- normal equations, Cholesky, substitutions of every lane's components (newton.rs:73-102), on every lane,
            // active or not, in one basic block: what depends on lambda alone (all of the factorisation of a linear
            // class) is formed once for all the slots of the wavefront, and the slots' dependent chains interleave.
            bool need_exact = false;
            seq.each([&](auto& s, auto* cls, int) {
                using C = typename class_of<decltype(cls)>::type;
                double cd = __builtin_nan("");
                bool ok = true;
                const bool cb = C::solve(s.J, s.r, lambda, s.d, cd, ok);
                s.exact = s.active && !ok && !cb;  // an operand outside the short division's range
                const bool take = s.active && !s.exact;  // (selects, not a branch: the slots stay one basic block)
                const double dm = fmax_nc(dmax, cd);
                dmax = take ? dm : dmax;
                lane_bad = lane_bad || (take && cb);
                need_exact = need_exact || s.exact;
            });
            if (need_exact) {  // (almost never: the same solves with plain divisions for the lanes that asked)
                seq.each([&](auto& s, auto* cls, int) {
                    using C = typename class_of<decltype(cls)>::type;
                    if (s.exact) {
                        double cd = __builtin_nan("");
                        double J2[C::ZJS > 0 ? C::ZJS : 1], r2[C::M > 0 ? C::M : 1];
                        opaque_copy(s.J, J2);
                        opaque_copy(s.r, r2);
                        const bool cb = C::solve_exact(J2, r2, lambda, s.d, cd);
                        lane_bad = lane_bad || cb;
                        dmax = fmax_nc(dmax, cd);
                    }
                });
            }
            // Pass 2 -- residual at the tentative values (newton.rs:111-116), speculative: x moves only after the rendezvous.
            // The sums continue the lane's running values and are taken over by one select per slot.
            seq.each([&](auto& s, auto* cls, int) {
                using C = typename class_of<decltype(cls)>::type;
                double xt[C::NV];
#pragma unroll
                for (int k = 0; k < C::NV; ++k) xt[k] = s.x[k] + s.d[k];
                s.wmask = 0;
                double sq_s = sq, mx_s = mx;
                C::residuals(xt, s.par, s.rn, true, sq_s, mx_s, s.wmask);
                if (s.active) {
                    sq = sq_s;
                    mx = mx_s;
                }
            });
            bool bad;
            if constexpr (FUSE) {
                bad = red.template step5<W>(sq, mx, dmax, eval_sq, eval_mx, lane_bad, lane, wave, NWAVES);
                if (eval_pending) {  // eval()'s verdict, one rendezvous late (newton.rs:45-60)
                    eval_pending = false;
                    residual_sq = eval_sq;
                    largest = eval_mx;
                    if (largest <= a.residual_tolerance) {  // converged before the first iteration: the speculative step is dropped
                        iterations = 0;
                        converged = 1;
                        break;
                    }
                }
            } else {
                bad = red.template step<W>(sq, mx, dmax, lane_bad, lane, wave, NWAVES);
            }
            if (bad) {  // numeric failure anywhere in the system => lambda *= 10, burn the iteration, x untouched
                lambda *= LM_LAMBDA_INCR;
                ++it;
                continue;
            }
            const double step_inf_norm = (a.n_row > 0) ? dmax : 0.0;
            const bool accept = sq < residual_sq;  // strict, newton.rs:118
            const uint32_t pass_res = pass++;
            const uint32_t pass_jac = pass;
            if (accept) ++pass;
            seq.each([&](auto& s, auto* cls, int) {
                using C = typename class_of<decltype(cls)>::type;
                if constexpr (!C::LINEAR) log_mask(s, cls, s.wmask, pass_res);
                if (accept) {
#pragma unroll
                    for (int k = 0; k < C::NV; ++k) s.x[k] = s.x[k] + s.d[k];
#pragma unroll
                    for (int k = 0; k < C::M; ++k) s.r[k] = s.rn[k];
                    if constexpr (!C::LINEAR) {
                        unsigned long long wm = 0;
                        C::jacobian(s.x, s.par, s.J, wm);
                        log_mask(s, cls, wm, pass_jac);
                    }
                } else {  // reject: x += d, x -= d like the reference (newton.rs:111-114,:124-131), not a copy
#pragma unroll
                    for (int k = 0; k < C::NV; ++k) s.x[k] = (s.x[k] + s.d[k]) - s.d[k];
                }
            });
            if (accept) {
                lambda *= LM_LAMBDA_DECR;
                residual_sq = sq;
                largest = mx;
                r_is_at_x = true;
            } else {
                r_is_at_x = false;  // x is now (x + d) - d, which may differ from the x of r in the last bit
                lambda *= LM_LAMBDA_INCR;
            }
            if (step_inf_norm <= a.step_tolerance) {  // newton.rs:134-139
                iterations = it;
                converged = 1;
                break;
            }
            ++it;
        }

        // ---- unsatisfied check (lib.rs:305-327, :358-370) and write-back -----------------------------------------------------------------
        const bool use_r = r_is_at_x && UNIT_W;
        const bool all_satisfied = use_r && largest < EPS && !isnan(residual_sq);
        double unsat_cnt = 0.0;
        double* xo = a.x_out + sys * a.n_row;
        uint8_t* mask = a.unsat_mask ? a.unsat_mask + sys * a.n_cons : nullptr;
        seq.each([&](auto& s, auto* cls, int) {
            using C = typename class_of<decltype(cls)>::type;
            if (all_satisfied) {
                if (mask && s.active)
#pragma unroll
                    for (int ci = 0; ci < C::NC; ++ci) mask[s.pos[(size_t)ci * C::STRIDE]] = 0;
            } else if (use_r) {
                C::unsatisfied_from_r(s.r, s.active, unsat_cnt, mask, s.pos);
            } else {
                C::unsatisfied(s.x, s.par, s.active, unsat_cnt, mask, s.pos);
            }
            if (s.active) {
#pragma unroll
                for (int k = 0; k < C::NV; ++k) xo[s.ids[k]] = s.x[k];
            }
        });
        if (!all_satisfied || ANY_NONLINEAR) {
            double none = __builtin_nan("");
            red.template sum_max<W>(unsat_cnt, none, lane, wave, NWAVES);
        }
        if (tid == 0 && grid_wg == 0) {
            EzpzStatus st;
            st.iterations = iterations;
            st.converged = converged;
            st.n_unsatisfied = (uint32_t)unsat_cnt;
            st.n_warnings = ANY_NONLINEAR ? (uint32_t)__hip_atomic_load(nwarn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            st.final_residual_inf = (a.n_rows_total > 0) ? largest : 0.0;
            st.final_lambda = lambda;
            if (red.grid && __hip_atomic_load(&red.grid->dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                st.iterations = EZPZ_ITERATIONS_TEAM_TIMEOUT;
                st.converged = 0;
            }
            a.status[sys] = st;
            // serves the system after next of this workgroup / slot (every wavefront passes a rendezvous of the next
            // system, which this thread joins only after the store, before it can touch the counter again)
            if (ANY_NONLINEAR) __hip_atomic_store(nwarn, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        sys_next = sys + n_slots;
        if (tickets) {
            sys_next = (uint64_t)n_slots + __builtin_amdgcn_readfirstlane(drawn_lds[parity]);
        }
    }
    if constexpr (RESIDENT) publish_done(done);
    } while (RESIDENT && resident_next(done, born, reinterpret_cast<unsigned long long*>(smem + kRedDoubles + 8)));
}

// The same kernel for a system spread over SEVERAL workgroups (the generator names it when CompPlan::jit_wgs > 1: the 200 000-variable
// ladder on ~100 workgroups of 4 wavefronts): solve_kernel's loop with the reductions' grid stage.  A function of its own so that
// nothing done here moves an instruction of the one-workgroup kernel above (at 168 of 168 registers that one notices code it never
// runs).  Every verdict of the LM control (eval(), each step) is a trip to workgroup 0 and back, ~3.5 us each
// (two hops through memory), during which nobody has anything to do: the ladder spent 11 of its 18 us per solve there
// (profiles/r06_ladder_stamps_before.txt) -- which is why linear systems go through solve_kernel_grid_fast first and come here only
// when that kernel says so: JitArgs::redo != null, the launch solves the systems LISTED there (redo[0] of them: redo[1], ...).
template <class SEQ, int NWAVES, bool ANY_NONLINEAR, bool UNIT_W, bool RESIDENT = false>
__device__ __forceinline__ void solve_kernel_grid(const JitArgs& a, double* smem) {
    using namespace ezpz::dev;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((uint32_t)tid >> 6);
    Red red;
    red.buf = smem;
    red.flags = reinterpret_cast<int*>(smem + kRedDoubles);
    red.flip = 0;
    red.turn = 0;
    int* nwarn2 = red.flags + 4;
    if (tid < 8) red.flags[tid] = 0;
    if (NWAVES > 1) __syncthreads();
    // workgroup g of the G that share a system, slot = which system in flight
    const uint32_t grid_wgs = a.grid ? a.grid_wgs : 1u;
    const uint32_t grid_wg = blockIdx.x % grid_wgs, grid_slot = blockIdx.x / grid_wgs, n_slots = gridDim.x / grid_wgs;
    const uint64_t n_systems = a.redo ? (uint64_t)a.redo[0] : a.batch;
    if (a.redo && blockIdx.x == 0 && tid == 0) {
        a.redo_next[0] = 0;
        if (a.redo_seen) *a.redo_seen = (unsigned int)n_systems;
    }
    if (a.redo && n_systems == 0 && !RESIDENT) return;  // (the usual case of a launch behind solve_kernel_grid_fast: nothing to do)
    red.grid = a.grid ? a.grid + grid_slot : nullptr;
    red.grid_wgs = grid_wgs;
    red.grid_wg = grid_wg;
    red.grid_seq = 0;
    if (red.grid) {  // continue the slot's sequence numbers where the previous launch left them (this workgroup's own)
        const gridchunk_t c0 = grid_peek(&red.grid->arr[0][0][grid_wg]), c1 = grid_peek(&red.grid->arr[1][0][grid_wg]);
        red.grid_seq = c0.z > c1.z ? c0.z : c1.z;
    }
    const uint32_t wave_global = grid_wg * NWAVES + wave;
    constexpr int W = NWAVES <= 2 ? 2 : NWAVES <= 4 ? 4 : NWAVES <= 8 ? 8 : 16;

    // ---- this wavefront's slots: what never changes from system to system lives in registers for the whole launch ----
    SEQ seq;
    seq.each([&](auto& s, auto* cls, int index) {
        using C = typename class_of<decltype(cls)>::type;
        const uint32_t* t = a.blob + a.o_slots + 4 * ((size_t)wave_global * SEQ::N + index);
        const uint32_t ids_off = t[0], par_off = t[1], pos_off = t[2], count = t[3];
        s.active = (uint32_t)lane < count;
#pragma unroll
        for (int k = 0; k < C::NV; ++k) s.ids[k] = a.blob[ids_off + (size_t)k * C::STRIDE + lane];
        const double* par = reinterpret_cast<const double*>(a.blob + par_off) + lane;
#pragma unroll
        for (int k = 0; k < C::NC; ++k) s.par[k] = par[(size_t)k * C::STRIDE];
        s.pos = a.blob + pos_off + lane;
        s.wmask = 0;
    });

    uint32_t parity = 0;
    DoneWord done = a.done;
    for (uint64_t item = grid_slot; item < n_systems; item += n_slots, parity ^= 1u) {
        const uint64_t sys = a.redo ? (uint64_t)a.redo[1 + item] : item;
        const double* x0 = a.x0 + sys * a.n_row;
        // (several workgroups: the counter is the scratch's, zeroed by workgroup 0 two systems ago)
        int* nwarn = red.grid ? &red.grid->nwarn[parity] : nwarn2 + parity;
        auto log_warning = [&](uint32_t pass, uint32_t pos) {  // Warning::Degenerate, every evaluation (solver.rs:340-346)
            const int idx = atomicAdd(nwarn, 1);
            if (a.warn_log && (uint32_t)idx < a.warn_cap) a.warn_log[sys * a.warn_cap + idx] = ((uint64_t)pass << 32) | pos;
        };
        auto log_mask = [&](auto& s, auto* cls, unsigned long long m, uint32_t pass) {
            using C = typename class_of<decltype(cls)>::type;
            if (!s.active) m = 0;
            while (m) {
                const int ci = __builtin_ctzll(m);
                m &= m - 1;
                log_warning(pass, s.pos[(size_t)ci * C::STRIDE]);
            }
        };

        // ---- load the initial values; eval() (newton.rs:45, :232-236) ------------------------------------------------------------
        double sq = 0.0, mx = __builtin_nan("");
        seq.each([&](auto& s, auto* cls, int index) {
            using C = typename class_of<decltype(cls)>::type;
#pragma unroll
            for (int k = 0; k < C::NV; ++k) s.x[k] = x0[s.ids[k]];
            unsigned long long wm = 0;
            double sq_s = sq, mx_s = mx;
            C::residuals(s.x, s.par, s.r, true, sq_s, mx_s, wm);
            if (s.active) {
                sq = sq_s;
                mx = mx_s;
            }
            if constexpr (!C::LINEAR) {
                log_mask(s, cls, wm, 0);
                wm = 0;
                C::jacobian(s.x, s.par, s.J, wm);
                log_mask(s, cls, wm, 1);
            }
        });
        red.template sum_max<W>(sq, mx, lane, wave, NWAVES);
        double residual_sq = sq, largest = mx;
        uint32_t pass = 2;
        double lambda = a.initial_lambda;
        uint32_t it = 0, iterations = a.max_iterations, converged = 0;
        bool r_is_at_x = true;

        // ---- the LM loop (newton.rs:47-139) ------------------------------------------------------------------------------------------
        for (;;) {
            if (it >= a.max_iterations) break;            // newton.rs:141-144
            if (largest <= a.residual_tolerance) {  // newton.rs:50-60
                iterations = it;
                converged = 1;
                break;
            }
            bool lane_bad = false;
            double dmax = __builtin_nan("");
            sq = 0.0;
            mx = __builtin_nan("");
            // Pass 1 -- normal equations, Cholesky, substitutions of every lane's components (newton.rs:73-102), on every lane,
            // active or not, in one basic block: what depends on lambda alone (all of the factorisation of a linear
            // class) is formed once for all the slots of the wavefront, and the slots' dependent chains interleave.
            bool need_exact = false;
            seq.each([&](auto& s, auto* cls, int) {
                using C = typename class_of<decltype(cls)>::type;
                double cd = __builtin_nan("");
                bool ok = true;
                const bool cb = C::solve(s.J, s.r, lambda, s.d, cd, ok);
                s.exact = s.active && !ok && !cb;  // an operand outside the short division's range
                const bool take = s.active && !s.exact;  // (selects, not a branch: the slots stay one basic block)
                const double dm = fmax_nc(dmax, cd);
                dmax = take ? dm : dmax;
                lane_bad = lane_bad || (take && cb);
                need_exact = need_exact || s.exact;
            });
            if (need_exact) {  // (almost never: the same solves with plain divisions for the lanes that asked)
                seq.each([&](auto& s, auto* cls, int) {
                    using C = typename class_of<decltype(cls)>::type;
                    if (s.exact) {
                        double cd = __builtin_nan("");
                        double J2[C::ZJS > 0 ? C::ZJS : 1], r2[C::M > 0 ? C::M : 1];
                        opaque_copy(s.J, J2);
                        opaque_copy(s.r, r2);
                        const bool cb = C::solve_exact(J2, r2, lambda, s.d, cd);
                        lane_bad = lane_bad || cb;
                        dmax = fmax_nc(dmax, cd);
                    }
                });
            }
            // Pass 2 -- residual at the tentative values (newton.rs:111-116), speculative: x moves only after the rendezvous.
            // The sums continue the lane's running values and are taken over by one select per slot.
            seq.each([&](auto& s, auto* cls, int) {
                using C = typename class_of<decltype(cls)>::type;
                double xt[C::NV];
#pragma unroll
                for (int k = 0; k < C::NV; ++k) xt[k] = s.x[k] + s.d[k];
                s.wmask = 0;
                double sq_s = sq, mx_s = mx;
                C::residuals(xt, s.par, s.rn, true, sq_s, mx_s, s.wmask);
                if (s.active) {
                    sq = sq_s;
                    mx = mx_s;
                }
            });
            const bool bad = red.template step<W>(sq, mx, dmax, lane_bad, lane, wave, NWAVES);
            if (bad) {  // numeric failure anywhere in the system => lambda *= 10, burn the iteration, x untouched
                lambda *= LM_LAMBDA_INCR;
                ++it;
                continue;
            }
            const double step_inf_norm = (a.n_row > 0) ? dmax : 0.0;
            const bool accept = sq < residual_sq;  // strict, newton.rs:118
            const uint32_t pass_res = pass++;
            const uint32_t pass_jac = pass;
            if (accept) ++pass;
            seq.each([&](auto& s, auto* cls, int) {
                using C = typename class_of<decltype(cls)>::type;
                if constexpr (!C::LINEAR) log_mask(s, cls, s.wmask, pass_res);
                if (accept) {
#pragma unroll
                    for (int k = 0; k < C::NV; ++k) s.x[k] = s.x[k] + s.d[k];
#pragma unroll
                    for (int k = 0; k < C::M; ++k) s.r[k] = s.rn[k];
                    if constexpr (!C::LINEAR) {
                        unsigned long long wm = 0;
                        C::jacobian(s.x, s.par, s.J, wm);
                        log_mask(s, cls, wm, pass_jac);
                    }
                } else {  // reject: x += d, x -= d like the reference (newton.rs:111-114,:124-131), not a copy
#pragma unroll
                    for (int k = 0; k < C::NV; ++k) s.x[k] = (s.x[k] + s.d[k]) - s.d[k];
                }
            });
            if (accept) {
                lambda *= LM_LAMBDA_DECR;
                residual_sq = sq;
                largest = mx;
                r_is_at_x = true;
            } else {
                r_is_at_x = false;  // x is now (x + d) - d, which may differ from the x of r in the last bit
                lambda *= LM_LAMBDA_INCR;
            }
            if (step_inf_norm <= a.step_tolerance) {  // newton.rs:134-139
                iterations = it;
                converged = 1;
                break;
            }
            ++it;
        }

        // ---- unsatisfied check (lib.rs:305-327, :358-370) and write-back -----------------------------------------------------------------
        const bool use_r = r_is_at_x && UNIT_W;
        const bool all_satisfied = use_r && largest < EPS && !isnan(residual_sq);
        double unsat_cnt = 0.0;
        double* xo = a.x_out + sys * a.n_row;
        uint8_t* mask = a.unsat_mask ? a.unsat_mask + sys * a.n_cons : nullptr;
        seq.each([&](auto& s, auto* cls, int) {
            using C = typename class_of<decltype(cls)>::type;
            if (all_satisfied) {
                if (mask && s.active)
#pragma unroll
                    for (int ci = 0; ci < C::NC; ++ci) mask[s.pos[(size_t)ci * C::STRIDE]] = 0;
            } else if (use_r) {
                C::unsatisfied_from_r(s.r, s.active, unsat_cnt, mask, s.pos);
            } else {
                C::unsatisfied(s.x, s.par, s.active, unsat_cnt, mask, s.pos);
            }
            if (s.active) {
#pragma unroll
                for (int k = 0; k < C::NV; ++k) xo[s.ids[k]] = s.x[k];
            }
        });
        if (!all_satisfied || ANY_NONLINEAR) {
            double none = __builtin_nan("");
            red.template sum_max<W>(unsat_cnt, none, lane, wave, NWAVES);
        }
        if (tid == 0 && grid_wg == 0) {
            EzpzStatus st;
            st.iterations = iterations;
            st.converged = converged;
            st.n_unsatisfied = (uint32_t)unsat_cnt;
            st.n_warnings = ANY_NONLINEAR ? (uint32_t)__hip_atomic_load(nwarn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            st.final_residual_inf = (a.n_rows_total > 0) ? largest : 0.0;
            st.final_lambda = lambda;
            if (red.grid && __hip_atomic_load(&red.grid->dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                st.iterations = EZPZ_ITERATIONS_TEAM_TIMEOUT;
                st.converged = 0;
            }
            a.status[sys] = st;
            // serves the system after next of this workgroup / slot (every wavefront passes a rendezvous of the next
            // system, which this thread joins only after the store, before it can touch the counter again)
            if (ANY_NONLINEAR) __hip_atomic_store(nwarn, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if constexpr (RESIDENT) publish_done(done);  // (several workgroups per system: a completion word, never resident)
}

// A value at a uniform base + a lane's 32-bit byte offset, as a buffer access: four scalar registers for the row's descriptor and ONE
// vector register per value for the offset.  (`base[index]` with a 64-bit index becomes a 64-bit address per value and use in vector
// registers -- recurrences over the systems of the launch: 48 registers for the ladder's loads, prefetches and stores.)
typedef int bufword2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_at(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0xFFFFFFFFu, 0x00020000);  // raw, no swizzle, 4 GB
}
__device__ __forceinline__ double load_at(__amdgpu_buffer_rsrc_t row, uint32_t byte_offset) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(row, (int)byte_offset, 0, 0));
}
__device__ __forceinline__ void store_at(__amdgpu_buffer_rsrc_t row, uint32_t byte_offset, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(bufword2_t, v), row, (int)byte_offset, 0, 0);
}

// ---- what the kernels that do not wait for the LM control's verdicts share (solve_kernel_fast, solve_kernel_grid_fast) -----------
// A wavefront's slots for those kernels: the caller's variable ids as BYTE offsets (n_row < 2^29: the host), the instances'
// parameters, and the factorisation of every class at the two lambdas of the expected path -- J^T J + lambda I of a linear class is
// the same matrix for every instance of every system: factorised once per launch (newton.rs:73-99: the operations of C::solve
// that do not involve the right-hand side, in its order), kept in scalar registers by the class's first slot.
template <class SEQ>
__device__ __forceinline__ void fast_setup(SEQ& seq, const JitArgs& a, const uint32_t wave_global, const int lane, const uint32_t first_byte = 0) {
    seq.each([&](auto& s, auto* cls, int index) {
        using C = typename class_of<decltype(cls)>::type;
        const uint32_t* t = a.blob + a.o_slots + 4 * ((size_t)wave_global * SEQ::N + index);
        const uint32_t ids_off = t[0], par_off = t[1], count = t[3];
        s.active = (uint32_t)lane < count;
        // (first_byte: the wavefront's piece of the row starts there -- offsets into the piece; a lane without an instance reads its start)
#pragma unroll
        for (int k = 0; k < C::NV; ++k) s.ids[k] = s.active ? a.blob[ids_off + (size_t)k * C::STRIDE + lane] * 8u - first_byte : 0u;
        const double* par = reinterpret_cast<const double*>(a.blob + par_off) + lane;
#pragma unroll
        for (int k = 0; k < C::NC; ++k) s.par[k] = par[(size_t)k * C::STRIDE];
    });
}
template <class SEQ>
__device__ __forceinline__ void fast_fetch(SEQ& seq, const JitArgs& a, const uint64_t sys) {  // Slot::xn <- the guesses of `sys`
    const __amdgpu_buffer_rsrc_t x0 = row_at(a.x0 + sys * a.n_row);
    seq.each([&](auto& s, auto* cls, int) {
        using C = typename class_of<decltype(cls)>::type;
#pragma unroll
        for (int i = 0; i < C::NV; ++i) s.xn[i] = load_at(x0, s.ids[i]);
    });
}
// A wavefront's CONTIGUOUS piece of a row -- `bytes` from `first_byte` on: the generator found that the variables of the wavefront's
// instances are exactly those (comp_program.cpp: wave ranges) -- moved as FULL LINES: lane l takes 16 bytes at 1024 p + 16 l, p = 0 ...
// PIECES - 1, consecutive lanes on consecutive bytes, instead of eight 8-byte accesses per lane at strides of 16 / 32 bytes in which a
// 128-byte line is shared by three or four instructions (tools/row_copy_bench.hip: 165 against 150 M rows/s of 16 KB is what memory
// allows the two patterns).  The slots read and write their values in an LDS copy of the piece (`buf`: the wavefront's own).
typedef int bufword4_t __attribute__((ext_vector_type(4)));
typedef bufword4_t __attribute__((may_alias)) lds_b128_t;  // (the piece is written as 16-byte words and read as doubles, and the other way round)
typedef double __attribute__((may_alias)) lds_f64_t;
template <int PIECES>
__device__ __forceinline__ void piece_load(bufword4_t (&q)[PIECES], const __amdgpu_buffer_rsrc_t row, const uint32_t first_byte, const uint32_t bytes, const int lane) {
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
        const uint32_t at = 1024u * p + 16u * lane;
        q[p] = bufword4_t{0, 0, 0, 0};
        if (at + 16u <= bytes) {
            q[p] = __builtin_amdgcn_raw_buffer_load_b128(row, (int)(first_byte + at), 0, 0);
        } else if (at + 8u <= bytes) {
            const bufword2_t h = __builtin_amdgcn_raw_buffer_load_b64(row, (int)(first_byte + at), 0, 0);
            q[p].x = h.x, q[p].y = h.y;
        }
    }
}
template <int PIECES>
__device__ __forceinline__ void piece_to_lds(const bufword4_t (&q)[PIECES], double* buf, const int lane) {
#pragma unroll
    for (int p = 0; p < PIECES; ++p) *reinterpret_cast<lds_b128_t*>(reinterpret_cast<char*>(buf) + 1024 * p + 16 * lane) = q[p];
}
template <int PIECES>
__device__ __forceinline__ void piece_store(const double* buf, const __amdgpu_buffer_rsrc_t row, const uint32_t first_byte, const uint32_t bytes, const int lane) {
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
        const uint32_t at = 1024u * p + 16u * lane;
        const bufword4_t q = *reinterpret_cast<const lds_b128_t*>(reinterpret_cast<const char*>(buf) + at);
        if (at + 16u <= bytes)
            __builtin_amdgcn_raw_buffer_store_b128(q, row, (int)(first_byte + at), 0, 0);
        else if (at + 8u <= bytes)
            __builtin_amdgcn_raw_buffer_store_b64(bufword2_t{q.x, q.y}, row, (int)(first_byte + at), 0, 0);
    }
}
template <class SEQ>
__device__ __forceinline__ void fast_factor(SEQ& seq, const JitArgs& a) {
    const double lambda1 = a.initial_lambda * ezpz::dev::LM_LAMBDA_DECR;
    seq.each([&](auto& s, auto* cls, int) {
        using C = typename class_of<decltype(cls)>::type;
        if (&s != &seq.template first<C>()) return;  // (once per class: its other slots read the first one's)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            double F[C::NF];
            bool ok = true;
            const bool bad = C::factor(st == 0 ? a.initial_lambda : lambda1, F, ok);
#pragma unroll
            for (int i = 0; i < C::NF; ++i) s.F[st][i] = uniform(F[i]);
            s.fbad[st] = __builtin_amdgcn_readfirstlane((int)bad) != 0;
            s.fok[st] = __builtin_amdgcn_readfirstlane((int)ok) != 0;
        }
    });
}
// One wavefront's share of system `sys` from the guesses in Slot::xn to the stored values: eval() (newton.rs:45, :232-236), then two
// iterations taken for accepted -- the loop's two passes (newton.rs:73-116), x += d, r = r_next.  SLOT BY SLOT: one slot's x, d and
// r are live at a time, and the place of a slot's guesses is free for those of the wavefront's NEXT system (`sys_n`, if `next`),
// asked for a whole system ahead of their use, as soon as it has read them.  The sums run over the slots in the loop's order:
// the same bits.  Leaves in v[], in the wavefront's last lane: sum r0^2, sum r1^2, sum r2^2, max|r2| -- what the verdict needs as
// NUMBERS (the accept tests compare sums, the status reports max|r2|).  The other four maxima of the LM control (max|r0|, |d1|,
// max|r1|, |d2|) are only ever compared with a tolerance (newton.rs:50-60, :134-139), and `max > tol` is `any lane's > tol`: they
// travel as two bits each instead of a cross-lane reduction of a double (20 instructions apiece, a fifth of the kernel's):
// kFastGt << 2 k: some value exceeded the tolerance, kFastFin << 2 k: some value was not NaN (the maxima ignore NaNs:
// newton.rs:53,:108 -- a maximum of NaNs only is NaN, and NaN <= tol is false).  Bits 0 / 1: a pivot failed in the first / second
// step, bit 2: an operand outside the short division's range (almost never: the system goes on the redo list, where
// C::solve_exact divides plainly).
constexpr unsigned int kFastGt = 8u, kFastFin = 16u;  // (k = 0: max|r0|, 1: |d1|, 2: max|r1|, 3: |d2|)
__device__ __forceinline__ bool fast_exceeds(unsigned int flags, int k) {  // !(maximum k <= its tolerance), as the loop tests it
    return (flags & (kFastGt << (2 * k))) != 0 || (flags & (kFastFin << (2 * k))) == 0;
}
// IO: how the values travel -- 0: every slot loads the next system's guesses and stores its values itself; 1: the values wait in LDS
// (`out_lds`) and are stored back to back; 2: the wavefront owns a contiguous piece of the row (`first_byte`, `bytes`): this
// system's values are in `out_lds`, the slots read and write them there, the next system's piece is loaded as full lines at the top
// and put into `next_lds` at the end, this one's is stored as full lines.
template <int IO, class SEQ>
__device__ __forceinline__ unsigned int fast_wave(SEQ& seq, const JitArgs& a, const uint64_t sys, const uint64_t sys_n, const bool next,
                                                  const uint32_t wave_global, const int lane, double (&v)[4], double* const out_lds,
                                                  double* const next_lds = nullptr, const uint32_t first_byte = 0, const uint32_t bytes = 0) {
    using namespace ezpz::dev;
    constexpr bool STAGE = IO == 1;
    constexpr int PIECES = (SEQ::NVS + 1) / 2;
    const __amdgpu_buffer_rsrc_t xo = row_at(a.x_out + sys * a.n_row);
    const __amdgpu_buffer_rsrc_t x0n = row_at(a.x0 + (next ? sys_n : sys) * a.n_row);
    bufword4_t ahead[IO == 2 ? PIECES : 1];
    if constexpr (IO == 2) {
        if (next) piece_load<PIECES>(ahead, x0n, first_byte, bytes, lane);
    }
    uint8_t* mask = a.unsat_mask ? a.unsat_mask + sys * a.n_cons : nullptr;
    const __amdgpu_buffer_rsrc_t table = row_at(a.blob);
    v[0] = v[1] = v[2] = 0.0;
    v[3] = __builtin_nan("");
    double m[4] = {__builtin_nan(""), __builtin_nan(""), __builtin_nan(""), __builtin_nan("")};  // this lane's max|r0|, |d1|, max|r1|, |d2|
    bool lane_bad1 = false, lane_bad2 = false, lane_redo = false;
    int var0 = 0;
    seq.each([&](auto& s, auto* cls, int index) {
        using C = typename class_of<decltype(cls)>::type;
        auto& f = seq.template first<C>();
        if constexpr (IO == 2) {
#pragma unroll
            for (int i = 0; i < C::NV; ++i) s.x[i] = *reinterpret_cast<const lds_f64_t*>(reinterpret_cast<const char*>(out_lds) + s.ids[i]);
        } else {
#pragma unroll
            for (int i = 0; i < C::NV; ++i) s.x[i] = s.xn[i];
            if (next) {
#pragma unroll
                for (int i = 0; i < C::NV; ++i) s.xn[i] = load_at(x0n, s.ids[i]);
            }
        }
        unsigned long long wm = 0;
        {
            double sq_s = v[0], mx_s = m[0];
            C::residuals(s.x, s.par, s.r, true, sq_s, mx_s, wm);
            v[0] = s.active ? sq_s : v[0];
            m[0] = s.active ? mx_s : m[0];
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            double cd = __builtin_nan("");
            bool ok = f.fok[st];
            C::solve_f(f.F[st], s.r, s.d, cd, ok);
            const bool cb = f.fbad[st];
            lane_redo = lane_redo || (s.active && !ok && !cb);
            double& dmax = m[1 + 2 * st];
            const double dm = fmax_nc(dmax, cd);
            dmax = s.active ? dm : dmax;
            if (st == 0)
                lane_bad1 = lane_bad1 || (s.active && cb);
            else
                lane_bad2 = lane_bad2 || (s.active && cb);
#pragma unroll
            for (int i = 0; i < C::NV; ++i) s.x[i] = s.x[i] + s.d[i];
            double& mx = st == 0 ? m[2] : v[3];
            double sq_s = v[1 + st], mx_s = mx;
            C::residuals(s.x, s.par, s.r, true, sq_s, mx_s, wm);
            v[1 + st] = s.active ? sq_s : v[1 + st];
            mx = s.active ? mx_s : mx;
        }
        if (mask) {
            const uint32_t pos_off = __builtin_amdgcn_readfirstlane(a.blob[a.o_slots + 4 * ((size_t)wave_global * SEQ::N + index) + 2]);
#pragma unroll
            for (int ci = 0; ci < C::NC; ++ci) {
                const uint32_t at = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(table, lane * 4, (int)((pos_off + (uint32_t)ci * C::STRIDE) * 4u), 0);
                if (s.active) mask[at] = 0;
            }
        }
        if constexpr (IO == 2) {
            if (s.active) {
#pragma unroll
                for (int i = 0; i < C::NV; ++i) *reinterpret_cast<lds_f64_t*>(reinterpret_cast<char*>(out_lds) + s.ids[i]) = s.x[i];
            }
        } else if constexpr (STAGE) {  // (the values wait in LDS -- this lane's own words, `out_lds` is the wavefront's -- for the stores below)
#pragma unroll
            for (int i = 0; i < C::NV; ++i) out_lds[(var0 + i) * 64 + lane] = s.x[i];
            var0 += C::NV;
        } else {
            if (s.active) {
#pragma unroll
                for (int i = 0; i < C::NV; ++i) store_at(xo, s.ids[i], s.x[i]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    // the values this leads to ("every constraint satisfied", lib.rs:305-327: largest < EPS is part of the verdict) -- the
    // wavefront's stores BACK TO BACK: a 128-byte line of the row is shared by three or four of them, and stored slot by slot, as
    // the values came, the pieces reached memory apart: 23.8 KB written and 18 KB read per 16 KB row of the 2000 x 2000 system,
    // 4.8 TB/s for 113.9 M solves/s; together 16.0 / 16.0 KB and 129 M (profiles/r06_fast_stores.txt).  STAGE false: stored as
    // they come (a system on several workgroups: 15 registers less let a fourth wavefront onto the SIMD, worth more to the
    // ladder -- issue-bound at 0.37 of the memory roof -- than 20 % of traffic)
    var0 = 0;
    if constexpr (STAGE) seq.each([&](auto& s, auto* cls, int) {
        using C = typename class_of<decltype(cls)>::type;
#pragma unroll
        for (int i = 0; i < C::NV; ++i) {
            const double x = out_lds[(var0 + i) * 64 + lane];
            if (s.active) store_at(xo, s.ids[i], x);
        }
        var0 += C::NV;
    });
    if constexpr (IO == 2) {
        piece_store<PIECES>(out_lds, xo, first_byte, bytes, lane);
        if (next) piece_to_lds<PIECES>(ahead, next_lds, lane);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = i < 3 ? reduce_wave_to_last_lane(v[i], OpSum()) : reduce_wave_to_last_lane(v[i], OpMax());
    unsigned int flags = (__ballot(lane_bad1) != 0 ? 1u : 0u) | (__ballot(lane_bad2) != 0 ? 2u : 0u) | (__ballot(lane_redo) != 0 ? 4u : 0u);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double tol = (k & 1) ? a.step_tolerance : a.residual_tolerance;
        flags |= (__ballot(m[k] > tol) != 0 ? kFastGt << (2 * k) : 0u) | (__ballot(m[k] == m[k]) != 0 ? kFastFin << (2 * k) : 0u);
    }
    return flags;
}
// The reference's decisions in the reference's order on a system's totals -- lane i < 4 of the calling wavefront holds value i of
// fast_wave's list, `flags` the OR of its bits over the system (bit 2 also: a workgroup's line never came) -- : eval() and the top of
// iteration 0 (newton.rs:45-60), its step (:93-99, :118-139), the top of iteration 1, its step, the top of iteration 2
// (max_iterations >= 3, says the host: the limit is not what ends it), every constraint satisfied (lib.rs:305-327: unit weights,
// r is at x).  If they are the expected ones, lane 0 writes the status; if not, it puts the system on the redo list.  Returns
// whether they were.
__device__ __forceinline__ bool fast_verdict(const JitArgs& a, const uint64_t sys, const double t, const unsigned int flags, const int lane) {
    using namespace ezpz::dev;
    auto at = [&](int src_lane) {
        const unsigned long long u = __builtin_bit_cast(unsigned long long, t);
        const unsigned int lo = __builtin_amdgcn_readlane((unsigned int)u, src_lane), hi = __builtin_amdgcn_readlane((unsigned int)(u >> 32), src_lane);
        return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    };
    const double sq_0 = at(0), sq_1 = at(1), sq_2 = at(2), mx_2 = at(3);
    // (a system without variables has step norm 0 <= any tolerance in the loop: n_row > 0 here, a plan has variables)
    const bool stands = !(flags & 4u) && fast_exceeds(flags, 0) && !(flags & 1u) && sq_1 < sq_0 && fast_exceeds(flags, 1) &&
                        fast_exceeds(flags, 2) && !(flags & 2u) && sq_2 < sq_1 && fast_exceeds(flags, 3) &&
                        mx_2 <= a.residual_tolerance && mx_2 < EPS && !isnan(sq_2) && a.n_row > 0;
    if (lane == 0) {
        if (stands) {
            EzpzStatus st;
            st.iterations = 2;
            st.converged = 1;
            st.n_unsatisfied = 0;
            st.n_warnings = 0;
            st.final_residual_inf = (a.n_rows_total > 0) ? mx_2 : 0.0;
            st.final_lambda = (a.initial_lambda * LM_LAMBDA_DECR) * LM_LAMBDA_DECR;
            a.status[sys] = st;
        } else {
            const unsigned int at_list = __hip_atomic_fetch_add(a.redo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.redo[1 + at_list] = (unsigned int)sys;
        }
    }
    return stands;
}

// VERDICTS NOT WAITED FOR, one workgroup per system (the 2000 x 2000 headline).  The same idea as solve_kernel_grid_fast below, where
// it is explained, without anything crossing workgroups: a wavefront takes its slots through eval() and both iterations (fast_wave),
// the wavefronts' partials meet in LDS at the ONE barrier a system costs, wavefront 0 makes the reference's decisions on the totals
// (fast_verdict) while the others are on the next system already.  solve_kernel spends three rendezvous per system on the LM
// control, each with every wavefront's state held in registers across it (168 of them, three wavefronts per SIMD, 43 % of wave
// cycles parked: profiles/r05_bench_massive.json); here nothing is live across the barrier but the next system's guesses.
// CONTIG (says the generator): every wavefront's variables are one contiguous piece of the row -- moved as full lines (fast_wave, IO 2).
template <class SEQ, int NWAVES, bool CONTIG = false>
__device__ __forceinline__ void solve_kernel_fast(const JitArgs& a) {
    using namespace ezpz::dev;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((uint32_t)tid >> 6);
    // Who solves which system: a workgroup's first is its own index; every further one is DRAWN (JitArgs::ticket, as in solve_kernel:
    // a workgroup that starts late -- a co-tenant held its place -- then has one system to catch up on, not a fixed share of the
    // batch: 100.9 -> 53.7 M solves/s beside one resident one-call kernel with fixed shares, DESIGN_HISTORY.md C.2) -- and drawn TWO
    // systems ahead, because the guesses are asked for one ahead: thread 0 asks the counter at the top of a system and leaves the
    // answer in LDS ahead of that system's barrier.  (Two draws per workgroup are in vain: the host counts on it.)
    const bool tickets = a.ticket != nullptr;
    const uint32_t ticket_c = blockIdx.x & 7u;
    __shared__ unsigned int fast_drawn[2];
    unsigned int drawn = 0;
    // (the compiler's atomic, so that it knows an answer is on its way -- solve_kernel's note on its draw)
    if (tickets && tid == 0) drawn = __hip_atomic_fetch_add(a.ticket + ticket_c * kTicketStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    SEQ seq;
    __shared__ double fast_part[2][4 * 16];  // [parity of k][value][wavefront]: the wavefronts' partials
    __shared__ int fast_pflag[2][16];
    // per wavefront: a system's values on their way out (fast_wave, IO 1) -- or, CONTIG, the wavefront's piece of the row: this
    // system's and the next one's (IO 2; a piece of NVS x 64 doubles, rounded up to whole 16-byte accesses of the 64 lanes)
    constexpr int PIECE = ((SEQ::NVS + 1) / 2) * 128;
    __shared__ __attribute__((aligned(16))) double fast_out[NWAVES * (CONTIG ? 2 * PIECE : SEQ::NVS * 64)];
    uint32_t first_byte = 0, piece_bytes = 0;
    if constexpr (CONTIG) {
        first_byte = a.blob[a.o_ranges + 2 * wave] * 8u;
        piece_bytes = a.blob[a.o_ranges + 2 * wave + 1] * 8u;
    }
    double* const row_lds = fast_out + wave * (CONTIG ? 2 * PIECE : SEQ::NVS * 64);
    fast_setup(seq, a, wave, lane, first_byte);
    if constexpr (CONTIG) {
        if (blockIdx.x < a.batch) {
            bufword4_t q[(SEQ::NVS + 1) / 2];
            piece_load<(SEQ::NVS + 1) / 2>(q, row_at(a.x0 + (uint64_t)blockIdx.x * a.n_row), first_byte, piece_bytes, lane);
            piece_to_lds<(SEQ::NVS + 1) / 2>(q, row_lds, lane);
        }
    } else {
        if (blockIdx.x < a.batch) fast_fetch(seq, a, blockIdx.x);
    }
    fast_factor(seq, a);
    auto drawn_system = [&](unsigned int d) { return (uint64_t)gridDim.x + (uint64_t)(d - a.ticket_base[ticket_c]) * 8u + ticket_c; };
    uint64_t sys = blockIdx.x, sys_n = sys + gridDim.x;
    if (tickets) {
        if (tid == 0) {
            asm volatile("" : "+v"(drawn));
            fast_drawn[1] = drawn;
        }
        __syncthreads();
        sys_n = drawn_system(__builtin_amdgcn_readfirstlane(fast_drawn[1]));
    }
    unsigned int kp = 0;
    for (; sys < a.batch; kp ^= 1u) {
        if (tickets && tid == 0) drawn = __hip_atomic_fetch_add(a.ticket + ticket_c * kTicketStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double v[4];
        unsigned int wave_flags;
        if constexpr (CONTIG)
            wave_flags = fast_wave<2>(seq, a, sys, sys_n, sys_n < a.batch, wave, lane, v, row_lds + kp * PIECE, row_lds + (kp ^ 1u) * PIECE, first_byte, piece_bytes);
        else
            wave_flags = fast_wave<1>(seq, a, sys, sys_n, sys_n < a.batch, wave, lane, v, row_lds);
        uint64_t sys_nn = sys_n + gridDim.x;
        if (tickets && tid == 0) {
            asm volatile("" : "+v"(drawn));
            if (NWAVES > 1) fast_drawn[kp] = drawn;
        }
        if constexpr (NWAVES == 1) {
            if (tickets) sys_nn = drawn_system(__builtin_amdgcn_readfirstlane(drawn));
            // (lane i <- value i from the last lane)
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned long long u = __builtin_bit_cast(unsigned long long, v[i]);
                const unsigned int lo = __builtin_amdgcn_readlane((unsigned int)u, 63), hi = __builtin_amdgcn_readlane((unsigned int)(u >> 32), 63);
                const double x = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
                t = lane == i ? x : t;
            }
            fast_verdict(a, sys, t, wave_flags, lane);
        } else {
            if (lane == 63) {
#pragma unroll
                for (int i = 0; i < 4; ++i) fast_part[kp][16 * i + wave] = v[i];
                fast_pflag[kp][wave] = (int)wave_flags;
            }
            __syncthreads();
            if (tickets) sys_nn = drawn_system(__builtin_amdgcn_readfirstlane(fast_drawn[kp]));
            if (wave == 0) {
                const int i = lane & 3;
                double t = fast_part[kp][16 * i];
                unsigned int fl = (unsigned int)fast_pflag[kp][0];
#pragma unroll
                for (int w2 = 1; w2 < NWAVES; ++w2) {
                    const double o = fast_part[kp][16 * i + w2];
                    const double sum = t + o, mxm = fmax_nc(t, o);
                    t = i < 3 ? sum : mxm;
                    fl |= (unsigned int)fast_pflag[kp][w2];
                }
                fast_verdict(a, sys, t, fl, lane);
            }
        }
        sys = sys_n;
        sys_n = sys_nn;
    }
}

// VERDICTS NOT WAITED FOR -- a LINEAR system with unit weights on several workgroups (the ladder).  What the verdicts of such a system
// decide is almost always the same -- not converged at the start, pivots fine, step accepted, not converged, step accepted,
// converged: 2 iterations (README.md:36-38) -- and a step needs nothing of the SYSTEM but lambda, which those verdicts fix
// (newton.rs:118-123: x 0.1 per accepted step).  So every workgroup takes both steps on its own share, STORES the values that
// result, publishes what the verdict needs of it (fast_wave: three sums of squares, the last maximum, and the other maxima and
// flags as bits) as four self-validating chunks of one line, and goes on to the next system with empty hands.  One system later (the
// lines are all there by then: no wait) ONE workgroup -- they take turns: sequence number mod G -- gathers the lines, makes the
// reference's decisions in the reference's order on the system's totals (newton.rs:50-60, :93-99, :118-139, lib.rs:305-327) and,
// if they are the expected ones, writes the status: iterations 2, converged, nothing unsatisfied.  Any other outcome (converged
// earlier, a failed pivot, a rejected step, the step tolerance met, not converged after two, a residual at or above EPSILON) puts
// the system on the launch's REDO LIST (JitArgs::redo), which the launch of solve_kernel_grid that follows on the stream solves
// from the caller's guesses with every verdict waited for -- so x0 must not be the buffer the values were stored to (the host
// runs that kernel alone when they overlap), and the arithmetic of the expected path is that kernel's operation for operation:
// the same bits (tests/test_gpu_components.py: every exit, mixed in one launch).
// Nothing waits in the steady state: a system costs a workgroup its loads (asked for a system ahead), ~600 vector instructions per
// wavefront, one barrier, its stores, and every G-th time a turn at the totals.  A kernel of its own because it needs a third of
// the registers of the loop (no r_next, no step kept across a rendezvous, nothing of the general evaluators): more systems in
// flight.  The only wait left is flow control: a workgroup may be four systems ahead of the totals (the ring's depth).
// IO (fast_wave): 0 = values stored slot by slot; 1 = they wait in LDS and a wavefront's stores go out back to back (the pieces of a
// 128-byte line reach the L2 together: written once); 2 = the wavefront's contiguous piece of the row as whole lines through LDS.
template <class SEQ, int NWAVES, int IO = 0>
__device__ __forceinline__ void solve_kernel_grid_fast(const JitArgs& a) {
    constexpr bool CONTIG = IO == 2;
    using namespace ezpz::dev;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((uint32_t)tid >> 6);
    const uint32_t grid_wgs = a.grid_wgs;
    const uint32_t grid_wg = blockIdx.x % grid_wgs, grid_slot = blockIdx.x / grid_wgs, n_slots = gridDim.x / grid_wgs;
    GridScratch* const gs0 = a.grid + grid_slot;
    const uint32_t wave_global = grid_wg * NWAVES + wave;
    SEQ seq;
    // (CONTIG: a wavefront's variables are one contiguous piece of the row -- moved as full lines through an LDS copy, fast_wave IO 2)
    constexpr int PIECE = ((SEQ::NVS + 1) / 2) * 128;
    __shared__ __attribute__((aligned(16))) double fast_row[CONTIG ? NWAVES * 2 * PIECE : IO == 1 ? NWAVES * SEQ::NVS * 64 : 2];
    uint32_t first_byte = 0, piece_bytes = 0;
    if constexpr (CONTIG) {
        first_byte = a.blob[a.o_ranges + 2 * wave_global] * 8u;
        piece_bytes = a.blob[a.o_ranges + 2 * wave_global + 1] * 8u;
    }
    double* const row_lds = fast_row + (CONTIG ? wave * 2 * PIECE : IO == 1 ? wave * SEQ::NVS * 64 : 0);
    fast_setup(seq, a, wave_global, lane, first_byte);
    if (grid_slot < a.batch) {  // (the guesses of the first system; every later one's a system ahead)
        if constexpr (CONTIG) {
            bufword4_t q[(SEQ::NVS + 1) / 2];
            piece_load<(SEQ::NVS + 1) / 2>(q, row_at(a.x0 + (uint64_t)grid_slot * a.n_row), first_byte, piece_bytes, lane);
            piece_to_lds<(SEQ::NVS + 1) / 2>(q, row_lds, lane);
        } else {
            fast_fetch(seq, a, grid_slot);
        }
    }
    fast_factor(seq, a);
    constexpr int GROUPS = NWAVES * 16;      // a turn at the totals: thread = (value tid & 3, group tid >> 2)
    __shared__ double fast_part[2][4 * 16];  // [parity of k][value][wavefront]: the wavefronts' partials
    __shared__ int fast_pflag[2][16];
    __shared__ double fast_gath[NWAVES * 64];  // the gather's first fold
    __shared__ int fast_gflag[NWAVES * 64];
    auto put = [&](gridchunk_t* p, double x, unsigned int w, unsigned int q) {
        const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
        gridchunk_t c;
        c.x = (unsigned int)u, c.y = (unsigned int)(u >> 32), c.z = q, c.w = w;
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(c) : "memory");
    };
    auto val = [](const gridchunk_t& c) { return __builtin_bit_cast(double, ((unsigned long long)c.y << 32) | c.x); };
    // sequence number of the last system this workgroup published (continues from launch to launch): the largest in its four ring
    // places, looked at by four lanes at once
    unsigned int ring_q;
    {
        gridchunk_t ring_c;
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(ring_c) : "v"(&gs0->ring_p[lane & 3][grid_wg][0]) : "memory");
        const unsigned int z0 = __builtin_amdgcn_readlane(ring_c.z, 0), z1 = __builtin_amdgcn_readlane(ring_c.z, 1);
        const unsigned int z2 = __builtin_amdgcn_readlane(ring_c.z, 2), z3 = __builtin_amdgcn_readlane(ring_c.z, 3);
        const unsigned int m0 = z0 > z1 ? z0 : z1, m1 = z2 > z3 ? z2 : z3;
        ring_q = m0 > m1 ? m0 : m1;
    }
    bool f1 = false, f2 = false;  // systems k - 1 / k - 2 of this slot were published: k - 1's totals may be this workgroup's turn
    uint64_t sys1 = 0;
    unsigned int q1 = 0, q2 = 0;
    for (uint64_t k = 0;; ++k) {
        const uint64_t sys = grid_slot + k * (uint64_t)n_slots;
        const bool have = sys < a.batch;
        // (past its last system a workgroup stays for its turn at the totals of that one, if it is its turn, and for nothing else)
        if (!have && !(f1 && q1 % grid_wgs == grid_wg)) break;
        const unsigned int kp = (unsigned int)k & 1u;
        unsigned int q0 = 0;
#ifdef EZPZ_JIT_STAMPS  // (diagnostic compilations only, tools/ladder_stamps.py: the pointer and the counter are registers the kernel has no room for)
        unsigned long long* const stamps = a.stamps && tid == 0 && have ? a.stamps + (sys * grid_wgs + grid_wg) * 16 : nullptr;
        int stamp_n = 0;
        auto stamp = [&]() {
            if (stamps && stamp_n < 16) stamps[stamp_n++] = (unsigned long long)wall_clock64();
        };
#else
        auto stamp = [] {};
#endif
        stamp();  // 0: start
        // (opaque, every time round: what the lanes' addresses into the scratch have in common would otherwise be computed once,
        // ahead of the loop, and held in vector registers across the slots -- 15 registers too many for four wavefronts per SIMD)
        GridScratch* gs = gs0;
        asm volatile("" : "+s"(gs));
        if (have) {
            const uint64_t sys_n = sys + n_slots;
            double v[4];
            unsigned int wave_flags;
            if constexpr (CONTIG)
                wave_flags = fast_wave<2>(seq, a, sys, sys_n, sys_n < a.batch, wave_global, lane, v, row_lds + kp * PIECE, row_lds + (kp ^ 1u) * PIECE, first_byte, piece_bytes);
            else
                wave_flags = fast_wave<IO>(seq, a, sys, sys_n, sys_n < a.batch, wave_global, lane, v, IO == 1 ? row_lds : nullptr);
            stamp();  // 1: both steps taken, stores issued
            if (lane == 63) {
#pragma unroll
                for (int i = 0; i < 4; ++i) fast_part[kp][16 * i + wave] = v[i];
                fast_pflag[kp][wave] = (int)wave_flags;
            }
        }
        // flow control: system k's line takes the ring place of system k - 4's, whose totals were taken three systems ago (k - 2's
        // "taken" mark is looked for: there at the first look, and it implies every earlier one)
        if (have && f2 && tid == 0 && q2 % grid_wgs != grid_wg) {
            for (unsigned int spins = 0;; ++spins) {
                const gridchunk_t c = grid_peek(&gs->ring_v[q2 & 3u][0]);
                if (c.z == q2) break;
                if (EZPZ_GRID_TIMED_OUT(spins, &gs->dead)) {
                    __hip_atomic_store(&gs->dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (have) __syncthreads();
        stamp();  // 2: the barrier
        if (have) {  // wavefront 0 folds the wavefronts' partials (lane = value) and publishes them
            q0 = ++ring_q;
            if (wave == 0) {
                const int i = lane & 3;
                double t = fast_part[kp][16 * i];
                unsigned int fl = (unsigned int)fast_pflag[kp][0];
#pragma unroll
                for (int w2 = 1; w2 < NWAVES; ++w2) {
                    const double o = fast_part[kp][16 * i + w2];
                    const double sum = t + o, mxm = fmax_nc(t, o);
                    t = i < 3 ? sum : mxm;
                    fl |= (unsigned int)fast_pflag[kp][w2];
                }
                if (lane < 4) put(&gs->ring_p[q0 & 3u][grid_wg][lane], t, lane == 0 ? fl : 0u, q0);
            }
        }
        stamp();  // 3: published
        // ---- this workgroup's turn: the totals of system k - 1 and the reference's verdict on them ------------------------------------
        if (f1 && q1 % grid_wgs == grid_wg) {
            const int i = tid & 3, j = tid >> 2;
            double acc = i < 3 ? 0.0 : __builtin_nan("");
            unsigned int fl = 0;
            bool dead = false;
            for (uint32_t g0 = (uint32_t)j; g0 < grid_wgs; g0 += 4u * GROUPS) {
                gridchunk_t c[4];
                const gridchunk_t* src = &gs->ring_p[q1 & 3u][g0][i];
                const bool in1 = g0 + GROUPS < grid_wgs, in2 = g0 + 2u * GROUPS < grid_wgs, in3 = g0 + 3u * GROUPS < grid_wgs;
                const gridchunk_t* s1 = in1 ? src + 8 * GROUPS : src;
                const gridchunk_t* s2 = in2 ? src + 16 * GROUPS : src;
                const gridchunk_t* s3 = in3 ? src + 24 * GROUPS : src;
                for (unsigned int spins = 0;; ++spins) {
                    asm volatile(
                        "global_load_dwordx4 %0, %4, off sc0 sc1\n\t"
                        "global_load_dwordx4 %1, %5, off sc0 sc1\n\t"
                        "global_load_dwordx4 %2, %6, off sc0 sc1\n\t"
                        "global_load_dwordx4 %3, %7, off sc0 sc1\n\t"
                        "s_waitcnt vmcnt(0)"
                        : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3])
                        : "v"(src), "v"(s1), "v"(s2), "v"(s3)
                        : "memory");
                    if (c[0].z == q1 && c[1].z == q1 && c[2].z == q1 && c[3].z == q1) break;
                    if (EZPZ_GRID_TIMED_OUT(spins, &gs->dead)) {
                        __hip_atomic_store(&gs->dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        dead = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                const bool in[4] = {true, in1, in2, in3};
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const double o = val(c[h]);
                    const double sum = acc + o, mxm = fmax_nc(acc, o);
                    acc = in[h] ? (i < 3 ? sum : mxm) : acc;
                    fl |= in[h] ? c[h].w : 0u;
                }
            }
            fast_gath[tid] = acc;
            fast_gflag[tid] = (int)(fl | (dead ? 4u : 0u));
            __syncthreads();
            if (wave == 0) {
                double t = fast_gath[i];
                unsigned int fl2 = (unsigned int)fast_gflag[i];
                for (int jj = 1; jj < GROUPS; ++jj) {
                    const double o = fast_gath[i + 4 * jj];
                    const double sum = t + o, mxm = fmax_nc(t, o);
                    t = i < 3 ? sum : mxm;
                    fl2 |= (unsigned int)fast_gflag[i + 4 * jj];
                }
                unsigned int flags = 0;
#pragma unroll
                for (int l2 = 0; l2 < 4; ++l2) flags |= (unsigned int)__builtin_amdgcn_readlane((int)fl2, l2);
                const bool stands = fast_verdict(a, sys1, t, flags, lane);
                if (lane == 0) {
                    gridchunk_t c;
                    c.x = stands ? 1u : 2u, c.y = 0, c.z = q1, c.w = 0;
                    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(&gs->ring_v[q1 & 3u][0]), "v"(c) : "memory");
                }
            }
        }
        stamp();  // 4: (its turn: the totals of k - 1 taken)
        f2 = f1, q2 = q1;
        f1 = have, sys1 = sys, q1 = q0;
    }
}

// ---- one LANE per system ------------------------------------------------------------------------------------------------------
// Batches of one small system (a sketch fixture: <= ~24 variables, one or a few components) need no cross-lane traffic at
// all: every lane runs the whole LM loop of newton.rs:29-145 on its own system with its own lambda / accept / iteration
// count, state in registers, and picks the next system of the batch when it is done (persistent lanes: a wavefront's
// lanes are at different iterations of different systems, none waits for the slowest).  Sums run in row order on one
// lane -- the reference's own order.  The class struct is the same straight-line code as above with the constraint
// parameters as literals (every system of the batch shares them); C::load / C::store map the caller's variable order.
// The Jacobian is not kept between iterations: it is re-evaluated at `xj`, the values of the last accepted step (where
// the reference refreshed it, newton.rs:121) -- the same bits, fewer live registers; its Degenerate warnings are logged
// the first time only, like the reference's one refresh.
struct LaneArgs {
    const double* x0;
    double* x_out;
    EzpzStatus* status;
    uint8_t* unsat_mask;  // optional
    uint64_t* warn_log;   // optional
    uint32_t warn_cap, max_iterations;
    uint32_t n_row, n_cons;
    uint64_t batch;
    double residual_tolerance, step_tolerance, initial_lambda;
    // a heterogeneous batch's systems of this topology, in place (mixed.hip): system i's values start at row_offset[i] of the
    // caller's ragged x0 / x_out and its status goes to status[sys_of[i]]; null: dense rows of n_row values, status[i]
    const uint64_t* row_offset;
    const uint32_t* sys_of;
    DoneWord done;  // one-call launches: the completion word (dev_types.hpp)
};
static_assert(sizeof(LaneArgs) == 160, "LaneArgs is restated on the host (jit.cpp: LaneArgsHost)");

template <class C, bool UNIT_W, bool RESIDENT = false>
__device__ __forceinline__ void lane_kernel(const LaneArgs& a) {
    using namespace ezpz::dev;
    constexpr int NV = C::NV, M = C::M > 0 ? C::M : 1, ZJ = C::ZJS > 0 ? C::ZJS : 1;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    // (a resident launch -- DoneWord::request, one workgroup -- serves one request after the other on the same buffers)
    __shared__ unsigned long long resident_word;
    const unsigned long long born = wall_clock64();
    DoneWord done = a.done;
    do {
    uint64_t next = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool have = false;
    uint64_t sys = 0;
    double x[NV], xj[NV], r[M];
    const double par[C::NC > 0 ? C::NC : 1] = {};  // unused: parameters are literals in the class code
    double residual_sq = 0.0, largest = 0.0, lambda = 0.0;
    uint32_t it = 0, pass = 0, pass_jac = 0, nwarn = 0;
    bool r_is_at_x = true, fresh = false;
    for (;;) {
        auto log_mask = [&](unsigned long long m, uint32_t p) {  // Warning::Degenerate, in evaluation order (solver.rs:340-346)
            while (m) {
                const int ci = __builtin_ctzll(m);
                m &= m - 1;
                if (a.warn_log && nwarn < a.warn_cap) a.warn_log[sys * a.warn_cap + nwarn] = ((uint64_t)p << 32) | C::pos_of(ci);
                ++nwarn;
            }
        };
        if (!have && next < a.batch) {  // ---- the next system of the batch: eval() (newton.rs:45, :232-236) ----
            sys = next;
            next += stride;
            have = true;
            C::load(a.x0 + (a.row_offset ? a.row_offset[sys] : sys * a.n_row), x);
#pragma unroll
            for (int k = 0; k < NV; ++k) xj[k] = x[k];
            nwarn = 0;
            double sq = 0.0, mx = __builtin_nan("");
            unsigned long long wm = 0;
            C::residuals(x, par, r, true, sq, mx, wm);
            log_mask(wm, 0);
            residual_sq = sq;
            largest = mx;
            lambda = a.initial_lambda;
            it = 0;
            pass = 2;
            pass_jac = 1;
            fresh = true;  // the Jacobian of eval() is evaluated (and its warnings logged) by the first iteration, or at the end
            r_is_at_x = true;
        }
        if (!__any(have)) break;
        if (!have) continue;
        bool finish = false;
        uint32_t iterations = a.max_iterations, converged = 0;
        if (it >= a.max_iterations) {  // newton.rs:141-144
            finish = true;
        } else if (largest <= a.residual_tolerance) {  // newton.rs:50-60
            iterations = it;
            converged = 1;
            finish = true;
        } else {
            double J[ZJ];
            if constexpr (!C::LINEAR) {
                unsigned long long wm = 0;
                C::jacobian(xj, par, J, wm);
                if (fresh) log_mask(wm, pass_jac);
                fresh = false;
            }
            double d[NV], dmax = __builtin_nan("");
            bool ok = true;
            bool bad = C::solve(J, r, lambda, d, dmax, ok);
            if (!ok && !bad) {  // an operand outside the short division's range: plain divisions for this lane
                dmax = __builtin_nan("");
                double J2[ZJ], x2[NV], r2[M];  // (the Jacobian evaluated again rather than kept alive across the short solve)
                opaque_copy(xj, x2);
                opaque_copy(r, r2);
                if constexpr (!C::LINEAR) {
                    unsigned long long wm = 0;
                    C::jacobian(x2, par, J2, wm);
                }
                bad = C::solve_exact(J2, r2, lambda, d, dmax);
            }
            if (bad) {  // LltError::Numeric: lambda *= 10, burn the iteration (newton.rs:93-99)
                lambda *= LM_LAMBDA_INCR;
                ++it;
            } else {
                double xt[NV], rn[M];
#pragma unroll
                for (int k = 0; k < NV; ++k) xt[k] = x[k] + d[k];  // newton.rs:111-114
                double sq = 0.0, mx = __builtin_nan("");
                unsigned long long wm = 0;
                C::residuals(xt, par, rn, true, sq, mx, wm);
                log_mask(wm, pass++);
                if (sq < residual_sq) {  // strict, newton.rs:118
#pragma unroll
                    for (int k = 0; k < NV; ++k) x[k] = xt[k], xj[k] = xt[k];
#pragma unroll
                    for (int k = 0; k < M; ++k) r[k] = rn[k];
                    lambda *= LM_LAMBDA_DECR;
                    residual_sq = sq;
                    largest = mx;
                    r_is_at_x = true;
                    fresh = !C::LINEAR;  // newton.rs:121: refresh_jacobian (evaluated lazily, see above)
                    pass_jac = pass++;
                } else {  // reject: x += d, x -= d like the reference (newton.rs:124-131), not a copy
#pragma unroll
                    for (int k = 0; k < NV; ++k) x[k] = xt[k] - d[k];
                    lambda *= LM_LAMBDA_INCR;
                    r_is_at_x = false;
                }
                if (dmax <= a.step_tolerance) {  // newton.rs:134-139
                    iterations = it;
                    converged = 1;
                    finish = true;
                } else {
                    ++it;
                }
            }
        }
        if (finish) {
            if constexpr (!C::LINEAR) {
                if (fresh) {  // the refresh of the last accepted step (or of eval()) still owes its warnings
                    double J[ZJ];
                    unsigned long long wm = 0;
                    C::jacobian(xj, par, J, wm);
                    log_mask(wm, pass_jac);
                    fresh = false;
                }
            }
            // unsatisfied check (lib.rs:305-327, :358-370)
            double unsat = 0.0;
            uint8_t* mask = a.unsat_mask ? a.unsat_mask + sys * a.n_cons : nullptr;
            const bool use_r = r_is_at_x && UNIT_W;
            if (use_r && largest < EPS && !isnan(residual_sq)) {
                if (mask)
#pragma unroll
                    for (int ci = 0; ci < C::NC; ++ci) mask[C::pos_of(ci)] = 0;
            } else if (use_r) {
                C::unsatisfied_from_r(r, true, unsat, mask, nullptr);
            } else {
                C::unsatisfied(x, par, true, unsat, mask, nullptr);
            }
            C::store(a.x_out + (a.row_offset ? a.row_offset[sys] : sys * a.n_row), x);
            EzpzStatus st;
            st.iterations = iterations;
            st.converged = converged;
            st.n_unsatisfied = (uint32_t)unsat;
            st.n_warnings = nwarn;
            st.final_residual_inf = (C::M > 0) ? largest : 0.0;
            st.final_lambda = lambda;
            a.status[a.sys_of ? (uint64_t)a.sys_of[sys] : sys] = st;
            have = false;
        }
    }
    if constexpr (RESIDENT) publish_done(done);
    } while (RESIDENT && resident_next(done, born, &resident_word));
}

// ---- one WAVEFRONT per system: the latency shape of a small system ----------------------------------------------------------------
// One solve() call of a sketch fixture is one lane of lane_kernel above walking ~2000 dependent instructions per LM
// iteration at one instruction per 6-8 cycles: `square` 6.7 us per iteration.  Of those, the constraint sweeps and the
// assembly of the normal equations are embarrassingly parallel -- a constraint per lane, a sum of JtJ / Jt r per lane --
// and only the elimination is a chain.  This kernel keeps the lane kernel's arithmetic OPERATION FOR OPERATION (same
// class program, same order of every sum: bit-identical results, tests compare the two) and spreads the parallel part over
// the 64 lanes of one wavefront:
//   sweeps      lane ci evaluates constraint ci (its DevCon in registers, from a table in the code object); lanes diverge
//               only by the kinds present (the generated dispatch lists those);
//   assembly    lane q forms quantity q -- the diagonal / right-hand side of variable v, or one entry of the strict lower
//               part -- as the lane kernel's sum over its (a, b) operand pairs, padded to the longest list;
//   elimination every lane runs the lane kernel's straight-line factorisation and substitutions on the assembled
//               quantities (uniform: one lane's cost), so the step needs no broadcast;
// values live in LDS (x, the Jacobian's slots followed by a zero and -r, the assembled quantities, d): one wavefront, so
// LDS operations are ordered without barriers.  C: the generated class (ClsW: the lane class plus tables and dispatch).
// Between two phases of the wavefront kernel whose lanes exchange values through LDS: the lanes run in lockstep and the LDS
// executes a wavefront's operations in issue order, so no barrier instruction is needed -- but the COMPILER must not move a
// lane's LDS loads above the stores other lanes make before them in program order (it would: a lane that stores nothing in
// a phase looks free to load early), nor keep values other lanes overwrite in registers.  A wavefront-scope fence is that
// promise in its memory model, and costs an s_waitcnt at most.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// lane `l`'s value of v on every lane (l a literal after unrolling: two v_readlane_b32 into a scalar register pair)
__device__ __forceinline__ double lane_value(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

template <class C, bool UNIT_W>
__device__ __forceinline__ void wave_kernel(const LaneArgs& a) {
    using namespace ezpz::dev;
    constexpr int NV = C::NV, M = C::M > 0 ? C::M : 1, ZJ = C::ZJ, NQ = C::NQ, NQR = (NQ + 63) / 64, PMAX = C::PMAX;
    constexpr int O_A = 0, O_NR = ZJ + 1, O_X = ZJ + 1 + M, O_XT = O_X + NV, O_R = O_XT + NV, O_RN = O_R + M, O_Q = O_RN + M,
                  O_D = O_Q + NQ + 1, LDS_DOUBLES = O_D + NV;  // (Q[NQ] = 0: the entries of L a row does not have)
    __shared__ double lds[LDS_DOUBLES];
    __shared__ unsigned long long resident_word;
    double* const A = lds + O_A;    // Jacobian slots, then one zero (the operand of padding pairs), then -r by row
    double* const nr = lds + O_NR;  // -r (the right-hand side's operands)
    double* const xs = lds + O_X;
    double* const xt = lds + O_XT;
    double* const r = lds + O_R;
    double* const rn = lds + O_RN;
    double* const Q = lds + O_Q;
    double* const dl = lds + O_D;
    const int lane = threadIdx.x;
    const bool is_con = lane < C::NC;
    const DevCon c = C::con(is_con ? lane : 0);
    uint32_t pr[NQR][PMAX];
#pragma unroll
    for (int k = 0; k < NQR; ++k)
#pragma unroll
        for (int p = 0; p < PMAX; ++p) pr[k][p] = C::pair(k, p, lane);
    if (lane == 0) A[ZJ] = 0.0, Q[NQ] = 0.0;
    uint32_t rs[C::NROW];  // where this lane's row of L sits among the assembled quantities (tail_wave)
#pragma unroll
    for (int k = 0; k < C::NROW; ++k) rs[k] = C::row_slot(k, lane);
    if constexpr (C::LINEAR) {
        for (int s2 = lane; s2 < ZJ; s2 += 64) A[s2] = C::jconst(s2);
    }
    wave_sync();
    const unsigned long long born = wall_clock64();
    DoneWord done = a.done;
    do {
    for (uint64_t sys = blockIdx.x; sys < a.batch; sys += gridDim.x) {
        uint32_t nwarn = 0;
        auto log_mask = [&](unsigned long long m, uint32_t p) {  // Warning::Degenerate, in evaluation order (solver.rs:340-346)
            while (m) {
                const int ci = __builtin_ctzll(m);
                m &= m - 1;
                if (lane == 0 && a.warn_log && nwarn < a.warn_cap) a.warn_log[sys * a.warn_cap + nwarn] = ((uint64_t)p << 32) | C::pos_of(ci);
                ++nwarn;
            }
        };
        // residuals of the constraints at `xv` into `out` (weighted) and, NEG, their negatives into nr; degenerate lanes
        auto residuals_at = [&](const double* xv, double* out, bool neg) -> unsigned long long {
            bool deg = false;
            if (is_con) {
                double r0, r1;
                deg = C::residual_of(c, xv, r0, r1);
                const double w0 = c.weight * r0;
                out[c.row0] = w0;
                if (neg) nr[c.row0] = -w0;
                if (c.nrows > 1) {
                    const double w1 = c.weight * r1;
                    out[c.row0 + 1] = w1;
                    if (neg) nr[c.row0 + 1] = -w1;
                }
            }
            wave_sync();
            return C::LINEAR ? 0ull : __ballot(deg);
        };
        auto jacobian_at = [&](const double* xv) -> unsigned long long {
            if constexpr (C::LINEAR) {
                return 0ull;
            } else {
                bool deg = false;
                if (is_con) {
                    JacWriter<double*> w;
                    w.jv = A;
                    w.jbase = c.jbase;
                    w.loc[0] = ((const uint32_t*)c.jloc)[0], w.loc[1] = ((const uint32_t*)c.jloc)[1];
                    w.loc[2] = ((const uint32_t*)c.jloc)[2], w.loc[3] = ((const uint32_t*)c.jloc)[3];
                    w.weight = c.weight;
                    deg = C::jacobian_of(c, xv, w);
                }
                wave_sync();
                return __ballot(deg);
            }
        };
        // the sum of squares and maximum in request order, on every lane alike (the reference's sequential sum over the rows:
        // the accept test `sum < previous` of a stalled solve hinges on its last bit)
        auto sum_rows = [&](const double* rows, double& sq, double& mx) {
            sq = 0.0;
            mx = __builtin_nan("");
            C::sum_rows(rows, sq, mx);
        };
        // ---- load; eval() (newton.rs:45, :232-236) ----------------------------------------------------------------------------
        wave_sync();  // (the previous system's last reads of the values)
        if (lane < NV) xs[lane] = a.x0[sys * a.n_row + C::var_of(lane)];
        wave_sync();
        log_mask(residuals_at(xs, r, true), 0);
        log_mask(jacobian_at(xs), 1);
        double residual_sq, largest;
        sum_rows(r, residual_sq, largest);
        double lambda = a.initial_lambda;
        uint32_t it = 0, pass = 2, iterations = a.max_iterations, converged = 0;
        bool r_is_at_x = true;
        for (;;) {
            if (it >= a.max_iterations) break;            // newton.rs:141-144
            if (largest <= a.residual_tolerance) {        // newton.rs:50-60
                iterations = it;
                converged = 1;
                break;
            }
            // normal equations: quantity q = the lane kernel's sum over its operand pairs (newton.rs:73-86)
#pragma unroll
            for (int k = 0; k < NQR; ++k) {
                double acc = 0.0;
#pragma unroll
                for (int p = 0; p < PMAX; ++p) acc += A[pr[k][p] & 0xFFFFu] * A[pr[k][p] >> 16];
                if (k * 64 + lane < NQ) Q[k * 64 + lane] = acc;
            }
            wave_sync();
            double dmax = __builtin_nan("");
            bool ok = true;
            bool bad;
            if constexpr (C::HAS_TAIL_WAVE) {  // the elimination across the lanes (comp_program.cpp: emit_tail_wave), same bits
                bool fin = true;
                bad = C::tail_wave(Q, lambda, rs, lane, dl, dmax, fin);
                if (!fin && !bad) {  // a right-hand side that is not finite: the serial order (comp_program.cpp: emit_tail_wave)
                    wave_sync();
                    dmax = __builtin_nan("");
                    bad = C::tail_exact(Q, lambda, dl, dmax);
                }
            } else {
                bad = C::tail(Q, lambda, dl, dmax, ok);
                if (!ok && !bad) {  // an operand outside the short division's range: plain divisions
                    dmax = __builtin_nan("");
                    bad = C::tail_exact(Q, lambda, dl, dmax);
                }
            }
            wave_sync();  // (every lane has written the same d: read back by lane index below)
            if (bad) {  // LltError::Numeric: lambda *= 10, burn the iteration (newton.rs:93-99)
                lambda *= LM_LAMBDA_INCR;
                ++it;
                continue;
            }
            if (lane < NV) xt[lane] = xs[lane] + dl[lane];  // newton.rs:111-114
            wave_sync();
            log_mask(residuals_at(xt, rn, false), pass++);
            double sq, mx;
            sum_rows(rn, sq, mx);
            if (sq < residual_sq) {  // strict, newton.rs:118
                if (lane < NV) xs[lane] = xt[lane];
                if (lane < C::M) {
                    const double w = rn[lane];
                    r[lane] = w;
                    nr[lane] = -w;
                }
                lambda *= LM_LAMBDA_DECR;
                residual_sq = sq;
                largest = mx;
                r_is_at_x = true;
                wave_sync();
                log_mask(jacobian_at(xs), pass++);  // newton.rs:121: refresh_jacobian
            } else {  // reject: x += d, x -= d like the reference (newton.rs:124-131), not a copy
                if (lane < NV) xs[lane] = xt[lane] - dl[lane];
                lambda *= LM_LAMBDA_INCR;
                r_is_at_x = false;
                wave_sync();
            }
            if (dmax <= a.step_tolerance) {  // newton.rs:134-139
                iterations = it;
                converged = 1;
                break;
            }
            ++it;
        }
        // ---- unsatisfied check (lib.rs:305-327, :358-370) and write-back --------------------------------------------------------------
        uint8_t* mask = a.unsat_mask ? a.unsat_mask + sys * a.n_cons : nullptr;
        const bool use_r = r_is_at_x && UNIT_W;
        bool unsat_lane = false;
        if (!(use_r && largest < EPS && !isnan(residual_sq)) && is_con) {
            double r0, r1;
            if (use_r) {
                r0 = r[c.row0];
                r1 = c.nrows > 1 ? r[c.row0 + 1] : 0.0;
            } else {
                C::residual_of(c, xs, r0, r1);
                if (c.nrows <= 1) r1 = 0.0;
            }
            unsat_lane = !(fabs(r0) < EPS) || !(fabs(r1) < EPS);
        }
        if (mask && is_con) mask[c.pos] = unsat_lane ? 1 : 0;
        const uint32_t n_unsat = (uint32_t)__popcll(__ballot(unsat_lane));
        if (lane < NV) a.x_out[sys * a.n_row + C::var_of(lane)] = xs[lane];
        if (lane == 0) {
            EzpzStatus st;
            st.iterations = iterations;
            st.converged = converged;
            st.n_unsatisfied = n_unsat;
            st.n_warnings = nwarn;
            st.final_residual_inf = (C::M > 0) ? largest : 0.0;
            st.final_lambda = lambda;
            a.status[sys] = st;
        }
    }
    publish_done(done);
    } while (resident_next(done, born, &resident_word));
}

}  // namespace jit
}  // namespace ezpz
