"""Test infrastructure: a numpy executor of the FRONTAL launch shape's plan blob (csrc/front_types.hpp, csrc/fronts.cpp),
operation for operation what front_kernel.hip.hpp does with it -- assembly streams, extend-add through the children's maps,
partial dense Cholesky of the pivot block, Schur complement, chunks between workgroups, backward substitution.  It checks the
HOST symbolic phase on the CPU (no GPU in the build container): the step it produces must equal the dense solve of
(JtJ + lambda I) d = -Jt r assembled from the oracle's Jacobian rows (reference: ezpz/src/solver/newton.rs:73-102).
Not part of the product."""
import ctypes as C

import numpy as np

import ezpz_amd as E
from oracle import oracle as O

FRONT_DESC = np.dtype([("K", "<u2"), ("S", "<u2"), ("n_child", "<u2"), ("flags", "<u2"), ("panel", "<u4"), ("upd", "<u4"),
                       ("rows", "<u4"), ("child0", "<u4"), ("src_off", "<u4"), ("src_n", "<u2"), ("n_kids_local", "<u2"),
                       ("src_v", "u1", (4,)), ("up_chunk", "<u4"), ("exp0", "<u4"), ("parent_local", "<u4")])
assert FRONT_DESC.itemsize == 48
FRONT_CHILD = np.dtype([("upd", "<u4"), ("rows", "<u2"), ("flags", "<u2"), ("map", "<u4"), ("pad", "<u4")])
assert FRONT_CHILD.itemsize == 16
FRONT_GHOST = np.dtype([("local", "<u4"), ("chunk", "<u4")])
WG_FIELDS = ["n_loc", "n_own", "n_ghost", "n_cons", "n_rows", "zj", "n_fronts", "n_levels", "o_var_glob", "o_cons", "o_tables",
             "tab_bytes", "t_level_ptr", "t_children", "t_rows", "t_exports", "t_maps", "t_stream", "asm_word0", "asm_trips", "t_cons", "o_ghosts", "l_x", "l_d", "l_r",
             "l_rn", "l_jv", "l_panels", "l_upool", "ws_doubles", "n_remote_children", "t_sched", "o_slotmap", "pad0", "pad1", "pad2"]
FRONT_WG = np.dtype([(f, "<u4") for f in WG_FIELDS])
assert FRONT_WG.itemsize == 144
DEVCON = np.dtype([("ids", "<u4", (8,)), ("param", "<f8"), ("weight", "<f8"), ("row0", "<u4"), ("jbase", "<u4"), ("pos", "<u4"),
                   ("kind", "u1"), ("tag", "u1"), ("nrows", "u1"), ("nslots", "u1"), ("jloc", "u1", (16,))])
assert DEVCON.itemsize == 80
FASM_DIAG, FASM_RHS, FASM_NOP = 1 << 17, 1 << 18, 1 << 19
FRONT_REMOTE_PARENT, FRONT_EXPORTS, FRONT_CHILD_REMOTE = 1, 2, 1


class Plan:
    def __init__(self, recs, n_vars, wgs=1, max_wgs=32, lds_bytes=160 * 1024):
        recs = np.ascontiguousarray(recs)
        self.recs, self.n_vars = recs, n_vars
        L = E.lib()
        info = np.zeros(16, np.uint64)
        size = L.ezpz_debug_front_plan(recs.ctypes.data, len(recs), n_vars, wgs, max_wgs, lds_bytes, None, 0, info.ctypes.data)
        self.ok = size > 0
        if not self.ok:
            return
        buf = np.zeros(size, np.uint8)
        L.ezpz_debug_front_plan(recs.ctypes.data, len(recs), n_vars, wgs, max_wgs, lds_bytes, buf.ctypes.data, size, info.ctypes.data)
        self.blob = buf
        (self.n_wgs, self.n_chunks, self.bad_chunk0, self.verdict_chunk, self.lds_bytes, self.n_fronts, self.n_levels, self.max_rows,
         self.max_pivots, self.threads, self.model_cycles, self.panel_doubles, self.update_doubles, self.ordering, self.n_components) = [int(v) for v in info[:15]]
        self.wgs = np.frombuffer(buf, FRONT_WG, self.n_wgs, 0)

    def arr(self, dtype, off, count):
        return np.frombuffer(self.blob, dtype, count, int(off))

    def wg_tables(self, g):
        W = self.wgs[g]
        t0 = int(W["o_tables"])
        nf = int(W["n_fronts"])
        descs = self.arr(FRONT_DESC, t0, nf)
        level_ptr = self.arr("<u4", t0 + int(W["t_level_ptr"]), int(W["n_levels"]) + 1)
        n_children = int(descs["child0"][-1]) + int(descs["n_child"][-1]) if nf else 0
        children = self.arr(FRONT_CHILD, t0 + int(W["t_children"]), n_children)
        rows = self.arr("<u2", t0 + int(W["t_rows"]), (int(W["t_exports"]) - int(W["t_rows"])) // 2)
        exports = self.arr("<u4", t0 + int(W["t_exports"]), (int(W["t_maps"]) - int(W["t_exports"])) // 4)
        maps = self.arr("u1", t0 + int(W["t_maps"]), int(W["t_stream"]) - int(W["t_maps"]))
        return descs, level_ptr, children, rows, exports, maps


def evaluate(plan, g, x_caller):
    """The local r (weighted) and Jacobian values of workgroup g's constraints at x (caller numbering), by the oracle's evaluators."""
    W = plan.wgs[g]
    cons = plan.arr(DEVCON, W["o_cons"], int(W["n_cons"]))
    r = np.zeros(int(W["n_rows"]) + 1)
    jv = np.zeros(int(W["zj"]) + 1)
    for d in cons:
        rec = plan.recs[int(d["pos"])]
        res, _ = O.residual(rec, x_caller)
        rows, _ = O.jacobian_rows(rec, x_caller)
        w = float(d["weight"])
        e = 0
        for k, row in enumerate(rows):
            r[int(d["row0"]) + k] = w * res[k]
            for (_vid, pd) in row:
                code = int(d["jloc"][e])
                slot = int(d["jbase"]) + (code & 0x7F)
                if code & 0x80:
                    jv[slot] += w * pd
                else:
                    jv[slot] = w * pd
                e += 1
    return r, jv


def tri(a, b):
    return a * (a + 1) // 2 + b


def linear_step(plan, x_caller, lam):
    """One linear solve through the plan: returns (d in caller numbering, bad flag)."""
    G = plan.n_wgs
    chunks = {}
    ws = [np.zeros(int(plan.wgs[g]["ws_doubles"])) for g in range(G)]
    bad = [False] * G
    evals = [evaluate(plan, g, x_caller) for g in range(G)]

    def factor(g):
        W = plan.wgs[g]
        descs, level_ptr, children, rows, exports, maps = plan.wg_tables(g)
        w = ws[g]
        r, jv = evals[g]
        t_end = int(W["t_cons"]) if int(W["t_cons"]) != 0xFFFFFFFF else int(W["tab_bytes"])
        stream = plan.arr("<u4", int(W["o_tables"]) + int(W["t_stream"]), (t_end - int(W["t_stream"])) // 4)
        pan = int(W["l_panels"])
        # the assembly of the linear solve: panels and update matrices zeroed, then the workgroup's assembly stream
        w[pan:] = 0.0
        offs = stream[int(W["asm_word0"]): int(W["asm_word0"]) + int(W["asm_trips"])]
        for tr in range(int(W["asm_trips"])):
            base = int(offs[tr])
            wdt = int(stream[base]) >> 24
            for l in range(64):
                hdr = int(stream[base + l])
                assert hdr >> 24 == wdt
                if hdr & FASM_NOP:
                    continue
                acc = 0.0
                for q in range(wdt):
                    op = int(stream[base + 64 * (1 + q) + l])
                    a, b = op & 0xFFFF, op >> 16
                    acc += jv[a] * (-r[b]) if hdr & FASM_RHS else jv[a] * jv[b]
                if hdr & FASM_DIAG:
                    acc += lam
                w[pan + (hdr & 0xFFFF)] = acc
        for lv in range(int(W["n_levels"])):
            for k in range(int(level_ptr[lv]), int(level_ptr[lv + 1])):
                d = descs[k]
                K, S = int(d["K"]), int(d["S"])
                S1, R = S + 1, S - K
                nU = (R + 1) * (R + 2) // 2
                P = w[int(d["panel"]): int(d["panel"]) + S1 * K]
                U = w[int(d["upd"]): int(d["upd"]) + nU] if R else w[0:0]  # (a root front has no update matrix: nothing may be written)
                # the front's source stream: elements of its local children's update matrices, gathered by destination
                base = int(d["src_off"])
                n_e = int(d["src_n"])
                for tr in range((n_e + 63) // 64):
                    vdt = int(d["src_v"][min(tr, 3)])
                    for l in range(64):
                        hdr = int(stream[base + l])
                        if hdr & FASM_NOP:
                            continue
                        acc = w[pan + (hdr & 0xFFFF)]
                        for q in range(vdt):
                            sw = int(stream[base + 64 * (1 + q) + l])
                            acc += w[pan + (sw & 0xFFFF)] + w[pan + (sw >> 16)]
                        w[pan + (hdr & 0xFFFF)] = acc
                    base += 64 * (1 + vdt)
                # children in other workgroups (this workgroup's own are sources of the stream)
                for c in children[int(d["child0"]): int(d["child0"]) + int(d["n_child"])]:
                    Rc1 = int(c["rows"])
                    m = maps[int(c["map"]): int(c["map"]) + Rc1]
                    for a in range(Rc1):
                        for b in range(a + 1):
                            e = tri(a, b)
                            if e == Rc1 * (Rc1 + 1) // 2 - 1:
                                continue  # the right-hand side's row against itself is nobody's
                            assert int(c["flags"]) & FRONT_CHILD_REMOTE
                            val = chunks[int(c["upd"]) + e]
                            i, j = int(m[a]), int(m[b])
                            assert i >= j, (i, j, a, b)
                            if j < K:
                                P[j * S1 + i] += val
                            else:
                                U[tri(i - K, j - K)] += val
                # partial factorisation: lane r holds row r of the K pivot columns; row S = right-hand side
                A = P.reshape(K, S1).T.copy()  # [row, col]
                for j in range(K):
                    piv = A[j, j]
                    if not piv > 0.0:
                        bad[g] = True
                    with np.errstate(all="ignore"):
                        rinv = 1.0 / np.sqrt(piv)
                        lcol = A[:, j] * rinv
                    lcol[:j + 1] = 0.0
                    A[j + 1:, j] = lcol[j + 1:]
                    A[j, j] = rinv  # (the factor's diagonal is kept as 1 / d_j)
                    for kk in range(j + 1, K):
                        A[kk:, kk] -= lcol[kk:] * lcol[kk]
                P[:] = A.T.reshape(-1)
                # Schur complement (row R = right-hand side)
                for a in range(R + 1):
                    for b in range(a + 1):
                        if a == R and b == R:
                            continue
                        acc = 0.0
                        for kk in range(K):
                            acc += P[kk * S1 + K + a] * P[kk * S1 + K + b]
                        U[tri(a, b)] -= acc
                if int(d["flags"]) & FRONT_REMOTE_PARENT:
                    for e in range(nU - 1):
                        chunks[int(d["up_chunk"]) + e] = U[e]

    def backward(g):
        W = plan.wgs[g]
        descs, level_ptr, children, rows, exports, maps = plan.wg_tables(g)
        w = ws[g]
        dv = w[int(W["l_d"]): int(W["l_d"]) + int(W["n_loc"])]
        ghosts = plan.arr(FRONT_GHOST, W["o_ghosts"], int(W["n_ghost"]))
        for gh in ghosts:
            dv[int(gh["local"])] = chunks[int(gh["chunk"])]
        for lv in reversed(range(int(W["n_levels"]))):
            for k in range(int(level_ptr[lv]), int(level_ptr[lv + 1])):
                d = descs[k]
                K, S = int(d["K"]), int(d["S"])
                S1 = S + 1
                P = w[int(d["panel"]): int(d["panel"]) + S1 * K]
                frow = rows[int(d["rows"]): int(d["rows"]) + S]
                t = np.array([P[kk * S1 + S] for kk in range(K)])
                for rr in range(K, S):
                    xr = dv[int(frow[rr])]
                    for kk in range(K):
                        t[kk] -= P[kk * S1 + rr] * xr
                xs = np.zeros(K)
                for j in reversed(range(K)):
                    xs[j] = t[j] * P[j * S1 + j]
                    for kk in range(j):
                        t[kk] -= P[kk * S1 + j] * xs[j]
                for kk in range(K):
                    dv[int(frow[kk])] = xs[kk]
                    if int(d["flags"]) & FRONT_EXPORTS and int(exports[int(d["exp0"]) + kk]) != 0xFFFFFFFF:
                        chunks[int(exports[int(d["exp0"]) + kk])] = xs[kk]

    for g in range(1, G):
        factor(g)
    factor(0)
    backward(0)
    for g in range(1, G):
        backward(g)
    out = np.full(plan.n_vars, np.nan)
    for g in range(G):
        W = plan.wgs[g]
        vg = plan.arr("<u4", W["o_var_glob"], int(W["n_loc"]))
        dv = ws[g][int(W["l_d"]): int(W["l_d"]) + int(W["n_loc"])]
        for k in range(int(W["n_own"])):
            assert np.isnan(out[int(vg[k])]), "a variable eliminated twice"
            out[int(vg[k])] = dv[k]
        for k in range(int(W["n_own"]), int(W["n_loc"])):  # ghosts carry the owner's value
            pass
    return out, any(bad)


def dense_step(recs, n_vars, x, lam):
    """(JtJ + lambda I) d = -Jt r from the oracle's rows, dense (newton.rs:73-102)."""
    rows_j, rs = [], []
    for rec in recs:
        res, _ = O.residual(rec, x)
        rows, _ = O.jacobian_rows(rec, x)
        w = float(rec["weight"])
        for k, row in enumerate(rows):
            jr = np.zeros(n_vars)
            for (vid, pd) in row:
                jr[int(vid)] += w * pd
            rows_j.append(jr)
            rs.append(w * res[k])
    J = np.array(rows_j)
    r = np.array(rs)
    A = J.T @ J + lam * np.eye(n_vars)
    return np.linalg.solve(A, -J.T @ r)
