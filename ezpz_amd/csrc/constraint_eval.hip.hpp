// Device evaluators for the 25 ezpz constraint kinds (gfx950, fp64).
//
// One lane evaluates one constraint: it gathers <= 8 variables from the system's value vector (LDS
// resident), computes 1-2 residuals (`con_residual`) or <= 16 partial derivatives (`con_jacobian`)
// in registers and stores them through the constraint's precomputed Jacobian slots.
//
// The math follows the reference line by line, including its guards, which differ between the residual
// and the Jacobian of the same kind (e.g. `<=` vs `<`):
//   residuals       ezpz/src/constraints.rs:499-950
//   jacobian rows   ezpz/src/constraints.rs:1000-2293, helpers :2361-2647
//   2-vector ops    ezpz/src/vector.rs:1-143
// Built with -ffp-contract=off: the Rust reference never fuses a*b+c, and the LM accept/reject and
// convergence tests compare sums of these values.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

#include "dev_types.hpp"

namespace ezpz {
namespace dev {

constexpr double EPS = 1e-4;  // lib.rs:43
constexpr double PI = 3.14159265358979323846264338327950288;

struct V2 {
    double x, y;
};
__device__ __forceinline__ V2 mk(double x, double y) { return V2{x, y}; }
__device__ __forceinline__ V2 operator-(V2 a, V2 b) { return V2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ V2 operator+(V2 a, V2 b) { return V2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ V2 operator*(V2 a, double s) { return V2{a.x * s, a.y * s}; }
__device__ __forceinline__ double mag(V2 a) { return hypot(a.x, a.y); }          // vector.rs:15-17
__device__ __forceinline__ double mag2(V2 a) { return a.x * a.x + a.y * a.y; }   // vector.rs:20-22 (pow(.,2))
__device__ __forceinline__ double dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ double cross(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
__device__ __forceinline__ V2 perp_ccw(V2 a) { return V2{-a.y, a.x}; }
__device__ __forceinline__ V2 perp_cw(V2 a) { return V2{a.y, -a.x}; }
__device__ __forceinline__ double signed_angle(V2 a, V2 b) { return atan2(cross(a, b), dot(a, b)); }  // :72-74
// reflect(b) = self - 2*(self - proj_b self), division by b.b unguarded (vector.rs:58-69)
__device__ __forceinline__ V2 reflect(V2 a, V2 b) {
    V2 proj = b * (dot(a, b) / dot(b, b));
    V2 rej = a - proj;
    return a - rej * 2.0;
}
// Rotation2: col0 = (cos, sin) (vector.rs:110-143)
struct Rot {
    double c, s;
};
__device__ __forceinline__ V2 rot_apply(Rot r, V2 v) { return V2{(r.c * v.x) - (r.s * v.y), (r.s * v.x) + (r.c * v.y)}; }
__device__ __forceinline__ Rot rot_inv(Rot r) { return Rot{r.c, -r.s}; }
// rotation_for_angle_kind, constraints.rs:2641-2647; Angle::to_radians, datatypes.rs:66-72
__device__ __noinline__ Rot sincos_of(double a) {  // {cos, sin} of a (libm::sin / libm::cos in the reference)
    double s, c;
    sincos(a, &s, &c);
    return Rot{c, s};
}
__device__ __noinline__ double pow_1p5(double v) { return pow(v, 1.5); }
// Not inlined: sincos / atan2 / fmod / pow bring large OCML bodies whose registers would otherwise be live across
// every branch of the evaluator switch (189 -> 124 VGPRs for the sweeps on gfx950).
__device__ __noinline__ Rot rot_for(uint32_t tag, double val) {
    if (tag == EZPZ_ANGLE_PARALLEL) return Rot{1.0, 0.0};
    if (tag == EZPZ_ANGLE_PERPENDICULAR) return Rot{0.0, 1.0};
    double rad = (tag == EZPZ_ANGLE_OTHER_DEG) ? val * (PI / 180.0) : val;
    return sincos_of(rad);
}
// f64::signum: +1 for +0.0, -1 for -0.0, NaN stays NaN
__device__ __forceinline__ double signum(double x) { return isnan(x) ? x : copysign(1.0, x); }
__device__ __forceinline__ double rem_euclid(double x, double m) {
    double r = fmod(x, m);
    return r < 0.0 ? r + fabs(m) : r;
}
// classify_point_arc_coincident, constraints.rs:2593-2606: 0 interior, 1 start, 2 end
__device__ __noinline__ int classify_pac(V2 s, V2 e, V2 p) {
    const double two_pi = 2.0 * PI;
    double a_sp = rem_euclid(signed_angle(s, p), two_pi);
    double a_se = rem_euclid(signed_angle(s, e), two_pi);
    if (a_sp < a_se) return 0;
    return (mag2(e - p) < mag2(s - p)) ? 2 : 1;
}

#define XV(i) (xs[c.ids[(i)]])

// ---------------------------------------------------------------------------------------------------
// Constraint::residual.  Unweighted residuals in r0/r1 (left 0 on a degenerate guard); returns the
// degenerate flag.
// ---------------------------------------------------------------------------------------------------
template <class XP>
__device__ __forceinline__ bool lines_at_angle_residual(V2 u, V2 v, uint32_t tag, double val, double& r0) {
    double len_u = mag(u), len_v = mag(v);
    if (len_u <= EPS || len_v <= EPS) return true;  // :632
    Rot rot = rot_for(tag, val);
    r0 = cross(u, rot_apply(rot_inv(rot), v)) / ((len_u + len_v) * 0.5);  // :639
    return false;
}

// LINEAR_ONLY builds the evaluator for topologies whose every constraint is one of the nine linear kinds (Fixed,
// ScalarEqual, Vertical, Horizontal, the two axis distances, CircleRadius, PointsCoincident, Midpoint): the other
// sixteen bodies are not instantiated.  Their mere presence costs the linear paths 20 % on the 2000 x 2000
// massive_parallel_system (145 vs 59 VGPRs, 70 vs 29 spilled SGPRs, 9.8 k vs 2.0 k instructions of kernel).
#define EZPZ_HEAVY_KIND        \
    if constexpr (LINEAR_ONLY) \
        return false;          \
    else

template <bool LINEAR_ONLY, class XP>
__device__ __forceinline__ bool con_residual(const DevCon& c, XP xs, double& r0, double& r1) {
    r0 = 0.0;
    r1 = 0.0;
    switch (c.kind) {
    case EZPZ_LINE_TANGENT_TO_CIRCLE:  // :509-544
        EZPZ_HEAVY_KIND {
        V2 p0 = mk(XV(0), XV(1)), p1 = mk(XV(2), XV(3)), cc = mk(XV(4), XV(5));
        double radius = fabs(XV(6));
        V2 u = p1 - p0;
        double mag_u = mag(u);
        if (mag_u <= EPS) return true;
        V2 v = cc - p0;
        double side_sign = (c.tag == EZPZ_LINE_RIGHT) ? -1.0 : 1.0;
        r0 = side_sign * cross(u, v) / mag_u - radius;
        return false;
    }
    case EZPZ_CIRCLE_TANGENT_TO_CIRCLE:  // :545-564
        EZPZ_HEAVY_KIND {
        V2 ac = mk(XV(0), XV(1)), bc = mk(XV(3), XV(4));
        double ar = fabs(XV(2)), br = fabs(XV(5));
        double dist = mag(ac - bc);
        r0 = (c.tag == EZPZ_CIRCLE_INTERIOR) ? fabs(ar - br) - dist : ar + br - dist;
        return false;
    }
    case EZPZ_DISTANCE:  // :565-574
        EZPZ_HEAVY_KIND
        r0 = mag(mk(XV(0), XV(1)) - mk(XV(2), XV(3))) - c.param;
        return false;
    case EZPZ_DISTANCE_VAR:  // :575-583
        EZPZ_HEAVY_KIND {
        double dx = XV(0) - XV(2), dy = XV(1) - XV(3);
        r0 = -XV(4) + sqrt(dx * dx + dy * dy);
        return false;
    }
    case EZPZ_VERTICAL_DISTANCE:  // :584-591
        r0 = (XV(1) - XV(3)) - c.param;
        return false;
    case EZPZ_HORIZONTAL_DISTANCE:  // :592-596
        r0 = (XV(0) - XV(2)) - c.param;
        return false;
    case EZPZ_VERTICAL:  // :597-601
        r0 = XV(0) - XV(2);
        return false;
    case EZPZ_HORIZONTAL:  // :602-606
        r0 = XV(1) - XV(3);
        return false;
    case EZPZ_FIXED:  // :607-610
        r0 = XV(0) - c.param;
        return false;
    case EZPZ_SCALAR_EQUAL:  // :611-616
        r0 = XV(0) - XV(1);
        return false;
    case EZPZ_LINES_AT_ANGLE:  // :617-640
        EZPZ_HEAVY_KIND
        return lines_at_angle_residual<XP>(mk(XV(2) - XV(0), XV(3) - XV(1)), mk(XV(6) - XV(4), XV(7) - XV(5)), c.tag,
                                           c.param, r0);
    case EZPZ_POINTS_COINCIDENT:  // :641-648
        r0 = XV(0) - XV(2);
        r1 = XV(1) - XV(3);
        return false;
    case EZPZ_CIRCLE_RADIUS:  // :649-652
        r0 = XV(2) - c.param;
        return false;
    case EZPZ_LINES_EQUAL_LENGTH:  // :653-658
        EZPZ_HEAVY_KIND
        r0 = mag(mk(XV(0), XV(1)) - mk(XV(2), XV(3))) - mag(mk(XV(4), XV(5)) - mk(XV(6), XV(7)));
        return false;
    case EZPZ_ARC_RADIUS:  // :659-682  Distance(center,start), Distance(center,end)
        EZPZ_HEAVY_KIND {
        V2 cc = mk(XV(0), XV(1));
        r0 = mag(cc - mk(XV(2), XV(3))) - c.param;
        r1 = mag(cc - mk(XV(4), XV(5))) - c.param;
        return false;
    }
    case EZPZ_ARC:  // :683-696
        EZPZ_HEAVY_KIND {
        double cx = XV(0), cy = XV(1);
        r0 = hypot(XV(2) - cx, XV(3) - cy) - hypot(XV(4) - cx, XV(5) - cy);
        return false;
    }
    case EZPZ_MIDPOINT:  // :697-711
        r0 = XV(4) - XV(0) / 2.0 - XV(2) / 2.0;
        r1 = XV(5) - XV(1) / 2.0 - XV(3) / 2.0;
        return false;
    case EZPZ_POINT_LINE_DISTANCE:  // :712-740, :2625-2639
        EZPZ_HEAVY_KIND {
        double px = XV(0), py = XV(1), lpx = XV(2), lpy = XV(3), lqx = XV(4), lqy = XV(5);
        double a = lpy - lqy, b = lqx - lpx, cc = (lpx * lqy) - (lqx * lpy);
        double den = hypot(a, b);
        if (den < EPS) return true;
        r0 = (a * px + b * py + cc) / den - c.param;
        return false;
    }
    case EZPZ_VERTICAL_POINT_LINE_DISTANCE:  // :741-762
        EZPZ_HEAVY_KIND {
        double ax = XV(0), ay = XV(1), px = XV(2), py = XV(3), qx = XV(4), qy = XV(5);
        double dx = qx - px, dy = qy - py;
        if (fabs(dx) <= EPS || (dx * dx + dy * dy) <= EPS * EPS) return true;
        r0 = ay - py - dy * (1.0 / dx) * (ax - px) - c.param;
        return false;
    }
    case EZPZ_HORIZONTAL_POINT_LINE_DISTANCE:  // :763-785
        EZPZ_HEAVY_KIND {
        double ax = XV(0), ay = XV(1), px = XV(2), py = XV(3), qx = XV(4), qy = XV(5);
        double dx = qx - px, dy = qy - py;
        if (fabs(dy) <= EPS || (dx * dx + dy * dy) <= EPS * EPS) return true;
        r0 = ax - px - dx * (1.0 / dy) * (ay - py) - c.param;
        return false;
    }
    case EZPZ_SYMMETRIC:  // :786-808  reflect(a - p, q - p) - b + p
        EZPZ_HEAVY_KIND {
        V2 p = mk(XV(0), XV(1)), q = mk(XV(2), XV(3)), a = mk(XV(4), XV(5)), b = mk(XV(6), XV(7));
        V2 res = (reflect(a - p, q - p) - b) + p;
        r0 = res.x;
        r1 = res.y;
        return false;
    }
    case EZPZ_POINT_ARC_COINCIDENT:  // :809-858
        EZPZ_HEAVY_KIND {
        V2 cc = mk(XV(0), XV(1));
        V2 s = mk(XV(2), XV(3)) - cc, e = mk(XV(4), XV(5)) - cc, p = mk(XV(6), XV(7)) - cc;
        double r = mag(s), r_e = mag(e), r_p = mag(p);
        if (r < EPS || r_e < EPS || r_p < EPS) return true;
        V2 e_proj = e * (r / r_e);
        int part = classify_pac(s, e_proj, p);
        V2 f = (part == 0) ? p * (r / r_p - 1.0) : ((part == 2) ? e_proj - p : s - p);
        r0 = f.x;
        r1 = f.y;
        return false;
    }
    case EZPZ_ARC_LENGTH:  // :859-896
        EZPZ_HEAVY_KIND {
        double cx = XV(0), cy = XV(1);
        double ux = XV(2) - cx, uy = XV(3) - cy;
        double r2 = ux * ux + uy * uy;
        if (r2 <= EPS * EPS) return true;
        double alpha = c.param / sqrt(r2);
        Rot sc = sincos_of(alpha);
        double sa = sc.s, ca = sc.c;
        r0 = (XV(4) - cx) - (ca * ux - sa * uy);
        r1 = (XV(5) - cy) - (sa * ux + ca * uy);
        return false;
    }
    case EZPZ_ARC_ANGLE:  // :897-915  LinesAtAngle(center->start, center->end, Other(angle))
        EZPZ_HEAVY_KIND {
        double cx = XV(0), cy = XV(1);
        return lines_at_angle_residual<XP>(mk(XV(2) - cx, XV(3) - cy), mk(XV(4) - cx, XV(5) - cy), c.tag, c.param, r0);
    }
    case EZPZ_POINTS_AT_ANGLE:  // :916-948
        EZPZ_HEAVY_KIND {
        V2 p0 = mk(XV(0), XV(1));
        V2 u = mk(XV(2), XV(3)) - p0, v = mk(XV(4), XV(5)) - p0;
        double len_u = mag(u), len_v = mag(v);
        if (len_u <= EPS || len_v <= EPS) return true;
        Rot rot = rot_for(c.tag, c.param);
        double s = (len_u + len_v) * 0.5;
        V2 res = (v * len_u - rot_apply(rot, u) * len_v) * (1.0 / s);
        r0 = res.x;
        r1 = res.y;
        return false;
    }
    default:
        return false;
    }
}

// ---------------------------------------------------------------------------------------------------
// Constraint::jacobian_rows.  Partial number E (emission order, kinds.hpp) is stored as
// weight*pd through jloc[E]; a degenerate guard leaves the affected row zero, like the reference's
// zero-filled value array (solver.rs:362).
// ---------------------------------------------------------------------------------------------------
template <class JP>
struct JacWriter {
    JP jv;            // Jacobian value array of this system
    uint32_t jbase;
    uint32_t loc[4];  // jloc[16] packed
    double weight;
    template <int E>
    __device__ __forceinline__ void put(double pd) const {
        uint32_t code = (loc[E >> 2] >> ((E & 3) * 8)) & 0xFFu;
        uint32_t slot = jbase + (code & 0x7Fu);
        double w = weight * pd;  // solver.rs:403
        if (code & 0x80u)
            jv[slot] = jv[slot] + w;  // duplicate column inside the row: accumulate (solver.rs:418)
        else
            jv[slot] = w;
    }
    template <int E0, int N>
    __device__ __forceinline__ void zero() const {
#pragma unroll
        for (int e = 0; e < N; ++e) {
            uint32_t code = (loc[(E0 + e) >> 2] >> (((E0 + e) & 3) * 8)) & 0xFFu;
            jv[jbase + (code & 0x7Fu)] = 0.0;
        }
    }
};

// Distance partials (:1160-1204) for points (ia..ia+1), (ib..ib+1) of c.ids, entries E0..E0+3
template <int E0, class XP, class JP>
__device__ __forceinline__ bool distance_jac(const DevCon& c, XP xs, const JacWriter<JP>& w, int ia, int ib) {
    double x0 = XV(ia), y0 = XV(ia + 1), x1 = XV(ib), y1 = XV(ib + 1);
    double dist = mag(mk(x0, y0) - mk(x1, y1));
    if (dist < EPS) {
        w.template zero<E0, 4>();
        return true;
    }
    w.template put<E0 + 0>((x0 - x1) / dist);
    w.template put<E0 + 1>((y0 - y1) / dist);
    w.template put<E0 + 2>((-x0 + x1) / dist);
    w.template put<E0 + 3>((-y0 + y1) / dist);
    return false;
}

// LinesAtAngle partials (:1358-1418) from u = l0.p1 - l0.p0, v = l1.p1 - l1.p0
template <class JP>
__device__ __forceinline__ bool lines_at_angle_jac(V2 u, V2 v, uint32_t tag, double val, const JacWriter<JP>& w) {
    double len_u = mag(u), len_v = mag(v);
    if ((len_u <= EPS) || (len_v <= EPS)) {
        w.template zero<0, 8>();
        return true;
    }
    V2 u_hat = u * (1.0 / len_u), v_hat = v * (1.0 / len_v);
    Rot rot = rot_for(tag, val);
    double s = (len_u + len_v) * 0.5;
    V2 riv = rot_apply(rot_inv(rot), v);
    double a = cross(u, riv);
    double inv_s = 1.0 / s;
    double t = a * inv_s * 0.5;
    V2 df_du = (perp_cw(riv) - u_hat * t) * inv_s;
    V2 df_dv = (perp_ccw(rot_apply(rot, u)) - v_hat * t) * inv_s;
    w.template put<0>(-df_du.x);
    w.template put<1>(-df_du.y);
    w.template put<2>(df_du.x);
    w.template put<3>(df_du.y);
    w.template put<4>(-df_dv.x);
    w.template put<5>(-df_dv.y);
    w.template put<6>(df_dv.x);
    w.template put<7>(df_dv.y);
    return false;
}

template <bool LINEAR_ONLY, class XP, class JP>
__device__ __forceinline__ bool con_jacobian(const DevCon& c, XP xs, const JacWriter<JP>& w) {
    switch (c.kind) {
    case EZPZ_LINE_TANGENT_TO_CIRCLE:  // :1010-1090
        EZPZ_HEAVY_KIND {
        V2 p0 = mk(XV(0), XV(1)), p1 = mk(XV(2), XV(3)), cc = mk(XV(4), XV(5));
        V2 u = p1 - p0;
        double mag_u = mag(u);
        if (mag_u <= EPS) {
            w.template zero<0, 7>();
            return true;
        }
        V2 v = cc - p0;
        double cross_uv = cross(u, v);
        double mag_u3 = mag_u * mag_u * mag_u;
        double sg = (c.tag == EZPZ_LINE_RIGHT) ? -1.0 : 1.0;
        double du_x = sg * (-(u.x * cross_uv) / mag_u3 + v.y / mag_u);
        double du_y = sg * (-(u.y * cross_uv) / mag_u3 - v.x / mag_u);
        double dv_x = sg * (-u.y / mag_u);
        double dv_y = sg * (u.x / mag_u);
        w.template put<0>(-(du_x + dv_x));
        w.template put<1>(-(du_y + dv_y));
        w.template put<2>(du_x);
        w.template put<3>(du_y);
        w.template put<4>(dv_x);
        w.template put<5>(dv_y);
        w.template put<6>(-signum(XV(6)));
        return false;
    }
    case EZPZ_CIRCLE_TANGENT_TO_CIRCLE:  // :1091-1159
        EZPZ_HEAVY_KIND {
        V2 ac = mk(XV(0), XV(1)), bc = mk(XV(3), XV(4));
        double a_r = XV(2), b_r = XV(5);
        V2 d = bc - ac;
        double mag_d = mag(d);
        if (mag_d <= EPS) {
            w.template zero<0, 6>();
            return true;
        }
        V2 u_d = d * (1.0 / mag_d);
        double a_sign = signum(a_r), b_sign = signum(b_r);
        double dar, dbr;
        if (c.tag == EZPZ_CIRCLE_INTERIOR) {
            double inner = signum(fabs(a_r) - fabs(b_r));
            dar = inner * a_sign;
            dbr = -inner * b_sign;
        } else {
            dar = a_sign;
            dbr = b_sign;
        }
        w.template put<0>(u_d.x);
        w.template put<1>(u_d.y);
        w.template put<2>(dar);
        w.template put<3>(-u_d.x);
        w.template put<4>(-u_d.y);
        w.template put<5>(dbr);
        return false;
    }
    case EZPZ_DISTANCE:  // :1160-1204
        EZPZ_HEAVY_KIND
        return distance_jac<0>(c, xs, w, 0, 2);
    case EZPZ_DISTANCE_VAR:  // :1205-1251
        EZPZ_HEAVY_KIND {
        double px = XV(0), py = XV(1), qx = XV(2), qy = XV(3);
        double dist = mag(mk(px, py) - mk(qx, qy));
        if (dist < EPS) {
            w.template zero<0, 5>();
            return true;
        }
        double inv = 1.0 / dist;
        w.template put<0>((px - qx) * inv);
        w.template put<1>((py - qy) * inv);
        w.template put<2>(-(px - qx) * inv);
        w.template put<3>(-(py - qy) * inv);
        w.template put<4>(-1.0);
        return false;
    }
    case EZPZ_VERTICAL_DISTANCE:    // :1252-1269
    case EZPZ_HORIZONTAL_DISTANCE:  // :1270-1287
    case EZPZ_VERTICAL:             // :1288-1311
    case EZPZ_HORIZONTAL:           // :1312-1335
    case EZPZ_SCALAR_EQUAL:         // :1345-1357
        w.template put<0>(1.0);
        w.template put<1>(-1.0);
        return false;
    case EZPZ_FIXED:          // :1336-1344
    case EZPZ_CIRCLE_RADIUS:  // :1505-1512
        w.template put<0>(1.0);
        return false;
    case EZPZ_LINES_AT_ANGLE:  // :1358-1418
        EZPZ_HEAVY_KIND
        return lines_at_angle_jac(mk(XV(2) - XV(0), XV(3) - XV(1)), mk(XV(6) - XV(4), XV(7) - XV(5)), c.tag, c.param, w);
    case EZPZ_LINES_EQUAL_LENGTH:  // :1419-1455
        EZPZ_HEAVY_KIND {
        double x0 = XV(0), y0 = XV(1), x1 = XV(2), y1 = XV(3), x2 = XV(4), y2 = XV(5), x3 = XV(6), y3 = XV(7);
        double len0 = mag(mk(x0, y0) - mk(x1, y1)), len1 = mag(mk(x2, y2) - mk(x3, y3));
        if (len0 < EPS || len1 < EPS) {
            w.template zero<0, 8>();
            return true;
        }
        w.template put<0>((x0 - x1) / len0);
        w.template put<1>((y0 - y1) / len0);
        w.template put<2>((-x0 + x1) / len0);
        w.template put<3>((-y0 + y1) / len0);
        w.template put<4>((-x2 + x3) / len1);
        w.template put<5>((-y2 + y3) / len1);
        w.template put<6>((x2 - x3) / len1);
        w.template put<7>((y2 - y3) / len1);
        return false;
    }
    case EZPZ_POINTS_COINCIDENT:  // :1456-1504
        w.template put<0>(1.0);
        w.template put<1>(-1.0);
        w.template put<2>(1.0);
        w.template put<3>(-1.0);
        return false;
    case EZPZ_ARC_RADIUS:  // :1513-1536: both Distance rows are evaluated even if one is degenerate
        EZPZ_HEAVY_KIND {
        bool d0 = distance_jac<0>(c, xs, w, 0, 2);
        bool d1 = distance_jac<4>(c, xs, w, 0, 4);
        return d0 || d1;
    }
    case EZPZ_ARC:  // :1537-1598
        EZPZ_HEAVY_KIND {
        double cx = XV(0), cy = XV(1);
        double usx = XV(2) - cx, usy = XV(3) - cy, uex = XV(4) - cx, uey = XV(5) - cy;
        double dist0 = hypot(usx, usy), dist1 = hypot(uex, uey);
        if (dist0 <= EPS || dist1 <= EPS) {
            w.template zero<0, 6>();
            return true;
        }
        w.template put<0>(usx / dist0);
        w.template put<1>(usy / dist0);
        w.template put<2>(-uex / dist1);
        w.template put<3>(-uey / dist1);
        w.template put<4>(-usx / dist0 + uex / dist1);
        w.template put<5>(-usy / dist0 + uey / dist1);
        return false;
    }
    case EZPZ_MIDPOINT:  // :1599-1642
        w.template put<0>(1.0);
        w.template put<1>(-0.5);
        w.template put<2>(-0.5);
        w.template put<3>(1.0);
        w.template put<4>(-0.5);
        w.template put<5>(-0.5);
        return false;
    case EZPZ_POINT_LINE_DISTANCE:  // :1643-1675 + pds_for_point_line :2435-2516 (no guard in the reference)
        EZPZ_HEAVY_KIND {
        double px = XV(0), py = XV(1), p0x = XV(2), p0y = XV(3), p1x = XV(4), p1y = XV(5);
        double ex = -p0x + p1x, ey = p0y - p1y;
        double euclid = hypot(ex, ey);
        double denom = pow_1p5(ex * ex + ey * ey);
        double common = (p0x * p1y - p0y * p1x + px * (p0y - p1y) + py * (-p0x + p1x));
        w.template put<0>((p0y - p1y) / euclid);
        w.template put<1>((-p0x + p1x) / euclid);
        w.template put<2>(((-p0x + p1x) * common) / denom + (p1y - py) / euclid);
        w.template put<3>(((-p0y + p1y) * common) / denom + (-p1x + px) / euclid);
        w.template put<4>(((p0x - p1x) * common) / denom + (-p0y + py) / euclid);
        w.template put<5>(((p0y - p1y) * common) / denom + (p0x - px) / euclid);
        return false;
    }
    case EZPZ_VERTICAL_POINT_LINE_DISTANCE:  // :1676-1733
        EZPZ_HEAVY_KIND {
        double ax = XV(0), px = XV(2), py = XV(3), qx = XV(4), qy = XV(5);
        double dx = qx - px, dy = qy - py;
        if (fabs(dx) <= EPS || (dx * dx + dy * dy) <= EPS * EPS) {
            w.template zero<0, 6>();
            return true;
        }
        double pq = px - qx;
        double inv2 = 1.0 / (pq * pq);  // pow(px - qx, -2)
        double inv = 1.0 / pq;
        w.template put<0>((-py + qy) * inv);
        w.template put<1>(1.0);
        w.template put<2>((ax - qx) * (py - qy) * inv2);
        w.template put<3>((-ax + qx) * inv);
        w.template put<4>(-(ax - px) * (py - qy) * inv2);
        w.template put<5>((ax - px) * inv);
        return false;
    }
    case EZPZ_HORIZONTAL_POINT_LINE_DISTANCE:  // :1734-1787 (`<` here, `<=` in the residual)
        EZPZ_HEAVY_KIND {
        double ay = XV(1), px = XV(2), py = XV(3), qx = XV(4), qy = XV(5);
        double dx = qx - px, dy = qy - py;
        if (fabs(dy) < EPS || (dx * dx + dy * dy) < EPS * EPS) {
            w.template zero<0, 6>();
            return true;
        }
        double pq = py - qy;
        double inv2 = 1.0 / (pq * pq);
        double inv = 1.0 / pq;
        w.template put<0>(1.0);
        w.template put<1>((-px + qx) * inv);
        w.template put<2>((-ay + qy) * inv);
        w.template put<3>((ay - qy) * (px - qx) * inv2);
        w.template put<4>((ay - py) * inv);
        w.template put<5>(-(ay - py) * (px - qx) * inv2);
        return false;
    }
    case EZPZ_SYMMETRIC:  // :1788-1879 + pds_from_symmetric :2361-2433
        EZPZ_HEAVY_KIND {
        double px = XV(0), py = XV(1), qx = XV(2), qy = XV(3), ax = XV(4), ay = XV(5);
        double dx = px - qx, dy = py - qy;
        double dx2 = dx * dx, dy2 = dy * dy;
        double r = dx2 + dy2;
        double r2 = r * r;
        if (r2 < EPS) {
            w.template zero<0, 16>();
            return true;
        }
        double sx = ax - px, sy = ay - py;
        double dt = sx * dx + sy * dy;
        w.template put<0>((-4.0 * dx2 * dt + 2.0 * r2 + 2.0 * r * (sx * dx + sy * dy + dx * (ax - 2.0 * px + qx))) / r2);
        w.template put<1>(dx * (-4.0 * dy * dt + 2.0 * r * (ay - 2.0 * py + qy)) / r2);
        w.template put<2>((4.0 * dx2 * dt - (4.0 * sx * dx + 2.0 * sy * dy) * r) / r2);
        w.template put<3>(dx * (-2.0 * sy * r + 4.0 * dy * dt) / r2);
        w.template put<4>(1.0 * (dx2 - dy2) / r);
        w.template put<5>(2.0 * dx * dy / r);
        w.template put<6>(-1.0);
        w.template put<7>(0.0);
        w.template put<8>(dy * (-4.0 * dx * dt + 2.0 * r * (ax - 2.0 * px + qx)) / r2);
        w.template put<9>((-4.0 * dy2 * dt + 2.0 * r2 + 2.0 * r * (sx * dx + sy * dy + dy * (ay - 2.0 * py + qy))) / r2);
        w.template put<10>(dy * (-2.0 * sx * r + 4.0 * dx * dt) / r2);
        w.template put<11>((4.0 * dy2 * dt - (2.0 * sx * dx + 4.0 * sy * dy) * r) / r2);
        w.template put<12>(2.0 * dx * dy / r);
        w.template put<13>(1.0 * (-dx2 + dy2) / r);
        w.template put<14>(0.0);
        w.template put<15>(-1.0);
        return false;
    }
    case EZPZ_POINT_ARC_COINCIDENT:  // :1880-2063
        EZPZ_HEAVY_KIND {
        V2 cc = mk(XV(0), XV(1));
        V2 s = mk(XV(2), XV(3)) - cc, e = mk(XV(4), XV(5)) - cc, p = mk(XV(6), XV(7)) - cc;
        double r = mag(s), r_e = mag(e), r_p = mag(p);
        if (r < EPS || r_e < EPS || r_p < EPS) {
            w.template zero<0, 16>();
            return true;
        }
        V2 u_s = s * (1.0 / r), u_e = e * (1.0 / r_e);
        V2 e_proj = e * (r / r_e);
        // j_x[i][k] = d f_k / d x_i
        double s00, s01, s10, s11, e00, e01, e10, e11, p00, p01, p10, p11;
        int part = classify_pac(s, e_proj, p);
        if (part == 0) {
            V2 u_p = p * (1.0 / r_p);
            double q = r / r_p;
            s00 = u_p.x * u_s.x;
            s01 = u_p.y * u_s.x;
            s10 = u_p.x * u_s.y;
            s11 = u_p.y * u_s.y;
            e00 = e01 = e10 = e11 = 0.0;
            p00 = (q - 1.0) - q * u_p.x * u_p.x;
            p01 = -q * u_p.y * u_p.x;
            p10 = -q * u_p.x * u_p.y;
            p11 = (q - 1.0) - q * u_p.y * u_p.y;
        } else if (part == 2) {
            double q = r / r_e;
            s00 = u_e.x * u_s.x;
            s01 = u_e.y * u_s.x;
            s10 = u_e.x * u_s.y;
            s11 = u_e.y * u_s.y;
            e00 = q * (1.0 - u_e.x * u_e.x);
            e01 = -q * u_e.y * u_e.x;
            e10 = -q * u_e.x * u_e.y;
            e11 = q * (1.0 - u_e.y * u_e.y);
            p00 = -1.0;
            p01 = 0.0;
            p10 = 0.0;
            p11 = -1.0;
        } else {
            s00 = 1.0;
            s01 = 0.0;
            s10 = 0.0;
            s11 = 1.0;
            e00 = e01 = e10 = e11 = 0.0;
            p00 = -1.0;
            p01 = 0.0;
            p10 = 0.0;
            p11 = -1.0;
        }
        double o00 = -(s00 + e00 + p00), o01 = -(s01 + e01 + p01), o10 = -(s10 + e10 + p10), o11 = -(s11 + e11 + p11);
        w.template put<0>(o00);
        w.template put<1>(o10);
        w.template put<2>(s00);
        w.template put<3>(s10);
        w.template put<4>(e00);
        w.template put<5>(e10);
        w.template put<6>(p00);
        w.template put<7>(p10);
        w.template put<8>(o01);
        w.template put<9>(o11);
        w.template put<10>(s01);
        w.template put<11>(s11);
        w.template put<12>(e01);
        w.template put<13>(e11);
        w.template put<14>(p01);
        w.template put<15>(p11);
        return false;
    }
    case EZPZ_ARC_LENGTH:  // :2064-2163
        EZPZ_HEAVY_KIND {
        double cx = XV(0), cy = XV(1);
        double ux = XV(2) - cx, uy = XV(3) - cy;
        double r2 = ux * ux + uy * uy;
        if (r2 <= EPS * EPS) {
            w.template zero<0, 12>();
            return true;
        }
        double d = c.param;
        double r = sqrt(r2);
        double alpha = d / r;
        Rot sc = sincos_of(alpha);
        double sa = sc.s, ca = sc.c;
        double rux = ca * ux - sa * uy, ruy = sa * ux + ca * uy;
        double k = d / (r2 * r);
        w.template put<0>(-ca - ruy * ux * k);
        w.template put<1>(sa - ruy * uy * k);
        w.template put<2>(1.0);
        w.template put<3>(0.0);
        w.template put<4>(-1.0 + ca + ruy * ux * k);
        w.template put<5>(-sa + ruy * uy * k);
        w.template put<6>(-sa + rux * ux * k);
        w.template put<7>(-ca + rux * uy * k);
        w.template put<8>(0.0);
        w.template put<9>(1.0);
        w.template put<10>(sa - rux * ux * k);
        w.template put<11>(-1.0 + ca - rux * uy * k);
        return false;
    }
    case EZPZ_ARC_ANGLE:  // :2164-2175
        EZPZ_HEAVY_KIND {
        double cx = XV(0), cy = XV(1);
        return lines_at_angle_jac(mk(XV(2) - cx, XV(3) - cy), mk(XV(4) - cx, XV(5) - cy), c.tag, c.param, w);
    }
    case EZPZ_POINTS_AT_ANGLE:  // :2176-2291
        EZPZ_HEAVY_KIND {
        V2 p0 = mk(XV(0), XV(1));
        V2 u = mk(XV(2), XV(3)) - p0, v = mk(XV(4), XV(5)) - p0;
        double len_u = mag(u), len_v = mag(v);
        if (len_u <= EPS || len_v <= EPS) {
            w.template zero<0, 12>();
            return true;
        }
        V2 u_hat = u * (1.0 / len_u), v_hat = v * (1.0 / len_v);
        Rot rot = rot_for(c.tag, c.param);
        double s = (len_u + len_v) * 0.5;
        V2 rot_e1 = rot_apply(rot, mk(1.0, 0.0)), rot_e2 = rot_apply(rot, mk(0.0, 1.0));
        double inv_s = 1.0 / s;
        V2 rot_u = rot_apply(rot, u);
        V2 res = (v * len_u - rot_u * len_v) * inv_s;
        V2 half = res * 0.5;
        V2 du0 = ((v - half) * u_hat.x - rot_e1 * len_v) * inv_s;
        V2 du1 = ((v - half) * u_hat.y - rot_e2 * len_v) * inv_s;
        V2 dv0 = (mk(len_u, 0.0) - (rot_u + half) * v_hat.x) * inv_s;
        V2 dv1 = (mk(0.0, len_u) - (rot_u + half) * v_hat.y) * inv_s;
        w.template put<0>(-(du0.x + dv0.x));
        w.template put<1>(-(du1.x + dv1.x));
        w.template put<2>(du0.x);
        w.template put<3>(du1.x);
        w.template put<4>(dv0.x);
        w.template put<5>(dv1.x);
        w.template put<6>(-(du0.y + dv0.y));
        w.template put<7>(-(du1.y + dv1.y));
        w.template put<8>(du0.y);
        w.template put<9>(du1.y);
        w.template put<10>(dv0.y);
        w.template put<11>(dv1.y);
        return false;
    }
    default:
        return false;
    }
}

#undef XV
#undef EZPZ_HEAVY_KIND

template <class XP>
__device__ __forceinline__ bool con_residual(const DevCon& c, XP xs, double& r0, double& r1) {
    return con_residual<false>(c, xs, r0, r1);
}
template <class XP, class JP>
__device__ __forceinline__ bool con_jacobian(const DevCon& c, XP xs, const JacWriter<JP>& w) {
    return con_jacobian<false>(c, xs, w);
}

}  // namespace dev
}  // namespace ezpz
