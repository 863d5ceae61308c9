/*
 * ezpz_oracle.c -- TEST INFRASTRUCTURE ONLY (see ezpz_oracle.h).
 *
 * Part 1: per-constraint math.  Restates
 *   ezpz/src/vector.rs:1-143            (V, Rotation2)
 *   ezpz/src/constraints.rs:146-193     (set_from_initial_values)
 *   ezpz/src/constraints.rs:378-491     (nonzeroes)
 *   ezpz/src/constraints.rs:499-950     (residual)
 *   ezpz/src/constraints.rs:954-993     (residual_dim)
 *   ezpz/src/constraints.rs:1000-2293   (jacobian_rows)
 *   ezpz/src/constraints.rs:2361-2647   (helpers)
 * Part 2 (ezpz_oracle_solve.c): Model::new, LM loop, solve(), solve_inner().
 *
 * Compile with -ffp-contract=off: the Rust reference never fuses a*b+c.
 */
#include "ezpz_oracle.h"

#include <math.h>
#include <string.h>

#define EPSILON 1e-4 /* lib.rs:43 */
#define ORC_PI 3.14159265358979323846264338327950288

/* ---- vector.rs ---------------------------------------------------------------------------- */
typedef struct {
    double x, y;
} V;

static inline V v_new(double x, double y) {
    V v = {x, y};
    return v;
}
static inline V v_sub(V a, V b) { return v_new(a.x - b.x, a.y - b.y); }
static inline V v_add(V a, V b) { return v_new(a.x + b.x, a.y + b.y); }
static inline V v_scale(V a, double s) { return v_new(a.x * s, a.y * s); }
/* vector.rs:15-17 */
static inline double v_magnitude(V a) { return hypot(a.x, a.y); }
/* vector.rs:20-22: pow(x,2)+pow(y,2) */
static inline double v_magnitude_squared(V a) { return pow(a.x, 2.0) + pow(a.y, 2.0); }
static inline double v_dot(V a, V b) { return a.x * b.x + a.y * b.y; }
static inline double v_euclidean_distance(V a, V b) { return v_magnitude(v_sub(a, b)); }
static inline double v_cross_2d(V a, V b) { return a.x * b.y - a.y * b.x; }
static inline V v_perp_ccw(V a) { return v_new(-a.y, a.x); }
static inline V v_perp_cw(V a) { return v_new(a.y, -a.x); }
/* vector.rs:58-69 (division by b.b is unguarded in the reference) */
static inline V v_project(V a, V b) { return v_scale(b, v_dot(a, b) / v_dot(b, b)); }
static inline V v_reject(V a, V b) { return v_sub(a, v_project(a, b)); }
static inline V v_reflect(V a, V b) { return v_sub(a, v_scale(v_reject(a, b), 2.0)); }
/* vector.rs:72-74 */
static inline double v_signed_angle(V a, V b) { return atan2(v_cross_2d(a, b), v_dot(a, b)); }

/* Rotation2, vector.rs:110-143: col0 = (cos, sin) */
typedef struct {
    V col0;
} Rot2;
static inline Rot2 rot_from_sincos(double s, double c) {
    Rot2 r;
    r.col0 = v_new(c, s);
    return r;
}
static inline Rot2 rot_from_angle_radians(double angle) {
    /* libm::sincos */
    return rot_from_sincos(sin(angle), cos(angle));
}
static inline V rot_apply(Rot2 r, V v) {
    return v_new((r.col0.x * v.x) - (r.col0.y * v.y), (r.col0.y * v.x) + (r.col0.x * v.y));
}
static inline Rot2 rot_inverse(Rot2 r) {
    Rot2 o;
    o.col0 = v_new(r.col0.x, -r.col0.y);
    return o;
}

/* Rust f64 helpers */
static inline double rs_signum(double x) { return isnan(x) ? x : copysign(1.0, x); }
static inline double rs_rem_euclid(double x, double rhs) {
    double r = fmod(x, rhs);
    return (r < 0.0) ? r + fabs(rhs) : r;
}
static inline double rs_to_radians(double deg) { return deg * (ORC_PI / 180.0); }
static inline double rs_to_degrees(double rad) { return rad * (180.0 / ORC_PI); }
static inline double rs_recip(double x) { return 1.0 / x; }

/* datatypes.rs:66-72 Angle::to_radians */
static inline double angle_to_radians(uint8_t tag, double val) {
    return (tag == ORC_ANGLE_OTHER_DEG) ? rs_to_radians(val) : val;
}
double orc_angle_to_degrees(uint8_t tag, double val) {
    return (tag == ORC_ANGLE_OTHER_DEG) ? val : rs_to_degrees(val);
}

/* constraints.rs:2641-2647 */
static Rot2 rotation_for_angle_kind(uint8_t tag, double val) {
    switch (tag) {
    case ORC_ANGLE_PARALLEL:
        return rot_from_sincos(0.0, 1.0);
    case ORC_ANGLE_PERPENDICULAR:
        return rot_from_sincos(1.0, 0.0);
    default:
        return rot_from_angle_radians(angle_to_radians(tag, val));
    }
}

/* constraints.rs:2593-2606 */
enum { PAC_INTERIOR = 0, PAC_START = 1, PAC_END = 2 };
static int classify_point_arc_coincident(V s, V e, V p) {
    const double two_pi = 2.0 * ORC_PI;
    double a_sp = rs_rem_euclid(v_signed_angle(s, p), two_pi);
    double a_se = rs_rem_euclid(v_signed_angle(s, e), two_pi);
    if (a_sp < a_se) {
        return PAC_INTERIOR;
    } else if (v_magnitude_squared(v_sub(e, p)) < v_magnitude_squared(v_sub(s, p))) {
        return PAC_END;
    } else {
        return PAC_START;
    }
}

#define X(i) (x[c->ids[(i)]])

/* ---- residual_dim, constraints.rs:954-993 -------------------------------------------------- */
int orc_residual_dim(const OrcConstraint* c) {
    switch (c->kind) {
    case ORC_POINTS_COINCIDENT:
    case ORC_ARC_RADIUS:
    case ORC_MIDPOINT:
    case ORC_SYMMETRIC:
    case ORC_POINT_ARC_COINCIDENT:
    case ORC_ARC_LENGTH:
    case ORC_POINTS_AT_ANGLE:
        return 2;
    default:
        return 1;
    }
}

/* ---- nonzeroes, constraints.rs:378-491 ------------------------------------------------------ */
static void push_ids(uint32_t* row, int* n, const OrcConstraint* c, const int* which, int k) {
    for (int i = 0; i < k; ++i) row[(*n)++] = c->ids[which[i]];
}
int orc_nonzeroes(const OrcConstraint* c, uint32_t row0[8], int* n0, uint32_t row1[8], int* n1) {
    *n0 = 0;
    *n1 = 0;
    static const int seq[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    switch (c->kind) {
    case ORC_LINE_TANGENT_TO_CIRCLE: /* :380-383 line.all, circle.all */
        push_ids(row0, n0, c, seq, 7);
        break;
    case ORC_CIRCLE_TANGENT_TO_CIRCLE: /* :384-387 */
        push_ids(row0, n0, c, seq, 6);
        break;
    case ORC_DISTANCE: /* :388-391 */
        push_ids(row0, n0, c, seq, 4);
        break;
    case ORC_DISTANCE_VAR: /* :392-396 */
        push_ids(row0, n0, c, seq, 5);
        break;
    case ORC_VERTICAL_DISTANCE: { /* :397-399 p0.y, p1.y */
        static const int w[2] = {1, 3};
        push_ids(row0, n0, c, w, 2);
        break;
    }
    case ORC_HORIZONTAL_DISTANCE: { /* :400-402 p0.x, p1.x */
        static const int w[2] = {0, 2};
        push_ids(row0, n0, c, w, 2);
        break;
    }
    case ORC_VERTICAL: { /* :403 */
        static const int w[2] = {0, 2};
        push_ids(row0, n0, c, w, 2);
        break;
    }
    case ORC_HORIZONTAL: { /* :404 */
        static const int w[2] = {1, 3};
        push_ids(row0, n0, c, w, 2);
        break;
    }
    case ORC_LINES_AT_ANGLE: /* :405-408 */
        push_ids(row0, n0, c, seq, 8);
        break;
    case ORC_FIXED: /* :409 */
        push_ids(row0, n0, c, seq, 1);
        break;
    case ORC_SCALAR_EQUAL: /* :410 */
        push_ids(row0, n0, c, seq, 2);
        break;
    case ORC_POINTS_COINCIDENT: { /* :411-416 */
        static const int w0[2] = {0, 2};
        static const int w1[2] = {1, 3};
        push_ids(row0, n0, c, w0, 2);
        push_ids(row1, n1, c, w1, 2);
        break;
    }
    case ORC_CIRCLE_RADIUS: { /* :417 */
        static const int w[1] = {2};
        push_ids(row0, n0, c, w, 1);
        break;
    }
    case ORC_LINES_EQUAL_LENGTH: /* :418-421 */
        push_ids(row0, n0, c, seq, 8);
        break;
    case ORC_ARC_RADIUS: { /* :422-431 Distance(center,start) -> row0, Distance(center,end) -> row1 */
        static const int w0[4] = {0, 1, 2, 3};
        static const int w1[4] = {0, 1, 4, 5};
        push_ids(row0, n0, c, w0, 4);
        push_ids(row1, n1, c, w1, 4);
        break;
    }
    case ORC_ARC: { /* :432-434 arc.all_variables = start, end, center (inputs.rs:180-191) */
        static const int w[6] = {2, 3, 4, 5, 0, 1};
        push_ids(row0, n0, c, w, 6);
        break;
    }
    case ORC_MIDPOINT: { /* :435-438 */
        static const int w0[3] = {0, 2, 4};
        static const int w1[3] = {1, 3, 5};
        push_ids(row0, n0, c, w0, 3);
        push_ids(row1, n1, c, w1, 3);
        break;
    }
    case ORC_POINT_LINE_DISTANCE: /* :439-442 point.all, line.all */
        push_ids(row0, n0, c, seq, 6);
        break;
    case ORC_VERTICAL_POINT_LINE_DISTANCE:     /* :443-446 line.all, point.all */
    case ORC_HORIZONTAL_POINT_LINE_DISTANCE: { /* :447-450 */
        static const int w[6] = {2, 3, 4, 5, 0, 1};
        push_ids(row0, n0, c, w, 6);
        break;
    }
    case ORC_SYMMETRIC: /* :451-459 */
        push_ids(row0, n0, c, seq, 8);
        push_ids(row1, n1, c, seq, 8);
        break;
    case ORC_POINT_ARC_COINCIDENT: { /* :460-465 arc.all (s,e,c), point.all */
        static const int w[8] = {2, 3, 4, 5, 0, 1, 6, 7};
        push_ids(row0, n0, c, w, 8);
        push_ids(row1, n1, c, w, 8);
        break;
    }
    case ORC_ARC_LENGTH: { /* :466-469 */
        static const int w[6] = {2, 3, 4, 5, 0, 1};
        push_ids(row0, n0, c, w, 6);
        push_ids(row1, n1, c, w, 6);
        break;
    }
    case ORC_ARC_ANGLE: { /* :470-481 LinesAtAngle(c->s, c->e) : c,s,c,e (duplicate c) */
        static const int w[8] = {0, 1, 2, 3, 0, 1, 4, 5};
        push_ids(row0, n0, c, w, 8);
        break;
    }
    case ORC_POINTS_AT_ANGLE: /* :482-489 */
        push_ids(row0, n0, c, seq, 6);
        push_ids(row1, n1, c, seq, 6);
        break;
    default:
        break;
    }
    return orc_residual_dim(c);
}

/* ---- set_from_initial_values, constraints.rs:146-193 ---------------------------------------- */
void orc_set_from_initial_values(OrcConstraint* c, const double* x) {
    if (c->kind == ORC_LINE_TANGENT_TO_CIRCLE && c->tag == ORC_SIDE_UNDEFINED) {
        V p0 = v_new(X(0), X(1));
        V p1 = v_new(X(2), X(3));
        V cc = v_new(X(4), X(5));
        c->tag = (v_cross_2d(v_sub(p1, p0), v_sub(cc, p0)) >= 0.0) ? ORC_LINE_LEFT : ORC_LINE_RIGHT;
    } else if (c->kind == ORC_CIRCLE_TANGENT_TO_CIRCLE && c->tag == ORC_SIDE_UNDEFINED) {
        V a_c = v_new(X(0), X(1));
        double a_r = X(2);
        V b_c = v_new(X(3), X(4));
        double b_r = X(5);
        double dist = v_magnitude(v_sub(a_c, b_c));
        double r_int = fabs(fabs(a_r - b_r) - dist);
        double r_ext = fabs(a_r + b_r - dist);
        c->tag = (r_int < r_ext) ? ORC_CIRCLE_INTERIOR : ORC_CIRCLE_EXTERIOR;
    }
}

/* ---- LinesAtAngle residual shared by ArcAngle, constraints.rs:617-640 ----------------------- */
static void lines_at_angle_residual(double x0, double y0, double x1, double y1, double x2, double y2, double x3,
                                    double y3, uint8_t tag, double val, double* r0, int* degenerate) {
    V u = v_new(x1 - x0, y1 - y0);
    V v = v_new(x3 - x2, y3 - y2);
    double len_u = v_magnitude(u);
    double len_v = v_magnitude(v);
    if (len_u <= EPSILON || len_v <= EPSILON) {
        *degenerate = 1;
        return;
    }
    Rot2 rot = rotation_for_angle_kind(tag, val);
    *r0 = v_cross_2d(u, rot_apply(rot_inverse(rot), v)) / ((len_u + len_v) * 0.5);
}

/* Distance residual, constraints.rs:565-574 */
static double distance_residual(double p0x, double p0y, double p1x, double p1y, double expected) {
    double actual = v_euclidean_distance(v_new(p0x, p0y), v_new(p1x, p1y));
    return actual - expected;
}

/* ---- residual, constraints.rs:499-950 -------------------------------------------------------- */
void orc_residual(const OrcConstraint* c, const double* x, double r[3], int* degenerate) {
    double* residual0 = &r[0];
    double* residual1 = &r[1];
    switch (c->kind) {
    case ORC_LINE_TANGENT_TO_CIRCLE: { /* :509-544 */
        V p0 = v_new(X(0), X(1));
        V p1 = v_new(X(2), X(3));
        V cc = v_new(X(4), X(5));
        double radius = fabs(X(6));
        V u = v_sub(p1, p0);
        double mag_u = v_magnitude(u);
        if (mag_u <= EPSILON) {
            *residual0 = 0.0;
            *degenerate = 1;
            return;
        }
        V v = v_sub(cc, p0);
        double cross_uv = v_cross_2d(u, v);
        double side_sign = (c->tag == ORC_LINE_RIGHT) ? -1.0 : 1.0;
        double cen_dist = side_sign * cross_uv / mag_u;
        *residual0 = cen_dist - radius;
        break;
    }
    case ORC_CIRCLE_TANGENT_TO_CIRCLE: { /* :545-564 */
        V a_c = v_new(X(0), X(1));
        double a_r = fabs(X(2));
        V b_c = v_new(X(3), X(4));
        double b_r = fabs(X(5));
        double dist = v_magnitude(v_sub(a_c, b_c));
        *residual0 = (c->tag == ORC_CIRCLE_INTERIOR) ? fabs(a_r - b_r) - dist : a_r + b_r - dist;
        break;
    }
    case ORC_DISTANCE: /* :565-574 */
        *residual0 = distance_residual(X(0), X(1), X(2), X(3), c->param);
        break;
    case ORC_DISTANCE_VAR: { /* :575-583 */
        double px = X(0), py = X(1), qx = X(2), qy = X(3), d = X(4);
        *residual0 = -d + sqrt(pow(px - qx, 2.0) + pow(py - qy, 2.0));
        break;
    }
    case ORC_VERTICAL_DISTANCE: /* :584-591 */
        *residual0 = (X(1) - X(3)) - c->param;
        break;
    case ORC_HORIZONTAL_DISTANCE: /* :592-596 */
        *residual0 = (X(0) - X(2)) - c->param;
        break;
    case ORC_VERTICAL: /* :597-601 */
        *residual0 = X(0) - X(2);
        break;
    case ORC_HORIZONTAL: /* :602-606 */
        *residual0 = X(1) - X(3);
        break;
    case ORC_FIXED: /* :607-610 */
        *residual0 = X(0) - c->param;
        break;
    case ORC_SCALAR_EQUAL: /* :611-616 */
        *residual0 = X(0) - X(1);
        break;
    case ORC_LINES_AT_ANGLE: /* :617-640 */
        lines_at_angle_residual(X(0), X(1), X(2), X(3), X(4), X(5), X(6), X(7), c->tag, c->param, residual0,
                                degenerate);
        break;
    case ORC_POINTS_COINCIDENT: /* :641-648 */
        *residual0 = X(0) - X(2);
        *residual1 = X(1) - X(3);
        break;
    case ORC_CIRCLE_RADIUS: /* :649-652 */
        *residual0 = X(2) - c->param;
        break;
    case ORC_LINES_EQUAL_LENGTH: { /* :653-658 */
        double len0 = v_euclidean_distance(v_new(X(0), X(1)), v_new(X(2), X(3)));
        double len1 = v_euclidean_distance(v_new(X(4), X(5)), v_new(X(6), X(7)));
        *residual0 = len0 - len1;
        break;
    }
    case ORC_ARC_RADIUS: /* :659-682 Distance(center,start), Distance(center,end) */
        *residual0 = distance_residual(X(0), X(1), X(2), X(3), c->param);
        *residual1 = distance_residual(X(0), X(1), X(4), X(5), c->param);
        break;
    case ORC_ARC: { /* :683-696 */
        double cx = X(0), cy = X(1), sx = X(2), sy = X(3), ex = X(4), ey = X(5);
        double dist0 = hypot(sx - cx, sy - cy);
        double dist1 = hypot(ex - cx, ey - cy);
        *residual0 = dist0 - dist1;
        break;
    }
    case ORC_MIDPOINT: { /* :697-711 */
        double px = X(0), py = X(1), qx = X(2), qy = X(3), ax = X(4), ay = X(5);
        *residual0 = ax - px / 2.0 - qx / 2.0;
        *residual1 = ay - py / 2.0 - qy / 2.0;
        break;
    }
    case ORC_POINT_LINE_DISTANCE: { /* :712-740, :2625-2639 */
        double px = X(0), py = X(1);
        double lpx = X(2), lpy = X(3), lqx = X(4), lqy = X(5);
        double a = lpy - lqy;
        double b = lqx - lpx;
        double cc = (lpx * lqy) - (lqx * lpy);
        double denominator = hypot(a, b);
        if (denominator < EPSILON) {
            *residual0 = 0.0;
            *degenerate = 1;
            return;
        }
        double actual_distance = (a * px + b * py + cc) / denominator;
        *residual0 = actual_distance - c->param;
        break;
    }
    case ORC_VERTICAL_POINT_LINE_DISTANCE: { /* :741-762 */
        double ax = X(0), ay = X(1), px = X(2), py = X(3), qx = X(4), qy = X(5);
        double dx = qx - px;
        double dy = qy - py;
        if (fabs(dx) <= EPSILON || (dx * dx + dy * dy) <= EPSILON * EPSILON) {
            *degenerate = 1;
            return;
        }
        *residual0 = ay - py - dy * rs_recip(dx) * (ax - px) - c->param;
        break;
    }
    case ORC_HORIZONTAL_POINT_LINE_DISTANCE: { /* :763-785 */
        double ax = X(0), ay = X(1), px = X(2), py = X(3), qx = X(4), qy = X(5);
        double dx = qx - px;
        double dy = qy - py;
        if (fabs(dy) <= EPSILON || (dx * dx + dy * dy) <= EPSILON * EPSILON) {
            *degenerate = 1;
            return;
        }
        *residual0 = ax - px - dx * rs_recip(dy) * (ay - py) - c->param;
        break;
    }
    case ORC_SYMMETRIC: { /* :786-808 */
        V p = v_new(X(0), X(1));
        V q = v_new(X(2), X(3));
        V a = v_new(X(4), X(5));
        V b = v_new(X(6), X(7));
        V res = v_add(v_sub(v_reflect(v_sub(a, p), v_sub(q, p)), b), p);
        *residual0 = res.x;
        *residual1 = res.y;
        break;
    }
    case ORC_POINT_ARC_COINCIDENT: { /* :809-858 */
        V cc = v_new(X(0), X(1));
        V s = v_sub(v_new(X(2), X(3)), cc);
        V e = v_sub(v_new(X(4), X(5)), cc);
        V p = v_sub(v_new(X(6), X(7)), cc);
        double rr = v_magnitude(s);
        double r_e = v_magnitude(e);
        double r_p = v_magnitude(p);
        if (rr < EPSILON || r_e < EPSILON || r_p < EPSILON) {
            *residual0 = 0.0;
            *residual1 = 0.0;
            *degenerate = 1;
            return;
        }
        V e_proj = v_scale(e, rr / r_e);
        V f;
        switch (classify_point_arc_coincident(s, e_proj, p)) {
        case PAC_INTERIOR:
            f = v_scale(p, rr / r_p - 1.0);
            break;
        case PAC_END:
            f = v_sub(e_proj, p);
            break;
        default:
            f = v_sub(s, p);
            break;
        }
        *residual0 = f.x;
        *residual1 = f.y;
        break;
    }
    case ORC_ARC_LENGTH: { /* :859-896 */
        double cx = X(0), cy = X(1), ax = X(2), ay = X(3), bx = X(4), by = X(5);
        double d = c->param;
        double ux = ax - cx;
        double uy = ay - cy;
        double r2 = ux * ux + uy * uy;
        if (r2 <= EPSILON * EPSILON) {
            *residual0 = 0.0;
            *residual1 = 0.0;
            *degenerate = 1;
            return;
        }
        double alpha = d / sqrt(r2);
        double sa = sin(alpha);
        double ca = cos(alpha);
        double rux = ca * ux - sa * uy;
        double ruy = sa * ux + ca * uy;
        *residual0 = (bx - cx) - rux;
        *residual1 = (by - cy) - ruy;
        break;
    }
    case ORC_ARC_ANGLE: /* :897-915 LinesAtAngle(center->start, center->end, Other(angle)) */
        lines_at_angle_residual(X(0), X(1), X(2), X(3), X(0), X(1), X(4), X(5), c->tag, c->param, residual0,
                                degenerate);
        break;
    case ORC_POINTS_AT_ANGLE: { /* :916-948 */
        V p0v = v_new(X(0), X(1));
        V p1v = v_new(X(2), X(3));
        V p2v = v_new(X(4), X(5));
        V u = v_sub(p1v, p0v);
        V v = v_sub(p2v, p0v);
        double len_u = v_magnitude(u);
        double len_v = v_magnitude(v);
        if (len_u <= EPSILON || len_v <= EPSILON) {
            *degenerate = 1;
            return;
        }
        Rot2 rot = rotation_for_angle_kind(c->tag, c->param);
        double s = (len_u + len_v) * 0.5;
        V res = v_scale(v_sub(v_scale(v, len_u), v_scale(rot_apply(rot, u), len_v)), 1.0 / s);
        *residual0 = res.x;
        *residual1 = res.y;
        break;
    }
    default:
        break;
    }
}

/* ---- Jacobian helpers ------------------------------------------------------------------------- */
typedef struct {
    uint32_t* ids;
    double* pd;
    int* n;
} Row;
static inline void row_push(Row r, uint32_t id, double pd) {
    r.ids[*r.n] = id;
    r.pd[*r.n] = pd;
    (*r.n)++;
}
#define ID(i) (c->ids[(i)])

/* Distance jacobian, constraints.rs:1160-1204.  idx = positions of p0x,p0y,p1x,p1y in c->ids */
static void distance_jacobian(const OrcConstraint* c, const double* x, int i0x, int i0y, int i1x, int i1y, Row row,
                              int* degenerate) {
    double x0 = X(i0x), y0 = X(i0y), x1 = X(i1x), y1 = X(i1y);
    double dist = v_euclidean_distance(v_new(x0, y0), v_new(x1, y1));
    if (dist < EPSILON) {
        *degenerate = 1;
        return;
    }
    row_push(row, ID(i0x), (x0 - x1) / dist);
    row_push(row, ID(i0y), (y0 - y1) / dist);
    row_push(row, ID(i1x), (-x0 + x1) / dist);
    row_push(row, ID(i1y), (-y0 + y1) / dist);
}

/* LinesAtAngle jacobian, constraints.rs:1358-1418; idx[8] = positions of x0,y0,..,y3 in c->ids */
static void lines_at_angle_jacobian(const OrcConstraint* c, const double* x, const int idx[8], Row row,
                                    int* degenerate) {
    double x0 = X(idx[0]), y0 = X(idx[1]), x1 = X(idx[2]), y1 = X(idx[3]);
    double x2 = X(idx[4]), y2 = X(idx[5]), x3 = X(idx[6]), y3 = X(idx[7]);
    V u = v_new(x1 - x0, y1 - y0);
    V v = v_new(x3 - x2, y3 - y2);
    double len_u = v_magnitude(u);
    double len_v = v_magnitude(v);
    if ((len_u <= EPSILON) || (len_v <= EPSILON)) {
        *degenerate = 1;
        return;
    }
    V u_hat = v_scale(u, 1.0 / len_u);
    V v_hat = v_scale(v, 1.0 / len_v);
    Rot2 rot = rotation_for_angle_kind(c->tag, c->param);
    double s = (len_u + len_v) * 0.5;
    double a = v_cross_2d(u, rot_apply(rot_inverse(rot), v));
    double inv_s = 1.0 / s;
    double t = a * inv_s * 0.5;
    V df_du = v_scale(v_sub(v_perp_cw(rot_apply(rot_inverse(rot), v)), v_scale(u_hat, t)), inv_s);
    V df_dv = v_scale(v_sub(v_perp_ccw(rot_apply(rot, u)), v_scale(v_hat, t)), inv_s);
    /* PartialDerivatives4Points::jvars :2532-2568 */
    row_push(row, ID(idx[0]), -df_du.x);
    row_push(row, ID(idx[1]), -df_du.y);
    row_push(row, ID(idx[2]), df_du.x);
    row_push(row, ID(idx[3]), df_du.y);
    row_push(row, ID(idx[4]), -df_dv.x);
    row_push(row, ID(idx[5]), -df_dv.y);
    row_push(row, ID(idx[6]), df_dv.x);
    row_push(row, ID(idx[7]), df_dv.y);
}

/* ---- jacobian_rows, constraints.rs:1000-2293 ------------------------------------------------- */
void orc_jacobian_rows(const OrcConstraint* c, const double* x, uint32_t ids0[8], double pd0[8], int* n0,
                       uint32_t ids1[8], double pd1[8], int* n1, int* degenerate) {
    *n0 = 0;
    *n1 = 0;
    Row row0 = {ids0, pd0, n0};
    Row row1 = {ids1, pd1, n1};
    switch (c->kind) {
    case ORC_LINE_TANGENT_TO_CIRCLE: { /* :1010-1090 */
        V p0 = v_new(X(0), X(1));
        V p1 = v_new(X(2), X(3));
        V cc = v_new(X(4), X(5));
        V u = v_sub(p1, p0);
        double mag_u = v_magnitude(u);
        if (mag_u <= EPSILON) {
            *degenerate = 1;
            return;
        }
        V v = v_sub(cc, p0);
        double cross_uv = v_cross_2d(u, v);
        double mag_u_cubed = mag_u * mag_u * mag_u;
        double side_sign = (c->tag == ORC_LINE_RIGHT) ? -1.0 : 1.0;
        double dr_du_x = side_sign * (-(u.x * cross_uv) / mag_u_cubed + v.y / mag_u);
        double dr_du_y = side_sign * (-(u.y * cross_uv) / mag_u_cubed - v.x / mag_u);
        double dr_dv_x = side_sign * (-u.y / mag_u);
        double dr_dv_y = side_sign * (u.x / mag_u);
        double radius = X(6);
        double dr_dr = -rs_signum(radius);
        row_push(row0, ID(0), -(dr_du_x + dr_dv_x));
        row_push(row0, ID(1), -(dr_du_y + dr_dv_y));
        row_push(row0, ID(2), dr_du_x);
        row_push(row0, ID(3), dr_du_y);
        row_push(row0, ID(4), dr_dv_x);
        row_push(row0, ID(5), dr_dv_y);
        row_push(row0, ID(6), dr_dr);
        break;
    }
    case ORC_CIRCLE_TANGENT_TO_CIRCLE: { /* :1091-1159 */
        V a_c = v_new(X(0), X(1));
        double a_r = X(2);
        V b_c = v_new(X(3), X(4));
        double b_r = X(5);
        V d = v_sub(b_c, a_c);
        double mag_d = v_magnitude(d);
        if (mag_d <= EPSILON) {
            *degenerate = 1;
            return;
        }
        V u_d = v_scale(d, rs_recip(mag_d));
        double a_sign = rs_signum(a_r);
        double b_sign = rs_signum(b_r);
        double dr_dar, dr_dbr;
        if (c->tag == ORC_CIRCLE_INTERIOR) {
            double inner = rs_signum(fabs(a_r) - fabs(b_r));
            dr_dar = inner * a_sign;
            dr_dbr = -inner * b_sign;
        } else {
            dr_dar = a_sign;
            dr_dbr = b_sign;
        }
        row_push(row0, ID(0), u_d.x);
        row_push(row0, ID(1), u_d.y);
        row_push(row0, ID(2), dr_dar);
        row_push(row0, ID(3), -u_d.x);
        row_push(row0, ID(4), -u_d.y);
        row_push(row0, ID(5), dr_dbr);
        break;
    }
    case ORC_DISTANCE: /* :1160-1204 */
        distance_jacobian(c, x, 0, 1, 2, 3, row0, degenerate);
        break;
    case ORC_DISTANCE_VAR: { /* :1205-1251 */
        double px = X(0), py = X(1), qx = X(2), qy = X(3);
        double dist = v_euclidean_distance(v_new(px, py), v_new(qx, qy));
        if (dist < EPSILON) {
            *degenerate = 1;
            return;
        }
        row_push(row0, ID(0), (px - qx) * rs_recip(dist));
        row_push(row0, ID(1), (py - qy) * rs_recip(dist));
        row_push(row0, ID(2), -(px - qx) * rs_recip(dist));
        row_push(row0, ID(3), -(py - qy) * rs_recip(dist));
        row_push(row0, ID(4), -1.0);
        break;
    }
    case ORC_VERTICAL_DISTANCE: /* :1252-1269 */
        row_push(row0, ID(1), 1.0);
        row_push(row0, ID(3), -1.0);
        break;
    case ORC_HORIZONTAL_DISTANCE: /* :1270-1287 */
        row_push(row0, ID(0), 1.0);
        row_push(row0, ID(2), -1.0);
        break;
    case ORC_VERTICAL: /* :1288-1311 */
        row_push(row0, ID(0), 1.0);
        row_push(row0, ID(2), -1.0);
        break;
    case ORC_HORIZONTAL: /* :1312-1335 */
        row_push(row0, ID(1), 1.0);
        row_push(row0, ID(3), -1.0);
        break;
    case ORC_FIXED: /* :1336-1344 */
        row_push(row0, ID(0), 1.0);
        break;
    case ORC_SCALAR_EQUAL: /* :1345-1357 */
        row_push(row0, ID(0), 1.0);
        row_push(row0, ID(1), -1.0);
        break;
    case ORC_LINES_AT_ANGLE: { /* :1358-1418 */
        static const int idx[8] = {0, 1, 2, 3, 4, 5, 6, 7};
        lines_at_angle_jacobian(c, x, idx, row0, degenerate);
        break;
    }
    case ORC_LINES_EQUAL_LENGTH: { /* :1419-1455 */
        double x0 = X(0), y0 = X(1), x1 = X(2), y1 = X(3), x2 = X(4), y2 = X(5), x3 = X(6), y3 = X(7);
        double len0 = v_euclidean_distance(v_new(x0, y0), v_new(x1, y1));
        double len1 = v_euclidean_distance(v_new(x2, y2), v_new(x3, y3));
        if (len0 < EPSILON || len1 < EPSILON) {
            *degenerate = 1;
            return;
        }
        row_push(row0, ID(0), (x0 - x1) / len0);
        row_push(row0, ID(1), (y0 - y1) / len0);
        row_push(row0, ID(2), (-x0 + x1) / len0);
        row_push(row0, ID(3), (-y0 + y1) / len0);
        row_push(row0, ID(4), (-x2 + x3) / len1);
        row_push(row0, ID(5), (-y2 + y3) / len1);
        row_push(row0, ID(6), (x2 - x3) / len1);
        row_push(row0, ID(7), (y2 - y3) / len1);
        break;
    }
    case ORC_POINTS_COINCIDENT: /* :1456-1504 */
        row_push(row0, ID(0), 1.0);
        row_push(row0, ID(2), -1.0);
        row_push(row1, ID(1), 1.0);
        row_push(row1, ID(3), -1.0);
        break;
    case ORC_CIRCLE_RADIUS: /* :1505-1512 */
        row_push(row0, ID(2), 1.0);
        break;
    case ORC_ARC_RADIUS: /* :1513-1536 */
        distance_jacobian(c, x, 0, 1, 2, 3, row0, degenerate);
        distance_jacobian(c, x, 0, 1, 4, 5, row1, degenerate);
        break;
    case ORC_ARC: { /* :1537-1598 */
        double cx = X(0), cy = X(1), sx = X(2), sy = X(3), ex = X(4), ey = X(5);
        double usx = sx - cx;
        double usy = sy - cy;
        double uex = ex - cx;
        double uey = ey - cy;
        double dist0 = hypot(usx, usy);
        double dist1 = hypot(uex, uey);
        if (dist0 <= EPSILON || dist1 <= EPSILON) {
            *degenerate = 1;
            return;
        }
        row_push(row0, ID(2), usx / dist0);
        row_push(row0, ID(3), usy / dist0);
        row_push(row0, ID(4), -uex / dist1);
        row_push(row0, ID(5), -uey / dist1);
        row_push(row0, ID(0), -usx / dist0 + uex / dist1);
        row_push(row0, ID(1), -usy / dist0 + uey / dist1);
        break;
    }
    case ORC_MIDPOINT: /* :1599-1642 */
        row_push(row0, ID(4), 1.0);
        row_push(row0, ID(0), -0.5);
        row_push(row0, ID(2), -0.5);
        row_push(row1, ID(5), 1.0);
        row_push(row1, ID(1), -0.5);
        row_push(row1, ID(3), -0.5);
        break;
    case ORC_POINT_LINE_DISTANCE: { /* :1643-1675, pds_for_point_line :2435-2516 (no degenerate guard) */
        double px = X(0), py = X(1), p0x = X(2), p0y = X(3), p1x = X(4), p1y = X(5);
        double euclid_dist = hypot(-p0x + p1x, p0y - p1y);
        double d_px = (p0y - p1y) / euclid_dist;
        double d_py = (-p0x + p1x) / euclid_dist;
        double denom = pow(pow(-p0x + p1x, 2.0) + pow(p0y - p1y, 2.0), 1.5);
        double common = (p0x * p1y - p0y * p1x + px * (p0y - p1y) + py * (-p0x + p1x));
        double d_p0x = ((-p0x + p1x) * common) / denom + (p1y - py) / euclid_dist;
        double d_p0y = ((-p0y + p1y) * common) / denom + (-p1x + px) / euclid_dist;
        double d_p1x = ((p0x - p1x) * common) / denom + (-p0y + py) / euclid_dist;
        double d_p1y = ((p0y - p1y) * common) / denom + (p0x - px) / euclid_dist;
        row_push(row0, ID(0), d_px);
        row_push(row0, ID(1), d_py);
        row_push(row0, ID(2), d_p0x);
        row_push(row0, ID(3), d_p0y);
        row_push(row0, ID(4), d_p1x);
        row_push(row0, ID(5), d_p1y);
        break;
    }
    case ORC_VERTICAL_POINT_LINE_DISTANCE: { /* :1676-1733 */
        double ax = X(0), px = X(2), py = X(3), qx = X(4), qy = X(5);
        double dx = qx - px;
        double dy = qy - py;
        if (fabs(dx) <= EPSILON || (dx * dx + dy * dy) <= EPSILON * EPSILON) {
            *degenerate = 1;
            return;
        }
        double dpx = (ax - qx) * (py - qy) * pow(px - qx, -2.0);
        double dpy = (-ax + qx) * rs_recip(px - qx);
        double dqx = -(ax - px) * (py - qy) * pow(px - qx, -2.0);
        double dqy = (ax - px) * rs_recip(px - qx);
        double dax = (-py + qy) * rs_recip(px - qx);
        double day = 1.0;
        row_push(row0, ID(0), dax);
        row_push(row0, ID(1), day);
        row_push(row0, ID(2), dpx);
        row_push(row0, ID(3), dpy);
        row_push(row0, ID(4), dqx);
        row_push(row0, ID(5), dqy);
        break;
    }
    case ORC_HORIZONTAL_POINT_LINE_DISTANCE: { /* :1734-1787 (note `<` here vs `<=` in the residual) */
        double ay = X(1), px = X(2), py = X(3), qx = X(4), qy = X(5);
        double dx = qx - px;
        double dy = qy - py;
        if (fabs(dy) < EPSILON || (dx * dx + dy * dy) < EPSILON * EPSILON) {
            *degenerate = 1;
            return;
        }
        double dpx = (-ay + qy) * rs_recip(py - qy);
        double dpy = (ay - qy) * (px - qx) * pow(py - qy, -2.0);
        double dqx = (ay - py) * rs_recip(py - qy);
        double dqy = -(ay - py) * (px - qx) * pow(py - qy, -2.0);
        double dax = 1.0;
        double day = (-px + qx) * rs_recip(py - qy);
        row_push(row0, ID(0), dax);
        row_push(row0, ID(1), day);
        row_push(row0, ID(2), dpx);
        row_push(row0, ID(3), dpy);
        row_push(row0, ID(4), dqx);
        row_push(row0, ID(5), dqy);
        break;
    }
    case ORC_SYMMETRIC: { /* :1788-1879, pds_from_symmetric :2361-2433 */
        double px = X(0), py = X(1), qx = X(2), qy = X(3), ax = X(4), ay = X(5);
        double dx = px - qx;
        double dy = py - qy;
        double dx2 = dx * dx;
        double dy2 = dy * dy;
        double r = dx2 + dy2;
        double r2 = pow(r, 2.0);
        if (r2 < EPSILON) {
            *degenerate = 1;
            return;
        }
        double sx = ax - px;
        double sy = ay - py;
        double dot = sx * dx + sy * dy;
        double dpx0 = (-4.0 * dx2 * dot + 2.0 * r2 + 2.0 * r * (sx * dx + sy * dy + dx * (ax - 2.0 * px + qx))) / r2;
        double dpx1 = dy * (-4.0 * dx * dot + 2.0 * r * (ax - 2.0 * px + qx)) / r2;
        double dpy0 = dx * (-4.0 * dy * dot + 2.0 * r * (ay - 2.0 * py + qy)) / r2;
        double dpy1 = (-4.0 * dy2 * dot + 2.0 * r2 + 2.0 * r * (sx * dx + sy * dy + dy * (ay - 2.0 * py + qy))) / r2;
        double dqx0 = (4.0 * dx2 * dot - (4.0 * sx * dx + 2.0 * sy * dy) * r) / r2;
        double dqx1 = dy * (-2.0 * sx * r + 4.0 * dx * dot) / r2;
        double dqy0 = dx * (-2.0 * sy * r + 4.0 * dy * dot) / r2;
        double dqy1 = (4.0 * dy2 * dot - (2.0 * sx * dx + 4.0 * sy * dy) * r) / r2;
        double dax0 = 1.0 * (dx2 - dy2) / r;
        double dax1 = 2.0 * dx * dy / r;
        double day0 = 2.0 * dx * dy / r;
        double day1 = 1.0 * (-dx2 + dy2) / r;
        row_push(row0, ID(0), dpx0);
        row_push(row0, ID(1), dpy0);
        row_push(row0, ID(2), dqx0);
        row_push(row0, ID(3), dqy0);
        row_push(row0, ID(4), dax0);
        row_push(row0, ID(5), day0);
        row_push(row0, ID(6), -1.0);
        row_push(row0, ID(7), 0.0);
        row_push(row1, ID(0), dpx1);
        row_push(row1, ID(1), dpy1);
        row_push(row1, ID(2), dqx1);
        row_push(row1, ID(3), dqy1);
        row_push(row1, ID(4), dax1);
        row_push(row1, ID(5), day1);
        row_push(row1, ID(6), 0.0);
        row_push(row1, ID(7), -1.0);
        break;
    }
    case ORC_POINT_ARC_COINCIDENT: { /* :1880-2063 */
        V cc = v_new(X(0), X(1));
        V s = v_sub(v_new(X(2), X(3)), cc);
        V e = v_sub(v_new(X(4), X(5)), cc);
        V p = v_sub(v_new(X(6), X(7)), cc);
        double r = v_magnitude(s);
        double r_e = v_magnitude(e);
        double r_p = v_magnitude(p);
        if (r < EPSILON || r_e < EPSILON || r_p < EPSILON) {
            *degenerate = 1;
            return;
        }
        V u_s = v_scale(s, rs_recip(r));
        V u_e = v_scale(e, rs_recip(r_e));
        V e_proj = v_scale(e, r / r_e);
        double j_s[2][2], j_e[2][2], j_p[2][2];
        switch (classify_point_arc_coincident(s, e_proj, p)) {
        case PAC_INTERIOR: {
            V u_p = v_scale(p, rs_recip(r_p));
            double r_over_rp = r / r_p;
            j_s[0][0] = u_p.x * u_s.x;
            j_s[0][1] = u_p.y * u_s.x;
            j_s[1][0] = u_p.x * u_s.y;
            j_s[1][1] = u_p.y * u_s.y;
            j_e[0][0] = 0.0;
            j_e[0][1] = 0.0;
            j_e[1][0] = 0.0;
            j_e[1][1] = 0.0;
            j_p[0][0] = (r_over_rp - 1.0) - r_over_rp * u_p.x * u_p.x;
            j_p[0][1] = -r_over_rp * u_p.y * u_p.x;
            j_p[1][0] = -r_over_rp * u_p.x * u_p.y;
            j_p[1][1] = (r_over_rp - 1.0) - r_over_rp * u_p.y * u_p.y;
            break;
        }
        case PAC_END: {
            double r_over_re = r / r_e;
            j_s[0][0] = u_e.x * u_s.x;
            j_s[0][1] = u_e.y * u_s.x;
            j_s[1][0] = u_e.x * u_s.y;
            j_s[1][1] = u_e.y * u_s.y;
            j_e[0][0] = r_over_re * (1.0 - u_e.x * u_e.x);
            j_e[0][1] = -r_over_re * u_e.y * u_e.x;
            j_e[1][0] = -r_over_re * u_e.x * u_e.y;
            j_e[1][1] = r_over_re * (1.0 - u_e.y * u_e.y);
            j_p[0][0] = -1.0;
            j_p[0][1] = 0.0;
            j_p[1][0] = 0.0;
            j_p[1][1] = -1.0;
            break;
        }
        default: {
            j_s[0][0] = 1.0;
            j_s[0][1] = 0.0;
            j_s[1][0] = 0.0;
            j_s[1][1] = 1.0;
            j_e[0][0] = 0.0;
            j_e[0][1] = 0.0;
            j_e[1][0] = 0.0;
            j_e[1][1] = 0.0;
            j_p[0][0] = -1.0;
            j_p[0][1] = 0.0;
            j_p[1][0] = 0.0;
            j_p[1][1] = -1.0;
            break;
        }
        }
        double j_o[2][2];
        j_o[0][0] = -(j_s[0][0] + j_e[0][0] + j_p[0][0]);
        j_o[0][1] = -(j_s[0][1] + j_e[0][1] + j_p[0][1]);
        j_o[1][0] = -(j_s[1][0] + j_e[1][0] + j_p[1][0]);
        j_o[1][1] = -(j_s[1][1] + j_e[1][1] + j_p[1][1]);
        row_push(row0, ID(0), j_o[0][0]);
        row_push(row0, ID(1), j_o[1][0]);
        row_push(row0, ID(2), j_s[0][0]);
        row_push(row0, ID(3), j_s[1][0]);
        row_push(row0, ID(4), j_e[0][0]);
        row_push(row0, ID(5), j_e[1][0]);
        row_push(row0, ID(6), j_p[0][0]);
        row_push(row0, ID(7), j_p[1][0]);
        row_push(row1, ID(0), j_o[0][1]);
        row_push(row1, ID(1), j_o[1][1]);
        row_push(row1, ID(2), j_s[0][1]);
        row_push(row1, ID(3), j_s[1][1]);
        row_push(row1, ID(4), j_e[0][1]);
        row_push(row1, ID(5), j_e[1][1]);
        row_push(row1, ID(6), j_p[0][1]);
        row_push(row1, ID(7), j_p[1][1]);
        break;
    }
    case ORC_ARC_LENGTH: { /* :2064-2163 */
        double cx = X(0), cy = X(1), ax = X(2), ay = X(3);
        double d = c->param;
        double ux = ax - cx;
        double uy = ay - cy;
        double r2 = ux * ux + uy * uy;
        if (r2 <= EPSILON * EPSILON) {
            *degenerate = 1;
            return;
        }
        double r = sqrt(r2);
        double alpha = d / r;
        double sa = sin(alpha);
        double ca = cos(alpha);
        double rux = ca * ux - sa * uy;
        double ruy = sa * ux + ca * uy;
        double k = d / (r2 * r);
        row_push(row0, ID(2), -ca - ruy * ux * k);
        row_push(row0, ID(3), sa - ruy * uy * k);
        row_push(row0, ID(4), 1.0);
        row_push(row0, ID(5), 0.0);
        row_push(row0, ID(0), -1.0 + ca + ruy * ux * k);
        row_push(row0, ID(1), -sa + ruy * uy * k);
        row_push(row1, ID(2), -sa + rux * ux * k);
        row_push(row1, ID(3), -ca + rux * uy * k);
        row_push(row1, ID(4), 0.0);
        row_push(row1, ID(5), 1.0);
        row_push(row1, ID(0), sa - rux * ux * k);
        row_push(row1, ID(1), -1.0 + ca - rux * uy * k);
        break;
    }
    case ORC_ARC_ANGLE: { /* :2164-2175 LinesAtAngle(center->start, center->end) */
        static const int idx[8] = {0, 1, 2, 3, 0, 1, 4, 5};
        lines_at_angle_jacobian(c, x, idx, row0, degenerate);
        break;
    }
    case ORC_POINTS_AT_ANGLE: { /* :2176-2291 */
        V p0v = v_new(X(0), X(1));
        V p1v = v_new(X(2), X(3));
        V p2v = v_new(X(4), X(5));
        V u = v_sub(p1v, p0v);
        V v = v_sub(p2v, p0v);
        double len_u = v_magnitude(u);
        double len_v = v_magnitude(v);
        if (len_u <= EPSILON || len_v <= EPSILON) {
            *degenerate = 1;
            return;
        }
        double inv_len_u = 1.0 / len_u;
        double inv_len_v = 1.0 / len_v;
        V u_hat = v_scale(u, inv_len_u);
        V v_hat = v_scale(v, inv_len_v);
        Rot2 rot = rotation_for_angle_kind(c->tag, c->param);
        double s = (len_u + len_v) * 0.5;
        V rot_e1 = rot_apply(rot, v_new(1.0, 0.0));
        V rot_e2 = rot_apply(rot, v_new(0.0, 1.0));
        double inv_s = 1.0 / s;
        V rot_u = rot_apply(rot, u);
        V res = v_scale(v_sub(v_scale(v, len_u), v_scale(rot_u, len_v)), inv_s);
        V half_res = v_scale(res, 0.5);
        V dr_du0 = v_scale(v_sub(v_scale(v_sub(v, half_res), u_hat.x), v_scale(rot_e1, len_v)), inv_s);
        V dr_du1 = v_scale(v_sub(v_scale(v_sub(v, half_res), u_hat.y), v_scale(rot_e2, len_v)), inv_s);
        V dr_dv0 = v_scale(v_sub(v_new(len_u, 0.0), v_scale(v_add(rot_u, half_res), v_hat.x)), inv_s);
        V dr_dv1 = v_scale(v_sub(v_new(0.0, len_u), v_scale(v_add(rot_u, half_res), v_hat.y)), inv_s);
        row_push(row0, ID(0), -(dr_du0.x + dr_dv0.x));
        row_push(row0, ID(1), -(dr_du1.x + dr_dv1.x));
        row_push(row0, ID(2), dr_du0.x);
        row_push(row0, ID(3), dr_du1.x);
        row_push(row0, ID(4), dr_dv0.x);
        row_push(row0, ID(5), dr_dv1.x);
        row_push(row1, ID(0), -(dr_du0.y + dr_dv0.y));
        row_push(row1, ID(1), -(dr_du1.y + dr_dv1.y));
        row_push(row1, ID(2), dr_du0.y);
        row_push(row1, ID(3), dr_du1.y);
        row_push(row1, ID(4), dr_dv0.y);
        row_push(row1, ID(5), dr_dv1.y);
        break;
    }
    default:
        break;
    }
}
