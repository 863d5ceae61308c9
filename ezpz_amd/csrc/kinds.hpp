// Static per-kind facts shared by the host symbolic phase and the device evaluators.
//
// For every constraint kind: how many ids it uses, how many residual rows it has
// (reference ezpz/src/constraints.rs:954-993), the ids `nonzeroes` reports per row (:378-491, used
// for validation in the reference's order) and the ids-index of every partial `jacobian_rows` emits,
// in emission order (:1000-2293).  The device evaluator in constraint_eval.hip.hpp stores partial
// number e of a kind through DevCon::jloc[e]; the two must agree, which tests/test_gpu_parity.py
// checks against the oracle for all 25 kinds.
#pragma once
#include <cstdint>

namespace ezpz {

struct KindInfo {
    uint8_t n_ids;
    uint8_t n_rows;
    uint8_t n_emit[2];      // partials emitted per row
    uint8_t emit[2][8];     // ids-index of each emitted partial
    uint8_t n_nz[2];        // `nonzeroes` ids per row
    uint8_t nz[2][8];
};

// clang-format off
static constexpr KindInfo kKinds[25] = {
    /* 0 LineTangentToCircle */ {7, 1, {7, 0}, {{0,1,2,3,4,5,6}, {}}, {7, 0}, {{0,1,2,3,4,5,6}, {}}},
    /* 1 CircleTangentToCircle */ {6, 1, {6, 0}, {{0,1,2,3,4,5}, {}}, {6, 0}, {{0,1,2,3,4,5}, {}}},
    /* 2 Distance */ {4, 1, {4, 0}, {{0,1,2,3}, {}}, {4, 0}, {{0,1,2,3}, {}}},
    /* 3 DistanceVar */ {5, 1, {5, 0}, {{0,1,2,3,4}, {}}, {5, 0}, {{0,1,2,3,4}, {}}},
    /* 4 VerticalDistance */ {4, 1, {2, 0}, {{1,3}, {}}, {2, 0}, {{1,3}, {}}},
    /* 5 HorizontalDistance */ {4, 1, {2, 0}, {{0,2}, {}}, {2, 0}, {{0,2}, {}}},
    /* 6 Vertical */ {4, 1, {2, 0}, {{0,2}, {}}, {2, 0}, {{0,2}, {}}},
    /* 7 Horizontal */ {4, 1, {2, 0}, {{1,3}, {}}, {2, 0}, {{1,3}, {}}},
    /* 8 LinesAtAngle */ {8, 1, {8, 0}, {{0,1,2,3,4,5,6,7}, {}}, {8, 0}, {{0,1,2,3,4,5,6,7}, {}}},
    /* 9 Fixed */ {1, 1, {1, 0}, {{0}, {}}, {1, 0}, {{0}, {}}},
    /* 10 ScalarEqual */ {2, 1, {2, 0}, {{0,1}, {}}, {2, 0}, {{0,1}, {}}},
    /* 11 PointsCoincident */ {4, 2, {2, 2}, {{0,2}, {1,3}}, {2, 2}, {{0,2}, {1,3}}},
    /* 12 CircleRadius */ {3, 1, {1, 0}, {{2}, {}}, {1, 0}, {{2}, {}}},
    /* 13 LinesEqualLength */ {8, 1, {8, 0}, {{0,1,2,3,4,5,6,7}, {}}, {8, 0}, {{0,1,2,3,4,5,6,7}, {}}},
    /* 14 ArcRadius */ {6, 2, {4, 4}, {{0,1,2,3}, {0,1,4,5}}, {4, 4}, {{0,1,2,3}, {0,1,4,5}}},
    /* 15 Arc */ {6, 1, {6, 0}, {{2,3,4,5,0,1}, {}}, {6, 0}, {{2,3,4,5,0,1}, {}}},
    /* 16 Midpoint */ {6, 2, {3, 3}, {{4,0,2}, {5,1,3}}, {3, 3}, {{0,2,4}, {1,3,5}}},
    /* 17 PointLineDistance */ {6, 1, {6, 0}, {{0,1,2,3,4,5}, {}}, {6, 0}, {{0,1,2,3,4,5}, {}}},
    /* 18 VerticalPointLineDistance */ {6, 1, {6, 0}, {{0,1,2,3,4,5}, {}}, {6, 0}, {{2,3,4,5,0,1}, {}}},
    /* 19 HorizontalPointLineDistance */ {6, 1, {6, 0}, {{0,1,2,3,4,5}, {}}, {6, 0}, {{2,3,4,5,0,1}, {}}},
    /* 20 Symmetric */ {8, 2, {8, 8}, {{0,1,2,3,4,5,6,7}, {0,1,2,3,4,5,6,7}}, {8, 8}, {{0,1,2,3,4,5,6,7}, {0,1,2,3,4,5,6,7}}},
    /* 21 PointArcCoincident */ {8, 2, {8, 8}, {{0,1,2,3,4,5,6,7}, {0,1,2,3,4,5,6,7}}, {8, 8}, {{2,3,4,5,0,1,6,7}, {2,3,4,5,0,1,6,7}}},
    /* 22 ArcLength */ {6, 2, {6, 6}, {{2,3,4,5,0,1}, {2,3,4,5,0,1}}, {6, 6}, {{2,3,4,5,0,1}, {2,3,4,5,0,1}}},
    /* 23 ArcAngle */ {6, 1, {8, 0}, {{0,1,2,3,0,1,4,5}, {}}, {8, 0}, {{0,1,2,3,0,1,4,5}, {}}},
    /* 24 PointsAtAngle */ {6, 2, {6, 6}, {{0,1,2,3,4,5}, {0,1,2,3,4,5}}, {6, 6}, {{0,1,2,3,4,5}, {0,1,2,3,4,5}}},
};
// clang-format on

// The nine kinds whose residual is linear in the variables (constant Jacobian, no guard, no libm call).
inline bool kind_is_linear(uint32_t kind) {
    switch (kind) {
    case EZPZ_FIXED:
    case EZPZ_SCALAR_EQUAL:
    case EZPZ_VERTICAL:
    case EZPZ_HORIZONTAL:
    case EZPZ_VERTICAL_DISTANCE:
    case EZPZ_HORIZONTAL_DISTANCE:
    case EZPZ_CIRCLE_RADIUS:
    case EZPZ_POINTS_COINCIDENT:
    case EZPZ_MIDPOINT:
        return true;
    default:
        return false;
    }
}

}  // namespace ezpz
