// guide.h — the per-scene guide object shared by guide.hip (tables, cost / gradient kernels) and success.hip (the
// geometric success check over the finished batch).
#pragma once
#include "common.h"

namespace edmp {

struct RobotConst {
    float dh[7][4];      // a, d, cos(alpha), sin(alpha)
    float sf[9][12];     // static frames, row-major 3x4
    float he[9][3];      // link half extents
    double qlo[7], qhi[7];
};

struct Guide {
    int no = 0, G = 0, T = 0;
    float* aabb = nullptr;  // [G][T+1][no][6]
    RobotConst rc{};
    // the obstacles as the simulator of the reference spawns them (lib/environment.py:230-268): oriented boxes / cylinders,
    // f64 [no][16] = world rotation (row-major 3x3, columns = axes), centre, half extents, pad; kind 0 cuboid / 1 cylinder
    double* obb = nullptr;
    int32_t* kind = nullptr;
    // rows
    int B = 0;
    int32_t* row_class = nullptr;
    float* method = nullptr;
    double* grad_norm = nullptr;
    double* sched = nullptr;  // [B][T]
    int rows_T = 0;
    // scratch
    float* graw = nullptr;    // [B][7][L] raw f32 gradient
    double* rowsq = nullptr;  // [B]
    double* sumsq = nullptr;  // [1]
    float* startgoal = nullptr;  // [14] f32
    int scratch_B = 0, scratch_L = 0;
    float* vol_rows = nullptr;  // [B] for best trajectory
    int32_t* flags = nullptr;   // [3][flags_B] success check: ok, first colliding waypoint, within limits; + [4] counts
    int flags_B = 0;
};

}  // namespace edmp
