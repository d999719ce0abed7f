// kernel_instances.h — every instance of the position-tile kernels (wide.hip, bf3.hip) and of the whole-level kernel (level.hip)
// the layer program can launch (unet.hip: launch_rcb / launch_wrs / launch_level), with the build shard that compiles it.
// The sharded build (__graft_entry__.build) compiles kernel_shard.hip once per shard in parallel (one instance takes
// 3-30 s of compile time, the lot in one translation unit a quarter of an hour); the core translation unit only sees
// `extern template` declarations.  Shards are balanced by compile time.
#pragma once

// X(shard, KIND, MS, CG, GS, LIN, RES)
#define EDMP_WIDE_INSTANCES(X)            \
    X(0, WK_K5K4, 32, 64, 64, 4, true)    \
    X(1, WK_K5K4, 32, 64, 64, 4, false)   \
    X(2, WK_K5K4, 32, 32, 32, 4, true)    \
    X(3, WK_K5K4, 32, 32, 32, 4, false)   \
    X(4, WK_K5, 16, 32, 16, 13, true)     \
    X(5, WK_K5, 16, 32, 16, 13, false)    \
    X(6, WK_K5, 32, 32, 32, 7, true)      \
    X(7, WK_K5, 32, 32, 32, 7, false)     \
    X(6, WK_K5, 16, 32, 16, 7, true)      \
    X(7, WK_K5, 16, 32, 16, 7, false)     \
    X(0, WK_K5K2, 32, 64, 64, 2, true)    \
    X(1, WK_K5K2, 32, 64, 64, 2, false)   \
    X(2, WK_K5, 32, 64, 64, 4, true)      \
    X(3, WK_K5, 32, 64, 64, 4, false)     \
    X(4, WK_K5, 32, 32, 32, 4, true)      \
    X(5, WK_K5, 32, 32, 32, 4, false)     \
    X(6, WK_K5, 32, 64, 64, 2, true)      \
    X(7, WK_K5, 32, 64, 64, 2, false)     \
    X(0, WK_DOWN, 32, 64, 64, 4, false)   \
    X(1, WK_DOWN, 32, 32, 32, 7, false)   \
    X(2, WK_DOWN, 16, 32, 16, 13, false)  \
    X(3, WK_UP, 32, 64, 64, 2, false)     \
    X(4, WK_UP, 32, 32, 32, 4, false)     \
    X(5, WK_UP, 16, 32, 16, 7, false)     \
    X(8, WK_K5, 16, 32, 32, 7, true)      \
    X(9, WK_K5, 16, 32, 32, 7, false)     \
    X(10, WK_DOWN, 16, 32, 32, 7, false)  \
    X(11, WK_UP, 16, 32, 32, 4, false)    \
    X(12, WK_DOWN, 16, 64, 64, 4, false)  \
    X(13, WK_UP, 16, 64, 64, 2, false)

// X(shard, KIND, MS, CG, GS, LIN, RES): the bf16x3 position-tile kernel (bf3.hip)
#define EDMP_BF3_INSTANCES(X)            \
    X(16, WK_K5, 32, 32, 32, 7, true)    \
    X(17, WK_K5, 32, 32, 32, 7, false)   \
    X(18, WK_K5, 16, 32, 16, 7, true)    \
    X(16, WK_K5, 16, 32, 16, 7, false)   \
    X(17, WK_K5, 16, 32, 16, 13, true)   \
    X(18, WK_K5, 16, 32, 16, 13, false)  \
    X(16, WK_DOWN, 32, 32, 32, 7, false) \
    X(17, WK_UP, 32, 32, 32, 4, false)   \
    X(18, WK_DOWN, 32, 32, 32, 4, false) \
    X(16, WK_UP, 32, 32, 32, 2, false)   \
    X(17, WK_DOWN, 16, 32, 16, 13, false) \
    X(18, WK_UP, 16, 32, 16, 7, false)    \
    X(19, WK_K5, 32, 64, 64, 4, true)     \
    X(20, WK_K5, 32, 64, 64, 4, false)    \
    X(19, WK_K5, 32, 32, 32, 4, true)     \
    X(20, WK_K5, 32, 32, 32, 4, false)

// X(shard, MODE, C, L, SB, CIN)
#define EDMP_LEVEL_INSTANCES(X)        \
    X(8, LV_DOWN, 32, 50, 4, 8)        \
    X(9, LV_DOWN, 64, 25, 4, 32)       \
    X(10, LV_UP, 64, 13, 4, 256)       \
    X(11, LV_UP_FINAL, 32, 25, 4, 128) \
    X(12, LV_DOWN, 32, 50, 2, 8)       \
    X(13, LV_DOWN, 64, 25, 2, 32)      \
    X(14, LV_UP, 64, 13, 2, 256)       \
    X(15, LV_UP_FINAL, 32, 25, 2, 128)

// X(shard, MA, CA, LA, CINA, MB, CB, LB, CINB, SB): two consecutive levels in one launch (level.hip: level2_kernel)
#define EDMP_LEVEL2_INSTANCES(X)                      \
    X(14, LV_DOWN, 32, 50, 8, LV_DOWN, 64, 25, 32, 2) \
    X(15, LV_UP, 64, 13, 256, LV_UP_FINAL, 32, 25, 128, 2)

#define EDMP_KERNEL_SHARDS 21
