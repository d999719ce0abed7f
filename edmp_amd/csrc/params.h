// params.h — launch-parameter structs of the UNet kernels (unet.hip, wide.hip, level.hip) and the phase-stamp macro;
// shared by the core translation unit and the kernel shards (kernel_shard.hip).
#pragma once
#include "common.h"
#include "tail.h"

namespace edmp {

struct ConvP {
    const float* src1;
    const float* src2;  // second half of a channel concat, or nullptr
    int C1, C2;         // channels of src1 / src2 (storage widths)
    int Lin, Lout;
    int ntaps, stride, pad, transposed;
    const float* W;  // [tap][Cout][C1 + C2]
    const float* bias;
    float* dst;  // [B][Lout][Cout]
    int Cout;
    int B;
};

struct GnP {
    float* y;  // [B][L][C], normalised in place
    const float* gamma;
    const float* beta;
    const float* add_res;    // [B][L][C] or nullptr
    const float* add_tbias;  // [C] (already offset to step t) or nullptr
    int L, C, B;
};

struct RcbP {
    const float* src1;
    const float* src2;
    int C1, C2;
    const float* W;  // [5][Cout][C1+C2]
    const float* bias;
    const float* gamma;
    const float* beta;
    const float* add_tb;   // [Cout] time bias of step t, or nullptr
    const float* add_res;  // [B][L][Cout] residual, or nullptr
    float* dst;            // [B][L][Cout]
    int Cout;
    int B;
    // folded residual 1x1 conv of the block input (wide_conv_kernel<.., RES = true>): its weights [Cout][Cin] sit right behind
    // the five conv taps in W (tap index 5); res_out [B][L][Cout] receives conv + res_bias for conv2's epilogue
    float* res_out;
    const float* res_bias;
    int gx_shift = -1, ng_shift = -1;  // wide_conv_kernel: log2 of the XCDs across the channel groups (wide.hip: xcd_split) and of the group count, set by the launcher; -1 = plain mapping
};

// whole-level kernel (level.hip)
enum LevelMode { LV_DOWN = 0, LV_UP = 1, LV_UP_FINAL = 2 };

struct LevelP {
    const float* src1;  // [B][L][C1]
    const float* src2;  // [B][L][C2] concatenated behind src1 on the channel axis, or nullptr
    int C1, C2;
    // weight fragment streams (pack_fragments, sw = 16: [C/16][K/16][slots][64][4])
    const float* w11;   // RCB1 conv1, K = KX, slots 0..4 = taps, slot 5 = the residual 1x1 conv
    const float* w12;   // RCB1 conv2, K = C
    const float* w21;   // RCB2 conv1
    const float* w22;   // RCB2 conv2
    const float* wrs;   // resampling conv: k3 s2 (3 slots) | ConvTranspose k4 s2 (4 slots)
    const float* wfin;  // final Conv1dBlock (LV_UP_FINAL)
    // per-channel vectors [C]: conv bias, GroupNorm gamma / beta, time bias of step t, residual-conv bias
    const float *b11, *g11, *be11, *tb1, *rb1;
    const float *b12, *g12, *be12;
    const float *b21, *g21, *be21, *tb2;
    const float *b22, *g22, *be22;
    const float* brs;
    const float *bfin, *gfin, *befin;
    float* skip_out;  // [B][L][C] output of the second block (the level's skip tensor), or nullptr
    float* out;       // [B][LOUT][C] (LV_UP_FINAL: the final Conv1dBlock's output at the up-sampled length)
    int B;
    TailP tail;       // LV_UP_FINAL in the device-resident loop: head 1x1 conv + posterior step on the tile still in LDS (tail.h)
};

#ifdef EDMP_STAMPS  // phase timing experiment (scratch builds only): one wave of one mid-grid workgroup stamps s_memtime
#ifdef EDMP_STAMPS_DEFINE
__device__ unsigned long long g_stamps[8][16];
#else
extern __device__ unsigned long long g_stamps[8][16];
#endif
#define EDMP_STAMP(k, i)                                                        \
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == gridDim.y / 2) { \
        g_stamps[k][2 * (i)] = clock64();                                       \
        g_stamps[k][2 * (i) + 1] = wall_clock64();                              \
    }
// per-workgroup dispatch timeline (tools/levelbench.hip): wall clock at entry / exit and the hardware id (XCC, SE, CU) of every workgroup
#ifdef EDMP_STAMPS_DEFINE
__device__ unsigned long long g_wg_times[1024][4];
#else
extern __device__ unsigned long long g_wg_times[1024][4];
#endif
#define EDMP_WG_STAMP(i)                                                                        \
    if (threadIdx.x == 0 && blockIdx.x < 1024) {                                                \
        g_wg_times[blockIdx.x][i] = wall_clock64();                                             \
        if ((i) == 0) {                                                                         \
            g_wg_times[blockIdx.x][2] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); /* HW_REG_XCC_ID */ \
            g_wg_times[blockIdx.x][3] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  /* HW_REG_HW_ID */  \
        }                                                                                       \
    }
#else
#define EDMP_STAMP(k, i)
#define EDMP_WG_STAMP(i)
#endif

}  // namespace edmp
