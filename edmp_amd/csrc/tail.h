// tail.h — the tail of one reverse step: the UNet's final 1x1 conv + the posterior step (+ conditioning + next UNet input),
// shared by head_psample_kernel (sampler.hip) and the last whole-level kernel of the layer program (level.hip, LV_UP_FINAL),
// which runs it on the activations it still holds in LDS when the device-resident loop asks for it (TailP::on).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace edmp {

// ---- device noise source (NOT the reference's NumPy stream: a separate, explicitly non-parity mode) --------------------
// Philox4x32-10 counter RNG (Salmon et al. 2011): counter = (element, step, block, 0), key = seed.  Eight standard
// normals per (sample, waypoint) and step via Box-Muller, one per joint channel.  Removes the 0.9 s host draw and the
// 734 MB upload per scene that the NumPy-stream contract costs.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

__device__ __forceinline__ void rng_normal8(uint64_t seed, uint32_t step, uint32_t elem, float (&z)[8]) {
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        uint32_t u[4];
        philox4x32_10(elem, step, (uint32_t)blk, 0u, k0, k1, u);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float u1 = ((float)u[2 * h] + 1.0f) * 2.3283064365386963e-10f;  // (0, 1]
            const float u2 = (float)u[2 * h + 1] * 2.3283064365386963e-10f;       // [0, 1)
            const float r = sqrtf(-2.0f * logf(u1));
            float sn, cs;
            sincospif(2.0f * u2, &sn, &cs);
            z[4 * blk + 2 * h] = r * cs;
            z[4 * blk + 2 * h + 1] = r * sn;
        }
    }
}

// what the tail of reverse step t needs (sampler.hip: step_a fills it; level.hip consumes it)
struct TailP {
    int on = 0;            // 0: the level kernel writes its activations to HBM and a separate launch does the tail
    const float* w = nullptr;     // final_conv.1 weight [C][32]
    const float* bias = nullptr;  // [C]
    double* X = nullptr;          // state (B, C, N) f64
    const double* z = nullptr;    // noise of this step (B, C, N) f64 (rng == 0)
    float* xin = nullptr;         // next UNet input [B][N][8] f32 (finish)
    const double* sg = nullptr;   // start (7) | goal (7)
    int C = 0, N = 0;
    double c1 = 0.0, sqrt_alpha = 1.0, beta = 0.0;
    int zero_row0 = 0;
    unsigned long long seed = 0;
    int rng_step = 0;
    int cond = 0;
    int finish = 0;  // steps without guidance: condition X and write the next UNet input
    int rng = 0;     // device noise (Philox) instead of z
};

// eps[c] = final 1x1 conv of the UNet (final_conv.1, temporalunet.py:36) on the CIN activations hv of (sample b, waypoint l);
// X <- (X - c1 eps)/sqrt(alpha) + beta z (diffusion.py:116-135); FINISH: X[:, :, 0] = start, X[:, :, -1] = goal
// (diffusion.py:347-349) and the next UNet input.  i = b * N + l.  One code path for both callers: bit-identical results.
// The state and noise values of the item are fetched by the caller (tail_fetch) - ahead of the work whose result hv is, so
// that their memory latency is off the tail's critical path.
__device__ __forceinline__ void tail_fetch(const double* __restrict__ X, const double* __restrict__ z, bool rng, int b, int l, int N, int C, double (&xv)[8],
                                           double (&zv)[8]) {
#pragma unroll
    for (int co = 0; co < 8; ++co) {
        xv[co] = 0.0;
        zv[co] = 0.0;
        if (co < C) {
            const size_t idx = ((size_t)b * C + co) * N + l;
            xv[co] = X[idx];
            if (!rng) zv[co] = z[idx];
        }
    }
}

template <bool FINISH, bool RNG, int CIN>
__device__ __forceinline__ void head_psample_item(const float4 (&hv)[CIN / 4], const double (&xv)[8], const double (&zv)[8], int i, int b, int l,
                                                  const float* __restrict__ w, const float* __restrict__ bias, double* __restrict__ X,
                                                  float* __restrict__ eps_out, float* __restrict__ xin, const double* __restrict__ sg, int N, int C, double c1,
                                                  double sqrt_alpha, double beta, int zero_row0, uint64_t seed, int rng_step, int cond) {
    float xo[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float zr[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (RNG) rng_normal8(seed, (uint32_t)rng_step, (uint32_t)i, zr);
    float acc[8];
#pragma unroll
    for (int co = 0; co < 8; ++co) acc[co] = (co < C) ? bias[co] : 0.0f;
#pragma unroll
    for (int q = 0; q < CIN / 4; ++q) {
#pragma unroll
        for (int co = 0; co < 8; ++co) {
            if (co < C) {
                const float* wr = w + co * CIN + 4 * q;
                acc[co] = fmaf(hv[q].x, wr[0], acc[co]);
                acc[co] = fmaf(hv[q].y, wr[1], acc[co]);
                acc[co] = fmaf(hv[q].z, wr[2], acc[co]);
                acc[co] = fmaf(hv[q].w, wr[3], acc[co]);
            }
        }
    }
#pragma unroll
    for (int co = 0; co < 8; ++co) {
        if (co >= C) break;
        const float a = acc[co];
        const size_t idx = ((size_t)b * C + co) * N + l;
        if (eps_out) eps_out[idx] = a;
        double zz = RNG ? (double)zr[co] : zv[co];
        if (zero_row0 && b == 0) zz = 0.0;
        double x = (xv[co] - c1 * (double)a) / sqrt_alpha + beta * zz;
        if (FINISH) {
            if (cond && l == 0) x = sg[co];
            if (cond && l == N - 1) x = sg[7 + co];
            xo[co] = (float)x;
        }
        X[idx] = x;
    }
    if (FINISH) {
        float4* o = reinterpret_cast<float4*>(xin + (size_t)i * 8);
        o[0] = make_float4(xo[0], xo[1], xo[2], xo[3]);
        o[1] = make_float4(xo[4], xo[5], xo[6], xo[7]);
    }
}

}  // namespace edmp
