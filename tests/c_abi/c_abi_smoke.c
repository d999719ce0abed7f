/* c_abi_smoke.c — a plain-C host that drives libedmp_hip.so through include/edmp_hip.h only (no Python, no torch):
 * the binding a non-Python caller would write.  Device buffers come from the HIP runtime API.
 *
 * usage: c_abi_smoke <in.bin> <out.bin>
 *   in.bin  (little endian): int32 B, n_obstacles, n_classes, T, n_params, t_stop; then
 *           float  params[n_params]                    state-dict blob (tiny UNet dims 16,16,32,32,64,64)
 *           double obstacle_config[n_obstacles*10], clearance[n_classes*T], expansion[n_classes*T]
 *           float  half_extents[27], dh[28], static_frames[108]
 *           int32  row_class[B]; float method[B]; double grad_norm[B]; double sched[B*T]
 *           double start[7], goal[7]; double noise[(T+1)*B*7*50]
 *   out.bin: double X[B*7*50] after reverse steps T..t_stop+1, int32 best_index, float volumes[B]
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "edmp_hip.h"

#define CHECK(x)                                                                  \
    do {                                                                          \
        int rc_ = (x);                                                            \
        if (rc_ != 0) {                                                           \
            fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, edmp_last_error()); \
            return 2;                                                             \
        }                                                                         \
    } while (0)
#define HIPCHECK(x)                                                  \
    do {                                                             \
        hipError_t e_ = (x);                                         \
        if (e_ != hipSuccess) {                                      \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            return 3;                                                \
        }                                                            \
    } while (0)

static void* rd(FILE* f, size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (fread(p, 1, bytes, f) != bytes) {
        fprintf(stderr, "short read\n");
        exit(4);
    }
    return p;
}

int main(int argc, char** argv) {
    if (argc != 3) return 1;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    int32_t hdr[6];
    if (fread(hdr, 4, 6, f) != 6) return 1;
    const int B = hdr[0], no = hdr[1], G = hdr[2], T = hdr[3], np_ = hdr[4], t_stop = hdr[5];
    float* params = rd(f, (size_t)np_ * 4);
    double* oc = rd(f, (size_t)no * 10 * 8);
    double* clr = rd(f, (size_t)G * T * 8);
    double* exp_ = rd(f, (size_t)G * T * 8);
    float* he = rd(f, 27 * 4);
    float* dh = rd(f, 28 * 4);
    float* sf = rd(f, 108 * 4);
    int32_t* rc = rd(f, (size_t)B * 4);
    float* method = rd(f, (size_t)B * 4);
    double* gn = rd(f, (size_t)B * 8);
    double* sched = rd(f, (size_t)B * T * 8);
    double* start = rd(f, 56);
    double* goal = rd(f, 56);
    const size_t n = (size_t)B * 7 * 50;
    double* noise = rd(f, (size_t)(T + 1) * n * 8);
    fclose(f);

    edmp_ctx* ctx = NULL;
    CHECK(edmp_ctx_create(0, &ctx));
    edmp_unet_desc d;
    memset(&d, 0, sizeof d);
    d.input_dim = 7;
    d.time_dim = 32;
    d.n_levels = 6;
    const int dims[6] = {16, 16, 32, 32, 64, 64};
    for (int i = 0; i < 6; ++i) d.dims[i] = dims[i];
    d.horizon = 50;
    d.T = T;
    if (edmp_unet_param_count(&d) != np_) {
        fprintf(stderr, "param count mismatch\n");
        return 5;
    }
    CHECK(edmp_unet_load(ctx, &d, params, np_, B));
    CHECK(edmp_sampler_init(ctx, T, 0.02));
    CHECK(edmp_scene_set(ctx, oc, no, clr, exp_, G, T, he, dh, sf));
    CHECK(edmp_rows_set(ctx, rc, method, gn, sched, B, T));

    double *d_noise = NULL, *d_X = NULL;
    float* d_vol = NULL;
    HIPCHECK(hipMalloc((void**)&d_noise, (size_t)(T + 1) * n * 8));
    HIPCHECK(hipMalloc((void**)&d_X, n * 8));
    HIPCHECK(hipMalloc((void**)&d_vol, (size_t)B * 4));
    HIPCHECK(hipMemcpy(d_noise, noise, (size_t)(T + 1) * n * 8, hipMemcpyHostToDevice));
    CHECK(edmp_denoise_guided_dev(ctx, d_noise, B, start, goal, 1, t_stop, 1, d_X));
    int best = -1;
    CHECK(edmp_row_swept_volumes_dev(ctx, d_X, B, 50, start, goal, d_vol, &best));
    CHECK(edmp_ctx_synchronize(ctx));
    double* X = malloc(n * 8);
    float* vol = malloc((size_t)B * 4);
    HIPCHECK(hipMemcpy(X, d_X, n * 8, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(vol, d_vol, (size_t)B * 4, hipMemcpyDeviceToHost));
    FILE* o = fopen(argv[2], "wb");
    fwrite(X, 8, n, o);
    int32_t b32 = best;
    fwrite(&b32, 4, 1, o);
    fwrite(vol, 4, B, o);
    /* plan success of every row (edmp_success_rows_dev): the last obstacle marked as a true cylinder; flags + counts appended */
    {
        int32_t* kinds = calloc((size_t)no, 4);
        kinds[no - 1] = 1;
        CHECK(edmp_scene_set_shapes(ctx, kinds, no));
        int32_t *d_flags = NULL, counts[4] = {0, 0, 0, 0};
        HIPCHECK(hipMalloc((void**)&d_flags, (size_t)3 * B * 4));
        CHECK(edmp_success_rows_dev(ctx, d_X, B, 50, 4, NULL, d_flags, d_flags + B, d_flags + 2 * B, counts));
        int32_t* flags = malloc((size_t)3 * B * 4);
        HIPCHECK(hipMemcpy(flags, d_flags, (size_t)3 * B * 4, hipMemcpyDeviceToHost));
        fwrite(flags, 4, (size_t)3 * B, o);
        fwrite(counts, 4, 4, o);
        if (edmp_scene_set_shapes(ctx, kinds, no + 1) == 0) return 8; /* wrong length is refused */
    }
    fclose(o);
    /* error behaviour: bad arguments are reported, not crashed on */
    if (edmp_unet_forward_dev(ctx, NULL, 1, 1, NULL) == 0) return 6;
    if (strlen(edmp_last_error()) == 0) return 7;
    edmp_ctx_destroy(ctx);
    printf("c_abi_smoke ok: B=%d best=%d\n", B, best);
    return 0;
}
