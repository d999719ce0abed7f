// xcd_map_probe.hip — is "workgroup id i runs on XCD i mod 8" still true when several kernels are dispatched at the same time?
// (the assumption behind the persistent layer chain's L2-local hand-over, edmp_amd/csrc/chain.hip, and behind xcd_split's traffic model)
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_map_probe.hip -o tools/xcd_map_probe && tools/xcd_map_probe
// Each launch: 256 workgroups x 256 threads that spin ~20 us and record HW_REG_XCC_ID; one stream alone, then 2 / 4 streams launching
// back to back.  Printed per case: launches whose map is exactly (i + c) mod 8 for one c, and workgroups off that pattern.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void probe(int* out, int spin) {
    const long long t0 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf);
    while (clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
}

int main() {
    const int NS = 4, NL = 50;
    hipStream_t st[NS];
    for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int* d;
    hipMalloc((void**)&d, NS * NL * 256 * sizeof(int));
    std::vector<int> h(NS * NL * 256);
    for (int ns : {1, 2, 4}) {
        hipMemset(d, 0xff, NS * NL * 256 * sizeof(int));
        hipDeviceSynchronize();
        for (int l = 0; l < NL; ++l)
            for (int s = 0; s < ns; ++s) hipLaunchKernelGGL(probe, dim3(256), dim3(256), 0, st[s], d + (s * NL + l) * 256, 48000);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost);
        int clean = 0, launches = 0;
        long off = 0;
        for (int s = 0; s < ns; ++s)
            for (int l = 0; l < NL; ++l) {
                const int* m = &h[(s * NL + l) * 256];
                // best rotation
                int best = 256;
                for (int c = 0; c < 8; ++c) {
                    int perm[8];
                    for (int k = 0; k < 8; ++k) perm[k] = m[(k + 8 - c) % 8 + 0];  // xcc of the first workgroup of each residue class under rotation c
                    int bad = 0;
                    for (int i = 0; i < 256; ++i) bad += m[i] != m[i & 7];
                    best = bad < best ? bad : best;
                    (void)perm;
                }
                ++launches;
                clean += best == 0;
                off += best;
            }
        printf("%d stream(s): %d of %d launches have workgroup i on the XCD of workgroup (i mod 8); %ld workgroups off that pattern\n", ns, clean, launches, off);
    }
    return 0;
}
