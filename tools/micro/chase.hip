// Random-access latency of HBM as a kernel sees it: one wave, lane 0 chases pointers through a buffer of `bytes` (a random cycle of
// 128-byte lines); and the same with 64 independent chases per wave (one per lane).   hipcc --offload-arch=gfx950 -O3 chase.hip -o chase
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void init_kernel(uint64_t* buf, uint64_t lines, uint64_t mult, uint64_t add) {   // line i -> line (i * mult + add) mod lines: one cycle when
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;                      // lines is a power of two, mult = 1 mod 4, add odd
    if (i < lines) buf[i * 16] = ((i * mult + add) & (lines - 1)) * 16;
}
__global__ void chase_kernel(const uint64_t* __restrict__ buf, uint64_t steps, uint64_t lanes, uint64_t lines, uint64_t* out) {
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if ((threadIdx.x & 63) >= lanes) return;
    uint64_t p = ((t * 0x9E3779B97F4A7C15ull) & (lines - 1)) * 16;
    for (uint64_t s = 0; s < steps; ++s) p = __builtin_nontemporal_load(buf + p);
    out[t] = p;
}
int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 16.0;
    uint64_t lines = 1;
    while (lines * 128 * 2 <= gb * 1e9) lines *= 2;
    uint64_t *buf, *out;
    CK(hipMalloc(&buf, lines * 128));
    CK(hipMalloc(&out, 1 << 24));
    hipLaunchKernelGGL(init_kernel, dim3((lines + 255) / 256), dim3(256), 0, 0, buf, lines, 0x5851F42D4C957F2Dull, 0x14057B7EF767814Full | 1ull);
    CK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    struct Case { uint32_t blocks, threads; uint64_t lanes, steps; const char* what; };
    const Case cases[] = {{1, 64, 1, 20000, "1 wave, 1 lane"}, {1, 64, 64, 20000, "1 wave, 64 lanes"}, {256, 64, 64, 20000, "256 waves x 64"},
                          {256 * 4, 256, 64, 4000, "4096 waves x 64"}, {256 * 8, 512, 64, 2000, "16384 waves x 64"}};
    for (const Case& c : cases) {
        hipLaunchKernelGGL(chase_kernel, dim3(c.blocks), dim3(c.threads), 0, 0, buf, 200, c.lanes, lines, out);   // warm
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(chase_kernel, dim3(c.blocks), dim3(c.threads), 0, 0, buf, c.steps, c.lanes, lines, out);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        const double chains = (double)c.blocks * c.threads / 64 * c.lanes;
        printf("%.1f GB  %-18s  %.0f ns per step   %.2f G requests/s\n", lines * 128 / 1e9, c.what, ms * 1e6 / c.steps, chains * c.steps / ms / 1e6);
    }
    return 0;
}
