// Microbenchmark: chip-wide rate of RANDOM 64-byte (and 128-byte) reads from a buffer larger than the
// Infinity Cache -- the honest ceiling for the fmsi query kernels, whose every LF step is one such read
// ("the gather ceiling is the honest one").  Each lane owns a chain of dependent or
// independent reads of whole blocks (4 x dwordx4 per 64 B), like the query kernel.
//   build: hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o gpurun_out/gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}

// GRAN = bytes per read (64 or 128); ILP = independent chains per lane; dependent: next address depends on loaded data
template <int GRAN, int ILP, bool DEP>
__global__ __launch_bounds__(256) void gather(const uint4* __restrict__ buf, uint64_t n_blocks, int iters, uint64_t* out) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t state[ILP];
    uint32_t acc = 0;
    for (int c = 0; c < ILP; ++c) state[c] = mix(tid * ILP + c);
    for (int it = 0; it < iters; ++it) {
        uint4 v[ILP][GRAN / 16];
#pragma unroll
        for (int c = 0; c < ILP; ++c) {
            const uint4* p = buf + (state[c] % n_blocks) * (GRAN / 16);
#pragma unroll
            for (int q = 0; q < GRAN / 16; ++q) v[c][q] = p[q];
        }
#pragma unroll
        for (int c = 0; c < ILP; ++c) {
            uint32_t s = 0;
#pragma unroll
            for (int q = 0; q < GRAN / 16; ++q) s += v[c][q].x ^ v[c][q].y ^ v[c][q].z ^ v[c][q].w;
            acc += s;
            state[c] = mix(state[c] + (DEP ? s : 0u));
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int GRAN, int ILP, bool DEP>
void run(const uint4* buf, uint64_t bytes, uint64_t* out, int waves_per_simd) {
    const uint64_t n_blocks = bytes / GRAN;
    const int iters = 64;
    const int blocks = 256 * waves_per_simd * 4;   // 256 CUs x (waves/SIMD) workgroups of 4 waves... x4 oversubscription
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    gather<GRAN, ILP, DEP><<<blocks, 256>>>(buf, n_blocks, 4, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    gather<GRAN, ILP, DEP><<<blocks, 256>>>(buf, n_blocks, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double reads = (double)blocks * 256 * ILP * iters;
    printf("buf %.2f GB  gran %3d B  ilp %d  dep %d  grid %d : %.3f ms  %.2f G reads/s  %.2f TB/s\n", bytes / 1e9, GRAN, ILP, (int)DEP, blocks, ms,
           reads / ms / 1e6, reads * GRAN / ms / 1e9);
}

int main(int argc, char** argv) {
    // usage: gather_bench [GB ...]   (default sweep)
    std::vector<double> sizes;
    for (int i = 1; i < argc; ++i) sizes.push_back(atof(argv[i]));
    if (sizes.empty()) sizes = {0.25, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 8.0, 16.0};
    double mx = 0;
    for (double g : sizes) mx = g > mx ? g : mx;
    uint4* buf; uint64_t* out;
    hipMalloc(&buf, (uint64_t)(mx * 1e9) + 4096); hipMalloc(&out, 8);
    hipMemset(buf, 1, (uint64_t)(mx * 1e9) + 4096);
    for (double gb : sizes) {
        const uint64_t bytes = (uint64_t)(gb * 1e9) / 512 * 512;
        run<64, 1, true>(buf, bytes, out, 8);
        run<128, 1, true>(buf, bytes, out, 8);
        run<256, 1, true>(buf, bytes, out, 8);
        run<512, 1, true>(buf, bytes, out, 4);
    }
    return 0;
}
