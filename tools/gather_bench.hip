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
#include <string>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}

// GRAN = bytes per read (64 or 128); ILP = independent chains per lane; dependent: next address depends on loaded data
template <int GRAN, int ILP, bool DEP>
__global__ __launch_bounds__(256) void gather(const uint4* __restrict__ buf, uint64_t n_blocks, int iters, uint64_t* out) {
    extern __shared__ uint32_t occupancy_limiter[];   // dynamic LDS only limits the resident workgroups per CU
    if (iters < 0) occupancy_limiter[threadIdx.x] = 1;
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

// COOP lanes share one chain: each lane reads GRAN/COOP bytes of the block (consecutive 16-B pieces), so one
// wave instruction touches 64/COOP distinct lines instead of 64.
template <int GRAN, int ILP, int COOP, bool NT = false>
__global__ __launch_bounds__(256) void gather_coop(const uint4* __restrict__ buf, uint64_t n_blocks, int iters, uint64_t* out) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t sub = threadIdx.x % COOP;
    constexpr int PIECES = GRAN / 16 / COOP;
    uint64_t state[ILP];
    uint32_t acc = 0;
    for (int c = 0; c < ILP; ++c) state[c] = mix((tid / COOP) * ILP + c);
    for (int it = 0; it < iters; ++it) {
        uint4 v[ILP][PIECES];
#pragma unroll
        for (int c = 0; c < ILP; ++c) {
            const uint4* p = buf + (state[c] % n_blocks) * (GRAN / 16) + sub * PIECES;
#pragma unroll
            for (int q = 0; q < PIECES; ++q) {
                if (NT) {   // non-temporal: the line is not wanted in the caches again
                    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                    const u4 t = __builtin_nontemporal_load(reinterpret_cast<const u4*>(p + q));
                    v[c][q] = make_uint4(t.x, t.y, t.z, t.w);
                } else {
                    v[c][q] = p[q];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < ILP; ++c) {
            uint32_t s = 0;
#pragma unroll
            for (int q = 0; q < PIECES; ++q) s += v[c][q].x ^ v[c][q].y ^ v[c][q].z ^ v[c][q].w;
#pragma unroll
            for (int m = 1; m < COOP; m <<= 1) s += __shfl_xor(s, m, 64);
            acc += s;
            state[c] = mix(state[c] + s);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int GRAN, int ILP, int COOP, bool NT = false>
void run_coop(const uint4* buf, uint64_t bytes, uint64_t* out, int blocks = 8192, int iters = 64) {
    const uint64_t n_blocks = bytes / GRAN;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    gather_coop<GRAN, ILP, COOP, NT><<<blocks, 256>>>(buf, n_blocks, 4, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    gather_coop<GRAN, ILP, COOP, NT><<<blocks, 256>>>(buf, n_blocks, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double reads = (double)blocks * 256 / COOP * ILP * iters;
    printf("buf %.2f GB  grid %5d  coop %d%s gran %3d B  ilp %d : %.3f ms  %.2f G reads/s  %.2f TB/s\n", bytes / 1e9, blocks, COOP, NT ? " nt" : "   ", GRAN, ILP, ms, reads / ms / 1e6,
           reads * GRAN / ms / 1e9);
}

template <int GRAN, int ILP, bool DEP>
void run(const uint4* buf, uint64_t bytes, uint64_t* out, int waves_per_simd, int resident_waves_per_simd = 8) {
    const uint64_t n_blocks = bytes / GRAN;
    const int iters = 64;
    const int blocks = 256 * waves_per_simd * 4;   // 256 CUs x (waves/SIMD) workgroups of 4 waves... x4 oversubscription
    // a 256-thread workgroup puts one wave on each SIMD: W resident workgroups per CU = W waves/SIMD; 160 KiB LDS per CU
    const size_t lds = resident_waves_per_simd >= 8 ? 0 : (size_t)(160 * 1024 / resident_waves_per_simd) / 1024 * 1024;
    if (lds) hipFuncSetAttribute((const void*)gather<GRAN, ILP, DEP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    gather<GRAN, ILP, DEP><<<blocks, 256, lds>>>(buf, n_blocks, 4, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    gather<GRAN, ILP, DEP><<<blocks, 256, lds>>>(buf, n_blocks, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double reads = (double)blocks * 256 * ILP * iters;
    printf("buf %.2f GB  occ %d  gran %3d B  ilp %d  dep %d  grid %d : %.3f ms  %.2f G reads/s  %.2f TB/s\n", bytes / 1e9, resident_waves_per_simd, GRAN, ILP, (int)DEP, blocks, ms,
           reads / ms / 1e6, reads * GRAN / ms / 1e9);
}

int main(int argc, char** argv) {
    // usage: gather_bench [GB ...]   (default sweep)   |   gather_bench occ GB : occupancy sweep at one size
    if (argc > 2 && std::string(argv[1]) == "coop") {
        const double gb = atof(argv[2]);
        const uint64_t bytes = (uint64_t)(gb * 1e9) / 512 * 512;
        uint4* buf; uint64_t* out;
        hipMalloc(&buf, bytes + 4096); hipMalloc(&out, 8);
        hipMemset(buf, 1, bytes + 4096);
        run_coop<64, 1, 1>(buf, bytes, out);  run_coop<64, 2, 1>(buf, bytes, out);  run_coop<64, 4, 1>(buf, bytes, out);
        run_coop<64, 1, 4>(buf, bytes, out);  run_coop<64, 2, 4>(buf, bytes, out);  run_coop<64, 4, 4>(buf, bytes, out);  run_coop<64, 8, 4>(buf, bytes, out);
        run_coop<128, 1, 1>(buf, bytes, out); run_coop<128, 2, 1>(buf, bytes, out);
        run_coop<128, 1, 2>(buf, bytes, out); run_coop<128, 2, 2>(buf, bytes, out); run_coop<128, 4, 2>(buf, bytes, out);
        run_coop<128, 1, 4>(buf, bytes, out); run_coop<128, 2, 4>(buf, bytes, out); run_coop<128, 4, 4>(buf, bytes, out); run_coop<128, 8, 4>(buf, bytes, out);
        run_coop<128, 1, 8>(buf, bytes, out); run_coop<128, 2, 8>(buf, bytes, out); run_coop<128, 4, 8>(buf, bytes, out); run_coop<128, 8, 8>(buf, bytes, out);
        return 0;
    }
    if (argc > 2 && std::string(argv[1]) == "entry") {
        // table lookups of the phased path: one 16-byte entry per key out of a table of `GB`; who loads what?
        //   <16,*,1> one lane per entry (64 lines per instruction)        <64,*,2> lane pair, 32 B each = the whole 64-B line (2 instr)
        //   <32,*,2> lane pair, 16 B each = the entry's 32-B half line      <64,*,4> lane quad, 16 B each = the whole line (1 instr)
        const double gb = atof(argv[2]);
        const uint64_t bytes = (uint64_t)(gb * 1e9) / 512 * 512;
        uint4* buf; uint64_t* out;
        if (hipMalloc(&buf, bytes + 4096) != hipSuccess) { printf("cannot allocate %.1f GB\n", gb); return 1; }
        hipMalloc(&out, 8);
        hipMemset(buf, 1, bytes + 4096);
        run_coop<16, 1, 1>(buf, bytes, out); run_coop<16, 2, 1>(buf, bytes, out); run_coop<16, 4, 1>(buf, bytes, out);
        run_coop<64, 1, 2>(buf, bytes, out); run_coop<64, 2, 2>(buf, bytes, out);
        run_coop<32, 1, 2>(buf, bytes, out); run_coop<32, 2, 2>(buf, bytes, out); run_coop<32, 4, 2>(buf, bytes, out);
        run_coop<64, 1, 4>(buf, bytes, out); run_coop<64, 2, 4>(buf, bytes, out); run_coop<64, 4, 4>(buf, bytes, out);
        run_coop<128, 1, 8>(buf, bytes, out); run_coop<128, 2, 8>(buf, bytes, out);
        // non-temporal loads
        run_coop<16, 1, 1, true>(buf, bytes, out); run_coop<16, 2, 1, true>(buf, bytes, out);
        run_coop<32, 1, 2, true>(buf, bytes, out); run_coop<32, 2, 2, true>(buf, bytes, out);
        run_coop<128, 2, 8, true>(buf, bytes, out);
        // one lane reading a 32-byte (64-byte) entry with two (four) loads of its own
        run_coop<32, 1, 1>(buf, bytes, out); run_coop<32, 2, 1>(buf, bytes, out);
        run_coop<64, 1, 1>(buf, bytes, out); run_coop<64, 2, 1>(buf, bytes, out);
        return 0;
    }
    if (argc > 2 && std::string(argv[1]) == "inflight") {   // resident waves per SIMD = grid / 256 (all workgroups resident)
        const double gb = atof(argv[2]);
        const uint64_t bytes = (uint64_t)(gb * 1e9) / 512 * 512;
        uint4* buf; uint64_t* out;
        hipMalloc(&buf, bytes + 4096); hipMalloc(&out, 8);
        hipMemset(buf, 1, bytes + 4096);
        for (int w : {1, 2, 3, 4, 5, 6, 8}) {
            run_coop<64, 1, 1>(buf, bytes, out, 256 * w, 2048);
            run_coop<128, 1, 2>(buf, bytes, out, 256 * w, 2048);
            run_coop<128, 2, 2>(buf, bytes, out, 256 * w, 1024);
        }
        return 0;
    }
    if (argc > 2 && std::string(argv[1]) == "occ") {
        const double gb = atof(argv[2]);
        const uint64_t bytes = (uint64_t)(gb * 1e9) / 512 * 512;
        uint4* buf; uint64_t* out;
        hipMalloc(&buf, bytes + 4096); hipMalloc(&out, 8);
        hipMemset(buf, 1, bytes + 4096);
        for (int occ : {8, 6, 5, 4, 3, 2, 1}) {
            run<64, 1, true>(buf, bytes, out, 8, occ);
            run<64, 2, true>(buf, bytes, out, 8, occ);
            run<128, 1, true>(buf, bytes, out, 8, occ);
            run<128, 2, true>(buf, bytes, out, 8, occ);
        }
        return 0;
    }
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
