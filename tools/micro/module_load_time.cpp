// How long does the HIP runtime take to load a code object of a given size?  (round 6: `fmsi query`'s load runs its first kernels out of
// build_index.hip's code object -- 6.7 MB, most of it rocPRIM's sorts for the index BUILDER -- and the load's "payloads read and copied"
// phase is 0.10 s where the reads and the copies are 0.05.)
//   hipcc tools/micro/module_load_time.cpp -o /tmp/mlt && /tmp/mlt a.hsaco b.hsaco ...
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <fstream>
#include <iterator>
#include <vector>

int main(int argc, char** argv) {
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    if (hipInit(0) != hipSuccess || hipSetDevice(0) != hipSuccess) return 1;
    void* p = nullptr;
    if (hipMalloc(&p, 1 << 20) != hipSuccess) return 1;
    printf("hipInit + first hipMalloc: %.3f s\n", now() - t0);
    for (int rep = 0; rep < 2; ++rep)
        for (int i = 1; i < argc; ++i) {
            std::ifstream f(argv[i], std::ios::binary);
            std::vector<char> image((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
            hipModule_t m;
            t0 = now();
            const hipError_t e = hipModuleLoadData(&m, image.data());
            const double dt = now() - t0;
            printf("%-28s %9zu bytes: hipModuleLoadData %.4f s (%s)\n", argv[i], image.size(), dt, hipGetErrorString(e));
            if (e == hipSuccess) (void)hipModuleUnload(m);
        }
    return 0;
}
