// What do hipMalloc / hipMemset / hipFree of large buffers cost on this box?  (the index load is mostly these)
//   hipcc --offload-arch=gfx950 -O2 tools/alloc_bench.hip -o tools/alloc_bench && ./tools/alloc_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void one(const char* tag, double gb) {
    const size_t bytes = static_cast<size_t>(gb * (1ull << 30));
    void* p = nullptr;
    double t0 = now();
    if (hipMalloc(&p, bytes) != hipSuccess) { printf("%s malloc %g GiB failed\n", tag, gb); return; }
    double t1 = now();
    (void)hipMemset(p, 0, bytes);
    (void)hipDeviceSynchronize();
    double t2 = now();
    (void)hipFree(p);
    double t3 = now();
    printf("%-28s %5.1f GiB: malloc %.3f s  memset %.3f s  free %.3f s\n", tag, gb, t1 - t0, t2 - t1, t3 - t2);
}
int main() {
    (void)hipFree(nullptr);
    for (double gb : {1.0, 16.0, 32.0, 33.0, 40.0, 48.0, 64.0}) one("single", gb);
    for (double gb : {32.0, 40.0}) one("single again", gb);
    // many allocations alive at once, touched, freed, and allocated again (the builder's pattern)
    for (int round = 0; round < 3; ++round) {
        std::vector<void*> live;
        double t0 = now();
        for (int i = 0; i < 8; ++i) {
            void* p = nullptr;
            if (hipMalloc(&p, 22ull << 30) == hipSuccess) {
                live.push_back(p);
                (void)hipMemset(p, 0, 22ull << 30);
            }
        }
        (void)hipDeviceSynchronize();
        double t1 = now();
        for (void* p : live) (void)hipFree(p);
        double t2 = now();
        printf("round %d: %zu x 22 GiB: malloc+memset %.3f s, free %.3f s\n", round, live.size(), t1 - t0, t2 - t1);
        if (round == 1) std::this_thread::sleep_for(std::chrono::seconds(3));
    }
    one("after the rounds", 30.0);
    return 0;
}
