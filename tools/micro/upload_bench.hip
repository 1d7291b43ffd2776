// How fast can a file in the page cache reach HBM?  (the index load's upload, build_index.hip upload_files)
//   hipcc --offload-arch=gfx950 -O2 tools/micro/upload_bench.hip -o tools/micro/upload_bench -lpthread && tools/micro/upload_bench FILE
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
using clk = std::chrono::steady_clock;
static double since(clk::time_point t) { return std::chrono::duration<double>(clk::now() - t).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char** argv) {
    if (argc < 2) return 1;
    auto t0 = clk::now();
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    printf("hip init                 %.4f s\n", since(t0));
    int fd = open(argv[1], O_RDONLY);
    struct stat st; fstat(fd, &st);
    const size_t bytes = st.st_size;
    char* d = nullptr;
    t0 = clk::now();
    CK(hipMalloc(&d, bytes));
    printf("hipMalloc %.2f GB        %.4f s\n", bytes / 1e9, since(t0));
    for (size_t mb : {1, 4, 16, 64}) {
        void* p;
        t0 = clk::now();
        CK(hipHostMalloc(&p, mb << 20, hipHostMallocDefault));
        const double a = since(t0);
        t0 = clk::now();
        CK(hipHostFree(p));
        printf("hipHostMalloc %3zu MB      %.4f s   (free %.4f s)\n", mb, a, since(t0));
    }
    { hipStream_t s; t0 = clk::now(); CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); printf("stream create            %.4f s\n", since(t0)); CK(hipStreamDestroy(s)); }
    // (a) one copy from a populated mapping
    t0 = clk::now();
    void* m = mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
    printf("mmap + populate          %.4f s\n", since(t0));
    t0 = clk::now();
    CK(hipMemcpy(d, m, bytes, hipMemcpyHostToDevice));
    printf("hipMemcpy from mapping   %.4f s  = %.1f GB/s\n", since(t0), bytes / since(t0) / 1e9);
    munmap(m, bytes);
    // (b) threads: pread -> pinned slot -> async copy
    for (unsigned threads : {4u, 8u, 16u})
        for (size_t piece : {size_t(1) << 20, size_t(4) << 20}) {
            t0 = clk::now();
            std::atomic<size_t> next{0};
            const size_t n = (bytes + piece - 1) / piece;
            std::vector<double> setup(threads);
            auto work = [&](unsigned t) {
                auto ts = clk::now();
                CK(hipSetDevice(0));
                hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
                void* slot[2]; hipEvent_t ev[2]; bool busy[2] = {false, false};
                for (int i = 0; i < 2; ++i) { CK(hipHostMalloc(&slot[i], piece, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
                setup[t] = since(ts);
                unsigned turn = 0;
                for (size_t i; (i = next.fetch_add(1)) < n; turn ^= 1) {
                    if (busy[turn]) CK(hipEventSynchronize(ev[turn]));
                    const size_t off = i * piece, len = std::min(piece, bytes - off);
                    size_t got = 0;
                    while (got < len) { ssize_t r = pread(fd, (char*)slot[turn] + got, len - got, off + got); if (r <= 0) exit(2); got += r; }
                    CK(hipMemcpyAsync(d + off, slot[turn], len, hipMemcpyHostToDevice, s));
                    CK(hipEventRecord(ev[turn], s));
                    busy[turn] = true;
                }
                CK(hipStreamSynchronize(s));
                for (int i = 0; i < 2; ++i) { CK(hipHostFree(slot[i])); CK(hipEventDestroy(ev[i])); }
                CK(hipStreamDestroy(s));
            };
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < threads; ++t) pool.emplace_back(work, t);
            work(0);
            for (auto& th : pool) th.join();
            const double all = since(t0);
            double smax = 0; for (double x : setup) smax = std::max(smax, x);
            printf("threads %2u piece %zu MB   %.4f s  = %.1f GB/s   (slowest thread's set-up %.4f s)\n", threads, piece >> 20, all, bytes / all / 1e9, smax);
        }
    // (c) one big pinned buffer filled by threads, one copy
    {
        t0 = clk::now();
        void* p; CK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
        const double a = since(t0);
        auto t1 = clk::now();
        std::vector<std::thread> pool;
        const unsigned T = 16;
        for (unsigned t = 0; t < T; ++t) pool.emplace_back([&, t] { const size_t lo = bytes / T * t, hi = t + 1 == T ? bytes : bytes / T * (t + 1); size_t got = lo; while (got < hi) { ssize_t r = pread(fd, (char*)p + got, hi - got, got); if (r <= 0) exit(2); got += r; } });
        for (auto& th : pool) th.join();
        const double b = since(t1);
        t1 = clk::now();
        CK(hipMemcpy(d, p, bytes, hipMemcpyHostToDevice));
        printf("one pinned buffer: alloc %.4f s, 16-thread read %.4f s, copy %.4f s (%.1f GB/s)\n", a, b, since(t1), bytes / since(t1) / 1e9);
    }
    return 0;
}
