// How fast can 1 GB of formatted output reach a file on this host?  (round 6: `write()` is the longest stage of the `fmsi query` process --
// 0.16-0.19 s per 10^9 windows on tmpfs -- and round 2 dropped the alternatives after measuring them once, with fewer threads.)
//   g++ -O2 -pthread tools/micro/write_bench.cpp -o tools/micro/write_bench && tools/micro/write_bench /dev/shm/wb.out [GB]
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <fcntl.h>
#include <linux/falloc.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "/dev/shm/wb.out";
    const size_t total = static_cast<size_t>((argc > 2 ? atof(argv[2]) : 1.0) * (1ull << 30));
    const size_t piece = 8u << 20;
    std::vector<char> src(piece);
    for (size_t i = 0; i < piece; ++i) src[i] = "01"[i % 7 == 0];
    auto fresh = [&] {
        unlink(path);
        const int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0644);
        if (fd < 0) { perror("open"); exit(1); }
        return fd;
    };
    for (int rep = 0; rep < 2; ++rep) {
        {   // 1. one thread, write()
            const int fd = fresh();
            const double t0 = now();
            for (size_t at = 0; at < total; at += piece)
                if (write(fd, src.data(), piece) != static_cast<ssize_t>(piece)) { perror("write"); return 1; }
            printf("write(), one thread:                         %.3f s\n", now() - t0);
            close(fd);
        }
        {   // 5. fallocate, then write()
            const int fd = fresh();
            double t0 = now();
            if (posix_fallocate(fd, 0, static_cast<off_t>(total)) != 0) perror("fallocate");
            const double tf = now() - t0;
            t0 = now();
            for (size_t at = 0; at < total; at += piece)
                if (write(fd, src.data(), piece) != static_cast<ssize_t>(piece)) { perror("write"); return 1; }
            printf("fallocate %.3f s, then write(), one thread:   %.3f s\n", tf, now() - t0);
            close(fd);
        }
        {   // 6. fallocate(KEEP_SIZE) -- pages beyond the end of the file, its size unchanged -- then write()
            const int fd = fresh();
            double t0 = now();
            if (fallocate(fd, FALLOC_FL_KEEP_SIZE, 0, static_cast<off_t>(total)) != 0) perror("fallocate keep_size");
            const double tf = now() - t0;
            t0 = now();
            for (size_t at = 0; at < total; at += piece)
                if (write(fd, src.data(), piece) != static_cast<ssize_t>(piece)) { perror("write"); return 1; }
            printf("fallocate(KEEP_SIZE) %.3f s, then write():     %.3f s\n", tf, now() - t0);
            close(fd);
        }
        {   // 7. a second thread preallocates 64 MB pieces AHEAD of the writer while it writes (both take the inode's lock)
            const int fd = fresh();
            const double t0 = now();
            std::thread pre([&] {
                for (size_t at = 0; at < total; at += (64u << 20))
                    if (fallocate(fd, FALLOC_FL_KEEP_SIZE, static_cast<off_t>(at), 64 << 20) != 0) break;
            });
            for (size_t at = 0; at < total; at += piece)
                if (write(fd, src.data(), piece) != static_cast<ssize_t>(piece)) { perror("write"); return 1; }
            const double tw = now() - t0;
            pre.join();
            printf("write() beside a preallocating thread:        %.3f s\n", tw);
            close(fd);
        }
        for (unsigned T : {2u, 4u, 8u, 16u}) {
            {   // 2. pwrite() from T threads
                const int fd = fresh();
                const double t0 = now();
                std::vector<std::thread> th;
                for (unsigned t = 0; t < T; ++t)
                    th.emplace_back([&, t] {
                        for (size_t at = t * piece; at < total; at += T * piece)
                            if (pwrite(fd, src.data(), piece, static_cast<off_t>(at)) != static_cast<ssize_t>(piece)) perror("pwrite");
                    });
                for (auto& x : th) x.join();
                printf("pwrite(), %2u threads:                        %.3f s\n", T, now() - t0);
                close(fd);
            }
            for (int populate = 0; populate < 2; ++populate) {   // 3 / 4. a shared mapping, T threads copy into it
                const int fd = fresh();
                const double t0 = now();
                if (ftruncate(fd, static_cast<off_t>(total)) != 0) perror("ftruncate");
                char* m = static_cast<char*>(mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
                if (m == MAP_FAILED) { perror("mmap"); return 1; }
                std::vector<std::thread> th;
                for (unsigned t = 0; t < T; ++t)
                    th.emplace_back([&, t] {
                        for (size_t at = t * piece; at < total; at += T * piece) {
                            if (populate && madvise(m + at, piece, MADV_POPULATE_WRITE) != 0) { /* older kernel: the copy faults the pages in */ }
                            std::memcpy(m + at, src.data(), piece);
                        }
                    });
                for (auto& x : th) x.join();
                const double tc = now() - t0;
                munmap(m, total);
                printf("mmap + memcpy%s, %2u threads:       %.3f s (+ munmap %.3f s)\n", populate ? " + POPULATE_WRITE" : "                 ", T, tc, now() - t0 - tc);
                close(fd);
            }
        }
        printf("\n");
    }
    unlink(path);
    return 0;
}
