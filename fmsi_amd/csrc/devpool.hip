#include "devpool.h"

#include <map>
#include <mutex>
#include <unordered_map>

namespace fmsi {
namespace {

constexpr size_t kMinCached = size_t(64) << 20;

struct Pool {
    std::mutex m;
    std::unordered_map<void*, size_t> live;          // blocks handed out (>= kMinCached only)
    std::multimap<size_t, void*> cached[16];         // per device: size -> freed block
    size_t cached_bytes[16] = {};
};
Pool& pool() {
    static Pool* p = new Pool;   // (never destroyed: handles may be freed during process exit)
    return *p;
}
int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return d < 0 || d >= 16 ? 0 : d;
}

size_t trim_locked(Pool& P, int d) {
    size_t freed = 0;
    for (auto& e : P.cached[d]) {
        (void)hipFree(e.second);
        freed += e.first;
    }
    P.cached[d].clear();
    P.cached_bytes[d] = 0;
    return freed;
}

}  // namespace

hipError_t dev_malloc(void** p, size_t bytes) {
    if (bytes < kMinCached) return hipMalloc(p, bytes ? bytes : 8);
    Pool& P = pool();
    const int d = current_device();
    std::lock_guard<std::mutex> lk(P.m);
    auto it = P.cached[d].lower_bound(bytes);
    if (it != P.cached[d].end() && it->first <= bytes + bytes / 4) {   // (best fit, at most a quarter more than asked for)
        *p = it->second;
        P.live[*p] = it->first;
        P.cached_bytes[d] -= it->first;
        P.cached[d].erase(it);
        return hipSuccess;
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess && !P.cached[d].empty()) {   // the cache may hold what this allocation needs
        (void)hipGetLastError();
        trim_locked(P, d);
        e = hipMalloc(p, bytes);
    }
    if (e == hipSuccess) P.live[*p] = bytes;
    return e;
}

hipError_t dev_free(void* p) {
    if (!p) return hipSuccess;
    Pool& P = pool();
    {
        std::lock_guard<std::mutex> lk(P.m);
        auto it = P.live.find(p);
        if (it != P.live.end()) {
            const int d = current_device();
            P.cached[d].emplace(it->second, p);
            P.cached_bytes[d] += it->second;
            P.live.erase(it);
            return hipSuccess;
        }
    }
    return hipFree(p);
}

size_t dev_trim() {
    Pool& P = pool();
    std::lock_guard<std::mutex> lk(P.m);
    return trim_locked(P, current_device());
}

size_t dev_cached_bytes() {
    Pool& P = pool();
    std::lock_guard<std::mutex> lk(P.m);
    return P.cached_bytes[current_device()];
}

}  // namespace fmsi
