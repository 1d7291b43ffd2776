#include "devpool.h"

#include <atomic>
#include <cstdint>
#include <map>
#include <mutex>
#include <unordered_map>

namespace fmsi {
namespace {

constexpr size_t kMinCached = size_t(64) << 20;
constexpr int kMaxDevices = 16;

struct Live {
    size_t bytes;
    int device;   // the device the block was allocated on: where it is filed when freed, whatever the thread's current device is then
};
struct Pool {
    std::mutex m;
    std::unordered_map<void*, Live> live;                    // blocks handed out (>= kMinCached only)
    struct Parked {
        void* p;
        uint64_t seq;   // the device's free count when the block was parked
    };
    std::multimap<size_t, Parked> cached[kMaxDevices];       // per device: size -> freed block
    size_t cached_bytes[kMaxDevices] = {};
    // hipFree waits for the device before it releases memory, and a parked block must offer its next owner the same: work enqueued on
    // the first owner's stream may still read or write it.  The wait is taken when a block is handed out AGAIN -- once for every block
    // parked until then -- not with every free (rounds 2-4: a device-wide wait per pooled free, whether or not anybody took the block)
    uint64_t free_seq[kMaxDevices] = {}, waited_seq[kMaxDevices] = {};
};
Pool& pool() {
    static Pool* p = new Pool;   // (never destroyed: handles may be freed during process exit)
    return *p;
}
std::atomic<int> g_keep{0};
thread_local int t_scope_depth = 0;

int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return d < 0 || d >= kMaxDevices ? 0 : d;
}

// runs fn with `device` current and puts the thread back where it was
template <class F>
void on_device(int device, F&& fn) {
    const int prev = current_device();
    if (prev != device && hipSetDevice(device) != hipSuccess) (void)hipGetLastError();
    fn();
    if (prev != device && hipSetDevice(prev) != hipSuccess) (void)hipGetLastError();
}

size_t trim_locked(Pool& P, int d) {
    size_t freed = 0;
    if (P.cached[d].empty()) return 0;
    on_device(d, [&] {
        for (auto& e : P.cached[d]) {   // (hipFree waits for the device itself)
            (void)hipFree(e.second.p);
            freed += e.first;
        }
    });
    P.cached[d].clear();
    P.cached_bytes[d] = 0;
    return freed;
}

}  // namespace

hipError_t dev_malloc(void** p, size_t bytes) {
    Pool& P = pool();
    const int d = current_device();
    if (bytes < kMinCached) {
        hipError_t e = hipMalloc(p, bytes ? bytes : 8);
        if (e != hipSuccess) {   // the cache may be holding the room a small allocation needs, too
            std::lock_guard<std::mutex> lk(P.m);
            if (!P.cached[d].empty()) {
                (void)hipGetLastError();
                trim_locked(P, d);
                e = hipMalloc(p, bytes ? bytes : 8);
            }
        }
        return e;
    }
    std::lock_guard<std::mutex> lk(P.m);
    auto it = P.cached[d].lower_bound(bytes);
    if (it != P.cached[d].end() && it->first <= bytes + bytes / 4) {   // (best fit, at most a quarter more than asked for)
        const Pool::Parked blk = it->second;
        const size_t blk_bytes = it->first;
        P.cached_bytes[d] -= blk_bytes;
        P.cached[d].erase(it);
        hipError_t w = hipSuccess;
        if (blk.seq > P.waited_seq[d]) {   // parked since the device was last waited for: whatever its first owner enqueued must be over
            const uint64_t upto = P.free_seq[d];
            on_device(d, [&] { w = hipDeviceSynchronize(); });
            if (w == hipSuccess) P.waited_seq[d] = upto;
        }
        if (w == hipSuccess) {
            *p = blk.p;
            P.live[*p] = Live{blk_bytes, d};
            return hipSuccess;
        }
        (void)hipGetLastError();   // a device that cannot be waited for: the block goes back to the driver, not to another owner
        on_device(d, [&] { (void)hipFree(blk.p); });
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess && !P.cached[d].empty()) {   // the cache may hold what this allocation needs
        (void)hipGetLastError();
        trim_locked(P, d);
        e = hipMalloc(p, bytes);
    }
    if (e == hipSuccess) P.live[*p] = Live{bytes, d};
    return e;
}

hipError_t dev_free(void* p) {
    if (!p) return hipSuccess;
    Pool& P = pool();
    Live l{0, -1};
    {
        std::lock_guard<std::mutex> lk(P.m);
        auto it = P.live.find(p);
        if (it != P.live.end()) {
            l = it->second;
            P.live.erase(it);
        }
    }
    if (l.device < 0) return hipFree(p);
    std::lock_guard<std::mutex> lk(P.m);   // (parked as it is: dev_malloc waits for the device before anybody gets it -- the Pool's comment)
    P.cached[l.device].emplace(l.bytes, Pool::Parked{p, ++P.free_seq[l.device]});
    P.cached_bytes[l.device] += l.bytes;
    return hipSuccess;
}

size_t dev_trim() {
    Pool& P = pool();
    std::lock_guard<std::mutex> lk(P.m);
    return trim_locked(P, current_device());
}

void dev_settle() {
    Pool& P = pool();
    const int d = current_device();
    std::lock_guard<std::mutex> lk(P.m);
    if (P.free_seq[d] == P.waited_seq[d] || P.cached[d].empty()) return;
    const uint64_t upto = P.free_seq[d];
    if (hipDeviceSynchronize() == hipSuccess) P.waited_seq[d] = upto;
    else (void)hipGetLastError();
}

size_t dev_cached_bytes() {
    Pool& P = pool();
    std::lock_guard<std::mutex> lk(P.m);
    return P.cached_bytes[current_device()];
}

void dev_keep_scratch(bool on) { g_keep.store(on ? 1 : 0); }
bool dev_keeps_scratch() { return g_keep.load() != 0; }

ScratchScope::ScratchScope(int device) : device_(device) { ++t_scope_depth; }
ScratchScope::~ScratchScope() {
    if (--t_scope_depth > 0 || g_keep.load()) return;   // (a builder inside a builder: the outermost one settles)
    if (device_ < 0 || device_ >= kMaxDevices) return;
    Pool& P = pool();
    std::lock_guard<std::mutex> lk(P.m);
    trim_locked(P, device_);
}

}  // namespace fmsi
