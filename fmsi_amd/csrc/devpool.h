// Device memory for the library: hipMalloc / hipFree behind a cache of large freed blocks.
//
// Why (tools/alloc_bench.hip): on this platform hipFree of a large buffer returns at once but the NEXT hipMallocs stall
// for about a second per 30-40 GB freed before them.  The index builder, the pair layout, the candidate tables and the dictionary's
// construction hand each other ~150 GB of temporaries of the same few sizes (8, 4 and 1 bytes per row), so freed blocks of 64 MB and more
// are kept and handed out again (best fit, at most a quarter more than the request).
//
// Rules (round 3, after the advisor's findings):
//   * a block is filed under the device it was ALLOCATED on, whatever the freeing thread's current device is;
//   * a block is parked only after that device has been waited for (what hipFree would have done): whoever gets it next -- another handle,
//     another stream -- cannot meet a kernel of the previous owner;
//   * every builder entry point (index build / load, pair layout, candidate tables, dictionary, export, file writer) holds a ScratchScope:
//     when the OUTERMOST one ends, the cache of its device goes back to the driver, so that what a construction left behind does not sit
//     invisibly on the device (fmsi_hip_index_info.device_bytes does not count it) and starve later small allocations or the host's own
//     allocator (torch).  A host that runs a CHAIN of builders -- bench.py, the CLI's load, the tests' fixtures -- says
//     fmsi_hip_keep_scratch(1) before and fmsi_hip_keep_scratch(0) + fmsi_hip_trim_scratch() after it, and the chain shares its blocks;
//   * any allocation that fails, large or small, returns the cache to the driver and is tried again.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>

namespace fmsi {

hipError_t dev_malloc(void** p, size_t bytes);
template <class T>
hipError_t dev_malloc(T** p, size_t bytes) { return dev_malloc(reinterpret_cast<void**>(p), bytes); }
hipError_t dev_free(void* p);
// returns the cached blocks of the current device to the driver; the bytes they held
size_t dev_trim();
size_t dev_cached_bytes();
// waits for the current device now if a block parked on it has not been waited for yet (the wait its next owner would take at the hand-out):
// for a caller about to enqueue long work that such a hand-out should not have to sit out
void dev_settle();
void dev_keep_scratch(bool on);
bool dev_keeps_scratch();

struct ScratchScope {
    explicit ScratchScope(int device);
    ~ScratchScope();
    ScratchScope(const ScratchScope&) = delete;
    ScratchScope& operator=(const ScratchScope&) = delete;
private:
    int device_;
};

}  // namespace fmsi
