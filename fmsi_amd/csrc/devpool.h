// Device memory for the library: hipMalloc / hipFree behind a cache of large freed blocks.
//
// Why (tools/alloc_bench.hip): on this platform hipFree of a large buffer returns at once but the NEXT hipMallocs stall
// for about a second per 30-40 GB freed before them.  The index builder, the pair layout, the candidate tables and the dictionary's
// construction hand each other ~150 GB of temporaries of the same few sizes (8, 4 and 1 bytes per row), so freed blocks of 64 MB and more
// are kept and handed out again (best fit, at most a quarter more than the request); they go back to the driver when an allocation fails (then the
// allocation is tried again), when the last handle of a device is destroyed, or on fmsi_hip_trim_scratch().
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

}  // namespace fmsi
