// Private definition of the device-index handle shared by the query kernels (fmsi_hip.hip) and the
// on-device index builder (build_index.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../include/fmsi_hip.h"
#include "common.h"

namespace fmsi {

// device-visible index descriptor (passed by value as a kernel argument -> SGPRs)
struct DevIndex {
    const uint4* blocks;      // n_blocks x 64 B, layout in common.h
    const uint32_t* mrank;    // n_blocks + 1 cumulative mask popcounts (OR-mode interval test, ORDERS)
    const uint2* ptable;      // optional 4^t-entry prefix table of (i, j) after t LF steps, or nullptr
    uint32_t n;               // BWT rows
    uint32_t dollar;          // dollar_position
    uint32_t C0, C1, C2, C3;  // counts[] (src/fms_index.h:451)
    uint32_t ptable_len;      // t
};

#define FMSI_HIP_TRY(expr)                                                                                 \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) {                                                                            \
            ::fmsi::set_last_error(std::string(#expr) + ": " + hipGetErrorString(e_));                    \
            return FMSI_HIP_EHIP;                                                                          \
        }                                                                                                  \
    } while (0)

}  // namespace fmsi

struct fmsi_hip_index {
    int device = 0;
    fmsi::DevIndex dev{};
    uint64_t n = 0;
    int k = 0;
    int has_klcp = 0;
    int prefix_len = 0;
    uint64_t n_blocks = 0;
    uint64_t device_bytes = 0;
    void* d_blocks = nullptr;
    void* d_mrank = nullptr;
    void* d_ptable = nullptr;
    void* d_klcp = nullptr;      // plain kLCP bit-vector (n bits, u64 words), when present
    void* d_text2 = nullptr;     // 2-bit packed superstring kept by the on-device builder (synthetic workloads)
    uint64_t text_len = 0;
    // staging for host-pointer calls
    void* d_in = nullptr; size_t d_in_cap = 0;
    void* d_aux = nullptr; size_t d_aux_cap = 0;
    void* d_out = nullptr; size_t d_out_cap = 0;
    void* d_kmers = nullptr; size_t d_kmers_cap = 0;   // windows -> packed k-mers
    unsigned long long* d_counters = nullptr;
    // profiling
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    double prof_ms = 0;
    uint64_t prof_launches = 0;
};

namespace fmsi {
// builds the optional prefix table on the device (fmsi_hip.hip)
int build_prefix_table(fmsi_hip_index* idx, int t);
}
