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
    uint32_t ptable_packed;   // entry format: 0 = (i, j); 1 = (i, tagged word) with next-base signatures, see decode_table_entry
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
    // optional pair layout (fmsi_pair.hip): 128-B blocks for two-base LF steps
    void* d_pblocks = nullptr;
    void* d_sbt = nullptr;
    void* d_c2 = nullptr;
    uint32_t pair_r1 = 0xFFFFFFFFu, pair_p1 = 0xFFFFFFFFu;
    void* d_text2 = nullptr;     // 2-bit packed superstring (kept by the on-device builder, or rebuilt by fmsi_hip_index_enable_verify)
    void* d_vtable = nullptr;    // text verification: the lookup kernel's 16-byte copy of the table entries with candidates 0 and 1
    void* d_stable = nullptr;    // ... and candidate 2 per entry
    bool vwide = false;          // d_vtable holds 32-byte entries (up to seven candidates in a double-stranded entry)
    uint64_t text_len = 0;
    // staging for host-pointer calls
    void* d_in = nullptr; size_t d_in_cap = 0;
    void* d_aux = nullptr; size_t d_aux_cap = 0;
    void* d_out = nullptr; size_t d_out_cap = 0;
    void* d_kmers = nullptr; size_t d_kmers_cap = 0;   // windows -> packed k-mers
    void* d_tasks = nullptr; size_t d_tasks_cap = 0;   // streaming task descriptors + records
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

// two-base-step kernel over the pair layout (fmsi_pair.hip); requires idx->d_pblocks
void launch_pair_query(fmsi_hip_index* idx, const uint64_t* d_kmers, uint64_t n, int k, uint32_t flags, bool use_table, uint8_t* o8,
                       long long* o64, unsigned long long* d_counters, hipStream_t s);

// k > 32: windows searched straight off the packed sequence (fmsi_longk.hip)
void launch_longk_query(fmsi_hip_index* idx, const uint64_t* d_seq, const uint64_t* d_window_start, uint64_t n, int k, uint32_t flags,
                        uint8_t* o8, long long* o64, unsigned long long* d_counters, hipStream_t s);

// phased form of the same (fmsi_pair.hip): table lookups in a kernel of their own, the searches still alive in a second;
// needs the pair layout and a prefix table with k >= t.  h_runs != nullptr: d_kmers is a packed sequence and the n windows
// are runs of consecutive windows of it -- h_runs (HOST) = n_runs first bases, then n_runs + 1 exclusive prefix sums of
// the run lengths; the lookup kernel cuts the windows itself
int launch_phased_query(fmsi_hip_index* idx, const uint64_t* d_kmers, uint64_t n, int k, uint32_t flags, uint8_t* o8, long long* o64,
                        unsigned long long* d_counters, hipStream_t s, const uint64_t* h_runs = nullptr, uint64_t n_runs = 0);

// the LF walk of `fmsi export` (fmsi_export.hip): letters with the mask's case into d_chars[0 .. n-2] (device), and
// optionally sa[row] = text position of the row's suffix for every row (device, n entries)
int export_walk(const fmsi_hip_index* idx, char* d_chars, uint32_t* d_sa);

// grow-only device staging buffer
inline int ensure_cap(void** p, size_t* cap, size_t need) {
    if (need <= *cap) return 0;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t want = need > (size_t(1) << 20) ? need : (size_t(1) << 20);
    FMSI_HIP_TRY(hipMalloc(p, want));
    *cap = want;
    return 0;
}

// HIP events around a kernel launch on the launch stream (fmsi_hip_profile_*)
inline int prof_begin(fmsi_hip_index* idx, hipStream_t s, size_t* slot) {
    *slot = static_cast<size_t>(-1);
    if (!idx->profiling) return 0;
    if (idx->events_used == idx->events.size()) {
        hipEvent_t a, b;
        FMSI_HIP_TRY(hipEventCreate(&a));
        FMSI_HIP_TRY(hipEventCreate(&b));
        idx->events.emplace_back(a, b);
    }
    *slot = idx->events_used++;
    FMSI_HIP_TRY(hipEventRecord(idx->events[*slot].first, s));
    return 0;
}
inline int prof_end(fmsi_hip_index* idx, hipStream_t s, size_t slot) {
    if (slot == static_cast<size_t>(-1)) return 0;
    FMSI_HIP_TRY(hipEventRecord(idx->events[slot].second, s));
    return 0;
}
}  // namespace fmsi
