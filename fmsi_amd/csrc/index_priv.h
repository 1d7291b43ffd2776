// Private definition of the device-index handle shared by the query kernels (fmsi_hip.hip) and the
// on-device index builder (build_index.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#include "../../include/fmsi_hip.h"
#include "common.h"
#include "devpool.h"

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

// the same blocks for an index of 2^32 rows and more (fmsi_wide.hip): u64 rows, absolute counts per superblock
struct DevIndexW {
    const uint4* blocks;
    const uint32_t* mrank;     // wrapped u32 cumulative mask popcounts
    const uint64_t* super;     // per superblock of 2^sb_log2 blocks: absolute #A #C #G #T #mask before it (+ 3 words of padding)
    uint32_t sb_log2, ptable_len;   // ptable_len: 0 = no prefix table
    uint64_t n, dollar, C0, C1, C2, C3;
    const ulonglong2* ptable;       // 4^ptable_len intervals [i, j) of the patterns' last ptable_len bases (fmsi_wide.hip)
};

// the k-mer dictionary (fmsi_dict.hip): up to four levels of 128-byte buckets, passed by value to the lookup kernel
constexpr uint32_t kDictMaxLevels = 4;
struct DictLevel {
    const uint4* buckets;   // 2^(64 - shift) buckets x 8 x 16 bytes
    uint32_t shift, pad;    // bucket = hash >> shift
    uint64_t slot_base;     // index of this level's first slot in the per-slot side arrays (chained lookups)
};
struct DictIndex {
    DictLevel lv[kDictMaxLevels];
    uint32_t levels;        // 0 = no dictionary
    int k;
    // chained lookups of reads (FMSI_HIP_DICT_CHAIN; null without): the 2-bit superstring, the answer nibble of the k-mer at every text
    // position (16 per u64 word) and, per dictionary slot, the text position of one occurrence of its k-mer
    const uint64_t* text2;
    const uint64_t* posans;
    const uint32_t* slot_pos;
    uint64_t text_len;
};

// the super-k-mer dictionary (skdict.inc): one level of 128-byte buckets of eight 16-byte records, passed by value to its kernel
struct SkIndex {
    const uint4* buckets;   // 2^bits buckets
    uint32_t bits;          // 0: none
    int k, m;
    uint32_t pad;
};

#define FMSI_HIP_TRY(expr)                                                                                 \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) {                                                                            \
            ::fmsi::set_last_error(std::string(#expr) + ": " + hipGetErrorString(e_));                    \
            return FMSI_HIP_EHIP;                                                                          \
        }                                                                                                  \
    } while (0)

// Library calls run on the handle's device and leave the calling thread's current device as they found it (a
// multi-GPU single-process caller -- torch, or the CLI's --devices threads -- is not silently moved to another GPU).
struct DeviceGuard {
    int prev = -1, target = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) : target(device) {
        if (hipGetDevice(&prev) != hipSuccess) {
            (void)hipGetLastError();
            prev = -1;
        }
        if (prev != device) err = hipSetDevice(device);
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != target) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define FMSI_HIP_ON_DEVICE(device)                  \
    ::fmsi::DeviceGuard device_guard_(device);      \
    FMSI_HIP_TRY(device_guard_.err)

// Page-locked staging for the small host tables a query call sends along (run starts / offsets): a ring of slots, each
// guarded by an event recorded behind the copy that reads it, so that the call stays asynchronous on the caller's stream
// (no hipStreamSynchronize to keep a local vector alive) and a slot is reused only when its copy is done.
struct MetaSlot {
    void* host = nullptr;
    size_t cap = 0;
    hipEvent_t done = nullptr;
    bool in_flight = false;
};
constexpr unsigned kMetaSlots = 4;

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
    // 64-bit rows (fmsi_wide.hip): n >= 2^32 - 128, or forced for tests; only the blocks, mrank and the superblock table exist
    bool wide = false;
    fmsi::DevIndexW devw{};
    void* d_super = nullptr;
    uint64_t counts64[4] = {0, 0, 0, 0}, dollar64 = 0;   // C array and dollar position in full width (any handle)
    void* d_blocks = nullptr;
    void* d_mrank = nullptr;
    void* d_ptable = nullptr;
    hipStream_t bg_stream = nullptr;   // the table's signature pass when it runs beside the first queries (build_prefix_table), else null
    bool bg_pending = false;
    // FMSI_HIP_LOAD_ASYNC_TABLE: the pass has not been started yet -- it starts with the handle's first streaming call (whose certificates are
    // what the signatures are mostly for) or once kSignAfterLookups k-mers have been looked up through the table (start_deferred_sign)
    bool sign_deferred = false;
    uint64_t lookups_unsigned = 0, streamed_unsigned = 0;
    int fn_kind = 0;             // demasking function of the current fmsi_hip_query_runs_fn call (FMSI_HIP_FN_*) and its bounds
    uint64_t fn_r = 0, fn_s = 0;
    void* d_klcp = nullptr;      // plain kLCP bit-vector (n bits, u64 words), when present
    void* d_sblocks = nullptr;   // streaming layout (fmsi_stream.hip): 128-byte blocks of 192 rows with the kLCP plane, built on first use
    uint64_t n_sblocks = 0;
    bool stream_layout_failed = false;   // no room for it: fmsi_hip_query_stream keeps the first kernel and does not try again
    // optional pair layout (fmsi_pair.hip): 128-B blocks for two-base LF steps
    void* d_pblocks = nullptr;
    void* d_sbt = nullptr;
    void* d_c2 = nullptr;
    uint32_t pair_r1 = 0xFFFFFFFFu, pair_p1 = 0xFFFFFFFFu;
    void* d_text2 = nullptr;     // 2-bit packed superstring (kept by the on-device builder, or rebuilt by fmsi_hip_index_enable_verify)
    void* d_vtable = nullptr;    // text verification: the lookup kernel's 16-byte copy of the table entries with candidates 0 and 1
    void* d_stable = nullptr;    // ... and candidate 2 per entry
    bool vwide = false;          // d_vtable holds 32-byte entries (up to seven candidates in a double-stranded entry)
    // optional k-mer dictionary (fmsi_dict.hip; fmsi_hip_index_enable_dict)
    fmsi::DictIndex dict{};
    void* d_dict[fmsi::kDictMaxLevels] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t dict_bytes = 0, dict_keys[fmsi::kDictMaxLevels] = {0, 0, 0, 0}, dict_ambiguous = 0;
    // optional super-k-mer dictionary (skdict.inc; FMSI_HIP_DICT_SUPERKMER) with its aux k-mer dictionary of what it does not hold
    fmsi::SkIndex sk{};
    void* d_sk = nullptr;
    fmsi::DictIndex sk_aux{};
    void* d_sk_aux[fmsi::kDictMaxLevels] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t sk_bytes = 0, sk_records = 0, sk_heavy = 0, sk_overflowed = 0, sk_aux_keys = 0;
    void* d_posans = nullptr;     // chained lookups: answer nibble per text position
    void* d_slot_pos = nullptr;   // ... and a text position per dictionary slot
    bool dict_owns_text = false;  // d_text2 was made by the dictionary's construction
    uint64_t dict_chain_bytes = 0;
    uint64_t text_len = 0;
    // staging for host-pointer calls
    void* d_in = nullptr; size_t d_in_cap = 0;
    void* d_aux = nullptr; size_t d_aux_cap = 0;
    void* d_out = nullptr; size_t d_out_cap = 0;
    void* d_kmers = nullptr; size_t d_kmers_cap = 0;   // windows -> packed k-mers
    void* d_tasks = nullptr; size_t d_tasks_cap = 0;   // streaming task descriptors + records
    void* d_bytes = nullptr; size_t d_bytes_cap = 0;   // FMSI_HIP_OUT_BITS: result bytes of the kernels that do not pack themselves
    void* d_case = nullptr; size_t d_case_cap = 0;     // fmsi_hip_query_runs_fn: the sequence's lower-case bits
    void* d_sorted = nullptr; size_t d_sorted_cap = 0; // FMSI_HIP_KERNEL_SORTED: items, sort buffers, hand-over slots, ternaries
    const uint64_t* fn_case = nullptr;                 // ... as the current call's kernel sees them (device pointer or nullptr)
    unsigned long long* d_counters = nullptr;
    fmsi::MetaSlot meta_ring[fmsi::kMetaSlots];
    unsigned meta_next = 0;
    // profiling
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    double prof_ms = 0;
    uint64_t prof_launches = 0;
};

namespace fmsi {
// builds the optional prefix table on the device (fmsi_hip.hip); sign_in_background: the signatures of its short entries are written on a
// stream of their own while the handle already answers -- wait_background() before anything that reads or replaces the table as a whole
int build_prefix_table(fmsi_hip_index* idx, int t, bool sign_in_background = false);
int wait_background(fmsi_hip_index* idx);
// round 6: the background pass is not started by the load but by what profits from it.  At human scale it is 0.17 s of device time for a
// 15-deep table; a query file of 10^8 isolated k-mers is 0.05 s of lookups, which ran at a third of their rate beside it (device stage
// 0.12-0.20 s, the `fmsi` binary's run 0.44-0.66 s where it is 0.40 without the pass) and gain nothing a pass that long can pay for.
void start_deferred_sign(fmsi_hip_index* idx);
constexpr uint64_t kSignAfterLookups = 250000000ull;
inline void note_lookups(fmsi_hip_index* idx, uint64_t n) {
    if (idx->sign_deferred && (idx->lookups_unsigned += n) >= kSignAfterLookups) start_deferred_sign(idx);
}
// ... and streamed windows wait for it longer still: the pass is 0.17 s of device time at human scale, the kLCP kernel without its certificates
// answers 10^9 windows of the benchmark's read mix in 0.055 s instead of 0.017 s -- the pass has paid for itself after 4.4 x 10^9 windows, and a
// job shorter than that is faster without it (the `fmsi` binary on 10^9 windows: device stage 0.10-0.16 s beside the pass, 0.06-0.09 s without;
// FMSI_HIP_SIGN_AFTER_WINDOWS sets the mark, 0 = the first streaming call as at the start of round 6)
uint64_t sign_after_windows();
inline void note_streamed(fmsi_hip_index* idx, uint64_t n) {
    if (idx->sign_deferred && (idx->streamed_unsigned += n) > sign_after_windows()) start_deferred_sign(idx);
}
// have the runtime load the code object of fmsi_hip.hip / fmsi_stream.hip now (it does so at the first use of one of their kernels, 10-20 ms
// each): the index load calls these while its payloads are still on their way
void warm_table_module();
void warm_stream_module();

// two-base-step kernel over the pair layout (fmsi_pair.hip); requires idx->d_pblocks
void launch_pair_query(fmsi_hip_index* idx, const uint64_t* d_kmers, uint64_t n, int k, uint32_t flags, bool use_table, uint8_t* o8,
                       long long* o64, unsigned long long* d_counters, hipStream_t s);

// k > 32: windows searched straight off the packed sequence (fmsi_longk.hip)
// `-f and|xor|r-s` (fmsi_fn.hip): fmsi_hip_query_runs_fn leaves the function in the handle and goes through fmsi_hip_query_runs with this
// private flag, which cuts the runs into window starts and launches the kernel below
// most items one rocprim::radix_sort call is given here (measured on ROCm 7.2: 2^32 - 135 items fail with "invalid configuration argument", 2^32 - 4099 sort)
constexpr uint64_t kMaxSortItems = (1ull << 32) - 8192;
constexpr uint32_t kModeFn = 0x80000000u;
void launch_fn_query(fmsi_hip_index* idx, const uint64_t* d_seq, const uint64_t* d_window_start, uint64_t n, int k, bool use_table, uint8_t* o8,
                     unsigned long long* d_counters, hipStream_t s);
void launch_longk_query(fmsi_hip_index* idx, const uint64_t* d_seq, const uint64_t* d_window_start, uint64_t n, int k, uint32_t flags,
                        uint8_t* o8, long long* o64, unsigned long long* d_counters, hipStream_t s);

// phased form of the same (fmsi_pair.hip): table lookups in a kernel of their own, the searches still alive in a second;
// needs the pair layout and a prefix table with k >= t.  h_runs != nullptr: d_kmers is a packed sequence and the n windows
// are runs of consecutive windows of it -- h_runs (HOST) = n_runs first bases, then n_runs + 1 exclusive prefix sums of
// the run lengths; the lookup kernel cuts the windows itself
int launch_phased_query(fmsi_hip_index* idx, const uint64_t* d_kmers, uint64_t n, int k, uint32_t flags, uint8_t* o8, long long* o64,
                        unsigned long long* d_counters, hipStream_t s, const uint64_t* h_runs = nullptr, uint64_t n_runs = 0);

// 64-bit rows (fmsi_wide.hip)
void wide_super_table(const uint32_t* blocks, const uint32_t* mrank, uint64_t n_blocks, uint32_t sb_log2, std::vector<uint64_t>& super);
int launch_wide_query(fmsi_hip_index* idx, const uint64_t* d_kmers, uint64_t n, int k, uint32_t flags, uint8_t* o8, long long* o64,
                      unsigned long long* d_counters, hipStream_t s);
int build_prefix_table_wide(fmsi_hip_index* idx, int t);   // (at most 14 deep: 16-byte entries)
// what a handle with 64-bit rows does not have
#define FMSI_HIP_NARROW_ONLY(idx, what)                                                                                          \
    do {                                                                                                                         \
        if ((idx)->wide) {                                                                                                       \
            ::fmsi::set_last_error(std::string(what) + " is not built for indexes of 2^32 rows or more (64-bit rows: plain search only)"); \
            return FMSI_HIP_ERANGE;                                                                                              \
        }                                                                                                                        \
    } while (0)

// FMSI_HIP_KERNEL_SORTED (fmsi_sorted.hip): the batch's strand searches sorted and walked in lock-step over the pair layout
bool sorted_walk_applies(const fmsi_hip_index* idx, int k, uint32_t flags);
int launch_sorted_query(fmsi_hip_index* idx, const uint64_t* d_kmers, uint64_t n, int k, uint32_t flags, uint8_t* o8, unsigned long long* d_counters, hipStream_t s,
                        bool chosen_by_auto = false);

// the k-mer dictionary answers this call (presence modes, the index's k, no NO_PREFIX_TABLE) -- and its launch; h_runs as for
// launch_phased_query.  The lookup kernel packs FMSI_HIP_OUT_BITS output itself.
bool dict_applies(const fmsi_hip_index* idx, int k, uint32_t flags);
// (uniform_runs != nullptr: {first base, stride, windows per run} of uniform runs -- no table, h_runs is not read)
int launch_dict_query(fmsi_hip_index* idx, const uint64_t* d_in, uint64_t n, int k, uint32_t flags, uint8_t* o8, unsigned long long* d_counters, hipStream_t s,
                      const uint64_t* h_runs = nullptr, uint64_t n_runs = 0, const uint64_t* uniform_runs = nullptr);
// uniform runs of a packed read batch through the chained form (needs the dictionary's chain data): a lane takes 32 consecutive
// windows, looks the first one up and follows the superstring from its occurrence
bool dict_chain_applies(const fmsi_hip_index* idx, int k, uint32_t flags);
// uniform runs through the super-k-mer dictionary (a bucket read per minimizer, not per window)
bool skdict_applies(const fmsi_hip_index* idx, int k, uint32_t flags);
// (uniform_runs, or the host-side run table of launch_dict_query: n_runs starts, then n_runs + 1 offsets; seq_words: the words of d_seq that
// hold bases -- the kernel reads no further)
int launch_skdict(fmsi_hip_index* idx, const uint64_t* d_seq, uint64_t seq_words, uint64_t n, int k, uint32_t flags, uint8_t* o8, unsigned long long* d_counters,
                  hipStream_t s, const uint64_t* uniform_runs, const uint64_t* h_runs = nullptr, uint64_t n_runs = 0);
// ... and isolated k-mers (8 bytes each) through the records: a lane per k-mer, one bucket read each (sk_kmers_kernel)
int launch_skdict_kmers(fmsi_hip_index* idx, const uint64_t* d_kmers, uint64_t n, uint32_t flags, uint8_t* o8, unsigned long long* d_counters, hipStream_t s);
int launch_dict_chain(fmsi_hip_index* idx, const uint64_t* d_seq, uint64_t n, int k, uint32_t flags, uint8_t* o8, unsigned long long* d_counters, hipStream_t s,
                      const uint64_t* uniform_runs);

// FMSI_HIP_OUT_BITS: does launch_phased_query write the packed output itself for this (handle, k, flags)?  (otherwise the
// caller has the kernels write bytes into a scratch buffer and packs them with launch_pack_results)
bool phased_packs_bits(const fmsi_hip_index* idx, int k, uint32_t flags);
// n result bytes (device) -> packed words at d_out (device, 8-byte aligned): 1 bit per window, `strict`: 4 (fmsi_hip.hip)
int launch_pack_results(const uint8_t* d_bytes, uint64_t n, bool strict, void* d_out, hipStream_t s);

// the LF walk of `fmsi export` (fmsi_export.hip): letters with the mask's case into d_chars[0 .. n-2] (device), and
// optionally sa[row] = text position of the row's suffix for every row (device, n entries)
int export_walk(const fmsi_hip_index* idx, char* d_chars, uint32_t* d_sa);

// grow-only device staging buffer
inline int ensure_cap(void** p, size_t* cap, size_t need) {
    if (need <= *cap) return 0;
    if (*p) (void)dev_free(*p);
    *p = nullptr;
    *cap = 0;
    const size_t want = need > (size_t(1) << 20) ? need : (size_t(1) << 20);
    FMSI_HIP_TRY(dev_malloc(p, want));
    *cap = want;
    return 0;
}

// a page-locked buffer of `bytes` for this call's host-side table; meta_commit() after the copy that reads it was enqueued
inline int meta_stage(fmsi_hip_index* idx, size_t bytes, void** host, unsigned* slot_out) {
    const unsigned slot = idx->meta_next++ % kMetaSlots;
    MetaSlot& m = idx->meta_ring[slot];
    if (m.in_flight) {
        FMSI_HIP_TRY(hipEventSynchronize(m.done));
        m.in_flight = false;
    }
    if (!m.done) FMSI_HIP_TRY(hipEventCreateWithFlags(&m.done, hipEventDisableTiming));
    if (m.cap < bytes) {
        if (m.host) (void)hipHostFree(m.host);
        m.host = nullptr;
        m.cap = 0;
        const size_t want = std::max<size_t>(bytes + bytes / 2, size_t(1) << 16);
        FMSI_HIP_TRY(hipHostMalloc(&m.host, want, hipHostMallocDefault));
        m.cap = want;
    }
    *host = m.host;
    *slot_out = slot;
    return 0;
}
inline int meta_commit(fmsi_hip_index* idx, unsigned slot, hipStream_t s) {
    MetaSlot& m = idx->meta_ring[slot];
    FMSI_HIP_TRY(hipEventRecord(m.done, s));
    m.in_flight = true;
    return 0;
}

// bytes of the result buffer of a query call over n_windows windows (fmsi_hip.hip)
size_t out_buffer_bytes(uint64_t n_windows, uint32_t flags);

// FMSI_HIP_OUT_BITS around kernels that write one result byte per window: begin() points them at the handle's scratch bytes
// and takes the flag out of `flags`, end() packs the bytes into the caller's buffer
struct PackAfter {
    bool on = false;
    uint8_t* bytes = nullptr;
    int begin(fmsi_hip_index* idx, uint32_t& flags, uint64_t n_windows, void* d_out) {
        const bool want = (flags & FMSI_HIP_OUT_BITS) && !(flags & FMSI_HIP_ORDERS);
        flags &= ~FMSI_HIP_OUT_BITS;
        bytes = static_cast<uint8_t*>(d_out);
        if (!want) return 0;
        if (reinterpret_cast<uintptr_t>(d_out) & 7u) {
            set_last_error("FMSI_HIP_OUT_BITS needs an 8-byte aligned output buffer");
            return FMSI_HIP_EINVAL;
        }
        int rc = ensure_cap(&idx->d_bytes, &idx->d_bytes_cap, n_windows);
        if (rc) return rc;
        on = true;
        bytes = static_cast<uint8_t*>(idx->d_bytes);
        return 0;
    }
    int end(uint64_t n_windows, uint32_t flags, void* d_out, hipStream_t s) const {
        return on ? launch_pack_results(bytes, n_windows, (flags & FMSI_HIP_STRICT) != 0u, d_out, s) : 0;
    }
};

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
// FMSI_HIP_VERBOSE: phase times of the load / the constructions on stderr (a device synchronisation per mark, only then)
struct PhaseClock {
    const char* tag = "";
    bool on = std::getenv("FMSI_HIP_VERBOSE") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(const char* what) {
        if (!on) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[fmsi %s] %-28s %8.3f s\n", tag, what, std::chrono::duration<double>(now - t).count());
        t = now;
    }
};

}  // namespace fmsi
