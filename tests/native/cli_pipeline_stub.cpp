// The `fmsi query` / `lookup` HOST pipeline (fmsi_amd/csrc/query_cli.cpp: readers, packing teams, device workers, in-order hand-over,
// predictor replay, formatting, sink) with the device taken out: the dozen library calls it makes are stood in for by functions that
// answer every window from a hash of its k-mer.  TEST INFRASTRUCTURE ONLY -- built by tests/test_host.py with ThreadSanitizer and with
// AddressSanitizer + UBSan, so that races and memory errors of the threaded pipeline show on a machine without a GPU, and so that
// its output can be held against itself across batch sizes, thread counts and handle counts (answers that depend on the k-mer alone
// must print the same however the input is cut: the strand predictor's state crosses the batches).  Nothing here is a query path:
// the answers are noise by construction.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/fmsi_hip.h"

extern "C" int fmsi_cli_query(int argc, char** argv, int output_orders);

struct fmsi_hip_index {
    int device;
};

namespace {
uint64_t kmer_hash(const uint64_t* seq, uint64_t start, int k) {
    uint64_t h = 0x9E3779B97F4A7C15ull;
    static const bool fast = std::getenv("FMSI_STUB_FAST") != nullptr;   // (timing runs of the host stages: the window's first word only)
    if (fast) return (seq[start >> 5] >> (start & 31)) * 0x9E3779B97F4A7C15ull >> 7;
    for (int t = 0; t < k; ++t) {
        const uint64_t p = start + static_cast<uint64_t>(t);
        h = (h ^ ((seq[p >> 5] >> (2 * (p & 31))) & 3u)) * 0x100000001B3ull;
        h ^= h >> 29;
    }
    return h;
}
void answer(const uint64_t* seq, uint64_t start, int k, uint32_t flags, int fn, void* out, uint64_t q) {
    const uint64_t h = kmer_hash(seq, start, k);
    if (flags & FMSI_HIP_ORDERS) {
        const int64_t f = h % 5 ? static_cast<int64_t>((h >> 8) % 3000000000ull) : -1, r = (h >> 3) % 4 == 0 ? static_cast<int64_t>((h >> 20) % 100000) : -1;
        int64_t* o = static_cast<int64_t*>(out);
        if (flags & FMSI_HIP_STRICT) {
            o[2 * q] = f;
            o[2 * q + 1] = r;
        } else {
            o[q] = f >= 0 ? f : r;
        }
        return;
    }
    const int f = static_cast<int>(h % 3) - 1, r = static_cast<int>((h / 3) % 3) - 1;
    const unsigned v = fn ? static_cast<unsigned>((h >> 7) & 1) : ((flags & FMSI_HIP_STRICT) ? static_cast<unsigned>((f + 1) | ((r + 1) << 2)) : static_cast<unsigned>(f == 1 || r == 1));
    if (flags & FMSI_HIP_OUT_BITS) {
        uint64_t* w = static_cast<uint64_t*>(out);
        if (flags & FMSI_HIP_STRICT) w[q >> 4] |= static_cast<uint64_t>(v) << (4 * (q & 15));
        else w[q >> 6] |= static_cast<uint64_t>(v) << (q & 63);
    } else {
        static_cast<uint8_t*>(out)[q] = static_cast<uint8_t>(v);
    }
}
int run_all(const uint64_t* seq, const uint64_t* run_start, const uint32_t* run_windows, uint64_t n_runs, uint64_t n_windows, int k, uint32_t flags, int fn,
            void* out) {
    if (flags & FMSI_HIP_OUT_BITS) std::memset(out, 0, ((flags & FMSI_HIP_STRICT) ? (n_windows + 15) / 16 : (n_windows + 63) / 64) * 8);
    uint64_t q = 0;
    for (uint64_t r = 0; r < n_runs; ++r)
        for (uint32_t i = 0; i < run_windows[r]; ++i, ++q) answer(seq, run_start[r] + i, k, flags, fn, out, q);
    return q == n_windows ? 0 : FMSI_HIP_EINVAL;
}
}  // namespace

extern "C" {
const char* fmsi_hip_last_error(void) { return "stub"; }
void* fmsi_hip_host_alloc(uint64_t bytes) { return std::malloc(bytes); }
void fmsi_hip_host_free(void* p) { std::free(p); }
int fmsi_hip_index_load(const char*, int, int device, int, fmsi_hip_index** out) {
    *out = new fmsi_hip_index{device};
    return 0;
}
void fmsi_hip_index_destroy(fmsi_hip_index* idx) { delete idx; }
int fmsi_hip_index_get_info(const fmsi_hip_index*, fmsi_hip_index_info* info) {
    std::memset(info, 0, sizeof *info);
    return 0;
}
int fmsi_hip_index_enable_dict(fmsi_hip_index*, uint32_t) { return 0; }
int fmsi_hip_index_enable_pair_layout(fmsi_hip_index*) { return 0; }
int fmsi_hip_index_enable_verify(fmsi_hip_index*) { return 0; }
int fmsi_hip_sync(fmsi_hip_index*, void*) { return 0; }
void fmsi_hip_keep_scratch(int) {}
int fmsi_hip_trim_scratch(int) { return 0; }
int fmsi_hip_query_kmers(fmsi_hip_index*, const uint64_t* kmers, uint64_t n_kmers, int k, uint32_t flags, void* out, uint64_t*, void*) {
    if (flags & FMSI_HIP_OUT_BITS) std::memset(out, 0, ((flags & FMSI_HIP_STRICT) ? (n_kmers + 15) / 16 : (n_kmers + 63) / 64) * 8);
    for (uint64_t q = 0; q < n_kmers; ++q) answer(kmers + q, 0, k, flags, 0, out, q);   // (the k-mer's word is a one-window sequence)
    return 0;
}
int fmsi_hip_query_runs(fmsi_hip_index*, const uint64_t* seq, uint64_t, const uint64_t* run_start, const uint32_t* run_windows, uint64_t n_runs,
                        uint64_t n_windows, int k, uint32_t flags, void* out, uint64_t*, void*) {
    return run_all(seq, run_start, run_windows, n_runs, n_windows, k, flags, 0, out);
}
int fmsi_hip_query_stream(fmsi_hip_index*, const uint64_t* seq, uint64_t, const uint64_t* run_start, const uint32_t* run_windows, uint64_t n_runs,
                          uint64_t n_windows, int k, uint32_t flags, void* out, uint64_t*, void*) {
    return run_all(seq, run_start, run_windows, n_runs, n_windows, k, flags, 0, out);
}
int fmsi_hip_query_runs_fn(fmsi_hip_index*, const uint64_t* seq, const uint64_t*, uint64_t, const uint64_t* run_start, const uint32_t* run_windows,
                           uint64_t n_runs, uint64_t n_windows, int k, int fn, uint64_t, uint64_t, uint32_t flags, void* out, uint64_t*, void*) {
    return run_all(seq, run_start, run_windows, n_runs, n_windows, k, flags, fn, out);
}
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const bool lookup = std::string(argv[1]) == "lookup";
    return fmsi_cli_query(argc - 1, argv + 1, lookup ? 1 : 0);
}
