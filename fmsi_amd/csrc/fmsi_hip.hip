// libfmsi_hip.so -- device side of the `fmsi query` hot path for MI355X (gfx950, wave64).
//
// What it replaces in the reference (citations relative to the reference root):
//   rank()                    src/fms_index.h:68-86    two dependent rank_support_v5 probes -> one 64-B block read
//   update_range()            src/fms_index.h:98-103   one LF / backward-search step
//   get_range_with_pattern()  src/fms_index.h:117-124  <= k LF steps with early exit on an empty interval
//   infer_presence<>()        src/fms_index.h:126-144  mask access (rrr_vector<63>::operator[]) -> bit of the same block
//   kmer_order_if_present()   src/fms_index.h:146-156  rank on the mask -> per-block cumulative count + popcount
//   query_kmers_single<>()    src/fms_index.h:263-331  strand schedule (without the sequential predictor, see STRICT)
//
// Integer / bit-manipulation path: no MFMA.  The limiter is dependent random 64-byte reads, so the
// design goals are (1) one aligned 64-B read per rank, (2) as many independent reads in flight per CU
// as the register file allows, (3) no lane idling while its neighbours still walk (refill kernel).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fmsi_hip.h"
#include "common.h"
#include "layout.h"

namespace fmsi {

// ------------------------------------------------------------------------------------------------
// device-visible index descriptor (passed by value as a kernel argument -> SGPRs)
// ------------------------------------------------------------------------------------------------
struct DevIndex {
    const uint4* blocks;      // n_blocks x 64 B, layout in common.h
    const uint32_t* mrank;    // n_blocks + 1 cumulative mask popcounts (OR-mode interval test, ORDERS)
    uint32_t n;               // BWT rows
    uint32_t dollar;          // dollar_position
    uint32_t C0, C1, C2, C3;  // counts[] (src/fms_index.h:451)
};

// ones in bit positions [0, s) of a 32-bit word, s may be <= 0 or >= 32
__device__ __forceinline__ uint32_t prefix32(int s) {
    return s >= 32 ? 0xFFFFFFFFu : (s <= 0 ? 0u : ((1u << s) - 1u));
}

// Keeps a value opaque to the optimiser.  Without it hipcc folds a chain of selects over the four
// components of a uint4 into a dynamically indexed vector, which it then spills to scratch
// (cdna_hip_programming.md rule 20).
__device__ __forceinline__ uint32_t opaque(uint32_t x) {
    asm volatile("" : "+v"(x));
    return x;
}

__device__ __forceinline__ uint32_t sel4(uint32_t c, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3) {
    const uint32_t lo = (c & 1u) ? a1 : a0;
    const uint32_t hi = (c & 1u) ? a3 : a2;
    return (c & 2u) ? hi : lo;
}

// occurrences of symbol c among the first p rows (p in [0,128]) of the block, plus the block's base count
__device__ __forceinline__ uint32_t rank_in_block(const uint4 cnt, const uint4 h, const uint4 l, uint32_t c, uint32_t p) {
    const uint32_t xh = (c & 2u) ? 0u : 0xFFFFFFFFu;
    const uint32_t xl = (c & 1u) ? 0u : 0xFFFFFFFFu;
    uint32_t r = sel4(c, opaque(cnt.x), opaque(cnt.y), opaque(cnt.z), opaque(cnt.w));
    const int s = static_cast<int>(p);
    r += __popc((h.x ^ xh) & (l.x ^ xl) & prefix32(s));
    r += __popc((h.y ^ xh) & (l.y ^ xl) & prefix32(s - 32));
    r += __popc((h.z ^ xh) & (l.z ^ xl) & prefix32(s - 64));
    r += __popc((h.w ^ xh) & (l.w ^ xl) & prefix32(s - 96));
    return r;
}

// One LF step on [i, j), i < j (update_range, src/fms_index.h:98-103 with rank() of :68-86).
// The upper boundary is ranked through the block of row j-1 (offset 1..128) so that a single-row
// interval needs ONE block.
__device__ __forceinline__ void lf_step(const DevIndex& ix, uint32_t c, uint32_t& i, uint32_t& j, uint32_t& blocks_read) {
    const uint32_t bi = i >> kBlockRowsLog2;
    const uint32_t bj = (j - 1u) >> kBlockRowsLog2;
    const uint4* pi = ix.blocks + (static_cast<size_t>(bi) << 2);
    const uint4* pj = ix.blocks + (static_cast<size_t>(bj) << 2);
    // both boundaries are fetched unconditionally: when bj == bi the second fetch is the same cache line
    const uint4 ci = pi[0], hi = pi[1], li = pi[2];
    const uint4 cj = pj[0], hj = pj[1], lj = pj[2];
    blocks_read += 1u + (bj != bi);
    uint32_t ri = rank_in_block(ci, hi, li, c, i & (kBlockRows - 1u));
    uint32_t rj = rank_in_block(cj, hj, lj, c, j - (bj << kBlockRowsLog2));
    if (c == 0u) {   // the '$' slot is stored as an 'A' and subtracted here (src/fms_index.h:81)
        ri -= (i > ix.dollar);
        rj -= (j > ix.dollar);
    }
    const uint32_t base = sel4(c, ix.C0, ix.C1, ix.C2, ix.C3);
    i = base + ri;
    j = base + rj;
}

// rank1 on the SA-transformed mask for a lower boundary (row x, offset 0..127) or an upper boundary
// (through row x-1, offset 1..128)
__device__ __forceinline__ uint32_t mask_rank(const DevIndex& ix, uint32_t b, uint32_t p) {
    const uint4 m = ix.blocks[(static_cast<size_t>(b) << 2) + 3];
    const int s = static_cast<int>(p);
    return ix.mrank[b] + __popc(m.x & prefix32(s)) + __popc(m.y & prefix32(s - 32)) + __popc(m.z & prefix32(s - 64)) +
           __popc(m.w & prefix32(s - 96));
}

__device__ __forceinline__ uint32_t mask_bit(const DevIndex& ix, uint32_t row) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(ix.blocks) + (static_cast<size_t>(row >> kBlockRowsLog2) << 4) + 12u;
    return (w[(row & (kBlockRows - 1u)) >> 5] >> (row & 31u)) & 1u;
}

// reverse complement of a 2-bit packed k-mer (base t in bits [2t,2t+1]); ReverseComplementString of src/kmers.h:53-59
__device__ __forceinline__ uint64_t revcomp_2bit(uint64_t f, int k) {
    uint64_t x = ~f;                                    // complement: 3 - c
    x = __brevll(x);                                    // reverse all bits ...
    x = ((x & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((x & 0x5555555555555555ull) << 1);   // ... and restore the order inside each pair
    x >>= (64 - 2 * k);
    return k == 32 ? x : (x & ((1ull << (2 * k)) - 1ull));
}

struct LaneCounters {
    uint32_t lf_steps = 0, blocks = 0, probes = 0, searches = 0;
};

// get_range_with_pattern + infer_presence for one strand.  Returns the ternary result (-1 empty, 0, 1);
// `order` receives kmer_order_if_present when ORDERS.
__device__ __forceinline__ int search_strand(const DevIndex& ix, uint64_t cur, int k, uint32_t mode_all, uint32_t want_orders,
                                             long long& order, LaneCounters& cn) {
    uint32_t i = 0, j = ix.n;
    cn.searches++;
    for (int t = k - 1; t >= 0 && i != j; --t) {
        const uint32_t c = static_cast<uint32_t>(cur >> (2 * t)) & 3u;
        lf_step(ix, c, i, j, cn.blocks);
        cn.lf_steps++;
    }
    order = -1;
    if (i == j) return -1;
    cn.probes++;
    cn.blocks++;
    if (mode_all && !want_orders) return static_cast<int>(mask_bit(ix, i));
    const uint32_t bi = i >> kBlockRowsLog2, bj = (j - 1u) >> kBlockRowsLog2;
    const uint32_t ri = mask_rank(ix, bi, i & (kBlockRows - 1u));
    const uint32_t rj = mask_rank(ix, bj, j - (bj << kBlockRowsLog2));
    const int any = rj > ri;
    if (want_orders && any) order = static_cast<long long>(ri);
    return any;
}

__device__ __forceinline__ uint64_t window_kmer(const uint64_t* __restrict__ seq, uint64_t start, int k) {
    const uint64_t w = start >> 5;
    const uint32_t sh = static_cast<uint32_t>(start & 31u) * 2u;
    uint64_t f = seq[w] >> sh;
    if (sh) f |= seq[w + 1] << (64u - sh);
    return k == 32 ? f : (f & ((1ull << (2 * k)) - 1ull));
}

__device__ __forceinline__ void flush_counters(const LaneCounters& cn, unsigned long long* counters, bool valid) {
    // wave reduction then one atomic per wave and counter
    uint32_t v[4] = {cn.lf_steps, cn.blocks, cn.probes, cn.searches};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t x = v[q];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(&counters[q], static_cast<unsigned long long>(x));
    }
    unsigned long long w = __popcll(__ballot(valid));
    if ((threadIdx.x & 63) == 0 && w) atomicAdd(&counters[6], w);
}

// ------------------------------------------------------------------------------------------------
// Kernel 1 ("lockstep"): one lane per window; the wave walks the k LF steps in lock-step and leaves the
// loop when its last lane is done.  INPUT 0: packed k-mers, 1: windows of a packed sequence.
// ------------------------------------------------------------------------------------------------
template <bool COUNT, int INPUT>
__global__ __launch_bounds__(256) void query_lockstep_kernel(DevIndex ix, const uint64_t* __restrict__ in,
                                                             const uint64_t* __restrict__ window_start, uint64_t n_windows,
                                                             int k, uint32_t flags, uint8_t* __restrict__ out,
                                                             long long* __restrict__ out_orders,
                                                             unsigned long long* __restrict__ counters) {
    const uint64_t q = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const bool valid = q < n_windows;
    const uint32_t mode_all = flags & FMSI_HIP_MODE_ALL;
    const uint32_t strict = flags & FMSI_HIP_STRICT;
    const uint32_t orders = flags & FMSI_HIP_ORDERS;
    LaneCounters cn;
    if (valid) {
        uint64_t f;
        if (INPUT == 0) f = in[q];
        else f = window_kmer(in, window_start ? window_start[q] : q, k);
        long long ord_f = -1, ord_r = -1;
        int res_f = search_strand(ix, f, k, mode_all, orders, ord_f, cn);
        int res_r = -1;
        bool second;
        if (strict) second = true;
        else if (orders) second = ord_f < 0;          // src/fms_index.h:283-288
        else if (mode_all) second = res_f == -1;      // src/fms_index.h:295-298
        else second = res_f != 1;                     // src/fms_index.h:290-293
        if (second) res_r = search_strand(ix, revcomp_2bit(f, k), k, mode_all, orders, ord_r, cn);
        if (orders) {
            if (strict) {
                out_orders[2 * q] = ord_f;
                out_orders[2 * q + 1] = ord_r;
            } else {
                out_orders[q] = second ? ord_r : ord_f;
            }
        } else if (strict) {
            out[q] = static_cast<uint8_t>((res_f + 1) | ((res_r + 1) << 2));
        } else {
            out[q] = static_cast<uint8_t>((second ? res_r : res_f) == 1);
        }
    }
    if (COUNT) flush_counters(cn, counters, valid);
}

// ------------------------------------------------------------------------------------------------
// host side: handle, layout builder, C-ABI
// ------------------------------------------------------------------------------------------------
#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) {                                                                            \
            set_last_error(std::string(#expr) + ": " + hipGetErrorString(e_));                            \
            return FMSI_HIP_EHIP;                                                                          \
        }                                                                                                  \
    } while (0)

}  // namespace fmsi

using namespace fmsi;

struct fmsi_hip_index {
    int device = 0;
    DevIndex dev{};
    uint64_t n = 0;
    int k = 0;
    int has_klcp = 0;
    int prefix_len = 0;
    uint64_t n_blocks = 0;
    uint64_t device_bytes = 0;
    void* d_blocks = nullptr;
    void* d_mrank = nullptr;
    // staging for host-pointer calls
    void* d_in = nullptr; size_t d_in_cap = 0;
    void* d_aux = nullptr; size_t d_aux_cap = 0;
    void* d_out = nullptr; size_t d_out_cap = 0;
    unsigned long long* d_counters = nullptr;
    // profiling
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    double prof_ms = 0;
    uint64_t prof_launches = 0;
};

namespace {

int ensure_cap(void** p, size_t* cap, size_t need) {
    if (need <= *cap) return 0;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    size_t want = std::max<size_t>(need, 1 << 20);
    HIP_TRY(hipMalloc(p, want));
    *cap = want;
    return 0;
}

int prof_begin(fmsi_hip_index* idx, hipStream_t s, size_t* slot) {
    *slot = static_cast<size_t>(-1);
    if (!idx->profiling) return 0;
    if (idx->events_used == idx->events.size()) {
        hipEvent_t a, b;
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&b));
        idx->events.emplace_back(a, b);
    }
    *slot = idx->events_used++;
    HIP_TRY(hipEventRecord(idx->events[*slot].first, s));
    return 0;
}
int prof_end(fmsi_hip_index* idx, hipStream_t s, size_t slot) {
    if (slot == static_cast<size_t>(-1)) return 0;
    HIP_TRY(hipEventRecord(idx->events[slot].second, s));
    return 0;
}

template <int INPUT>
int launch_query(fmsi_hip_index* idx, const uint64_t* d_in, const uint64_t* d_wstart, uint64_t n_windows, int k, uint32_t flags,
                 void* d_out, unsigned long long* d_counters, hipStream_t s) {
    if (n_windows == 0) return 0;
    const uint32_t threads = 256;
    const uint64_t blocks64 = (n_windows + threads - 1) / threads;
    if (blocks64 > 0x7FFFFFFFull) {
        set_last_error("too many windows for one launch");
        return FMSI_HIP_ERANGE;
    }
    const dim3 grid(static_cast<uint32_t>(blocks64)), block(threads);
    uint8_t* o8 = static_cast<uint8_t*>(d_out);
    long long* o64 = static_cast<long long*>(d_out);
    size_t slot;
    int rc = prof_begin(idx, s, &slot);
    if (rc) return rc;
    if (d_counters)
        hipLaunchKernelGGL((query_lockstep_kernel<true, INPUT>), grid, block, 0, s, idx->dev, d_in, d_wstart, n_windows, k, flags, o8,
                           o64, d_counters);
    else
        hipLaunchKernelGGL((query_lockstep_kernel<false, INPUT>), grid, block, 0, s, idx->dev, d_in, d_wstart, n_windows, k, flags, o8,
                           o64, d_counters);
    HIP_TRY(hipGetLastError());
    return prof_end(idx, s, slot);
}

size_t out_elem_bytes(uint32_t flags) {
    if (flags & FMSI_HIP_ORDERS) return (flags & FMSI_HIP_STRICT) ? 16 : 8;
    return 1;
}

int check_query_args(const fmsi_hip_index* idx, const void* in, const void* out, int k, uint32_t flags) {
    if (!idx || !in || !out) {
        set_last_error("null argument");
        return FMSI_HIP_EINVAL;
    }
    if (k < 1 || k > 32) {
        set_last_error("k must be in [1, 32] in this build");
        return FMSI_HIP_ERANGE;
    }
    (void)flags;
    return 0;
}

}  // namespace

extern "C" {

int fmsi_hip_abi_version(void) { return FMSI_HIP_ABI_VERSION; }
const char* fmsi_hip_last_error(void) { return fmsi::last_error(); }

int fmsi_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int fmsi_hip_index_create(const fmsi_index_view* view, int device, fmsi_hip_index** out) {
    if (!view || !out || !view->ac_gt || !view->ac || !view->gt || !view->mask) {
        set_last_error("null argument");
        return FMSI_HIP_EINVAL;
    }
    if (view->mask_bits == 0) {
        set_last_error("empty index");
        return FMSI_HIP_EINVAL;
    }
    if (view->mask_bits >= (1ull << 32) - kBlockRows) {
        set_last_error("index has >= 2^32 rows; the 64-bit row variant is not built");
        return FMSI_HIP_ERANGE;
    }
    int ndev = fmsi_hip_device_count();
    if (ndev <= 0 || device < 0 || device >= ndev) {
        set_last_error("no usable HIP device (libfmsi_hip has no CPU fallback)");
        return FMSI_HIP_ENODEV;
    }
    HIP_TRY(hipSetDevice(device));
    HostLayout L;
    int rc = build_layout(*view, L);
    if (rc) return rc;
    auto* idx = new fmsi_hip_index();
    idx->device = device;
    idx->n = view->mask_bits;
    idx->k = view->k;
    idx->n_blocks = L.n_blocks;
    const size_t blocks_bytes = L.blocks.size() * 4, mrank_bytes = L.mrank.size() * 4;
    auto fail = [&](int code) {
        fmsi_hip_index_destroy(idx);
        return code;
    };
    if (hipMalloc(&idx->d_blocks, blocks_bytes) != hipSuccess || hipMalloc(&idx->d_mrank, mrank_bytes) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&idx->d_counters), FMSI_HIP_N_COUNTERS * 8) != hipSuccess) {
        set_last_error("hipMalloc failed for the device layout");
        return fail(FMSI_HIP_ENOMEM);
    }
    if (hipMemcpy(idx->d_blocks, L.blocks.data(), blocks_bytes, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(idx->d_mrank, L.mrank.data(), mrank_bytes, hipMemcpyHostToDevice) != hipSuccess) {
        set_last_error("hipMemcpy of the device layout failed");
        return fail(FMSI_HIP_EHIP);
    }
    idx->device_bytes = blocks_bytes + mrank_bytes;
    idx->dev.blocks = static_cast<const uint4*>(idx->d_blocks);
    idx->dev.mrank = static_cast<const uint32_t*>(idx->d_mrank);
    idx->dev.n = static_cast<uint32_t>(idx->n);
    idx->dev.dollar = static_cast<uint32_t>(view->dollar_position);
    idx->dev.C0 = static_cast<uint32_t>(view->counts[0]);
    idx->dev.C1 = static_cast<uint32_t>(view->counts[1]);
    idx->dev.C2 = static_cast<uint32_t>(view->counts[2]);
    idx->dev.C3 = static_cast<uint32_t>(view->counts[3]);
    *out = idx;
    return 0;
}

void fmsi_hip_index_destroy(fmsi_hip_index* idx) {
    if (!idx) return;
    (void)hipSetDevice(idx->device);
    for (void* p : {idx->d_blocks, idx->d_mrank, idx->d_in, idx->d_aux, idx->d_out, static_cast<void*>(idx->d_counters)})
        if (p) (void)hipFree(p);
    for (auto& e : idx->events) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    delete idx;
}

int fmsi_hip_index_get_info(const fmsi_hip_index* idx, fmsi_hip_index_info* info) {
    if (!idx || !info) return FMSI_HIP_EINVAL;
    info->n = idx->n;
    info->k = idx->k;
    info->has_klcp = idx->has_klcp;
    info->device = idx->device;
    info->prefix_len = idx->prefix_len;
    info->block_bytes = kBlockBytes;
    info->block_symbols = kBlockRows;
    info->n_blocks = idx->n_blocks;
    info->device_bytes = idx->device_bytes;
    return 0;
}

static int run_query(fmsi_hip_index* idx, int input_kind, const uint64_t* in, size_t in_bytes, const uint64_t* wstart,
                     uint64_t n_windows, int k, uint32_t flags, void* out, uint64_t* counters, void* stream) {
    HIP_TRY(hipSetDevice(idx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t out_bytes = n_windows * out_elem_bytes(flags);
    const uint64_t* d_in = in;
    const uint64_t* d_ws = wstart;
    void* d_out = out;
    unsigned long long* d_cnt = reinterpret_cast<unsigned long long*>(counters);
    const bool dev_ptrs = flags & FMSI_HIP_DEVICE_PTRS;
    if (!dev_ptrs) {
        int rc;
        if ((rc = ensure_cap(&idx->d_in, &idx->d_in_cap, in_bytes))) return rc;
        if ((rc = ensure_cap(&idx->d_out, &idx->d_out_cap, out_bytes))) return rc;
        HIP_TRY(hipMemcpyAsync(idx->d_in, in, in_bytes, hipMemcpyHostToDevice, s));
        d_in = static_cast<const uint64_t*>(idx->d_in);
        if (wstart) {
            if ((rc = ensure_cap(&idx->d_aux, &idx->d_aux_cap, n_windows * 8))) return rc;
            HIP_TRY(hipMemcpyAsync(idx->d_aux, wstart, n_windows * 8, hipMemcpyHostToDevice, s));
            d_ws = static_cast<const uint64_t*>(idx->d_aux);
        }
        d_out = idx->d_out;
        if (counters) {
            HIP_TRY(hipMemcpyAsync(idx->d_counters, counters, FMSI_HIP_N_COUNTERS * 8, hipMemcpyHostToDevice, s));
            d_cnt = idx->d_counters;
        }
    }
    int rc = input_kind == 0 ? launch_query<0>(idx, d_in, d_ws, n_windows, k, flags, d_out, d_cnt, s)
                             : launch_query<1>(idx, d_in, d_ws, n_windows, k, flags, d_out, d_cnt, s);
    if (rc) return rc;
    if (!dev_ptrs) {
        if (out_bytes) HIP_TRY(hipMemcpyAsync(out, idx->d_out, out_bytes, hipMemcpyDeviceToHost, s));
        if (counters) HIP_TRY(hipMemcpyAsync(counters, idx->d_counters, FMSI_HIP_N_COUNTERS * 8, hipMemcpyDeviceToHost, s));
    }
    return 0;
}

int fmsi_hip_query_kmers(fmsi_hip_index* idx, const uint64_t* kmers, uint64_t n_kmers, int k, uint32_t flags, void* out,
                         uint64_t* counters, void* stream) {
    if (n_kmers == 0) return 0;
    int rc = check_query_args(idx, kmers, out, k, flags);
    if (rc) return rc;
    return run_query(idx, 0, kmers, n_kmers * 8, nullptr, n_kmers, k, flags, out, counters, stream);
}

int fmsi_hip_query_windows(fmsi_hip_index* idx, const uint64_t* packed_seq, uint64_t seq_bases, const uint64_t* window_start,
                           uint64_t n_windows, int k, uint32_t flags, void* out, uint64_t* counters, void* stream) {
    if (n_windows == 0) return 0;
    int rc = check_query_args(idx, packed_seq, out, k, flags);
    if (rc) return rc;
    if (!window_start && n_windows + static_cast<uint64_t>(k) - 1 > seq_bases) {
        set_last_error("implicit windows run past the end of the sequence");
        return FMSI_HIP_EINVAL;
    }
    return run_query(idx, 1, packed_seq, (seq_bases / 32 + 2) * 8, window_start, n_windows, k, flags, out, counters, stream);
}

int fmsi_hip_query_stream(fmsi_hip_index* idx, const uint64_t* packed_seq, uint64_t seq_bases, const uint64_t* run_start,
                          const uint32_t* run_windows, uint64_t n_runs, uint64_t n_windows, int k, uint32_t flags, void* out,
                          uint64_t* counters, void* stream) {
    (void)idx; (void)packed_seq; (void)seq_bases; (void)run_start; (void)run_windows; (void)n_runs; (void)n_windows;
    (void)k; (void)flags; (void)out; (void)counters; (void)stream;
    set_last_error("fmsi_hip_query_stream: kLCP streaming kernel not built yet");
    return FMSI_HIP_EINVAL;
}

int fmsi_hip_sync(fmsi_hip_index* idx, void* stream) {
    if (!idx) return FMSI_HIP_EINVAL;
    HIP_TRY(hipSetDevice(idx->device));
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return 0;
}

int fmsi_hip_profile_enable(fmsi_hip_index* idx, int on) {
    if (!idx) return FMSI_HIP_EINVAL;
    idx->profiling = on != 0;
    return 0;
}

int fmsi_hip_profile_read(fmsi_hip_index* idx, uint64_t* n_launches, double* total_ms, int reset) {
    if (!idx) return FMSI_HIP_EINVAL;
    HIP_TRY(hipSetDevice(idx->device));
    for (size_t e = 0; e < idx->events_used; ++e) {
        HIP_TRY(hipEventSynchronize(idx->events[e].second));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, idx->events[e].first, idx->events[e].second));
        idx->prof_ms += ms;
        idx->prof_launches++;
    }
    idx->events_used = 0;
    if (n_launches) *n_launches = idx->prof_launches;
    if (total_ms) *total_ms = idx->prof_ms;
    if (reset) {
        idx->prof_ms = 0;
        idx->prof_launches = 0;
    }
    return 0;
}

}  // extern "C"
