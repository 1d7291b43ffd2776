// `fmsi query -f and|xor|r-s`: f-masked superstrings (the reference's query_mode::general, src/fms_index.h:316-327 with
// single_query_general(), :171-179, and the demasking functions of src/functions.h:8-19).
//
// A window's answer is f(ones, total) over BOTH strands' SA intervals: total = the number of occurrences of the k-mer and of its
// reverse complement in the superstring, ones = how many of them carry an ON mask bit; a self-complementary k-mer is counted once
// (:319).  The strand predictor is consulted but never told anything in this mode and the two strands are summed, so -- unlike the
// presence modes -- nothing depends on the order of the windows: no replay, every window independent.
//
// "Counted once" is the reference's `!AreStringsEqual(kmer, rc_kmer, k)` on the query's CHARACTERS, and its reverse complement keeps
// the case of every character (src/kmers.h:21-38, 53-68): a self-complementary k-mer typed with asymmetric case -- acGT, whose reverse
// complement reads ACgt -- is not equal to it as a string and IS counted twice (which flips an xor and doubles the count a range
// looks at).  The 2-bit sequence has no case, so the caller hands over one bit per base (lower case or not) next to it.
//
// The search is the k-agnostic walk of fmsi_longk.hip: a lane walks its window straight off the 2-bit packed sequence, the forward
// strand right-to-left, the reverse complement left-to-right with complemented bases, starting from the prefix table's interval
// when the handle has one; the ON bits of an interval [i, j) are two mask ranks (block count + popcount below the row: the
// mask plane of the interleaved block), where the reference adds up j - i single accesses of its RRR vector.
// The kernel is a template over the handle's view: 32-bit rows, or 64-bit rows (indexes of 2^32 rows and more, device_fm_wide.h).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "../../include/fmsi_hip.h"
#include "common.h"
#include "device_fm.h"
#include "device_fm_wide.h"
#include "index_priv.h"

namespace fmsi {
namespace {

struct SeqCache {   // one cached word of the packed sequence
    const uint64_t* seq;
    uint64_t idx = ~0ull, word = 0;
    __device__ __forceinline__ uint32_t base(uint64_t p) {
        const uint64_t w = p >> 5;
        if (w != idx) {
            word = seq[w];
            idx = w;
        }
        return static_cast<uint32_t>(word >> (2u * static_cast<uint32_t>(p & 31u))) & 3u;
    }
};

struct FnCounters {
    uint32_t lf_steps = 0, blocks = 0, probes = 0, searches = 0, lookups = 0;
};

// get_range_with_pattern (src/fms_index.h:117-124) of one strand of the window at base p
template <class IX>
__device__ __forceinline__ void strand_range(const IX& ix, SeqCache& sc, uint64_t p, int k, bool rc, typename RowOf<IX>::type& i, typename RowOf<IX>::type& j,
                                             FnCounters& cn) {
    cn.searches++;
    const int T = static_cast<int>(ix.ptable_len);
    int t = k - 1;
    i = 0;
    j = ix.n;
    if (view_has_table(ix, k)) {
        // the T bases the search consumes first: the window's last T (forward) or its first T, complemented and reversed
        const uint64_t w = window_kmer(sc.seq, rc ? p : p + static_cast<uint64_t>(k - T), T);
        view_table_entry(ix, rc ? revcomp_2bit(w, T) : w, i, j);
        cn.lookups++;
        t = k - T - 1;
    }
    for (; t >= 0 && i != j; --t) {
        const uint32_t c = rc ? 3u - sc.base(p + static_cast<uint64_t>(k - 1 - t)) : sc.base(p + static_cast<uint64_t>(t));
        view_lf_step(ix, c, i, j, cn.blocks);
        cn.lf_steps++;
    }
}

// ON bits among the rows [i, j), i < j
template <class IX, class ROW>
__device__ __forceinline__ uint64_t ones_in(const IX& ix, ROW i, ROW j, FnCounters& cn) {
    cn.probes++;
    return view_ones_in(ix, i, j, cn.blocks);
}

// f(ones, total) the way query_kmers_single<general> reaches it: through std::function<bool(int, int)> wrappers, i.e. with both
// counts truncated to int and widened again (src/functions.h:21, src/main.cpp:251, src/fms_index.h:334)
__device__ __forceinline__ bool apply_fn(int fn, uint64_t r, uint64_t s, uint64_t ones, uint64_t total) {
    const uint64_t o = static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(static_cast<uint32_t>(ones))));
    const uint64_t t = static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(static_cast<uint32_t>(total))));
    if (fn == FMSI_HIP_FN_AND) return t != 0 && o == t;   // f_and, src/functions.h:11
    if (fn == FMSI_HIP_FN_XOR) return (o & 1u) != 0;      // f_xor, :14
    return o <= s && o >= r;                              // f_r_to_s, :17-19
}

// is the case of the window's characters symmetric (character t like character k-1-t)?  lower: bit p = base p is lower case
__device__ __forceinline__ bool case_symmetric(const uint64_t* __restrict__ lower, uint64_t p, int k) {
    for (int t = 0; t < k / 2; ++t) {
        const uint64_t a = p + static_cast<uint64_t>(t), b = p + static_cast<uint64_t>(k - 1 - t);
        if (((lower[a >> 6] >> (a & 63u)) ^ (lower[b >> 6] >> (b & 63u))) & 1u) return false;
    }
    return true;
}

template <bool COUNT, class IX>
__global__ __launch_bounds__(256) void query_fn_kernel(IX ix, const uint64_t* __restrict__ seq, const uint64_t* __restrict__ lower,
                                                       const uint64_t* __restrict__ window_start, uint64_t n_windows, int k, int fn, uint64_t fn_r,
                                                       uint64_t fn_s, uint8_t* __restrict__ out, unsigned long long* __restrict__ counters) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    FnCounters cn;
    uint32_t done = 0;
    SeqCache sc{seq};
    for (uint64_t q = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; q < n_windows; q += stride) {
        const uint64_t p = window_start[q];
        typename RowOf<IX>::type fi, fj, ri, rj;
        strand_range(ix, sc, p, k, false, fi, fj, cn);
        strand_range(ix, sc, p, k, true, ri, rj, cn);
        uint64_t ones = 0, total = 0;
        if (fi != fj) {
            ones = ones_in(ix, fi, fj, cn);
            total = fj - fi;
        }
        // "Do not count self complementary k-mers twice" (:319): equal k-mers have equal intervals, and different k-mers that both
        // occur have different ones; two empty intervals add nothing either way.  Equal k-mers in asymmetric case are different
        // strings to the reference, though (see the top of this file).
        if (ri != rj) {
            if (!(ri == fi && rj == fj)) {
                ones += ones_in(ix, ri, rj, cn);
                total += rj - ri;
            } else if (lower && !case_symmetric(lower, p, k)) {
                ones += ones;
                total += total;
            }
        }
        out[q] = apply_fn(fn, fn_r, fn_s, ones, total) ? 1 : 0;
        ++done;
    }
    if (COUNT) {
        uint32_t v[6] = {cn.lf_steps, cn.blocks, cn.probes, cn.searches, cn.lookups, done};
        const int slot[6] = {0, 1, 2, 3, 4, 6};
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            uint32_t x = v[s];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if ((threadIdx.x & 63) == 0 && x) atomicAdd(&counters[slot[s]], static_cast<unsigned long long>(x));
        }
    }
}

}  // namespace

void launch_fn_query(fmsi_hip_index* idx, const uint64_t* d_seq, const uint64_t* d_window_start, uint64_t n, int k, bool use_table, uint8_t* o8,
                     unsigned long long* d_counters, hipStream_t s) {
    const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>((n + 255) / 256, 256ull * 32));
    auto go = [&](auto view) {
        if (!use_table) view.ptable = nullptr;
        typedef decltype(view) IX;
        if (d_counters)
            hipLaunchKernelGGL((query_fn_kernel<true, IX>), dim3(blocks), dim3(256), 0, s, view, d_seq, idx->fn_case, d_window_start, n, k, idx->fn_kind, idx->fn_r,
                               idx->fn_s, o8, d_counters);
        else
            hipLaunchKernelGGL((query_fn_kernel<false, IX>), dim3(blocks), dim3(256), 0, s, view, d_seq, idx->fn_case, d_window_start, n, k, idx->fn_kind, idx->fn_r,
                               idx->fn_s, o8, d_counters);
    };
    if (idx->wide) go(idx->devw);
    else go(idx->dev);
}

}  // namespace fmsi

using namespace fmsi;

extern "C" int fmsi_hip_query_runs_fn(fmsi_hip_index* idx, const uint64_t* packed_seq, const uint64_t* lower_case, uint64_t seq_bases,
                                      const uint64_t* run_start, const uint32_t* run_windows, uint64_t n_runs, uint64_t n_windows, int k, int fn, uint64_t r,
                                      uint64_t s, uint32_t flags, void* out, uint64_t* counters, void* stream) {
    if (!idx) return FMSI_HIP_EINVAL;
    if (fn != FMSI_HIP_FN_AND && fn != FMSI_HIP_FN_XOR && fn != FMSI_HIP_FN_RANGE) {
        set_last_error("unknown demasking function");
        return FMSI_HIP_EINVAL;
    }
    if (flags & (FMSI_HIP_STRICT | FMSI_HIP_ORDERS)) {
        set_last_error("FMSI_HIP_STRICT / FMSI_HIP_ORDERS have no meaning with a demasking function");   // src/main.cpp:297-300
        return FMSI_HIP_EINVAL;
    }
    idx->fn_kind = fn;
    idx->fn_r = r;
    idx->fn_s = s;
    idx->fn_case = nullptr;
    if (lower_case && k % 2 == 0 && n_windows) {   // (no k-mer of odd length is its own reverse complement)
        if (flags & FMSI_HIP_DEVICE_PTRS) {
            idx->fn_case = lower_case;
        } else {
            FMSI_HIP_ON_DEVICE(idx->device);
            const size_t bytes = (seq_bases / 64 + 1) * 8;
            int rc = ensure_cap(&idx->d_case, &idx->d_case_cap, bytes);
            if (rc) return rc;
            FMSI_HIP_TRY(hipMemcpyAsync(idx->d_case, lower_case, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
            idx->fn_case = static_cast<const uint64_t*>(idx->d_case);
        }
    }
    return fmsi_hip_query_runs(idx, packed_seq, seq_bases, run_start, run_windows, n_runs, n_windows, k,
                               (flags & (FMSI_HIP_DEVICE_PTRS | FMSI_HIP_OUT_BITS | FMSI_HIP_NO_PREFIX_TABLE)) | kModeFn, out, counters, stream);
}
