// k > 32: windows whose k-mer does not fit one 64-bit word.  The reference's search is k-agnostic (it walks the query
// characters, src/fms_index.h:117-124), so `fmsi query` must be too; the fast kernels of fmsi_hip.hip / fmsi_pair.hip
// keep a packed k-mer in a register pair and stop at k = 32.  Here a lane walks its window straight off the 2-bit packed
// sequence in HBM: the forward strand right-to-left (bases p+k-1 .. p), the reverse-complement strand left-to-right with
// complemented bases (bases p .. p+k-1) -- base t of the reverse complement is 3 - S[p + k-1-t].  One 64-bit word of
// the sequence (32 bases) is cached per lane, so the sequence costs one (coalesced, neighbouring lanes share it) read
// per 32 LF steps; every LF step is the same 64-B block read(s) as everywhere else (device_fm.h: lf_step).
// Lock-step waves: neighbouring lanes hold neighbouring windows of a read, whose searches end at about the same depth.
// The kernel is a template over the handle's view: 32-bit rows, or 64-bit rows (indexes of 2^32 rows and more, device_fm_wide.h).
#include <hip/hip_runtime.h>

#include "../../include/fmsi_hip.h"
#include "common.h"
#include "device_fm.h"
#include "device_fm_wide.h"
#include "index_priv.h"

namespace fmsi {
namespace {

struct WideCounters {
    uint32_t lf_steps = 0, blocks = 0, probes = 0, searches = 0;
};

struct SeqWord {   // one cached word of the packed sequence
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

// get_range_with_pattern (src/fms_index.h:117-124) + infer_presence (:126-144) / kmer_order_if_present (:146-156)
template <class IX>
__device__ __forceinline__ int search_window_strand(const IX& ix, SeqWord& sw, uint64_t p, int k, bool rc, uint32_t mode_all,
                                                    uint32_t want_orders, long long& order, WideCounters& cn) {
    typename RowOf<IX>::type i = 0, j = ix.n;
    cn.searches++;
    for (int t = k - 1; t >= 0 && i != j; --t) {
        const uint32_t c = rc ? 3u - sw.base(p + static_cast<uint64_t>(k - 1 - t)) : sw.base(p + static_cast<uint64_t>(t));
        view_lf_step(ix, c, i, j, cn.blocks);
        cn.lf_steps++;
    }
    order = -1;
    if (i == j) return -1;
    cn.probes++;
    if (mode_all && !want_orders) {
        cn.blocks++;
        return static_cast<int>(view_mask_bit(ix, i));
    }
    uint64_t ri = 0;
    const int any = view_ones_in(ix, i, j, cn.blocks, &ri) != 0;
    if (want_orders && any) order = static_cast<long long>(ri);
    return any;
}

template <bool COUNT, class IX>
__global__ __launch_bounds__(256) void query_longk_kernel(IX ix, const uint64_t* __restrict__ seq,
                                                          const uint64_t* __restrict__ window_start, uint64_t n_windows, int k,
                                                          uint32_t flags, uint8_t* __restrict__ out, long long* __restrict__ out_orders,
                                                          unsigned long long* __restrict__ counters) {
    const uint32_t mode_all = flags & FMSI_HIP_MODE_ALL;
    const uint32_t strict = flags & FMSI_HIP_STRICT;
    const uint32_t orders = flags & FMSI_HIP_ORDERS;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    WideCounters cn;
    uint32_t done = 0;
    SeqWord sw{seq};
    for (uint64_t q = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; q < n_windows; q += stride) {
        const uint64_t p = window_start ? window_start[q] : q;
        long long ord_f = -1, ord_r = -1;
        const int res_f = search_window_strand(ix, sw, p, k, false, mode_all, orders, ord_f, cn);
        int res_r = -1;
        // second strand needed?  src/fms_index.h:283-299 with the forward strand first (no predictor swap)
        const bool second = strict ? true : (orders ? ord_f < 0 : (mode_all ? res_f == -1 : res_f != 1));
        if (second) res_r = search_window_strand(ix, sw, p, k, true, mode_all, orders, ord_r, cn);
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
        ++done;
    }
    if (COUNT) {
        uint32_t v[5] = {cn.lf_steps, cn.blocks, cn.probes, cn.searches, done};
        const int slot[5] = {0, 1, 2, 3, 6};
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            uint32_t x = v[s];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if ((threadIdx.x & 63) == 0 && x) atomicAdd(&counters[slot[s]], static_cast<unsigned long long>(x));
        }
    }
}

}  // namespace

void launch_longk_query(fmsi_hip_index* idx, const uint64_t* d_seq, const uint64_t* d_window_start, uint64_t n, int k, uint32_t flags,
                        uint8_t* o8, long long* o64, unsigned long long* d_counters, hipStream_t s) {
    const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>((n + 255) / 256, 256ull * 32));
    if (idx->wide) {
        if (d_counters)
            hipLaunchKernelGGL((query_longk_kernel<true, DevIndexW>), dim3(blocks), dim3(256), 0, s, idx->devw, d_seq, d_window_start, n, k, flags, o8, o64, d_counters);
        else
            hipLaunchKernelGGL((query_longk_kernel<false, DevIndexW>), dim3(blocks), dim3(256), 0, s, idx->devw, d_seq, d_window_start, n, k, flags, o8, o64, d_counters);
    } else if (d_counters) {
        hipLaunchKernelGGL((query_longk_kernel<true, DevIndex>), dim3(blocks), dim3(256), 0, s, idx->dev, d_seq, d_window_start, n, k, flags, o8, o64, d_counters);
    } else {
        hipLaunchKernelGGL((query_longk_kernel<false, DevIndex>), dim3(blocks), dim3(256), 0, s, idx->dev, d_seq, d_window_start, n, k, flags, o8, o64, d_counters);
    }
}

}  // namespace fmsi
