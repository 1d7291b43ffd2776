// 64-bit rows: indexes of 2^32 rows and more (the reference's intervals are size_t, src/fms_index.h:98-103).
//
// Every configuration of BASELINE.json has n < 2^32, and the fast kernels (fmsi_hip.hip, fmsi_pair.hip, fmsi_stream.hip,
// fmsi_dict.hip) keep rows, counts and table entries in 32 bits.  A larger index -- a pangenome superstring beyond 4.29 G bases --
// loads into the SAME 64-byte blocks: their u32 cumulative counts simply wrap, and a small table of absolute u64 counts at every
// superblock start (2^24 blocks = 2^31 rows) restores them: count = super + (u32)(block count - (u32)super), exact as long as a
// superblock spans < 2^32 rows.  The search below is the lock-step form of query_lockstep_kernel with u64 rows -- update_range()
// (src/fms_index.h:98-103), get_range_with_pattern() (:117-124), infer_presence() (:126-144), kmer_order_if_present()
// (:146-156) -- for isolated k-mers and, through the library's window / run expansion, for windows and runs; presence modes,
// STRICT, ORDERS and FMSI_HIP_OUT_BITS as everywhere.  A prefix table of plain 16-byte [i, j) entries, at most 14 deep, gives the search its
// first bases in one read (2.6 x on a 4.5 x 10^9-row index).  k > 32 and the demasking functions (-f) take the k-agnostic walks of
// fmsi_longk.hip / fmsi_fn.hip, which are templates over the row width.  Not built for such handles: pair layout, candidate tables,
// dictionaries and the kLCP kernel (FMSI_HIP_ERANGE says so); export walks 64-bit rows (fmsi_export.hip, device_fm_wide.h), and the
// CLI answers -S through this search (the kLCP array is the reference's short cut: the answers are the same).  Such handles come from index files (the host path) or from
// the on-device builder for 64-bit rows (build_index.hip, build_wide_from_device_text: a 4.5 x 10^9-base index is built and queried
// against the oracle in tests/test_gpu_wide.py).
// FMSI_HIP_FORCE_ROWS64 (environment, at fmsi_hip_index_create) sends any index through this path with superblocks of 8 blocks,
// which is how tests/test_gpu_wide.py holds it against the oracle on the golden indexes.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/fmsi_hip.h"
#include "common.h"
#include "device_fm.h"
#include "index_priv.h"
#include "device_fm_wide.h"

namespace fmsi {

namespace {

struct WideCounters {
    uint32_t lf_steps = 0, blocks = 0, probes = 0, searches = 0, lookups = 0;
};

// T: the depth of the prefix table to start from (0: none) -- the interval of the pattern's last T bases is one 16-byte read
__device__ __forceinline__ int search_strand_w(const DevIndexW& ix, uint64_t cur, int k, int T, uint32_t mode_all, uint32_t want_orders, long long& order,
                                               WideCounters& cn) {
    uint64_t i = 0, j = ix.n;
    int t = k - 1;
    cn.searches++;
    if (T > 0) {
        const ulonglong2 e = ix.ptable[(cur >> (2 * (k - T))) & ((1ull << (2 * T)) - 1ull)];
        i = e.x;
        j = e.y;
        t -= T;
        cn.lookups++;
        cn.blocks++;
    }
    for (; t >= 0 && i != j; --t) {
        lf_step_w(ix, static_cast<uint32_t>(cur >> (2 * t)) & 3u, i, j, cn.blocks);
        cn.lf_steps++;
    }
    order = -1;
    if (i == j) return -1;
    cn.probes++;
    cn.blocks++;
    if (mode_all && !want_orders) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(ix.blocks) + ((i >> kBlockRowsLog2) << 4) + 12u;
        return static_cast<int>((w[(i & (kBlockRows - 1u)) >> 5] >> (i & 31u)) & 1u);
    }
    const uint64_t bi = i >> kBlockRowsLog2, bj = (j - 1u) >> kBlockRowsLog2;
    const uint64_t ri = mask_rank_w(ix, bi, static_cast<uint32_t>(i & (kBlockRows - 1u)));
    const uint64_t rj = mask_rank_w(ix, bj, static_cast<uint32_t>(j - (bj << kBlockRowsLog2)));
    const int any = rj > ri;
    if (want_orders && any) order = static_cast<long long>(ri);
    return any;
}

template <bool COUNT>
__global__ __launch_bounds__(256) void query_wide_kernel(DevIndexW ix, const uint64_t* __restrict__ kmers, uint64_t n_windows, int k, uint32_t flags,
                                                         uint8_t* __restrict__ out, long long* __restrict__ out_orders,
                                                         unsigned long long* __restrict__ counters) {
    const uint64_t q = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const bool valid = q < n_windows;
    const uint32_t mode_all = flags & FMSI_HIP_MODE_ALL, strict = flags & FMSI_HIP_STRICT, orders = flags & FMSI_HIP_ORDERS;
    const int T = (flags & FMSI_HIP_NO_PREFIX_TABLE) || k < static_cast<int>(ix.ptable_len) ? 0 : static_cast<int>(ix.ptable_len);
    WideCounters cn;
    if (valid) {
        const uint64_t f = kmers[q];
        long long ord_f = -1, ord_r = -1;
        const int res_f = search_strand_w(ix, f, k, T, mode_all, orders, ord_f, cn);
        int res_r = -1;
        // second strand: src/fms_index.h:283-299 with the forward strand first
        const bool second = strict ? true : (orders ? ord_f < 0 : (mode_all ? res_f == -1 : res_f != 1));
        if (second) res_r = search_strand_w(ix, revcomp_2bit(f, k), k, T, mode_all, orders, ord_r, cn);
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
    if (COUNT) {
        uint32_t v[6] = {cn.lf_steps, cn.blocks, cn.probes, cn.searches, valid ? 1u : 0u, cn.lookups};
        const int slot[6] = {0, 1, 2, 3, 6, 4};
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            uint32_t x = v[s];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if ((threadIdx.x & 63) == 0 && x) atomicAdd(&counters[slot[s]], static_cast<unsigned long long>(x));
        }
    }
}

// one level of the prefix table: entry idx of level L = LF step with base (idx & 3) on entry idx >> 2 of level L - 1 (as ptable_level_kernel, fmsi_hip.hip)
__global__ __launch_bounds__(256) void ptable_level_w_kernel(DevIndexW ix, const ulonglong2* __restrict__ prev, ulonglong2* __restrict__ cur, uint64_t count) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t idx = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < count; idx += stride) {
        ulonglong2 e = prev ? prev[idx >> 2] : make_ulonglong2(0ull, ix.n);
        if (e.x != e.y) {
            uint32_t dummy = 0;
            uint64_t a = e.x, b = e.y;
            lf_step_w(ix, static_cast<uint32_t>(idx & 3u), a, b, dummy);
            e = make_ulonglong2(a, b);
        }
        cur[idx] = e;
    }
}

}  // namespace

// The prefix table of a handle with 64-bit rows: 4^t plain [i, j) pairs of 16 bytes (t <= 14: 4.3 GB), built level by level like the 32-bit one.
int build_prefix_table_wide(fmsi_hip_index* idx, int t) {
    if (t <= 0) return 0;
    t = std::min(t, 14);
    while (t > 1 && (1ull << (2 * t)) > 4 * idx->n) --t;   // (a table far larger than the index answers nothing faster)
    const uint64_t final_count = 1ull << (2 * t);
    ulonglong2 *a = nullptr, *b = nullptr;
    if (dev_malloc(reinterpret_cast<void**>(&a), final_count * sizeof(ulonglong2)) != hipSuccess ||
        (t > 1 && dev_malloc(reinterpret_cast<void**>(&b), (final_count / 4) * sizeof(ulonglong2)) != hipSuccess)) {
        (void)hipGetLastError();
        if (a) (void)dev_free(a);
        set_last_error("hipMalloc failed for the prefix table (" + std::to_string(final_count * sizeof(ulonglong2) >> 20) + " MiB)");
        return FMSI_HIP_ENOMEM;
    }
    ulonglong2* bufs[2] = {a, b};   // levels alternate between the two buffers so that the last level lands in `a`
    int cur = (t % 2 == 1) ? 0 : 1;
    const ulonglong2* prev = nullptr;
    for (int level = 1; level <= t; ++level) {
        const uint64_t count = 1ull << (2 * level);
        const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>((count + 255) / 256, 1u << 22));
        hipLaunchKernelGGL(ptable_level_w_kernel, dim3(blocks), dim3(256), 0, 0, idx->devw, prev, bufs[cur], count);
        prev = bufs[cur];
        cur ^= 1;
    }
    const hipError_t e = hipGetLastError() != hipSuccess ? hipErrorUnknown : hipDeviceSynchronize();
    if (b) (void)dev_free(b);
    if (e != hipSuccess) {
        (void)dev_free(a);
        set_last_error(std::string("prefix table build: ") + hipGetErrorString(e));
        return FMSI_HIP_EHIP;
    }
    idx->d_ptable = a;
    idx->devw.ptable = a;
    idx->devw.ptable_len = static_cast<uint32_t>(t);
    idx->prefix_len = t;
    idx->device_bytes += final_count * sizeof(ulonglong2);
    return 0;
}

// absolute counts at every superblock start from the wrapped u32 counts of the host layout: per block the count of a symbol is
// (next block's cumulative - this block's) mod 2^32 <= 128
void wide_super_table(const uint32_t* blocks, const uint32_t* mrank, uint64_t n_blocks, uint32_t sb_log2, std::vector<uint64_t>& super) {
    const uint64_t n_super = ((n_blocks - 1) >> sb_log2) + 1;
    super.assign(n_super * 8, 0);
    uint64_t run[5] = {0, 0, 0, 0, 0};
    for (uint64_t b = 0; b < n_blocks; ++b) {
        if ((b & ((1ull << sb_log2) - 1)) == 0)
            for (int s = 0; s < 5; ++s) super[(b >> sb_log2) * 8 + s] = run[s];
        if (b + 1 < n_blocks) {
            for (int s = 0; s < 4; ++s) run[s] += static_cast<uint32_t>(blocks[(b + 1) * kBlockDwords + s] - blocks[b * kBlockDwords + s]);
            run[4] += static_cast<uint32_t>(mrank[b + 1] - mrank[b]);
        }
    }
}

int launch_wide_query(fmsi_hip_index* idx, const uint64_t* d_kmers, uint64_t n, int k, uint32_t flags, uint8_t* o8, long long* o64,
                      unsigned long long* d_counters, hipStream_t s) {
    const uint64_t blocks64 = (n + 255) / 256;
    if (blocks64 > 0x7FFFFFFFull) {
        set_last_error("too many windows for one launch");
        return FMSI_HIP_ERANGE;
    }
    const dim3 grid(static_cast<uint32_t>(blocks64)), block(256);
    if (d_counters) hipLaunchKernelGGL((query_wide_kernel<true>), grid, block, 0, s, idx->devw, d_kmers, n, k, flags, o8, o64, d_counters);
    else hipLaunchKernelGGL((query_wide_kernel<false>), grid, block, 0, s, idx->devw, d_kmers, n, k, flags, o8, o64, d_counters);
    FMSI_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace fmsi
