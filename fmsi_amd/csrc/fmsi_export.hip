// `fmsi export` on the device: counterpart of export_ms() (reference src/fms_index.h:462-473), SURVEY.md 8(f).4.
//
// The reference walks ONE chain: row 0 (the empty suffix) -> LF -> ... N steps, writing the text back to front
// (letter = access(row), case = mask bit of the row it lands on).  The walk is a single path through all rows, so it
// can be cut anywhere: every S-th row starts a chain that walks until it lands on another start row (or on the '$'
// row, the end of the path).  Chain lengths are geometric with mean S because consecutive text positions sit at
// unrelated rows.  Pass 1 measures every chain (length, landing row); the chains form one linked list from row 0, and
// their offsets in the text are a list ranking -- pointer jumping on the device, log2(chains) rounds over two small
// arrays --; pass 2 walks again and writes the letters.  2 N dependent 64-byte block reads spread over millions of lanes
// instead of N on one core.  With the ranking on the device the chains can be short (mean 256 steps at human scale,
// 11 million of them): the longest chain, which every pass has to wait for, is about ln(chains) mean lengths -- the dictionary's
// text at human scale in 0.20 s instead of 0.28 s.  The kernels are templates over the handle's view: 32-bit rows (DevIndex) or, for indexes of
// 2^32 rows and more, 64-bit rows over the same blocks (DevIndexW, device_fm_wide.h) -- chains are numbered in 32 bits either way (at most
// 2^24 of them).  (Measured and dropped: persistent lanes that take the next chain when theirs ends,
// ballot + mbcnt as in the refill kernel -- 0.21 s: the walk is bound by the dependent reads, not by lanes idling in their waves.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/fmsi_hip.h"
#include "common.h"
#include "device_fm.h"
#include "device_fm_wide.h"
#include "index_priv.h"
#include "trace.h"

namespace fmsi {
namespace {

template <class ROW>
struct StepOut {
    uint32_t letter, mask_at_next;
    ROW next_row;
};

// access() + one LF step + the mask bit of the landing row (src/fms_index.h:88-95, 98-103, 468-470)
__device__ __forceinline__ StepOut<uint32_t> export_step(const DevIndex& ix, uint32_t row) {
    const uint32_t* blk = reinterpret_cast<const uint32_t*>(ix.blocks) + (static_cast<size_t>(row >> kBlockRowsLog2) << 4);
    const uint32_t o = row & (kBlockRows - 1u), w = o >> 5, sh = o & 31u;
    StepOut<uint32_t> s;
    s.letter = (((blk[4 + w] >> sh) & 1u) << 1) | ((blk[8 + w] >> sh) & 1u);
    uint32_t a = row, b = row + 1u, dummy = 0;
    lf_step(ix, s.letter, a, b, dummy);
    s.next_row = a;
    s.mask_at_next = mask_bit(ix, a);
    return s;
}
__device__ __forceinline__ StepOut<uint64_t> export_step(const DevIndexW& ix, uint64_t row) {
    const uint32_t* blk = reinterpret_cast<const uint32_t*>(ix.blocks) + ((row >> kBlockRowsLog2) << 4);
    const uint32_t o = static_cast<uint32_t>(row) & (kBlockRows - 1u), w = o >> 5, sh = o & 31u;
    StepOut<uint64_t> s;
    s.letter = (((blk[4 + w] >> sh) & 1u) << 1) | ((blk[8 + w] >> sh) & 1u);
    uint64_t a = row, b = row + 1u;
    uint32_t dummy = 0;
    lf_step_w(ix, s.letter, a, b, dummy);
    s.next_row = a;
    const uint32_t* m = reinterpret_cast<const uint32_t*>(ix.blocks) + ((a >> kBlockRowsLog2) << 4) + 12u;
    s.mask_at_next = (m[(static_cast<uint32_t>(a) & (kBlockRows - 1u)) >> 5] >> (a & 31u)) & 1u;
    return s;
}

// chain c starts at row c * stride; it ends when it lands on a start row or on the '$' row
template <class IX>
__global__ __launch_bounds__(256) void export_measure_kernel(IX ix, uint32_t stride, uint64_t n_chains, uint32_t* __restrict__ len,
                                                             typename RowOf<IX>::type* __restrict__ landing) {
    typedef typename RowOf<IX>::type Row;
    const uint64_t c = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= n_chains) return;
    Row row = static_cast<Row>(c * stride);
    uint32_t steps = 0;
    if (row != ix.dollar) {
        do {
            row = export_step(ix, row).next_row;
            ++steps;
        } while (row != ix.dollar && row % stride != 0u);
    }
    len[c] = steps;
    landing[c] = row;
}

// ---- list ranking: offset of every chain along the walk ------------------------------------------------------------------
constexpr uint32_t kListEnd = 0xFFFFFFFFu;
// nxt[c] = the chain that follows c on the walk (the one starting at c's landing row), dist[c] = len[c]
template <class ROW>
__global__ __launch_bounds__(256) void rank_init_kernel(const uint32_t* __restrict__ len, const ROW* __restrict__ landing, uint32_t stride, ROW dollar,
                                                        uint64_t n_chains, uint32_t* __restrict__ nxt, uint64_t* __restrict__ dist) {
    const uint64_t c = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= n_chains) return;
    const ROW row = landing[c];
    nxt[c] = row == dollar ? kListEnd : static_cast<uint32_t>(row / stride);
    dist[c] = len[c];
}
// one round of pointer jumping: dist[c] becomes the steps from c's start to the start of nxt[c]'s successor, ... after
// ceil(log2(chains)) rounds to the end of the walk
__global__ __launch_bounds__(256) void rank_jump_kernel(const uint32_t* __restrict__ nxt_in, const uint64_t* __restrict__ dist_in, uint64_t n_chains,
                                                        uint32_t* __restrict__ nxt_out, uint64_t* __restrict__ dist_out) {
    const uint64_t c = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= n_chains) return;
    const uint32_t n = nxt_in[c];
    if (n == kListEnd || n >= n_chains) {
        nxt_out[c] = kListEnd;
        dist_out[c] = dist_in[c];
    } else {
        nxt_out[c] = nxt_in[n];
        dist_out[c] = dist_in[c] + dist_in[n];
    }
}
// dist[c] = steps from c's start to the end of the walk  ->  last[c] = text position of the chain's first letter
__global__ __launch_bounds__(256) void rank_last_kernel(const uint64_t* __restrict__ dist, uint64_t n_chains, uint64_t text_len, uint64_t* __restrict__ last) {
    const uint64_t c = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= n_chains) return;
    last[c] = dist[c] - 1u;   // offset = text_len - dist[c]; last = text_len - 1 - offset
    (void)text_len;
}

// the chain's i-th letter is text position  last[c] - i  (last[c] = N - 1 - offset of the chain along the walk)
template <class IX>
__global__ __launch_bounds__(256) void export_write_kernel(IX ix, uint32_t stride, uint64_t n_chains, const uint32_t* __restrict__ len,
                                                           const uint64_t* __restrict__ last, char* __restrict__ out,
                                                           uint32_t* __restrict__ sa /* optional (32-bit rows): sa[row] = text position of the row's suffix */) {
    typedef typename RowOf<IX>::type Row;
    const uint64_t c = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= n_chains) return;
    const uint32_t steps = len[c];
    Row row = static_cast<Row>(c * stride);
    uint64_t p = last[c];
    for (uint32_t i = 0; i < steps; ++i, --p) {
        const StepOut<Row> s = export_step(ix, row);
        // masked_letters = "acgtACGT", src/fms_index.h:463
        const uint32_t lower = s.letter == 0u ? 'a' : (s.letter == 1u ? 'c' : (s.letter == 2u ? 'g' : 't'));
        out[p] = static_cast<char>(s.mask_at_next ? lower - 32u : lower);
        if (sa) sa[row] = static_cast<uint32_t>(p + 1u);   // the letter before the suffix of `row` sits at p
        row = s.next_row;
    }
}

}  // namespace
}  // namespace fmsi

using namespace fmsi;

namespace fmsi {

// The walk: letters (with the mask's case) into d_chars[0 .. n-2], and optionally sa[row] for every row (sa[dollar row] = 0).
template <class IX>
static int export_walk_t(const fmsi_hip_index* idx, const IX& view, char* d_chars, uint32_t* d_sa) {
    typedef typename RowOf<IX>::type Row;
    const uint64_t n = idx->n, text_len = n - 1;
    if (text_len == 0) return 0;
    TraceRange trace_range("fmsi: LF walk (export)");
    // at most 2^24 chains, 16 rows apart or more (human scale: 256 rows, 11 million chains; FMSI_HIP_EXPORT_STRIDE: another distance, for A/B runs)
    uint64_t stride64 = 16;
    if (const char* e = std::getenv("FMSI_HIP_EXPORT_STRIDE")) stride64 = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
    while ((n + stride64 - 1) / stride64 > (1ull << 24)) stride64 *= 2;
    const uint32_t stride = static_cast<uint32_t>(stride64);
    const uint64_t n_chains = (n + stride64 - 1) / stride64;
    uint32_t *d_len = nullptr, *d_nxt[2] = {nullptr, nullptr};
    Row* d_landing = nullptr;
    uint64_t *d_last = nullptr, *d_dist[2] = {nullptr, nullptr};
    auto cleanup = [&]() {
        for (void* p : {static_cast<void*>(d_len), static_cast<void*>(d_landing), static_cast<void*>(d_last), static_cast<void*>(d_nxt[0]), static_cast<void*>(d_nxt[1]),
                        static_cast<void*>(d_dist[0]), static_cast<void*>(d_dist[1])})
            if (p) (void)dev_free(p);
    };
    auto fail = [&](int code, const std::string& msg) {
        cleanup();
        set_last_error(msg);
        return code;
    };
    if (dev_malloc(reinterpret_cast<void**>(&d_len), n_chains * 4) != hipSuccess || dev_malloc(reinterpret_cast<void**>(&d_landing), n_chains * sizeof(Row)) != hipSuccess ||
        dev_malloc(reinterpret_cast<void**>(&d_last), n_chains * 8) != hipSuccess || dev_malloc(reinterpret_cast<void**>(&d_nxt[0]), n_chains * 4) != hipSuccess ||
        dev_malloc(reinterpret_cast<void**>(&d_nxt[1]), n_chains * 4) != hipSuccess || dev_malloc(reinterpret_cast<void**>(&d_dist[0]), n_chains * 8) != hipSuccess ||
        dev_malloc(reinterpret_cast<void**>(&d_dist[1]), n_chains * 8) != hipSuccess)
        return fail(FMSI_HIP_ENOMEM, "hipMalloc failed for the export buffers");
    const dim3 grid(static_cast<uint32_t>((n_chains + 255) / 256)), block(256);
    hipLaunchKernelGGL((export_measure_kernel<IX>), grid, block, 0, 0, view, stride, n_chains, d_len, d_landing);
    // rank the list of chains: distance of every chain's start from the end of the walk
    hipLaunchKernelGGL((rank_init_kernel<Row>), grid, block, 0, 0, d_len, d_landing, stride, static_cast<Row>(view.dollar), n_chains, d_nxt[0], d_dist[0]);
    int cur = 0;
    for (uint64_t reach = 1; reach < n_chains; reach *= 2, cur ^= 1)
        hipLaunchKernelGGL(rank_jump_kernel, grid, block, 0, 0, d_nxt[cur], d_dist[cur], n_chains, d_nxt[cur ^ 1], d_dist[cur ^ 1]);
    hipLaunchKernelGGL(rank_last_kernel, grid, block, 0, 0, d_dist[cur], n_chains, text_len, d_last);
    // the walk from row 0 (chain 0) must be the whole text, every chain on it: then the distances are a permutation-free cover
    uint64_t head = 0;
    uint32_t head_next = 0;
    if (hipGetLastError() != hipSuccess || hipMemcpy(&head, d_dist[cur], 8, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(&head_next, d_nxt[cur], 4, hipMemcpyDeviceToHost) != hipSuccess)
        return fail(FMSI_HIP_EHIP, "export: measuring the chains failed");
    if (head != text_len || head_next != kListEnd) return fail(FMSI_HIP_EHIP, "export: the LF walk from row 0 does not cover the text (corrupt index?)");
    if (d_sa && hipMemset(d_sa + view.dollar, 0, 4) != hipSuccess) return fail(FMSI_HIP_EHIP, "export: memset failed");
    hipLaunchKernelGGL((export_write_kernel<IX>), grid, block, 0, 0, view, stride, n_chains, d_len, d_last, d_chars, d_sa);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(FMSI_HIP_EHIP, "export: writing the text failed");
    cleanup();
    return 0;
}

int export_walk(const fmsi_hip_index* idx, char* d_chars, uint32_t* d_sa) {
    if (!idx->wide) return export_walk_t(idx, idx->dev, d_chars, d_sa);
    if (d_sa) {
        set_last_error("export: row positions are kept in 32 bits (an index of 2^32 rows or more has none)");
        return FMSI_HIP_ERANGE;
    }
    return export_walk_t(idx, idx->devw, d_chars, nullptr);
}

}  // namespace fmsi

extern "C" int fmsi_hip_index_export(const fmsi_hip_index* idx, char* out, uint64_t out_capacity, uint64_t* out_len) {
    ::fmsi::ScratchScope scratch_scope_(idx ? idx->device : -1);   // (devpool.h: the outermost builder returns the block cache to the driver)
    if (!idx || !out || !out_len) {
        set_last_error("null argument");
        return FMSI_HIP_EINVAL;
    }
    const uint64_t text_len = idx->n - 1;
    *out_len = text_len;
    if (out_capacity < text_len) {
        set_last_error("output buffer too small: the superstring has " + std::to_string(text_len) + " bases");
        return FMSI_HIP_EINVAL;
    }
    if (text_len == 0) return 0;
    FMSI_HIP_ON_DEVICE(idx->device);
    char* d_out = nullptr;
    if (dev_malloc(reinterpret_cast<void**>(&d_out), text_len) != hipSuccess) {
        set_last_error("hipMalloc failed for the export buffers");
        return FMSI_HIP_ENOMEM;
    }
    int rc = export_walk(idx, d_out, nullptr);
    if (rc == 0 && hipMemcpy(out, d_out, text_len, hipMemcpyDeviceToHost) != hipSuccess) {
        set_last_error("export: copying the text back failed");
        rc = FMSI_HIP_EHIP;
    }
    (void)dev_free(d_out);
    return rc;
}
