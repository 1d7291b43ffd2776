// On-device index construction: counterpart of construct<T>() (reference src/fms_index.h:397-460) that
// goes from a masked superstring straight to the interleaved device layout, with the suffix sort done
// on the GPU.  It exists so that human-genome-scale synthetic indexes can be made on the box in seconds
// (the reference's qsufsort needs ~1 h and 45 GB there, SURVEY.md 7 hard part 5).
//
// Pipeline (N = |MS|, n = N + 1 rows):
//   1. 2-bit text in HBM (uploaded, or generated in place from a counter-based PRNG for synthetic runs)
//   2. key[p] = the 32 bases starting at p, first base in the top bits (zero padded past the end)
//   3. rocPRIM radix sort of (key, p) pairs -- a library sort, not part of the query hot path
//   4. rows whose key equals a neighbour's key (true repeats longer than 32, or padding ambiguity at the
//      text end) are re-sorted on the host by full suffix comparison.  Random texts have a handful of
//      such rows; highly repetitive inputs make this step slow (GPU prefix doubling is the follow-up)
//   5. one pass over the suffix array emits BWT symbol, SA-transformed mask bit and kLCP bit per row;
//      wave ballots pack them into the block planes; per-block counts are prefix-summed on the device.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fmsi_hip.h"
#include "common.h"
#include "index_priv.h"

namespace fmsi {
namespace {

__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// word w of the synthetic superstring: 32 uniform bases (see fmsi_hip_synthetic_text_word in the header)
__host__ __device__ inline uint64_t synth_word(uint64_t seed, uint64_t w) { return splitmix64(seed ^ (w * 0xD1B54A32D192ED03ull)); }

__global__ void synth_text_kernel(uint64_t* __restrict__ text2, uint64_t n_words, uint64_t len, uint64_t seed) {
    const uint64_t w = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint64_t v = synth_word(seed, w);
    const uint64_t first = w * 32;
    if (first >= len) v = 0;
    else if (first + 32 > len) v &= (1ull << (2 * (len - first))) - 1ull;
    text2[w] = v;
}

__device__ __forceinline__ uint64_t bases_at(const uint64_t* __restrict__ text2, uint64_t p) {   // 32 bases from p, base p in bits 0..1
    const uint64_t w = p >> 5;
    const uint32_t sh = static_cast<uint32_t>(p & 31u) * 2u;
    uint64_t f = text2[w] >> sh;
    if (sh) f |= text2[w + 1] << (64u - sh);
    return f;
}

__device__ __forceinline__ uint64_t to_lex_key(uint64_t f) {   // first base to the top bits: integer order == lexicographic order
    uint64_t x = __brevll(f);
    return ((x & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((x & 0x5555555555555555ull) << 1);
}

__global__ void make_keys_kernel(const uint64_t* __restrict__ text2, uint64_t len, uint64_t* __restrict__ keys, uint32_t* __restrict__ pos) {
    const uint64_t p = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p >= len) return;
    keys[p] = to_lex_key(bases_at(text2, p));   // text2 is zero past `len`, so short suffixes are zero padded
    pos[p] = static_cast<uint32_t>(p);
}

// flags rows of the sorted order whose key equals a neighbour's
__global__ void count_ties_kernel(const uint64_t* __restrict__ keys, uint64_t len, unsigned long long* __restrict__ n_ties) {
    const uint64_t r = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    bool tie = false;
    if (r < len) {
        const uint64_t kk = keys[r];
        tie = (r > 0 && keys[r - 1] == kk) || (r + 1 < len && keys[r + 1] == kk);
    }
    const unsigned long long m = __ballot(tie);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_ties, static_cast<unsigned long long>(__popcll(m)));
}
__global__ void collect_ties_kernel(const uint64_t* __restrict__ keys, uint64_t len, uint32_t* __restrict__ rows,
                                    unsigned long long* __restrict__ cursor) {
    const uint64_t r = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (r >= len) return;
    const uint64_t kk = keys[r];
    if ((r > 0 && keys[r - 1] == kk) || (r + 1 < len && keys[r + 1] == kk)) rows[atomicAdd(cursor, 1ull)] = static_cast<uint32_t>(r);
}

struct U4 {
    uint32_t a, c, g, t;
};
struct U4Plus {
    __host__ __device__ U4 operator()(const U4& x, const U4& y) const { return U4{x.a + y.a, x.c + y.c, x.g + y.g, x.t + y.t}; }
};

// kLCP bit of row r: the suffixes of rows r and r+1 share their first k-1 bases (construct_klcp, src/fms_index.h:357-385;
// rows whose suffix is shorter than k-1 get 0).  One thread per row, one 64-bit ballot per wave.  Kept apart from
// emit_rows_kernel: fused into it, the compare gave run-to-run differing bits on gfx950 for rows >= 2^16
// (tests/test_gpu_builder.py::test_klcp_equals_oracle pins this kernel).
__global__ __launch_bounds__(256) void klcp_kernel(const uint64_t* __restrict__ text2, const uint32_t* __restrict__ sa_sorted,
                                                   uint64_t len, int k, uint64_t* __restrict__ klcp) {
    const uint64_t row = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    bool kbit = false;
    if (row < len) {
        const uint64_t s = row == 0 ? len : sa_sorted[row - 1];
        const uint64_t s2 = sa_sorted[row];
        const uint64_t km1 = static_cast<uint64_t>(k - 1);
        if (km1 <= len && s <= len - km1 && s2 <= len - km1) {
            const uint64_t m = km1 >= 32 ? ~0ull : ((1ull << (2 * km1)) - 1ull);
            kbit = km1 == 0 || ((bases_at(text2, s) ^ bases_at(text2, s2)) & m) == 0;
            if (kbit && km1 > 32) {   // k in 34..64: the reference switches to __uint128_t k-mers (src/main.cpp:229-230)
                const uint64_t m2 = km1 >= 64 ? ~0ull : ((1ull << (2 * (km1 - 32))) - 1ull);
                kbit = ((bases_at(text2, s + 32) ^ bases_at(text2, s2 + 32)) & m2) == 0;
            }
        }
    }
    const uint64_t kk = __ballot(kbit);
    if ((threadIdx.x & 63) == 0 && row <= len) klcp[row >> 6] = kk;
}

// One thread per row.  sa_sorted[r] (r = 0..N-1) are the suffixes 0..N-1 in order; row 0 is the empty suffix.
// 256 threads = 4 waves = 2 blocks of the layout.
__global__ __launch_bounds__(256) void emit_rows_kernel(const uint64_t* __restrict__ text2, const uint64_t* __restrict__ mask_text,
                                                        const uint32_t* __restrict__ sa_sorted, uint64_t len, int k, int all_ones_mask,
                                                        uint32_t* __restrict__ blocks, U4* __restrict__ block_counts,
                                                        uint32_t* __restrict__ mask_counts, uint32_t* __restrict__ dollar_row) {
    __shared__ uint32_t s_cnt[4][5];
    const uint64_t n = len + 1;
    const uint64_t row = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint32_t wave_in_block = threadIdx.x >> 6;
    uint32_t sym = 0, mbit = 0;
    const bool valid = row < n;
    if (valid) {
        const uint64_t s = row == 0 ? len : sa_sorted[row - 1];
        if (s == 0) *dollar_row = static_cast<uint32_t>(row);                         // fms_index.h:414-415
        else sym = static_cast<uint32_t>(text2[(s - 1) >> 5] >> (2 * ((s - 1) & 31))) & 3u;   // :417
        if (s != len) {                                                               // :419-421
            if (all_ones_mask) mbit = s + static_cast<uint64_t>(k) <= len;            // synthetic MS: every k-mer ON
            else mbit = static_cast<uint32_t>(mask_text[s >> 6] >> (s & 63)) & 1u;
        }
    }
    const uint64_t hi = __ballot(valid && (sym & 2u));
    const uint64_t lo = __ballot(valid && (sym & 1u));
    const uint64_t mm = __ballot(valid && mbit);
    const uint64_t vv = __ballot(valid);
    if ((threadIdx.x & 63) == 0) {
        const uint64_t w = row >> 6;          // 64-row word index
        if (vv) {
            uint32_t* blk = blocks + (w >> 1) * kBlockDwords;
            const uint32_t half = static_cast<uint32_t>(w & 1) * 2;
            blk[4 + half] = static_cast<uint32_t>(hi);
            blk[5 + half] = static_cast<uint32_t>(hi >> 32);
            blk[8 + half] = static_cast<uint32_t>(lo);
            blk[9 + half] = static_cast<uint32_t>(lo >> 32);
            blk[12 + half] = static_cast<uint32_t>(mm);
            blk[13 + half] = static_cast<uint32_t>(mm >> 32);
        }
        s_cnt[wave_in_block][0] = __popcll(vv & ~hi & ~lo);
        s_cnt[wave_in_block][1] = __popcll(vv & ~hi & lo);
        s_cnt[wave_in_block][2] = __popcll(hi & ~lo);
        s_cnt[wave_in_block][3] = __popcll(hi & lo);
        s_cnt[wave_in_block][4] = __popcll(mm);
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const uint64_t b = static_cast<uint64_t>(blockIdx.x) * 2 + threadIdx.x;
        const uint32_t w0 = threadIdx.x * 2, w1 = w0 + 1;
        block_counts[b] = U4{s_cnt[w0][0] + s_cnt[w1][0], s_cnt[w0][1] + s_cnt[w1][1], s_cnt[w0][2] + s_cnt[w1][2],
                             s_cnt[w0][3] + s_cnt[w1][3]};
        mask_counts[b] = s_cnt[w0][4] + s_cnt[w1][4];
    }
}

__global__ void scatter_counts_kernel(const U4* __restrict__ scanned, uint32_t* __restrict__ blocks, uint64_t n_blocks) {
    const uint64_t b = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const U4 v = scanned[b];
    uint32_t* blk = blocks + b * kBlockDwords;
    blk[0] = v.a;
    blk[1] = v.c;
    blk[2] = v.g;
    blk[3] = v.t;
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t bytes) {
        if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) {
            set_last_error("hipMalloc failed in the index builder (" + std::to_string(bytes) + " bytes)");
            return FMSI_HIP_ENOMEM;
        }
        return 0;
    }
    template <class T>
    T* as() { return static_cast<T*>(p); }
    void* release() {
        void* q = p;
        p = nullptr;
        return q;
    }
};

inline unsigned host_base(const uint64_t* text2, uint64_t p) { return static_cast<unsigned>(text2[p >> 5] >> (2 * (p & 31))) & 3u; }

// a < b for suffixes a != b of the text ('$' = end of text is the smallest symbol)
bool suffix_less(const uint64_t* text2, uint64_t len, uint64_t a, uint64_t b) {
    while (a < len && b < len) {
        const unsigned x = host_base(text2, a), y = host_base(text2, b);
        if (x != y) return x < y;
        ++a;
        ++b;
    }
    return a == len && b != len;
}

uint32_t grid_for(uint64_t items, uint32_t threads = 256) { return static_cast<uint32_t>((items + threads - 1) / threads); }

// Shared tail of both entry points: text2 (device, zero padded, + mask_text or all-ones) -> device index.
int build_from_device_text(DevBuf& text2, uint64_t len, DevBuf* mask_text, int k, int with_klcp, int device, int prefix_len,
                           fmsi_hip_index** out) {
    const uint64_t n = len + 1;
    if (len == 0 || n >= (1ull << 32) - kBlockRows) {
        set_last_error("superstring length must be in [1, 2^32 - 130)");
        return FMSI_HIP_ERANGE;
    }
    const uint64_t n_blocks = ((n + kBlockRows - 1) >> kBlockRowsLog2) + 1;
    int rc;
    DevBuf keys_a, keys_b, pos_a, pos_b, temp, ties;
    if ((rc = keys_a.alloc(len * 8)) || (rc = keys_b.alloc(len * 8)) || (rc = pos_a.alloc(len * 4)) || (rc = pos_b.alloc(len * 4)) ||
        (rc = ties.alloc(16)))
        return rc;
    hipLaunchKernelGGL(make_keys_kernel, dim3(grid_for(len)), dim3(256), 0, 0, text2.as<uint64_t>(), len, keys_a.as<uint64_t>(),
                       pos_a.as<uint32_t>());
    rocprim::double_buffer<uint64_t> dk(keys_a.as<uint64_t>(), keys_b.as<uint64_t>());
    rocprim::double_buffer<uint32_t> dv(pos_a.as<uint32_t>(), pos_b.as<uint32_t>());
    size_t temp_bytes = 0;
    FMSI_HIP_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, dk, dv, static_cast<size_t>(len), 0, 64, 0));
    if ((rc = temp.alloc(temp_bytes))) return rc;
    FMSI_HIP_TRY(rocprim::radix_sort_pairs(temp.p, temp_bytes, dk, dv, static_cast<size_t>(len), 0, 64, 0));
    uint64_t* sorted_keys = dk.current();
    uint32_t* sorted_pos = dv.current();

    // ties -> host resolution
    FMSI_HIP_TRY(hipMemset(ties.p, 0, 16));
    hipLaunchKernelGGL(count_ties_kernel, dim3(grid_for(len)), dim3(256), 0, 0, sorted_keys, len, ties.as<unsigned long long>());
    unsigned long long n_ties = 0;
    FMSI_HIP_TRY(hipMemcpy(&n_ties, ties.p, 8, hipMemcpyDeviceToHost));
    if (n_ties) {
        DevBuf rows;
        if ((rc = rows.alloc(n_ties * 4))) return rc;
        hipLaunchKernelGGL(collect_ties_kernel, dim3(grid_for(len)), dim3(256), 0, 0, sorted_keys, len, rows.as<uint32_t>(),
                           ties.as<unsigned long long>() + 1);
        std::vector<uint32_t> h_rows(n_ties);
        FMSI_HIP_TRY(hipMemcpy(h_rows.data(), rows.p, n_ties * 4, hipMemcpyDeviceToHost));
        std::sort(h_rows.begin(), h_rows.end());
        std::vector<uint64_t> h_text((len >> 5) + 2);
        FMSI_HIP_TRY(hipMemcpy(h_text.data(), text2.p, h_text.size() * 8, hipMemcpyDeviceToHost));
        // consecutive rows form groups of equal keys (a group never contains a gap, by construction)
        std::vector<uint32_t> grp;
        size_t a = 0;
        while (a < h_rows.size()) {
            size_t b = a + 1;
            while (b < h_rows.size() && h_rows[b] == h_rows[b - 1] + 1) ++b;
            // rows a..b-1 are contiguous, but may hold several key groups back to back: fetch keys to split
            const size_t cnt = b - a;
            std::vector<uint64_t> gk(cnt);
            grp.resize(cnt);
            FMSI_HIP_TRY(hipMemcpy(gk.data(), sorted_keys + h_rows[a], cnt * 8, hipMemcpyDeviceToHost));
            FMSI_HIP_TRY(hipMemcpy(grp.data(), sorted_pos + h_rows[a], cnt * 4, hipMemcpyDeviceToHost));
            size_t g0 = 0;
            while (g0 < cnt) {
                size_t g1 = g0 + 1;
                while (g1 < cnt && gk[g1] == gk[g0]) ++g1;
                std::sort(grp.begin() + g0, grp.begin() + g1,
                          [&](uint32_t x, uint32_t y) { return suffix_less(h_text.data(), len, x, y); });
                g0 = g1;
            }
            FMSI_HIP_TRY(hipMemcpy(sorted_pos + h_rows[a], grp.data(), cnt * 4, hipMemcpyHostToDevice));
            a = b;
        }
    }
    // the keys are no longer needed: reuse nothing, just drop them before the layout allocations
    // (sorted_pos lives in pos_a or pos_b)
    DevBuf blocks, counts, counts_scanned, mcounts, mrank, klcp, dollar, scan_temp;
    const size_t blocks_bytes = n_blocks * kBlockBytes;
    if ((rc = blocks.alloc(blocks_bytes)) || (rc = counts.alloc((n_blocks + 1) * sizeof(U4))) ||
        (rc = counts_scanned.alloc((n_blocks + 1) * sizeof(U4))) || (rc = mcounts.alloc((n_blocks + 1) * 4)) ||
        (rc = mrank.alloc((n_blocks + 2) * 4)) || (rc = dollar.alloc(4)))
        return rc;
    const uint64_t klcp_words = (n + 63) / 64 + 1;
    if (with_klcp && (rc = klcp.alloc(klcp_words * 8))) return rc;
    FMSI_HIP_TRY(hipMemset(blocks.p, 0, blocks_bytes));
    FMSI_HIP_TRY(hipMemset(counts.p, 0, (n_blocks + 1) * sizeof(U4)));
    FMSI_HIP_TRY(hipMemset(mcounts.p, 0, (n_blocks + 1) * 4));
    if (with_klcp) FMSI_HIP_TRY(hipMemset(klcp.p, 0, klcp_words * 8));
    const uint32_t emit_grid = grid_for(n);   // 256 rows = 2 layout blocks per workgroup; grid covers <= n_blocks - 1 blocks
    hipLaunchKernelGGL(emit_rows_kernel, dim3(emit_grid), dim3(256), 0, 0, text2.as<uint64_t>(),
                       mask_text ? mask_text->as<uint64_t>() : nullptr, sorted_pos, len, k, mask_text ? 0 : 1, blocks.as<uint32_t>(),
                       counts.as<U4>(), mcounts.as<uint32_t>(), dollar.as<uint32_t>());
    FMSI_HIP_TRY(hipGetLastError());
    if (with_klcp) {
        hipLaunchKernelGGL(klcp_kernel, dim3(emit_grid), dim3(256), 0, 0, text2.as<uint64_t>(), sorted_pos, len, k, klcp.as<uint64_t>());
        FMSI_HIP_TRY(hipGetLastError());
    }
    // exclusive prefix sums over blocks (+1 element: totals)
    size_t st_bytes = 0;
    FMSI_HIP_TRY(rocprim::exclusive_scan(nullptr, st_bytes, counts.as<U4>(), counts_scanned.as<U4>(), U4{0, 0, 0, 0},
                                         static_cast<size_t>(n_blocks + 1), U4Plus()));
    size_t st2 = 0;
    FMSI_HIP_TRY(rocprim::exclusive_scan(nullptr, st2, mcounts.as<uint32_t>(), mrank.as<uint32_t>(), 0u,
                                         static_cast<size_t>(n_blocks + 1), rocprim::plus<uint32_t>()));
    if ((rc = scan_temp.alloc(std::max(st_bytes, st2)))) return rc;
    FMSI_HIP_TRY(rocprim::exclusive_scan(scan_temp.p, st_bytes, counts.as<U4>(), counts_scanned.as<U4>(), U4{0, 0, 0, 0},
                                         static_cast<size_t>(n_blocks + 1), U4Plus()));
    FMSI_HIP_TRY(rocprim::exclusive_scan(scan_temp.p, st2, mcounts.as<uint32_t>(), mrank.as<uint32_t>(), 0u,
                                         static_cast<size_t>(n_blocks + 1), rocprim::plus<uint32_t>()));
    hipLaunchKernelGGL(scatter_counts_kernel, dim3(grid_for(n_blocks)), dim3(256), 0, 0, counts_scanned.as<U4>(), blocks.as<uint32_t>(),
                       n_blocks);
    U4 totals;
    uint32_t dollar_row = 0;
    FMSI_HIP_TRY(hipMemcpy(&totals, counts_scanned.as<U4>() + n_blocks, sizeof(U4), hipMemcpyDeviceToHost));
    FMSI_HIP_TRY(hipMemcpy(&dollar_row, dollar.p, 4, hipMemcpyDeviceToHost));
    FMSI_HIP_TRY(hipDeviceSynchronize());

    auto* idx = new fmsi_hip_index();
    idx->device = device;
    idx->n = n;
    idx->k = k;
    idx->n_blocks = n_blocks;
    idx->d_blocks = blocks.release();
    idx->d_mrank = mrank.release();
    idx->d_text2 = text2.release();
    idx->text_len = len;
    idx->device_bytes = blocks_bytes + (n_blocks + 2) * 4 + ((len >> 5) + 2) * 8;
    if (with_klcp) {
        idx->d_klcp = klcp.release();
        idx->has_klcp = 1;
        idx->device_bytes += klcp_words * 8;
    }
    if (hipMalloc(reinterpret_cast<void**>(&idx->d_counters), FMSI_HIP_N_COUNTERS * 8) != hipSuccess) {
        fmsi_hip_index_destroy(idx);
        set_last_error("hipMalloc failed");
        return FMSI_HIP_ENOMEM;
    }
    idx->dev.blocks = static_cast<const uint4*>(idx->d_blocks);
    idx->dev.mrank = static_cast<const uint32_t*>(idx->d_mrank);
    idx->dev.ptable = nullptr;
    idx->dev.ptable_len = 0;
    idx->dev.n = static_cast<uint32_t>(n);
    idx->dev.dollar = dollar_row;
    idx->dev.C0 = 1;                                            // counts = {1, #A+1, ...}, fms_index.h:451
    idx->dev.C1 = totals.a;                                     // the '$' slot is stored (and counted) as an 'A'
    idx->dev.C2 = totals.a + totals.c;
    idx->dev.C3 = totals.a + totals.c + totals.g;
    if (prefix_len > 0 && (rc = build_prefix_table(idx, prefix_len))) {
        fmsi_hip_index_destroy(idx);
        return rc;
    }
    *out = idx;
    return 0;
}

int check_device(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        set_last_error("no usable HIP device (libfmsi_hip has no CPU fallback)");
        return FMSI_HIP_ENODEV;
    }
    FMSI_HIP_TRY(hipSetDevice(device));
    return 0;
}

// synthetic queries: see fmsi_hip_synthetic_queries in the header
__global__ void synth_queries_kernel(const uint64_t* __restrict__ text2, uint64_t len, int k, uint64_t seed, uint64_t first, uint64_t n,
                                     uint64_t* __restrict__ kmers) {
    const uint64_t q = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const uint64_t r = splitmix64(seed ^ ((first + q) * 0x9FB21C651E98DF25ull));
    const uint64_t r2 = splitmix64(r);
    const uint64_t m = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    uint64_t f;
    if (r & 1ull) {                         // positive: a k-mer of the superstring ...
        const uint64_t p = r2 % (len - static_cast<uint64_t>(k) + 1);
        f = bases_at(text2, p) & m;
        if (r & 2ull) {                     // ... half of them reverse-complemented
            uint64_t x = __brevll(~f);
            x = ((x & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((x & 0x5555555555555555ull) << 1);
            f = (x >> (64 - 2 * k)) & m;
        }
    } else {
        f = r2 & m;                         // uniform random k-mer
    }
    kmers[q] = f;
}

// synthetic reads: see fmsi_hip_synthetic_reads in the header.  One thread per 32-base output word.
__global__ void synth_reads_kernel(const uint64_t* __restrict__ text2, uint64_t len, uint64_t read_len, uint64_t seed, uint64_t first_read,
                                   uint64_t n_reads, uint32_t subst_per_million, uint64_t n_words, uint64_t* __restrict__ packed) {
    const uint64_t w = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint64_t v = 0;
    const uint64_t total = n_reads * read_len;
    for (uint32_t b = 0; b < 32; ++b) {
        const uint64_t pos = w * 32 + b;
        if (pos >= total) break;
        const uint64_t r = pos / read_len, off = pos - r * read_len;
        const uint64_t h = splitmix64(seed ^ ((first_read + r) * 0xA24BAED4963EE407ull));
        uint32_t c;
        if (h & 1ull) {                                   // a substring of the superstring ...
            const uint64_t p0 = splitmix64(h) % (len - read_len + 1);
            if (h & 2ull) {                               // ... half of them reverse-complemented
                const uint64_t p = p0 + (read_len - 1 - off);
                c = 3u - (static_cast<uint32_t>(text2[p >> 5] >> (2 * (p & 31))) & 3u);
            } else {
                const uint64_t p = p0 + off;
                c = static_cast<uint32_t>(text2[p >> 5] >> (2 * (p & 31))) & 3u;
            }
            if (subst_per_million) {
                const uint64_t hs = splitmix64(h ^ (off * 0x9E3779B97F4A7C15ull));
                if (hs % 1000000ull < subst_per_million) c = static_cast<uint32_t>(hs >> 40) & 3u;
            }
        } else {
            c = static_cast<uint32_t>(splitmix64(h ^ (off * 0xC2B2AE3D27D4EB4Full)) >> 17) & 3u;   // uniform random read
        }
        v |= static_cast<uint64_t>(c) << (2 * b);
    }
    packed[w] = v;
}

}  // namespace
}  // namespace fmsi

using namespace fmsi;

extern "C" {

int fmsi_hip_index_build(const char* ms, uint64_t len, int k, int with_klcp, int device, int prefix_len, fmsi_hip_index** out) {
    if (!ms || !out || k < 1 || (with_klcp && k > 64)) {
        set_last_error("bad argument (on-device kLCP construction needs k <= 64)");
        return FMSI_HIP_EINVAL;
    }
    int rc = check_device(device);
    if (rc) return rc;
    const uint64_t n_words = (len >> 5) + 2, m_words = (len >> 6) + 2;
    std::vector<uint64_t> h_text(n_words, 0), h_mask(m_words, 0);
    for (uint64_t p = 0; p < len; ++p) {
        unsigned c;
        switch (ms[p]) {   // nucleotideToInt, src/kmers.h:3-20
        case 'A': case 'a': c = 0; break;
        case 'C': case 'c': c = 1; break;
        case 'G': case 'g': c = 2; break;
        case 'T': case 't': c = 3; break;
        default:
            set_last_error("masked superstring contains a non-ACGT character at " + std::to_string(p));
            return FMSI_HIP_EINVAL;
        }
        h_text[p >> 5] |= static_cast<uint64_t>(c) << (2 * (p & 31));
        if (ms[p] >= 'A' && ms[p] <= 'Z') h_mask[p >> 6] |= 1ull << (p & 63);   // is_upper, src/kmers.h:71-73
    }
    DevBuf text2, mask_text;
    if ((rc = text2.alloc(n_words * 8)) || (rc = mask_text.alloc(m_words * 8))) return rc;
    FMSI_HIP_TRY(hipMemcpy(text2.p, h_text.data(), n_words * 8, hipMemcpyHostToDevice));
    FMSI_HIP_TRY(hipMemcpy(mask_text.p, h_mask.data(), m_words * 8, hipMemcpyHostToDevice));
    return build_from_device_text(text2, len, &mask_text, k, with_klcp, device, prefix_len, out);
}

int fmsi_hip_index_build_synthetic(uint64_t len, uint64_t seed, int k, int with_klcp, int device, int prefix_len, fmsi_hip_index** out) {
    if (!out || k < 1 || len < static_cast<uint64_t>(k) || (with_klcp && k > 64)) {
        set_last_error("bad argument (on-device kLCP construction needs k <= 64)");
        return FMSI_HIP_EINVAL;
    }
    int rc = check_device(device);
    if (rc) return rc;
    const uint64_t n_words = (len >> 5) + 2;
    DevBuf text2;
    if ((rc = text2.alloc(n_words * 8))) return rc;
    hipLaunchKernelGGL(synth_text_kernel, dim3(grid_for(n_words)), dim3(256), 0, 0, text2.as<uint64_t>(), n_words, len, seed);
    FMSI_HIP_TRY(hipGetLastError());
    return build_from_device_text(text2, len, nullptr, k, with_klcp, device, prefix_len, out);
}

uint64_t fmsi_hip_synthetic_text_word(uint64_t seed, uint64_t w) { return synth_word(seed, w); }

int fmsi_hip_synthetic_queries(fmsi_hip_index* idx, int k, uint64_t seed, uint64_t first, uint64_t n, uint64_t* d_kmers, void* stream) {
    if (!idx || !d_kmers || !idx->d_text2 || k < 1 || k > 32 || idx->text_len < static_cast<uint64_t>(k)) {
        set_last_error("synthetic queries need an index built by fmsi_hip_index_build*, k <= 32");
        return FMSI_HIP_EINVAL;
    }
    if (n == 0) return 0;
    FMSI_HIP_TRY(hipSetDevice(idx->device));
    hipLaunchKernelGGL(synth_queries_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint64_t*>(idx->d_text2), idx->text_len, k, seed, first, n, d_kmers);
    FMSI_HIP_TRY(hipGetLastError());
    return 0;
}

int fmsi_hip_synthetic_reads(fmsi_hip_index* idx, uint64_t read_len, uint64_t seed, uint64_t first_read, uint64_t n_reads,
                             uint32_t subst_per_million, uint64_t* d_packed, void* stream) {
    if (!idx || !d_packed || !idx->d_text2 || read_len == 0 || idx->text_len < read_len) {
        set_last_error("synthetic reads need an index built by fmsi_hip_index_build* and read_len <= |MS|");
        return FMSI_HIP_EINVAL;
    }
    if (n_reads == 0) return 0;
    FMSI_HIP_TRY(hipSetDevice(idx->device));
    const uint64_t n_words = (n_reads * read_len) / 32 + 2;
    hipLaunchKernelGGL(synth_reads_kernel, dim3(grid_for(n_words)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint64_t*>(idx->d_text2), idx->text_len, read_len, seed, first_read, n_reads, subst_per_million,
                       n_words, d_packed);
    FMSI_HIP_TRY(hipGetLastError());
    return 0;
}

int fmsi_hip_index_download_text(const fmsi_hip_index* idx, uint64_t* words) {
    if (!idx || !words || !idx->d_text2) return FMSI_HIP_EINVAL;
    FMSI_HIP_TRY(hipSetDevice(idx->device));
    FMSI_HIP_TRY(hipMemcpy(words, idx->d_text2, ((idx->text_len >> 5) + 2) * 8, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
