// On-device index construction: counterpart of construct<T>() (reference src/fms_index.h:397-460) that
// goes from a masked superstring straight to the interleaved device layout, with the suffix sort done
// on the GPU.  It exists so that human-genome-scale synthetic indexes can be made on the box in seconds
// (the reference's qsufsort needs ~1 h and 45 GB there, SURVEY.md 7 hard part 5).
//
// Pipeline (N = |MS|, n = N + 1 rows):
//   1. 2-bit text in HBM (uploaded, or generated in place from a counter-based PRNG for synthetic runs)
//   2. key[p] = the 29 bases starting at p, first base in the top bits (zero padded past the end), and in the low
//      6 bits min(N - p, 29): a suffix that ends inside the window sorts before every longer one ('$' is smallest)
//   3. rocPRIM radix sort of (key, p) pairs -- a library sort, not part of the query hot path
//   4. rows whose key equals a neighbour's (repeats of 29 bases or more) are refined by prefix doubling on the
//      device (Manber-Myers on the tied rows only): rank[p] = first row of p's group; round with depth L sorts the
//      tied rows by (group, rank[p + L]), splits the groups, updates the ranks, L doubles.  <= log2(N / 29) rounds
//      whatever the input; random texts need one
//   5. one pass over the suffix array emits BWT symbol and SA-transformed mask bit per row, wave ballots pack them
//      into the block planes, per-block counts are prefix-summed on the device; a second pass makes the kLCP bits.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fmsi_hip.h"
#include "common.h"
#include "host_index.h"
#include "index_priv.h"
#include "trace.h"

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

// ... or, with repeat families (fmsi_hip_index_build_synthetic_repeats), of a text in which a share of the 2048-base
// segments are diverged copies of one of `families` consensus segments: a stand-in for the skewed k-mer spectrum of a
// real genome, where many distinct k-mers share their last t bases (the prefix-table key) -- which uniform text lacks.
__host__ __device__ inline uint64_t synth_word_repeats(uint64_t seed, uint64_t w, uint32_t repeat_permille, uint32_t families,
                                                       uint32_t divergence_permille) {
    const uint64_t seg = w >> 6;   // 64 words = 2048 bases
    const uint64_t h = splitmix64(seed ^ 0x5EC7105EEDull ^ (seg * 0xA24BAED4963EE407ull));
    if (repeat_permille == 0 || families == 0 || (h % 1000u) >= repeat_permille) return synth_word(seed, w);
    const uint64_t fam = (h >> 20) % families;
    uint64_t v = synth_word(seed ^ 0xFA3117ull, fam * 64u + (w & 63u));   // the family's consensus word
    for (uint32_t b = 0; b < 32u; ++b) {                                  // this copy's substitutions
        const uint64_t r = splitmix64(seed ^ 0xD17E26Eull ^ ((w * 32u + b) * 0x9E3779B97F4A7C15ull));
        if ((r % 1000u) < divergence_permille) {
            const uint64_t c = (v >> (2u * b)) & 3u;
            const uint64_t d = (c + 1u + ((r >> 32) % 3u)) & 3u;          // one of the three other bases
            v = (v & ~(3ull << (2u * b))) | (d << (2u * b));
        }
    }
    return v;
}

__global__ void synth_text_kernel(uint64_t* __restrict__ text2, uint64_t n_words, uint64_t len, uint64_t seed, uint32_t repeat_permille,
                                  uint32_t families, uint32_t divergence_permille) {
    const uint64_t w = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint64_t v = synth_word_repeats(seed, w, repeat_permille, families, divergence_permille);
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

constexpr uint64_t kKeyBases = 29;   // bases per round-0 key; the low 6 bits hold min(remaining length, 29)

__global__ void make_keys_kernel(const uint64_t* __restrict__ text2, uint64_t len, uint64_t* __restrict__ keys, uint32_t* __restrict__ pos) {
    const uint64_t p = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p >= len) return;
    const uint64_t rem = len - p < kKeyBases ? len - p : kKeyBases;
    keys[p] = (to_lex_key(bases_at(text2, p)) & ~63ull) | rem;   // text2 is zero past `len`
    pos[p] = static_cast<uint32_t>(p);
}

// ---- prefix doubling on the tied rows ---------------------------------------------------------------------------
// head[i] = element i starts a group (its key differs from its predecessor's); start[i] = its own row if so, else 0
// (an inclusive max-scan of start[] then gives every element the first row of its group)
__global__ void group_heads_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ rows /* or null: row = i */,
                                   uint64_t count, uint32_t* __restrict__ start) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const bool head = i == 0 || keys[i] != keys[i - 1];
    start[i] = head ? (rows ? rows[i] : static_cast<uint32_t>(i)) : 0u;
}
// rank of a suffix = 1 + first row of its group (0 is the empty suffix); tied[i] = its group has more than one element.
// grp[] is non-decreasing and equal exactly within a group, so neighbours tell.
__global__ void rank_and_ties_kernel(const uint32_t* __restrict__ grp, const uint32_t* __restrict__ pos, uint64_t count,
                                     uint32_t* __restrict__ rank, uint32_t* __restrict__ tied) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t g = grp[i];
    rank[pos[i]] = g + 1u;
    tied[i] = (i > 0 && grp[i - 1] == g) || (i + 1 < count && grp[i + 1] == g);
}
// keeps the tied elements: (row, position, group) triples, order preserved
__global__ void compact_ties_kernel(const uint32_t* __restrict__ tied, const uint32_t* __restrict__ slot, const uint32_t* __restrict__ rows,
                                    const uint32_t* __restrict__ pos, const uint32_t* __restrict__ grp, uint64_t count,
                                    uint32_t* __restrict__ out_rows, uint32_t* __restrict__ out_pos, uint32_t* __restrict__ out_grp) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count || !tied[i]) return;
    const uint32_t o = slot[i];
    out_rows[o] = rows ? rows[i] : static_cast<uint32_t>(i);
    out_pos[o] = pos[i];
    out_grp[o] = grp[i];
}
// key of a tied suffix in the round of depth L: (its group, rank of the suffix L bases further on)
__global__ void doubling_keys_kernel(const uint32_t* __restrict__ pos, const uint32_t* __restrict__ grp, const uint32_t* __restrict__ rank,
                                     uint64_t count, uint64_t depth, uint64_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    // a tied suffix is at least `depth` long (shorter ones were unique by their length field), so pos + depth <= len
    keys[i] = (static_cast<uint64_t>(grp[i]) << 32) | rank[static_cast<uint64_t>(pos[i]) + depth];
    idx[i] = static_cast<uint32_t>(i);
}
// the tied rows keep their rows (ascending, groups contiguous); the sorted positions go back into the suffix array
__global__ void apply_order_kernel(const uint32_t* __restrict__ order, const uint32_t* __restrict__ pos, const uint32_t* __restrict__ rows,
                                   uint64_t count, uint32_t* __restrict__ new_pos, uint32_t* __restrict__ sa) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t p = pos[order[i]];
    new_pos[i] = p;
    sa[rows[i]] = p;
}

struct U4 {
    uint32_t a, c, g, t;
};
struct U32Max {
    __host__ __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};
struct U4Plus {
    __host__ __device__ U4 operator()(const U4& x, const U4& y) const { return U4{x.a + y.a, x.c + y.c, x.g + y.g, x.t + y.t}; }
};

// kLCP bit of row r: the suffixes of rows r and r+1 share their first k-1 bases (construct_klcp, src/fms_index.h:357-385;
// rows whose suffix is shorter than k-1 get 0).  One thread per row, one 64-bit ballot per wave.  Kept apart from
// emit_rows_kernel: fused into it, the compare gave run-to-run differing bits on gfx950 for rows >= 2^16
// (tests/test_gpu_builder.py::test_klcp_equals_oracle pins this kernel).
// (POS: 32- or 64-bit suffix positions; first_row: launches hold fewer than 2^32 threads, the builder for 64-bit rows comes in pieces)
template <class POS>
__global__ __launch_bounds__(256) void klcp_kernel(const uint64_t* __restrict__ text2, const POS* __restrict__ sa_sorted,
                                                   uint64_t len, int k, uint64_t* __restrict__ klcp, uint64_t first_row) {
    const uint64_t row = first_row + static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
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
        if (p) (void)dev_free(p);
    }
    int alloc(size_t bytes) {
        if (dev_malloc(&p, bytes ? bytes : 8) != hipSuccess) {
            set_last_error("hipMalloc failed in the index builder (" + std::to_string(bytes) + " bytes)");
            return FMSI_HIP_ENOMEM;
        }
        return 0;
    }
    template <class T>
    T* as() { return static_cast<T*>(p); }
    void drop() {
        if (p) (void)dev_free(p);
        p = nullptr;
    }
    void* release() {
        void* q = p;
        p = nullptr;
        return q;
    }
};

uint32_t grid_for(uint64_t items, uint32_t threads = 256) { return static_cast<uint32_t>((items + threads - 1) / threads); }

int build_wide_from_device_text(DevBuf& text2, uint64_t len, DevBuf* mask_text, int k, int with_klcp, int device, int prefix_len, fmsi_hip_index** out);

// Shared tail of both entry points: text2 (device, zero padded, + mask_text or all-ones) -> device index.
int build_from_device_text(DevBuf& text2, uint64_t len, DevBuf* mask_text, int k, int with_klcp, int device, int prefix_len,
                           fmsi_hip_index** out) {
    TraceRange trace_range("fmsi: build index on device");
    const uint64_t n = len + 1;
    // (2^32 - 130 is where 32-bit rows end; rocPRIM's radix sort stops a few thousand items earlier -- its launch arithmetic is 32-bit --
    // so the on-device builder does, too.  An index of that size made elsewhere still loads: the loader sorts nothing.)
    if (len == 0) {
        set_last_error("empty superstring");
        return FMSI_HIP_ERANGE;
    }
    if (len > kMaxSortItems || std::getenv("FMSI_HIP_FORCE_WIDE_BUILD")) return build_wide_from_device_text(text2, len, mask_text, k, with_klcp, device, prefix_len, out);
    const uint64_t n_blocks = ((n + kBlockRows - 1) >> kBlockRowsLog2) + 1;
    int rc;
    DevBuf keys_a, keys_b, pos_a, pos_b, temp;
    if ((rc = keys_a.alloc(len * 8)) || (rc = keys_b.alloc(len * 8)) || (rc = pos_a.alloc(len * 4)) || (rc = pos_b.alloc(len * 4))) return rc;
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

    // ---- ties: prefix doubling on the tied rows --------------------------------------------------------------------
    {
        DevBuf start, grp, tied, slot, rank;
        if ((rc = start.alloc(len * 4)) || (rc = grp.alloc(len * 4)) || (rc = tied.alloc((len + 1) * 4)) || (rc = slot.alloc((len + 1) * 4)) ||
            (rc = rank.alloc((len + 1) * 4)))
            return rc;
        size_t sb1 = 0, sb2 = 0;
        FMSI_HIP_TRY(rocprim::inclusive_scan(nullptr, sb1, start.as<uint32_t>(), grp.as<uint32_t>(), static_cast<size_t>(len), U32Max()));
        FMSI_HIP_TRY(rocprim::exclusive_scan(nullptr, sb2, tied.as<uint32_t>(), slot.as<uint32_t>(), 0u, static_cast<size_t>(len + 1),
                                             rocprim::plus<uint32_t>()));
        DevBuf scan_tmp;
        if ((rc = scan_tmp.alloc(std::max(sb1, sb2)))) return rc;
        auto group_and_compact = [&](const uint64_t* d_keys, const uint32_t* d_rows, const uint32_t* d_pos, uint64_t count, DevBuf& o_rows,
                                     DevBuf& o_pos, DevBuf& o_grp, uint64_t* n_tied) -> int {
            // groups of equal keys -> ranks; the elements of groups with more than one member are kept for the next round
            hipLaunchKernelGGL(group_heads_kernel, dim3(grid_for(count)), dim3(256), 0, 0, d_keys, d_rows, count, start.as<uint32_t>());
            size_t b1 = sb1, b2 = sb2;
            FMSI_HIP_TRY(rocprim::inclusive_scan(scan_tmp.p, b1, start.as<uint32_t>(), grp.as<uint32_t>(), static_cast<size_t>(count), U32Max()));
            hipLaunchKernelGGL(rank_and_ties_kernel, dim3(grid_for(count)), dim3(256), 0, 0, grp.as<uint32_t>(), d_pos, count,
                               rank.as<uint32_t>(), tied.as<uint32_t>());
            FMSI_HIP_TRY(hipMemsetAsync(tied.as<uint32_t>() + count, 0, 4, 0));
            FMSI_HIP_TRY(rocprim::exclusive_scan(scan_tmp.p, b2, tied.as<uint32_t>(), slot.as<uint32_t>(), 0u, static_cast<size_t>(count + 1),
                                                 rocprim::plus<uint32_t>()));
            uint32_t t = 0;
            FMSI_HIP_TRY(hipMemcpy(&t, slot.as<uint32_t>() + count, 4, hipMemcpyDeviceToHost));
            *n_tied = t;
            if (!t) return 0;
            int rc2;
            if ((rc2 = o_rows.alloc(static_cast<size_t>(t) * 4)) || (rc2 = o_pos.alloc(static_cast<size_t>(t) * 4)) ||
                (rc2 = o_grp.alloc(static_cast<size_t>(t) * 4)))
                return rc2;
            hipLaunchKernelGGL(compact_ties_kernel, dim3(grid_for(count)), dim3(256), 0, 0, tied.as<uint32_t>(), slot.as<uint32_t>(), d_rows, d_pos,
                               grp.as<uint32_t>(), count, o_rows.as<uint32_t>(), o_pos.as<uint32_t>(), o_grp.as<uint32_t>());
            FMSI_HIP_TRY(hipGetLastError());
            return 0;
        };
        FMSI_HIP_TRY(hipMemset(rank.p, 0, (len + 1) * 4));   // rank[len] = 0: the empty suffix
        DevBuf t_rows, t_pos, t_grp;
        uint64_t n_tied = 0;
        if ((rc = group_and_compact(sorted_keys, nullptr, sorted_pos, len, t_rows, t_pos, t_grp, &n_tied))) return rc;
        // round 0's key buffers are free now (22 GB each at human scale); the one that is not the suffix array's
        // neighbour keeps serving as scratch for nothing -- drop both
        FMSI_HIP_TRY(hipDeviceSynchronize());
        const bool pos_in_a = sorted_pos == pos_a.as<uint32_t>();
        (void)dev_free(keys_a.release());
        (void)dev_free(keys_b.release());
        (void)dev_free(pos_in_a ? pos_b.release() : pos_a.release());
        for (uint64_t depth = kKeyBases; n_tied; depth *= 2) {
            DevBuf k_a, k_b, i_a, i_b, new_pos, sort_tmp, n_rows, n_pos, n_grp;
            if ((rc = k_a.alloc(n_tied * 8)) || (rc = k_b.alloc(n_tied * 8)) || (rc = i_a.alloc(n_tied * 4)) || (rc = i_b.alloc(n_tied * 4)) ||
                (rc = new_pos.alloc(n_tied * 4)))
                return rc;
            hipLaunchKernelGGL(doubling_keys_kernel, dim3(grid_for(n_tied)), dim3(256), 0, 0, t_pos.as<uint32_t>(), t_grp.as<uint32_t>(),
                               rank.as<uint32_t>(), n_tied, depth, k_a.as<uint64_t>(), i_a.as<uint32_t>());
            rocprim::double_buffer<uint64_t> dk2(k_a.as<uint64_t>(), k_b.as<uint64_t>());
            rocprim::double_buffer<uint32_t> dv2(i_a.as<uint32_t>(), i_b.as<uint32_t>());
            size_t tb = 0;
            FMSI_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, dk2, dv2, static_cast<size_t>(n_tied), 0, 64, 0));
            if ((rc = sort_tmp.alloc(tb))) return rc;
            FMSI_HIP_TRY(rocprim::radix_sort_pairs(sort_tmp.p, tb, dk2, dv2, static_cast<size_t>(n_tied), 0, 64, 0));
            hipLaunchKernelGGL(apply_order_kernel, dim3(grid_for(n_tied)), dim3(256), 0, 0, dv2.current(), t_pos.as<uint32_t>(),
                               t_rows.as<uint32_t>(), n_tied, new_pos.as<uint32_t>(), sorted_pos);
            uint64_t still = 0;
            if ((rc = group_and_compact(dk2.current(), t_rows.as<uint32_t>(), new_pos.as<uint32_t>(), n_tied, n_rows, n_pos, n_grp, &still)))
                return rc;
            FMSI_HIP_TRY(hipDeviceSynchronize());
            std::swap(t_rows.p, n_rows.p);
            std::swap(t_pos.p, n_pos.p);
            std::swap(t_grp.p, n_grp.p);
            n_tied = still;
            if (depth > 2 * len) {
                set_last_error("suffix sort: prefix doubling did not converge");
                return FMSI_HIP_EHIP;
            }
        }
    }
    (void)dev_free(temp.release());
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
        hipLaunchKernelGGL((klcp_kernel<uint32_t>), dim3(emit_grid), dim3(256), 0, 0, text2.as<uint64_t>(), sorted_pos, len, k, klcp.as<uint64_t>(), 0ull);
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
    if (dev_malloc(reinterpret_cast<void**>(&idx->d_counters), FMSI_HIP_N_COUNTERS * 8) != hipSuccess) {
        fmsi_hip_index_destroy(idx);
        set_last_error("hipMalloc failed");
        return FMSI_HIP_ENOMEM;
    }
    idx->dev.blocks = static_cast<const uint4*>(idx->d_blocks);
    idx->dev.mrank = static_cast<const uint32_t*>(idx->d_mrank);
    idx->dev.ptable = nullptr;
    idx->dev.ptable_len = 0;
    idx->dev.ptable_packed = 0;
    idx->dev.n = static_cast<uint32_t>(n);
    idx->dev.dollar = dollar_row;
    idx->dev.C0 = 1;                                            // counts = {1, #A+1, ...}, fms_index.h:451
    idx->dev.C1 = totals.a;                                     // the '$' slot is stored (and counted) as an 'A'
    idx->dev.C2 = totals.a + totals.c;
    idx->dev.C3 = totals.a + totals.c + totals.g;
    idx->counts64[0] = idx->dev.C0;
    idx->counts64[1] = idx->dev.C1;
    idx->counts64[2] = idx->dev.C2;
    idx->counts64[3] = idx->dev.C3;
    idx->dollar64 = dollar_row;
    if (prefix_len > 0 && (rc = build_prefix_table(idx, prefix_len))) {
        fmsi_hip_index_destroy(idx);
        return rc;
    }
    *out = idx;
    return 0;
}

// ---- indexes of 2^32 rows and more (fmsi_wide.hip's handles) built on the device -------------------------------------------------------
// rocPRIM's radix sort takes fewer than 2^32 items and the builder above keeps suffix positions in 32 bits; beyond that the suffixes are
// dealt into 16 buckets by their first two bases (a histogram pass and a scatter pass with wave-aggregated cursors), every bucket -- a
// sixteenth of the text or so -- is sorted by its 29-base keys with 64-bit positions as values, and the buckets in order are the suffix
// array by 29 bases.  Suffixes that still tie (a handful in a text without long repeats: n^2 / 4^29 pairs) are ordered by comparing
// their next 29-base keys directly, group by group; a text with tied groups of more than kWideTieCap suffixes -- a repetitive one, i.e. any
// real genome of this size -- gets prefix doubling with 64-bit ranks over all tied suffixes (wide_prefix_doubling, round 4).  Rows, blocks and counts as above -- the u32 cumulative
// counts simply wrap -- plus the table of absolute counts per superblock that query_wide_kernel reads.  No kLCP array, no tables.
constexpr uint32_t kWideTieCap = 64;

__device__ __forceinline__ uint64_t wide_key(const uint64_t* __restrict__ text2, uint64_t len, uint64_t p) {
    const uint64_t rem = len - p < kKeyBases ? len - p : kKeyBases;
    return (to_lex_key(bases_at(text2, p)) & ~63ull) | rem;   // text2 is zero past `len`
}

__global__ __launch_bounds__(256) void wide_hist_kernel(const uint64_t* __restrict__ text2, uint64_t len, unsigned long long* __restrict__ hist /* 16 buckets + 4 bases */) {
    __shared__ unsigned long long s_h[20];
    if (threadIdx.x < 20) s_h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    uint32_t loc[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) loc[i] = 0;
    for (uint64_t p = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; p < len; p += stride) {
        const uint32_t b = static_cast<uint32_t>(wide_key(text2, len, p) >> 60);
        const uint32_t c = static_cast<uint32_t>(text2[p >> 5] >> (2 * (p & 31))) & 3u;
#pragma unroll
        for (int i = 0; i < 16; ++i) loc[i] += (b == static_cast<uint32_t>(i));
#pragma unroll
        for (int i = 0; i < 4; ++i) loc[16 + i] += (c == static_cast<uint32_t>(i));
    }
#pragma unroll
    for (int i = 0; i < 20; ++i)
        if (loc[i]) atomicAdd(&s_h[i], static_cast<unsigned long long>(loc[i]));
    __syncthreads();
    if (threadIdx.x < 20 && s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_h[threadIdx.x]);
}

// (a launch holds fewer than 2^32 threads: the kernels over all positions / rows are launched in pieces of 2^30, `first` says where a piece starts)
__global__ __launch_bounds__(256) void wide_scatter_kernel(const uint64_t* __restrict__ text2, uint64_t len, unsigned long long* __restrict__ cursor,
                                                           uint64_t* __restrict__ keys, uint64_t* __restrict__ pos, uint64_t first_item) {
    const uint64_t p = first_item + static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const bool valid = p < len;
    uint64_t key = 0;
    uint32_t b = 16;
    if (valid) {
        key = wide_key(text2, len, p);
        b = static_cast<uint32_t>(key >> 60);
    }
    uint64_t todo = __ballot(valid);
    const uint32_t lane = threadIdx.x & 63u;
    while (todo) {   // one cursor bump per bucket present in the wave
        const uint32_t first = static_cast<uint32_t>(__ffsll(static_cast<unsigned long long>(todo))) - 1u;
        const uint32_t b0 = static_cast<uint32_t>(__shfl(static_cast<int>(b), static_cast<int>(first), 64));
        const uint64_t m = __ballot(valid && b == b0);
        unsigned long long base = 0;
        if (lane == first) base = atomicAdd(&cursor[b0], static_cast<unsigned long long>(__popcll(m)));
        base = static_cast<unsigned long long>(__shfl(static_cast<long long>(base), static_cast<int>(first), 64));
        if (valid && b == b0) {
            const uint64_t at = base + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
            keys[at] = key;
            pos[at] = p;
        }
        todo &= ~m;
    }
}

// is suffix p before suffix q, both known to agree on their first `from` bases?  (key by key, the way the sort orders them)
__device__ __forceinline__ bool wide_suffix_less(const uint64_t* __restrict__ text2, uint64_t len, uint64_t p, uint64_t q, uint64_t from) {
    for (uint64_t off = from;; off += kKeyBases) {
        if (p + off >= len || q + off >= len) return p > q;   // the shorter suffix (the later position) first
        const uint64_t a = wide_key(text2, len, p + off), b = wide_key(text2, len, q + off);
        if (a != b) return a < b;
    }
}

// one thread per head of a group of equal keys: the group's positions in suffix order (insertion sort; groups are tiny)
__global__ __launch_bounds__(256) void wide_ties_kernel(const uint64_t* __restrict__ keys, uint64_t* __restrict__ sa, uint64_t count, const uint64_t* __restrict__ text2,
                                                        uint64_t len, unsigned long long* __restrict__ stats /* tied suffixes, groups over the cap */, uint64_t first_item) {
    const uint64_t i = first_item + static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint64_t key = keys[i];
    if ((i > 0 && keys[i - 1] == key) || i + 1 >= count || keys[i + 1] != key) return;
    uint64_t g = 2;
    while (i + g < count && keys[i + g] == key && g <= kWideTieCap) ++g;
    atomicAdd(&stats[0], static_cast<unsigned long long>(g));
    if (g > kWideTieCap) {
        atomicAdd(&stats[1], 1ull);
        return;
    }
    for (uint64_t a = 1; a < g; ++a) {
        const uint64_t v = sa[i + a];
        uint64_t b = a;
        while (b > 0 && wide_suffix_less(text2, len, v, sa[i + b - 1], kKeyBases)) {
            sa[i + b] = sa[i + b - 1];
            --b;
        }
        sa[i + b] = v;
    }
}

// ---- prefix doubling with 64-bit rows (round 4): tied groups of any size ---------------------------------------------------------------
// The narrow builder's scheme (group_heads / rank_and_ties / compact_ties / doubling_keys / apply_order above) with u64 rows, positions and
// ranks: rank[p] = 1 + first row of the group of suffix p (0: the empty suffix); the rows whose 29-base key equals a neighbour's are kept as
// (row, position, group) triples and sorted by (group, rank[position + depth]) -- a 128-bit key -- with the depth doubling until every
// group is one suffix.  The first pass runs over ALL len > 2^32 sorted suffixes: kernels and scans in pieces of 2^30 (a launch and a rocPRIM
// call take fewer than 2^32 items), carries handed from piece to piece; the rounds run over the tied suffixes only (fewer than 2^32 - 8192,
// or FMSI_HIP_ERANGE).
struct U64Max {
    __host__ __device__ uint64_t operator()(uint64_t a, uint64_t b) const { return a > b ? a : b; }
};
// start[i] = its own row when element i starts a group of equal keys, else 0; the first element of a piece takes the carry (the group's
// first row as the piece before left it) when it continues that piece's last group
__global__ __launch_bounds__(256) void wide_heads_kernel(const uint64_t* __restrict__ keys, uint64_t count, uint64_t first_item, uint64_t* __restrict__ start) {
    const uint64_t i = first_item + static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const bool head = i == 0 || keys[i] != keys[i - 1];
    start[i] = head ? i : ((i == first_item) ? start[i - 1] : 0ull);   // (start[i - 1]: the piece before is scanned by now -- its group's first row)
}
// rank of every suffix, and whether its group has more than one member
__global__ __launch_bounds__(256) void wide_rank_ties_kernel(const uint64_t* __restrict__ grp, const uint64_t* __restrict__ pos, uint64_t count, uint64_t first_item,
                                                             uint64_t* __restrict__ rank, uint32_t* __restrict__ tied /* this piece's flags, from 0 */) {
    const uint64_t i = first_item + static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint64_t g = grp[i];
    rank[pos[i]] = g + 1u;
    tied[i - first_item] = ((i > 0 && grp[i - 1] == g) || (i + 1 < count && grp[i + 1] == g)) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void wide_compact_kernel(const uint32_t* __restrict__ tied, const uint32_t* __restrict__ slot, const uint64_t* __restrict__ pos,
                                                           const uint64_t* __restrict__ grp, uint64_t count, uint64_t first_item, uint64_t out_base,
                                                           uint64_t* __restrict__ out_rows, uint64_t* __restrict__ out_pos, uint64_t* __restrict__ out_grp) {
    const uint64_t i = first_item + static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count || !tied[i - first_item]) return;
    const uint64_t o = out_base + slot[i - first_item];
    out_rows[o] = i;
    out_pos[o] = pos[i];
    out_grp[o] = grp[i];
}
// ---- the rounds over the tied suffixes (count < 2^32)
__global__ __launch_bounds__(256) void wide_doubling_keys_kernel(const uint64_t* __restrict__ pos, const uint64_t* __restrict__ grp, const uint64_t* __restrict__ rank,
                                                                 uint64_t count, uint64_t depth, __uint128_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    // (a tied suffix is at least `depth` long -- shorter ones were unique by their length field -- so pos + depth <= len: rank[len] = 0)
    keys[i] = (static_cast<__uint128_t>(grp[i]) << 64) | rank[pos[i] + depth];
    idx[i] = static_cast<uint32_t>(i);
}
__global__ __launch_bounds__(256) void wide_apply_order_kernel(const uint32_t* __restrict__ order, const uint64_t* __restrict__ pos, const uint64_t* __restrict__ rows,
                                                               uint64_t count, uint64_t* __restrict__ new_pos, uint64_t* __restrict__ sa) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint64_t p = pos[order[i]];
    new_pos[i] = p;
    sa[rows[i]] = p;
}
__global__ __launch_bounds__(256) void wide_round_heads_kernel(const __uint128_t* __restrict__ keys, const uint64_t* __restrict__ rows, uint64_t count,
                                                               uint64_t* __restrict__ start) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    start[i] = (i == 0 || keys[i] != keys[i - 1]) ? rows[i] : 0ull;
}
__global__ __launch_bounds__(256) void wide_round_rank_ties_kernel(const uint64_t* __restrict__ grp, const uint64_t* __restrict__ pos, uint64_t count,
                                                                   uint64_t* __restrict__ rank, uint32_t* __restrict__ tied) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint64_t g = grp[i];
    rank[pos[i]] = g + 1u;
    tied[i] = ((i > 0 && grp[i - 1] == g) || (i + 1 < count && grp[i + 1] == g)) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void wide_round_compact_kernel(const uint32_t* __restrict__ tied, const uint32_t* __restrict__ slot, const uint64_t* __restrict__ rows,
                                                                 const uint64_t* __restrict__ pos, const uint64_t* __restrict__ grp, uint64_t count,
                                                                 uint64_t* __restrict__ out_rows, uint64_t* __restrict__ out_pos, uint64_t* __restrict__ out_grp) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count || !tied[i]) return;
    const uint32_t o = slot[i];
    out_rows[o] = rows[i];
    out_pos[o] = pos[i];
    out_grp[o] = grp[i];
}

// keys: the sorted 29-base keys (consumed: freed here), sa: the positions in that order -- on return the suffix array
int wide_prefix_doubling(DevBuf& keys, uint64_t* sa, uint64_t len, PhaseClock& clock) {
    constexpr uint64_t kPiece = 1ull << 30;
    int rc;
    DevBuf grp, rank, tied, slot, scan_tmp;
    if ((rc = grp.alloc(len * 8)) || (rc = rank.alloc((len + 1) * 8)) || (rc = tied.alloc((kPiece + 1) * 4)) || (rc = slot.alloc((kPiece + 1) * 4))) return rc;
    size_t sb1 = 0, sb2 = 0;
    FMSI_HIP_TRY(rocprim::inclusive_scan(nullptr, sb1, grp.as<uint64_t>(), grp.as<uint64_t>(), static_cast<size_t>(kPiece), U64Max()));
    FMSI_HIP_TRY(rocprim::exclusive_scan(nullptr, sb2, tied.as<uint32_t>(), slot.as<uint32_t>(), 0u, static_cast<size_t>(kPiece + 1), rocprim::plus<uint32_t>()));
    if ((rc = scan_tmp.alloc(std::max(sb1, sb2)))) return rc;
    // ---- groups of equal keys over all suffixes: first rows by a max-scan, piece by piece
    for (uint64_t at = 0; at < len; at += kPiece) {
        const uint64_t cnt = std::min(kPiece, len - at);
        hipLaunchKernelGGL(wide_heads_kernel, dim3(grid_for(cnt)), dim3(256), 0, 0, keys.as<uint64_t>(), len, at, grp.as<uint64_t>());
        size_t b1 = sb1;
        FMSI_HIP_TRY(rocprim::inclusive_scan(scan_tmp.p, b1, grp.as<uint64_t>() + at, grp.as<uint64_t>() + at, static_cast<size_t>(cnt), U64Max()));
    }
    FMSI_HIP_TRY(hipDeviceSynchronize());
    (void)dev_free(keys.release());   // (only their equality was needed)
    FMSI_HIP_TRY(hipMemset(rank.p, 0, (len + 1) * 8));   // rank[len] = 0: the empty suffix
    // ---- ranks, and the tied suffixes counted (pass 0) and kept (pass 1)
    DevBuf t_rows, t_pos, t_grp;
    uint64_t n_tied = 0;
    for (int pass = 0; pass < 2; ++pass) {
        uint64_t base = 0;
        for (uint64_t at = 0; at < len; at += kPiece) {
            const uint64_t cnt = std::min(kPiece, len - at);
            hipLaunchKernelGGL(wide_rank_ties_kernel, dim3(grid_for(cnt)), dim3(256), 0, 0, grp.as<uint64_t>(), sa, len, at, rank.as<uint64_t>(), tied.as<uint32_t>());
            FMSI_HIP_TRY(hipMemsetAsync(tied.as<uint32_t>() + cnt, 0, 4, 0));
            size_t b2 = sb2;
            FMSI_HIP_TRY(rocprim::exclusive_scan(scan_tmp.p, b2, tied.as<uint32_t>(), slot.as<uint32_t>(), 0u, static_cast<size_t>(cnt + 1), rocprim::plus<uint32_t>()));
            uint32_t t = 0;
            FMSI_HIP_TRY(hipMemcpy(&t, slot.as<uint32_t>() + cnt, 4, hipMemcpyDeviceToHost));
            if (pass == 1 && t) {
                hipLaunchKernelGGL(wide_compact_kernel, dim3(grid_for(cnt)), dim3(256), 0, 0, tied.as<uint32_t>(), slot.as<uint32_t>(), sa, grp.as<uint64_t>(), len, at, base,
                                   t_rows.as<uint64_t>(), t_pos.as<uint64_t>(), t_grp.as<uint64_t>());
                FMSI_HIP_TRY(hipGetLastError());
            }
            base += t;
        }
        if (pass == 0) {
            n_tied = base;
            if (n_tied == 0) return 0;
            if (n_tied > kMaxSortItems) {
                set_last_error("the on-device builder for 2^32 rows and more: " + std::to_string(n_tied) + " suffixes share their first 29 bases with another -- more than one sort takes");
                return FMSI_HIP_ERANGE;
            }
            if ((rc = t_rows.alloc(n_tied * 8)) || (rc = t_pos.alloc(n_tied * 8)) || (rc = t_grp.alloc(n_tied * 8))) return rc;
        }
    }
    FMSI_HIP_TRY(hipDeviceSynchronize());
    (void)dev_free(grp.release());
    clock.mark("ties: ranks + tied suffixes");
    // ---- the rounds
    for (uint64_t depth = kKeyBases; n_tied; depth *= 2) {
        DevBuf k_a, k_b, i_a, i_b, new_pos, sort_tmp, r_grp, r_tied, r_slot, n_rows, n_pos, n_grp, r_scan;
        if ((rc = k_a.alloc(n_tied * 16)) || (rc = k_b.alloc(n_tied * 16)) || (rc = i_a.alloc(n_tied * 4)) || (rc = i_b.alloc(n_tied * 4)) || (rc = new_pos.alloc(n_tied * 8)) ||
            (rc = r_grp.alloc(n_tied * 8)) || (rc = r_tied.alloc((n_tied + 1) * 4)) || (rc = r_slot.alloc((n_tied + 1) * 4)))
            return rc;
        hipLaunchKernelGGL(wide_doubling_keys_kernel, dim3(grid_for(n_tied)), dim3(256), 0, 0, t_pos.as<uint64_t>(), t_grp.as<uint64_t>(), rank.as<uint64_t>(), n_tied, depth,
                           k_a.as<__uint128_t>(), i_a.as<uint32_t>());
        rocprim::double_buffer<__uint128_t> dk2(k_a.as<__uint128_t>(), k_b.as<__uint128_t>());
        rocprim::double_buffer<uint32_t> dv2(i_a.as<uint32_t>(), i_b.as<uint32_t>());
        size_t tb = 0;
        FMSI_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, dk2, dv2, static_cast<size_t>(n_tied), 0, 128, 0));
        if ((rc = sort_tmp.alloc(tb))) return rc;
        FMSI_HIP_TRY(rocprim::radix_sort_pairs(sort_tmp.p, tb, dk2, dv2, static_cast<size_t>(n_tied), 0, 128, 0));
        hipLaunchKernelGGL(wide_apply_order_kernel, dim3(grid_for(n_tied)), dim3(256), 0, 0, dv2.current(), t_pos.as<uint64_t>(), t_rows.as<uint64_t>(), n_tied,
                           new_pos.as<uint64_t>(), sa);
        // the groups of this round (rows stay in place: t_rows), ranks of the settled ones, what is still tied
        hipLaunchKernelGGL(wide_round_heads_kernel, dim3(grid_for(n_tied)), dim3(256), 0, 0, dk2.current(), t_rows.as<uint64_t>(), n_tied, r_grp.as<uint64_t>());
        size_t s1 = 0, s2 = 0;
        FMSI_HIP_TRY(rocprim::inclusive_scan(nullptr, s1, r_grp.as<uint64_t>(), r_grp.as<uint64_t>(), static_cast<size_t>(n_tied), U64Max()));
        FMSI_HIP_TRY(rocprim::exclusive_scan(nullptr, s2, r_tied.as<uint32_t>(), r_slot.as<uint32_t>(), 0u, static_cast<size_t>(n_tied + 1), rocprim::plus<uint32_t>()));
        if ((rc = r_scan.alloc(std::max(s1, s2)))) return rc;
        FMSI_HIP_TRY(rocprim::inclusive_scan(r_scan.p, s1, r_grp.as<uint64_t>(), r_grp.as<uint64_t>(), static_cast<size_t>(n_tied), U64Max()));
        hipLaunchKernelGGL(wide_round_rank_ties_kernel, dim3(grid_for(n_tied)), dim3(256), 0, 0, r_grp.as<uint64_t>(), new_pos.as<uint64_t>(), n_tied, rank.as<uint64_t>(),
                           r_tied.as<uint32_t>());
        FMSI_HIP_TRY(hipMemsetAsync(r_tied.as<uint32_t>() + n_tied, 0, 4, 0));
        FMSI_HIP_TRY(rocprim::exclusive_scan(r_scan.p, s2, r_tied.as<uint32_t>(), r_slot.as<uint32_t>(), 0u, static_cast<size_t>(n_tied + 1), rocprim::plus<uint32_t>()));
        uint32_t still = 0;
        FMSI_HIP_TRY(hipMemcpy(&still, r_slot.as<uint32_t>() + n_tied, 4, hipMemcpyDeviceToHost));
        if (still) {
            if ((rc = n_rows.alloc(static_cast<size_t>(still) * 8)) || (rc = n_pos.alloc(static_cast<size_t>(still) * 8)) || (rc = n_grp.alloc(static_cast<size_t>(still) * 8))) return rc;
            hipLaunchKernelGGL(wide_round_compact_kernel, dim3(grid_for(n_tied)), dim3(256), 0, 0, r_tied.as<uint32_t>(), r_slot.as<uint32_t>(), t_rows.as<uint64_t>(),
                               new_pos.as<uint64_t>(), r_grp.as<uint64_t>(), n_tied, n_rows.as<uint64_t>(), n_pos.as<uint64_t>(), n_grp.as<uint64_t>());
            FMSI_HIP_TRY(hipGetLastError());
        }
        FMSI_HIP_TRY(hipDeviceSynchronize());
        std::swap(t_rows.p, n_rows.p);
        std::swap(t_pos.p, n_pos.p);
        std::swap(t_grp.p, n_grp.p);
        n_tied = still;
        if (depth > 2 * len) {
            set_last_error("suffix sort (64-bit rows): prefix doubling did not converge");
            return FMSI_HIP_EHIP;
        }
    }
    clock.mark("ties: prefix doubling");
    return 0;
}

__global__ __launch_bounds__(256) void emit_rows_wide_kernel(const uint64_t* __restrict__ text2, const uint64_t* __restrict__ mask_text,
                                                             const uint64_t* __restrict__ sa_sorted, uint64_t len, int k, int all_ones_mask,
                                                             uint32_t* __restrict__ blocks, U4* __restrict__ block_counts,
                                                             uint32_t* __restrict__ mask_counts, unsigned long long* __restrict__ dollar_row, uint64_t first_row) {
    __shared__ uint32_t s_cnt[4][5];
    const uint64_t n = len + 1;
    const uint64_t row = first_row + static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint32_t wave_in_block = threadIdx.x >> 6;
    uint32_t sym = 0, mbit = 0;
    const bool valid = row < n;
    if (valid) {
        const uint64_t s = row == 0 ? len : sa_sorted[row - 1];
        if (s == 0) *dollar_row = row;                                                          // fms_index.h:414-415
        else sym = static_cast<uint32_t>(text2[(s - 1) >> 5] >> (2 * ((s - 1) & 31))) & 3u;   // :417
        if (s != len) {                                                                         // :419-421
            if (all_ones_mask) mbit = s + static_cast<uint64_t>(k) <= len;
            else mbit = static_cast<uint32_t>(mask_text[s >> 6] >> (s & 63)) & 1u;
        }
    }
    const uint64_t hi = __ballot(valid && (sym & 2u));
    const uint64_t lo = __ballot(valid && (sym & 1u));
    const uint64_t mm = __ballot(valid && mbit);
    const uint64_t vv = __ballot(valid);
    if ((threadIdx.x & 63) == 0) {
        const uint64_t w = row >> 6;
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
        const uint64_t b = (first_row >> kBlockRowsLog2) + static_cast<uint64_t>(blockIdx.x) * 2 + threadIdx.x;
        const uint32_t w0 = threadIdx.x * 2, w1 = w0 + 1;
        block_counts[b] = U4{s_cnt[w0][0] + s_cnt[w1][0], s_cnt[w0][1] + s_cnt[w1][1], s_cnt[w0][2] + s_cnt[w1][2],
                             s_cnt[w0][3] + s_cnt[w1][3]};
        mask_counts[b] = s_cnt[w0][4] + s_cnt[w1][4];
    }
}

int build_wide_from_device_text(DevBuf& text2, uint64_t len, DevBuf* mask_text, int k, int with_klcp, int device, int prefix_len, fmsi_hip_index** out) {
    TraceRange trace_range("fmsi: build index on device (64-bit rows)");
    PhaseClock clock{"wide build"};
    if (len > (1ull << 34)) {
        set_last_error("superstring too long for the on-device builder (2^34 bases)");
        return FMSI_HIP_ERANGE;
    }
    const uint64_t n = len + 1;
    const uint64_t n_blocks = ((n + kBlockRows - 1) >> kBlockRowsLog2) + 1;
    int rc;
    // ---- the suffixes into 16 buckets by their first two bases
    DevBuf hist, cursor, k0, p0, k1, p1;
    if ((rc = hist.alloc(20 * 8)) || (rc = cursor.alloc(16 * 8)) || (rc = k0.alloc(len * 8)) || (rc = p0.alloc(len * 8)) || (rc = k1.alloc(len * 8)) ||
        (rc = p1.alloc(len * 8)))
        return rc;
    FMSI_HIP_TRY(hipMemset(hist.p, 0, 20 * 8));
    hipLaunchKernelGGL(wide_hist_kernel, dim3(256 * 32), dim3(256), 0, 0, text2.as<uint64_t>(), len, hist.as<unsigned long long>());
    FMSI_HIP_TRY(hipGetLastError());
    unsigned long long h_hist[20], h_off[17];
    FMSI_HIP_TRY(hipMemcpy(h_hist, hist.p, 20 * 8, hipMemcpyDeviceToHost));
    h_off[0] = 0;
    for (int b = 0; b < 16; ++b) {
        h_off[b + 1] = h_off[b] + h_hist[b];
        if (h_hist[b] > kMaxSortItems) {
            set_last_error("a first-two-bases bucket of the suffixes holds 2^32 items or more: not for this builder");
            return FMSI_HIP_ERANGE;
        }
    }
    if (h_off[16] != len) {
        set_last_error("wide builder: the bucket histogram does not add up");
        return FMSI_HIP_EHIP;
    }
    FMSI_HIP_TRY(hipMemcpy(cursor.p, h_off, 16 * 8, hipMemcpyHostToDevice));
    constexpr uint64_t kPiece = 1ull << 30;
    for (uint64_t at = 0; at < len; at += kPiece) {
        const uint64_t cnt = std::min(kPiece, len - at);
        hipLaunchKernelGGL(wide_scatter_kernel, dim3(static_cast<uint32_t>((cnt + 255) / 256)), dim3(256), 0, 0, text2.as<uint64_t>(), len, cursor.as<unsigned long long>(),
                           k0.as<uint64_t>(), p0.as<uint64_t>(), at);
    }
    FMSI_HIP_TRY(hipGetLastError());
    clock.mark("suffixes dealt into 16 buckets");
    // ---- every bucket sorted by its 29-base keys
    {
        size_t tb_max = 0;
        for (int b = 0; b < 16; ++b) {
            size_t tb = 0;
            if (h_hist[b]) FMSI_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, k0.as<uint64_t>(), k1.as<uint64_t>(), p0.as<uint64_t>(), p1.as<uint64_t>(), static_cast<size_t>(h_hist[b]), 0, 60, 0));
            tb_max = std::max(tb_max, tb);
        }
        DevBuf temp;
        if ((rc = temp.alloc(tb_max))) return rc;
        for (int b = 0; b < 16; ++b) {
            if (!h_hist[b]) continue;
            size_t tb = tb_max;
            FMSI_HIP_TRY(rocprim::radix_sort_pairs(temp.p, tb, k0.as<uint64_t>() + h_off[b], k1.as<uint64_t>() + h_off[b], p0.as<uint64_t>() + h_off[b],
                                                   p1.as<uint64_t>() + h_off[b], static_cast<size_t>(h_hist[b]), 0, 60, 0));
        }
        FMSI_HIP_TRY(hipDeviceSynchronize());
    }
    (void)dev_free(k0.release());
    (void)dev_free(p0.release());
    clock.mark("buckets sorted");
    // ---- ties
    {
        DevBuf stats;
        if ((rc = stats.alloc(16))) return rc;
        FMSI_HIP_TRY(hipMemset(stats.p, 0, 16));
        for (uint64_t at = 0; at < len; at += kPiece) {
            const uint64_t cnt = std::min(kPiece, len - at);
            hipLaunchKernelGGL(wide_ties_kernel, dim3(static_cast<uint32_t>((cnt + 255) / 256)), dim3(256), 0, 0, k1.as<uint64_t>(), p1.as<uint64_t>(), len, text2.as<uint64_t>(), len,
                               stats.as<unsigned long long>(), at);
        }
        FMSI_HIP_TRY(hipGetLastError());
        unsigned long long h_stats[2] = {0, 0};
        FMSI_HIP_TRY(hipMemcpy(h_stats, stats.p, 16, hipMemcpyDeviceToHost));
        // groups beyond the insertion sort's cap (a repetitive text: any real genome of this size), or on request: prefix doubling over
        // ALL tied suffixes (what the group-by-group pass reordered inside its groups does not matter to it)
        static const bool force_doubling = std::getenv("FMSI_HIP_WIDE_DOUBLING") != nullptr;
        if (h_stats[1] || force_doubling) {
            clock.mark("ties (group by group)");
            if ((rc = wide_prefix_doubling(k1, p1.as<uint64_t>(), len, clock))) return rc;
        }
    }
    if (k1.p) (void)dev_free(k1.release());
    clock.mark("ties");
    // ---- rows -> blocks
    DevBuf blocks, counts, counts_scanned, mcounts, mrank, dollar, scan_temp, klcp;
    const size_t blocks_bytes = n_blocks * kBlockBytes;
    const uint64_t klcp_words = (n + 63) / 64 + 1;
    if ((rc = blocks.alloc(blocks_bytes)) || (rc = counts.alloc((n_blocks + 1) * sizeof(U4))) || (rc = counts_scanned.alloc((n_blocks + 1) * sizeof(U4))) ||
        (rc = mcounts.alloc((n_blocks + 1) * 4)) || (rc = mrank.alloc((n_blocks + 2) * 4)) || (rc = dollar.alloc(8)) || (with_klcp && (rc = klcp.alloc(klcp_words * 8))))
        return rc;
    if (with_klcp) {   // the kLCP array (what `fmsi index` writes as .klcp; this handle's own search does not read it)
        FMSI_HIP_TRY(hipMemset(klcp.p, 0, klcp_words * 8));
        for (uint64_t at = 0; at < n; at += kPiece) {
            const uint64_t cnt = std::min(kPiece, n - at);
            hipLaunchKernelGGL((klcp_kernel<uint64_t>), dim3(static_cast<uint32_t>((cnt + 255) / 256)), dim3(256), 0, 0, text2.as<uint64_t>(), p1.as<uint64_t>(), len, k,
                               klcp.as<uint64_t>(), at);
        }
        FMSI_HIP_TRY(hipGetLastError());
    }
    FMSI_HIP_TRY(hipMemset(blocks.p, 0, blocks_bytes));
    FMSI_HIP_TRY(hipMemset(counts.p, 0, (n_blocks + 1) * sizeof(U4)));
    FMSI_HIP_TRY(hipMemset(mcounts.p, 0, (n_blocks + 1) * 4));
    for (uint64_t at = 0; at < n; at += kPiece) {   // (kPiece is a multiple of 256 rows = two blocks)
        const uint64_t cnt = std::min(kPiece, n - at);
        hipLaunchKernelGGL(emit_rows_wide_kernel, dim3(static_cast<uint32_t>((cnt + 255) / 256)), dim3(256), 0, 0, text2.as<uint64_t>(),
                           mask_text ? mask_text->as<uint64_t>() : nullptr, p1.as<uint64_t>(), len, k, mask_text ? 0 : 1, blocks.as<uint32_t>(), counts.as<U4>(),
                           mcounts.as<uint32_t>(), dollar.as<unsigned long long>(), at);
    }
    FMSI_HIP_TRY(hipGetLastError());
    size_t st1 = 0, st2 = 0;
    FMSI_HIP_TRY(rocprim::exclusive_scan(nullptr, st1, counts.as<U4>(), counts_scanned.as<U4>(), U4{0, 0, 0, 0}, static_cast<size_t>(n_blocks + 1), U4Plus()));
    FMSI_HIP_TRY(rocprim::exclusive_scan(nullptr, st2, mcounts.as<uint32_t>(), mrank.as<uint32_t>(), 0u, static_cast<size_t>(n_blocks + 1), rocprim::plus<uint32_t>()));
    if ((rc = scan_temp.alloc(std::max(st1, st2)))) return rc;
    FMSI_HIP_TRY(rocprim::exclusive_scan(scan_temp.p, st1, counts.as<U4>(), counts_scanned.as<U4>(), U4{0, 0, 0, 0}, static_cast<size_t>(n_blocks + 1), U4Plus()));
    FMSI_HIP_TRY(rocprim::exclusive_scan(scan_temp.p, st2, mcounts.as<uint32_t>(), mrank.as<uint32_t>(), 0u, static_cast<size_t>(n_blocks + 1), rocprim::plus<uint32_t>()));
    hipLaunchKernelGGL(scatter_counts_kernel, dim3(grid_for(n_blocks)), dim3(256), 0, 0, counts_scanned.as<U4>(), blocks.as<uint32_t>(), n_blocks);
    FMSI_HIP_TRY(hipGetLastError());
    unsigned long long dollar_row = 0;
    FMSI_HIP_TRY(hipMemcpy(&dollar_row, dollar.p, 8, hipMemcpyDeviceToHost));
    FMSI_HIP_TRY(hipDeviceSynchronize());
    (void)dev_free(p1.release());
    clock.mark("rows -> blocks");
    // ---- the table of absolute counts per superblock (on the host, from the wrapped counts: fmsi_wide.hip)
    std::vector<uint64_t> super;
    {
        std::vector<uint32_t> h_cnt(n_blocks * 4), h_mrank(n_blocks + 1);
        FMSI_HIP_TRY(hipMemcpy2D(h_cnt.data(), 16, blocks.p, kBlockBytes, 16, n_blocks, hipMemcpyDeviceToHost));   // dw 0..3 of every block
        FMSI_HIP_TRY(hipMemcpy(h_mrank.data(), mrank.p, (n_blocks + 1) * 4, hipMemcpyDeviceToHost));
        const uint32_t sb_log2 = 24;
        const uint64_t n_super = ((n_blocks - 1) >> sb_log2) + 1;
        super.assign(n_super * 8, 0);
        uint64_t run[5] = {0, 0, 0, 0, 0};
        for (uint64_t b = 0; b < n_blocks; ++b) {
            if ((b & ((1ull << sb_log2) - 1)) == 0)
                for (int s5 = 0; s5 < 5; ++s5) super[(b >> sb_log2) * 8 + s5] = run[s5];
            if (b + 1 < n_blocks) {
                for (int c4 = 0; c4 < 4; ++c4) run[c4] += static_cast<uint32_t>(h_cnt[(b + 1) * 4 + c4] - h_cnt[b * 4 + c4]);
                run[4] += static_cast<uint32_t>(h_mrank[b + 1] - h_mrank[b]);
            }
        }
    }
    auto* idx = new fmsi_hip_index();
    idx->device = device;
    idx->n = n;
    idx->k = k;
    idx->n_blocks = n_blocks;
    idx->d_blocks = blocks.release();
    idx->d_mrank = mrank.release();
    idx->d_text2 = text2.release();
    idx->text_len = len;
    idx->device_bytes = blocks_bytes + (n_blocks + 2) * 4 + ((len >> 5) + 2) * 8 + super.size() * 8;
    idx->wide = true;
    if (with_klcp) {
        idx->d_klcp = klcp.release();
        idx->has_klcp = 1;
        idx->device_bytes += klcp_words * 8;
    }
    if (dev_malloc(&idx->d_super, super.size() * 8) != hipSuccess || hipMemcpy(idx->d_super, super.data(), super.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
        dev_malloc(reinterpret_cast<void**>(&idx->d_counters), FMSI_HIP_N_COUNTERS * 8) != hipSuccess) {
        fmsi_hip_index_destroy(idx);
        set_last_error("hipMalloc failed");
        return FMSI_HIP_ENOMEM;
    }
    idx->counts64[0] = 1;                                  // counts = {1, #A+1, ...}, fms_index.h:451 (the `$` slot counts as an 'A' row)
    idx->counts64[1] = h_hist[16] + 1;
    idx->counts64[2] = idx->counts64[1] + h_hist[17];
    idx->counts64[3] = idx->counts64[2] + h_hist[18];
    idx->dollar64 = dollar_row;
    idx->devw.blocks = static_cast<const uint4*>(idx->d_blocks);
    idx->devw.mrank = static_cast<const uint32_t*>(idx->d_mrank);
    idx->devw.super = static_cast<const uint64_t*>(idx->d_super);
    idx->devw.sb_log2 = 24;
    idx->devw.n = n;
    idx->devw.dollar = dollar_row;
    idx->devw.C0 = idx->counts64[0];
    idx->devw.C1 = idx->counts64[1];
    idx->devw.C2 = idx->counts64[2];
    idx->devw.C3 = idx->counts64[3];
    idx->devw.ptable = nullptr;
    idx->devw.ptable_len = 0;
    clock.mark("superblock table, handle");
    if (prefix_len > 0 && (rc = build_prefix_table_wide(idx, prefix_len))) {
        fmsi_hip_index_destroy(idx);
        return rc;
    }
    clock.mark("prefix table");
    *out = idx;
    return 0;
}

int check_device(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        set_last_error("no usable HIP device (libfmsi_hip has no CPU fallback)");
        return FMSI_HIP_ENODEV;
    }
    return 0;
}

// synthetic mask: position p < len - k + 1 is ON with probability mask_permille / 1000 (a hash of the seed and p), the last
// k - 1 positions are OFF (no k-mer starts there).  One thread per 64-position word.
__host__ __device__ inline uint32_t synth_mask_bit(uint64_t seed, uint64_t p, uint32_t mask_permille) {
    if (mask_permille >= 1000u) return 1u;
    return (splitmix64(seed ^ 0x3A5C0FF5E7ull ^ (p * 0xC2B2AE3D27D4EB4Full)) % 1000ull) < mask_permille ? 1u : 0u;
}
__global__ void synth_mask_kernel(uint64_t* __restrict__ mask, uint64_t m_words, uint64_t len, int k, uint64_t seed, uint32_t mask_permille) {
    const uint64_t w = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (w >= m_words) return;
    uint64_t v = 0;
    for (uint32_t b = 0; b < 64u; ++b) {
        const uint64_t p = w * 64u + b;
        if (p + static_cast<uint64_t>(k) <= len && synth_mask_bit(seed, p, mask_permille)) v |= 1ull << b;
    }
    mask[w] = v;
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

namespace {
// characters [0, count) of a slice that starts at text position `at` (a multiple of 64) -> 2-bit text words and mask words; one thread
// per 64 positions; *first_bad = the smallest position holding anything but ACGTacgt
__global__ __launch_bounds__(256) void pack_chars_kernel(const char* __restrict__ chars, uint64_t count, uint64_t at, uint64_t* __restrict__ text2,
                                                         uint64_t* __restrict__ maskw, unsigned long long* __restrict__ first_bad) {
    const uint64_t w = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (w * 64u >= count) return;
    uint64_t t0 = 0, t1 = 0, m = 0;
    for (uint32_t b = 0; b < 64u; ++b) {
        const uint64_t p = w * 64u + b;
        if (p >= count) break;
        const uint32_t ch = static_cast<uint8_t>(chars[p]);
        const uint32_t lc = ch | 0x20u;
        const uint64_t c = lc == 'a' ? 0u : (lc == 'c' ? 1u : (lc == 'g' ? 2u : 3u));
        if (c == 3u && lc != 't') atomicMin(first_bad, static_cast<unsigned long long>(at + p));
        if (b < 32u) t0 |= c << (2u * b);
        else t1 |= c << (2u * (b - 32u));
        if (ch >= 'A' && ch <= 'Z') m |= 1ull << b;
    }
    const uint64_t gw = (at >> 6) + w;
    text2[2 * gw] = t0;
    text2[2 * gw + 1] = t1;
    maskw[gw] = m;
}
}  // namespace

int fmsi_hip_index_build(const char* ms, uint64_t len, int k, int with_klcp, int device, int prefix_len, fmsi_hip_index** out) {
    ::fmsi::ScratchScope scratch_scope_(device);   // (devpool.h: the outermost builder returns the block cache to the driver)
    if (!ms || !out || k < 1 || (with_klcp && k > 64)) {
        set_last_error("bad argument (on-device kLCP construction needs k <= 64)");
        return FMSI_HIP_EINVAL;
    }
    int rc = check_device(device);
    if (rc) return rc;
    FMSI_HIP_ON_DEVICE(device);
    // the characters go up as they are (in slices, so that a superstring of gigabytes needs no second copy of itself on the device) and a
    // kernel packs them: 2-bit codes (nucleotideToInt, src/kmers.h:3-20), mask bit = upper case (is_upper, :71-73), and the position of
    // the first character that is not ACGTacgt, if there is one
    const uint64_t n_words = (len >> 5) + 2, m_words = (len >> 6) + 2;
    DevBuf text2, mask_text, bad;
    // (through two page-locked staging buffers filled by a few host threads: the caller's memory may be anything -- the CLI hands over
    // a mapping of the file, which the driver copies from at a fraction of the link's rate)
    const uint64_t slice = std::min<uint64_t>((len + 63) / 64 * 64, 1ull << 26);   // a multiple of 64
    if ((rc = text2.alloc(n_words * 8)) || (rc = mask_text.alloc(m_words * 8)) || (rc = bad.alloc(8))) return rc;
    FMSI_HIP_TRY(hipMemset(text2.p, 0, n_words * 8));
    FMSI_HIP_TRY(hipMemset(mask_text.p, 0, m_words * 8));
    FMSI_HIP_TRY(hipMemset(bad.p, 0xFF, 8));
    struct Stage {
        void* host = nullptr;
        DevBuf dev;
        hipEvent_t done = nullptr;
        ~Stage() {
            if (done) (void)hipEventDestroy(done);
            if (host) (void)hipHostFree(host);
        }
    } stage[2];
    for (Stage& st : stage) {
        if (hipHostMalloc(&st.host, slice, hipHostMallocDefault) != hipSuccess) st.host = nullptr;   // (then straight from the caller's memory)
        if ((rc = st.dev.alloc(slice + 64))) return rc;
        FMSI_HIP_TRY(hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
    }
    const unsigned copiers = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    unsigned turn = 0;
    for (uint64_t at = 0; at < len; at += slice, turn ^= 1u) {
        Stage& st = stage[turn];
        const uint64_t count = std::min(slice, len - at);
        FMSI_HIP_TRY(hipEventSynchronize(st.done));   // the kernel that read this buffer two slices ago
        const char* src = ms + at;
        if (st.host) {
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < copiers; ++t) {
                const uint64_t lo = count / copiers * t, hi = t + 1 == copiers ? count : count / copiers * (t + 1);
                pool.emplace_back([&, lo, hi] { std::memcpy(static_cast<char*>(st.host) + lo, ms + at + lo, hi - lo); });
            }
            for (std::thread& th : pool) th.join();
            src = static_cast<const char*>(st.host);
        }
        FMSI_HIP_TRY(hipMemcpyAsync(st.dev.p, src, count, hipMemcpyHostToDevice, 0));
        hipLaunchKernelGGL(pack_chars_kernel, dim3(grid_for((count + 63) / 64)), dim3(256), 0, 0, st.dev.as<char>(), count, at, text2.as<uint64_t>(),
                           mask_text.as<uint64_t>(), bad.as<unsigned long long>());
        FMSI_HIP_TRY(hipGetLastError());
        FMSI_HIP_TRY(hipEventRecord(st.done, 0));
        if (!st.host) FMSI_HIP_TRY(hipDeviceSynchronize());   // (pageable source: the copy must have read it before the caller goes on)
    }
    FMSI_HIP_TRY(hipDeviceSynchronize());
    unsigned long long first_bad = ~0ull;
    FMSI_HIP_TRY(hipMemcpy(&first_bad, bad.p, 8, hipMemcpyDeviceToHost));
    if (first_bad != ~0ull) {
        set_last_error("masked superstring contains a non-ACGT character at " + std::to_string(first_bad));
        return FMSI_HIP_EINVAL;
    }
    return build_from_device_text(text2, len, &mask_text, k, with_klcp, device, prefix_len, out);
}

int fmsi_hip_index_build_synthetic(uint64_t len, uint64_t seed, int k, int with_klcp, int device, int prefix_len, fmsi_hip_index** out) {
    return fmsi_hip_index_build_synthetic_repeats(len, seed, k, with_klcp, device, prefix_len, 0, 0, 0, out);
}

int fmsi_hip_index_build_synthetic_repeats(uint64_t len, uint64_t seed, int k, int with_klcp, int device, int prefix_len,
                                           uint32_t repeat_permille, uint32_t families, uint32_t divergence_permille, fmsi_hip_index** out) {
    if (!out || k < 1 || len < static_cast<uint64_t>(k) || (with_klcp && k > 64) || repeat_permille > 1000 || divergence_permille > 1000) {
        set_last_error("bad argument (on-device kLCP construction needs k <= 64)");
        return FMSI_HIP_EINVAL;
    }
    fmsi_hip_synthetic_spec spec;
    spec.len = len;
    spec.seed = seed;
    spec.k = k;
    spec.with_klcp = with_klcp;
    spec.prefix_len = prefix_len;
    spec.repeat_permille = repeat_permille;
    spec.families = families;
    spec.divergence_permille = divergence_permille;
    spec.mask_permille = 1000;
    return fmsi_hip_index_build_synthetic_spec(&spec, device, out);
}

int fmsi_hip_index_build_synthetic_spec(const fmsi_hip_synthetic_spec* spec, int device, fmsi_hip_index** out) {
    ::fmsi::ScratchScope scratch_scope_(device);   // (devpool.h: the outermost builder returns the block cache to the driver)
    if (!spec || !out || spec->k < 1 || spec->len < static_cast<uint64_t>(spec->k) || (spec->with_klcp && spec->k > 64) ||
        spec->repeat_permille > 1000 || spec->divergence_permille > 1000 || spec->mask_permille > 1000) {
        set_last_error("bad argument (on-device kLCP construction needs k <= 64)");
        return FMSI_HIP_EINVAL;
    }
    int rc = check_device(device);
    if (rc) return rc;
    FMSI_HIP_ON_DEVICE(device);
    const uint64_t len = spec->len, n_words = (len >> 5) + 2, m_words = (len >> 6) + 2;
    DevBuf text2, mask_text;
    if ((rc = text2.alloc(n_words * 8))) return rc;
    hipLaunchKernelGGL(synth_text_kernel, dim3(grid_for(n_words)), dim3(256), 0, 0, text2.as<uint64_t>(), n_words, len, spec->seed,
                       spec->repeat_permille, spec->families, spec->divergence_permille);
    FMSI_HIP_TRY(hipGetLastError());
    if (spec->mask_permille < 1000) {   // a mixed mask: an explicit mask bit-vector, as for a superstring given as text
        if ((rc = mask_text.alloc(m_words * 8))) return rc;
        hipLaunchKernelGGL(synth_mask_kernel, dim3(grid_for(m_words)), dim3(256), 0, 0, mask_text.as<uint64_t>(), m_words, len, spec->k, spec->seed,
                           spec->mask_permille);
        FMSI_HIP_TRY(hipGetLastError());
    }
    return build_from_device_text(text2, len, spec->mask_permille < 1000 ? &mask_text : nullptr, spec->k, spec->with_klcp, device,
                                  spec->prefix_len, out);
}

uint32_t fmsi_hip_synthetic_mask_bit(uint64_t seed, uint64_t p, uint32_t mask_permille) { return synth_mask_bit(seed, p, mask_permille); }

uint64_t fmsi_hip_synthetic_text_word(uint64_t seed, uint64_t w) { return synth_word(seed, w); }
uint64_t fmsi_hip_synthetic_text_word_repeats(uint64_t seed, uint64_t w, uint32_t repeat_permille, uint32_t families, uint32_t divergence_permille) {
    return synth_word_repeats(seed, w, repeat_permille, families, divergence_permille);
}

int fmsi_hip_synthetic_queries(fmsi_hip_index* idx, int k, uint64_t seed, uint64_t first, uint64_t n, uint64_t* d_kmers, void* stream) {
    if (!idx || !d_kmers || !idx->d_text2 || k < 1 || k > 32 || idx->text_len < static_cast<uint64_t>(k)) {
        set_last_error("synthetic queries need an index built by fmsi_hip_index_build*, k <= 32");
        return FMSI_HIP_EINVAL;
    }
    if (n == 0) return 0;
    FMSI_HIP_ON_DEVICE(idx->device);
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
    FMSI_HIP_ON_DEVICE(idx->device);
    const uint64_t n_words = (n_reads * read_len) / 32 + 2;
    hipLaunchKernelGGL(synth_reads_kernel, dim3(grid_for(n_words)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint64_t*>(idx->d_text2), idx->text_len, read_len, seed, first_read, n_reads, subst_per_million,
                       n_words, d_packed);
    FMSI_HIP_TRY(hipGetLastError());
    return 0;
}

int fmsi_hip_index_download_text(const fmsi_hip_index* idx, uint64_t* words) {
    if (!idx || !words || !idx->d_text2) return FMSI_HIP_EINVAL;
    FMSI_HIP_ON_DEVICE(idx->device);
    FMSI_HIP_TRY(hipMemcpy(words, idx->d_text2, ((idx->text_len >> 5) + 2) * 8, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"

// ---- load_index() straight into HBM (fmsi_hip_index_load) ---------------------------------------------------------------------------
// The host path (fmsi_host_index_load + fmsi_hip_index_create) decodes the RRR-63 mask and interleaves the planes with 16 host
// threads, then uploads the 64-byte blocks: 0.7-0.9 s at human scale, most of the CLI's whole-process time on a 1 GB query file.
// Here the files' payload goes to the device as it is (1.1 GB instead of 1.4 GB) and three kernels do the rest:
//   rrr_decode_kernel   a wave per 64 RRR blocks (two superblocks): lane l sums the offset widths of the blocks before it in its
//                       superblock (shuffle scan) from the superblock's sampled pointer, unranks its 63 bits by the combinatorial
//                       number system (rrr_helper.hpp:304-320; the binomials sit in LDS), and the wave emits 63 whole words
//   word popcounts + rocPRIM scan: where a 64-row word's G/T bits start in `gt`
//   layout_words_kernel a wave per 64-row word, a lane per row: its second-level bit comes from `gt` or `ac` at the lane's rank among
//                       the word's G/T (A/C) rows -- the pdep of the host builder as one ballot --, planes and per-block counts as
//                       emit_rows_kernel writes them; the same scans and scatter finish the blocks
// The result is dword for dword what fmsi_hip_index_create builds (tests/test_gpu_builder.py::test_device_load_equals_host_load).
namespace fmsi {
namespace {

struct Mapped {   // a file of the index: mapped for its headers (a few pages are ever touched), open for the payload's reads (UploadTeam)
    const uint8_t* data = nullptr;
    size_t size = 0;
    int fd = -1;
    ~Mapped() {
        if (data) munmap(const_cast<uint8_t*>(data), size);
        if (fd >= 0) ::close(fd);
    }
    bool open_file(const std::string& path) {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        bool ok = fstat(fd, &st) == 0 && st.st_size > 0;
        if (ok) {
            void* p = mmap(nullptr, static_cast<size_t>(st.st_size), PROT_READ, MAP_PRIVATE, fd, 0);
            ok = p != MAP_FAILED;
            if (ok) {
                data = static_cast<const uint8_t*>(p);
                size = static_cast<size_t>(st.st_size);
            }
        }
        return ok;
    }
};

// The payloads' way into HBM.  One hipMemcpy from the mapping moves 11 GB/s -- the runtime stages pageable memory through one buffer on
// one thread, which also takes the mapping's page faults --, a third of the whole load at human scale (1.4 GB: 0.13 s).  Here a few
// threads read() pieces of the files into page-locked slots of their own (no mapping, no faults: the page cache's pages are copied once)
// and enqueue each piece's copy; a slot is reused when its copy is done.  The link's rate, not one core's (tools/micro/upload_bench.hip:
// 16 threads read 1.4 GB in 0.024 s, the link moves it in 0.028 s).  All copies go to the default stream: a stream of its own per thread
// cost more than the whole upload (hipStreamCreate: 20 ms, 150 ms the first time on a box).
struct UploadJob {
    int fd;
    uint64_t file_off;
    char* dst;   // device
    uint64_t bytes;
};
class UploadTeam {   // start(): the threads begin; finish(): all pieces are in HBM (or the first error)
public:
    void start(const std::vector<UploadJob>& jobs, int device) {
        constexpr uint64_t kPiece = 4ull << 20;
        for (const UploadJob& j : jobs)
            for (uint64_t o = 0; o < j.bytes; o += kPiece) pieces_.push_back({&j, o, std::min(kPiece, j.bytes - o)});
        if (pieces_.empty()) return;
        const unsigned threads = static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>({8u, std::max(1u, std::thread::hardware_concurrency()), pieces_.size()})));
        for (unsigned t = 0; t < threads; ++t) pool_.emplace_back([this, device] { work(device); });
    }
    int finish() {
        for (std::thread& th : pool_) th.join();
        pool_.clear();
        if (failed_.load()) {
            (void)hipGetLastError();
            set_last_error(first_error_);
            return first_error_.rfind("short read", 0) == 0 ? FMSI_HIP_EIO : FMSI_HIP_EHIP;
        }
        return 0;
    }
    ~UploadTeam() {
        failed_.store(1);   // (an early return of the load: the threads stop at their next piece)
        for (std::thread& th : pool_) th.join();
    }

private:
    struct Piece {
        const UploadJob* job;
        uint64_t off, len;
    };
    std::vector<Piece> pieces_;
    std::vector<std::thread> pool_;
    std::atomic<size_t> next_{0};
    std::atomic<int> failed_{0};
    std::string first_error_;
    std::mutex err_m_;
    void fail(const std::string& what) {
        std::lock_guard<std::mutex> lk(err_m_);
        if (!failed_.exchange(1)) first_error_ = what;
    }
    void work(int device) {
        constexpr uint64_t kPiece = 4ull << 20;
        if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice failed in an upload thread");
        const hipStream_t stream = nullptr;
        void* slot[2] = {nullptr, nullptr};
        hipEvent_t done[2] = {nullptr, nullptr};
        bool busy[2] = {false, false};
        bool ok = true;
        for (int i = 0; ok && i < 2; ++i)
            ok = hipHostMalloc(&slot[i], kPiece, hipHostMallocDefault) == hipSuccess && hipEventCreateWithFlags(&done[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) fail("page-locked staging for the index upload could not be set up");
        unsigned turn = 0;
        for (size_t i; ok && !failed_.load() && (i = next_.fetch_add(1)) < pieces_.size(); turn ^= 1u) {
            const Piece& pc = pieces_[i];
            if (busy[turn] && hipEventSynchronize(done[turn]) != hipSuccess) {
                fail("hipEventSynchronize failed in the index upload");
                break;
            }
            uint64_t got = 0;
            while (got < pc.len) {
                const ssize_t r = pread(pc.job->fd, static_cast<char*>(slot[turn]) + got, pc.len - got, static_cast<off_t>(pc.job->file_off + pc.off + got));
                if (r < 0 && errno == EINTR) continue;
                if (r <= 0) break;
                got += static_cast<uint64_t>(r);
            }
            if (got != pc.len) {
                fail("short read of an index file");
                break;
            }
            if (hipMemcpyAsync(pc.job->dst + pc.off, slot[turn], pc.len, hipMemcpyHostToDevice, stream) != hipSuccess || hipEventRecord(done[turn], stream) != hipSuccess) {
                fail("hipMemcpyAsync failed in the index upload");
                break;
            }
            busy[turn] = true;
        }
        for (int i = 0; i < 2; ++i) {
            if (busy[i] && hipEventSynchronize(done[i]) != hipSuccess) fail("index upload: hipEventSynchronize failed");
            if (done[i]) (void)hipEventDestroy(done[i]);
            if (slot[i]) (void)hipHostFree(slot[i]);
        }
    }
};

struct Span {   // a run of u64 words inside a mapped file
    const uint8_t* p = nullptr;
    uint64_t bits = 0, words = 0;
    uint8_t width = 0;
};
struct FileCursor {
    const uint8_t* p;
    size_t left;
    bool u64(uint64_t& v) {
        if (left < 8) return false;
        std::memcpy(&v, p, 8);
        p += 8;
        left -= 8;
        return true;
    }
    bool bit_vector(Span& s) {   // int_vector<1>: u64 bit count + words
        if (!u64(s.bits)) return false;
        s.words = (s.bits + 63) >> 6;
        if (left < s.words * 8) return false;
        s.p = p;
        p += s.words * 8;
        left -= s.words * 8;
        return true;
    }
    bool int_vector(Span& s) {   // int_vector<0>: u64 bit count + u8 width + words
        if (!u64(s.bits) || left < 1) return false;
        s.width = *p;
        ++p;
        --left;
        s.words = (s.bits + 63) >> 6;
        if (left < s.words * 8) return false;
        s.p = p;
        p += s.words * 8;
        left -= s.words * 8;
        return true;
    }
};

__device__ __forceinline__ uint64_t dev_read_bits(const uint64_t* __restrict__ words, uint64_t pos, uint32_t len) {
    if (len == 0) return 0;
    const uint64_t* w = words + (pos >> 6);
    const uint32_t off = static_cast<uint32_t>(pos & 63u);
    uint64_t v = w[0] >> off;
    if (off + len > 64u) v |= w[1] << (64u - off);
    return len >= 64u ? v : (v & ((1ull << len) - 1ull));
}

__global__ __launch_bounds__(256) void rrr_decode_kernel(const uint64_t* __restrict__ bt, const uint64_t* __restrict__ btnr, uint64_t btnr_bits,
                                                         const uint64_t* __restrict__ btnrp, uint32_t btnrp_w, const uint64_t* __restrict__ invert,
                                                         const uint64_t* __restrict__ binom, const uint8_t* __restrict__ space_tab, uint64_t n,
                                                         uint64_t n_blocks, uint64_t n_super, uint64_t* __restrict__ mask_words, uint64_t n_words,
                                                         uint32_t* __restrict__ corrupt) {
    __shared__ uint64_t s_c[64 * 64];
    __shared__ uint8_t s_space[64];
    for (uint32_t i = threadIdx.x; i < 64u * 64u; i += blockDim.x) s_c[i] = binom[i];
    if (threadIdx.x < 64u) s_space[threadIdx.x] = space_tab[threadIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;   // 64 blocks = 2 superblocks = 63 words
    const uint64_t b = wave * 64u + lane, s = b >> 5;
    uint64_t v = 0;
    uint32_t width = 0, cls = 0;
    const bool have = b < n_blocks;
    if (have) {
        const uint32_t stored = static_cast<uint32_t>(dev_read_bits(bt, b * 6u, 6u));
        const bool inv = (invert[s >> 6] >> (s & 63u)) & 1ull;
        cls = inv ? 63u - stored : stored;   // rrr_vector.hpp:262-263
        width = s_space[stored];
    }
    // offset of this block's number: the superblock's sampled pointer + the widths of the blocks before it in the superblock
    uint32_t incl = width;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
        const uint32_t up = static_cast<uint32_t>(__shfl_up(static_cast<int>(incl), d, 32));
        if ((lane & 31u) >= static_cast<uint32_t>(d)) incl += up;
    }
    if (have && b * 63u < n) {   // (the dummy block of n % 63 == 0 holds nothing)
        const uint64_t base = s < n_super ? dev_read_bits(btnrp, s * btnrp_w, btnrp_w) : 0ull;
        const uint64_t pos = base + (incl - width);
        if (pos + width > btnr_bits) {
            *corrupt = 1u;
        } else if (cls == 63u) {
            v = (1ull << 63) - 1ull;
        } else if (cls != 0u) {
            uint64_t nr = dev_read_bits(btnr, pos, width);
            uint32_t left = cls;
            for (uint32_t p = 0, remaining = 63u; p < 63u && left > 0u; ++p, --remaining) {   // rrr_helper.hpp:304-320, positions from the LSB
                const uint64_t skip = s_c[(remaining - 1u) * 64u + left];
                if (nr >= skip) {
                    v |= 1ull << p;
                    nr -= skip;
                    --left;
                }
            }
        }
        const uint64_t first = b * 63u;
        if (first + 63u > n) v &= (1ull << (n - first)) - 1ull;
    }
    // 64 x 63 bits -> 63 words: word w of the wave = block w from its bit w on, then the first w + 1 bits of block w + 1
    const uint64_t nxt = static_cast<uint64_t>(__shfl_down(static_cast<long long>(v), 1, 64));
    const uint64_t out_w = wave * 63u + lane;
    if (lane < 63u && out_w < n_words) mask_words[out_w] = (v >> lane) | (nxt << (63u - lane));
}

__global__ __launch_bounds__(256) void word_popcount_kernel(const uint64_t* __restrict__ words, uint64_t n_words, uint64_t n_bits, uint32_t* __restrict__ pc) {
    const uint64_t w = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint64_t x = words[w];
    if (w == n_words - 1 && (n_bits & 63u)) x &= (1ull << (n_bits & 63u)) - 1ull;
    pc[w] = static_cast<uint32_t>(__popcll(x));
}

// a wave per 64-row word (256 threads = 4 words = 2 blocks, as emit_rows_kernel)
__global__ __launch_bounds__(256) void layout_words_kernel(const uint64_t* __restrict__ ac_gt, const uint64_t* __restrict__ ac, const uint64_t* __restrict__ gt,
                                                           const uint64_t* __restrict__ mask, const uint32_t* __restrict__ gt_off, uint64_t n,
                                                           uint64_t n_words, uint32_t* __restrict__ blocks, U4* __restrict__ block_counts,
                                                           uint32_t* __restrict__ mask_counts) {
    __shared__ uint32_t s_cnt[4][5];
    const uint32_t lane = threadIdx.x & 63u, wave_in_block = threadIdx.x >> 6;
    const uint64_t w = static_cast<uint64_t>(blockIdx.x) * 4u + wave_in_block;
    const uint64_t row = w * 64u + lane;
    const bool valid = w < n_words && row < n;
    uint64_t g = 0, m = 0, gpos = 0;
    if (w < n_words) {
        g = ac_gt[w];
        m = mask[w];
        gpos = gt_off[w];
    }
    const uint64_t vv = __ballot(valid);
    g &= vv;
    m &= vv;
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t lo_bit = 0;
    if (valid) {
        if ((g >> lane) & 1ull) {
            const uint64_t p = gpos + static_cast<uint32_t>(__popcll(g & below));
            lo_bit = static_cast<uint32_t>(gt[p >> 6] >> (p & 63u)) & 1u;
        } else {
            const uint64_t p = (w * 64u - gpos) + static_cast<uint32_t>(__popcll(~g & vv & below));
            lo_bit = static_cast<uint32_t>(ac[p >> 6] >> (p & 63u)) & 1u;
        }
    }
    const uint64_t lo = __ballot(valid && lo_bit);
    if (lane == 0) {
        if (vv) {
            uint32_t* blk = blocks + (w >> 1) * kBlockDwords;
            const uint32_t half = static_cast<uint32_t>(w & 1) * 2;
            blk[4 + half] = static_cast<uint32_t>(g);
            blk[5 + half] = static_cast<uint32_t>(g >> 32);
            blk[8 + half] = static_cast<uint32_t>(lo);
            blk[9 + half] = static_cast<uint32_t>(lo >> 32);
            blk[12 + half] = static_cast<uint32_t>(m);
            blk[13 + half] = static_cast<uint32_t>(m >> 32);
        }
        s_cnt[wave_in_block][0] = __popcll(vv & ~g & ~lo);
        s_cnt[wave_in_block][1] = __popcll(vv & ~g & lo);
        s_cnt[wave_in_block][2] = __popcll(g & ~lo);
        s_cnt[wave_in_block][3] = __popcll(g & lo);
        s_cnt[wave_in_block][4] = __popcll(m);
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const uint64_t b = static_cast<uint64_t>(blockIdx.x) * 2 + threadIdx.x;
        const uint32_t w0 = threadIdx.x * 2, w1 = w0 + 1;
        block_counts[b] = U4{s_cnt[w0][0] + s_cnt[w1][0], s_cnt[w0][1] + s_cnt[w1][1], s_cnt[w0][2] + s_cnt[w1][2], s_cnt[w0][3] + s_cnt[w1][3]};
        mask_counts[b] = s_cnt[w0][4] + s_cnt[w1][4];
    }
}

int load_fail(int code, const std::string& msg) {
    set_last_error(msg);
    return code;
}

}  // namespace
}  // namespace fmsi

extern "C" int fmsi_hip_index_load(const char* prefix_c, int load_flags, int device, int prefix_len, fmsi_hip_index** out) {
    const int use_klcp = load_flags & FMSI_HIP_LOAD_KLCP;
    const bool async_table = (load_flags & FMSI_HIP_LOAD_ASYNC_TABLE) != 0;
    ::fmsi::ScratchScope scratch_scope_(device);   // (devpool.h: the outermost builder returns the block cache to the driver)
    if (!prefix_c || !out) return load_fail(FMSI_HIP_EINVAL, "null argument");
    TraceRange trace_range("fmsi: load index files into HBM");
    PhaseClock clock{"load"};
    const std::string base = std::string(prefix_c) + ".fmsi";
    // ---- the files: headers parsed on the host, payloads stay where the page cache has them ----------------------------------
    Mapped f_acgt, f_ac, f_gt, f_mask, f_klcp;
    if (!f_acgt.open_file(base + ".ac_gt")) return load_fail(FMSI_HIP_EIO, "cannot read " + base + ".ac_gt");
    if (!f_ac.open_file(base + ".ac")) return load_fail(FMSI_HIP_EIO, "cannot read " + base + ".ac");
    if (!f_gt.open_file(base + ".gt")) return load_fail(FMSI_HIP_EIO, "cannot read " + base + ".gt");
    if (!f_mask.open_file(base + ".mask")) return load_fail(FMSI_HIP_EIO, "cannot read " + base + ".mask");
    const bool with_klcp = use_klcp && f_klcp.open_file(base + ".klcp");   // fms_index.h:513-515: only when the file exists AND -S was given
    Span acgt, ac, gt, klcp, bt, btnr, btnrp, rank_samples, invert;
    {
        FileCursor c{f_acgt.data, f_acgt.size};
        if (!c.bit_vector(acgt)) return load_fail(FMSI_HIP_EIO, "malformed bit vector " + base + ".ac_gt");
        c = FileCursor{f_ac.data, f_ac.size};
        if (!c.bit_vector(ac)) return load_fail(FMSI_HIP_EIO, "malformed bit vector " + base + ".ac");
        c = FileCursor{f_gt.data, f_gt.size};
        if (!c.bit_vector(gt)) return load_fail(FMSI_HIP_EIO, "malformed bit vector " + base + ".gt");
        if (with_klcp) {
            c = FileCursor{f_klcp.data, f_klcp.size};
            if (!c.bit_vector(klcp)) return load_fail(FMSI_HIP_EIO, "malformed bit vector " + base + ".klcp");
        }
    }
    uint64_t n = 0;
    {
        FileCursor c{f_mask.data, f_mask.size};
        if (!c.u64(n) || !c.int_vector(bt) || !c.bit_vector(btnr) || !c.int_vector(btnrp) || !c.int_vector(rank_samples) || !c.bit_vector(invert))
            return load_fail(FMSI_HIP_EIO, "truncated rrr_vector<63> image");
    }
    const uint64_t n_rrr_blocks = (n + 63) / 63, n_super = (n_rrr_blocks + 31) / 32;   // incl. the dummy block when n % 63 == 0 (rrr_vector.hpp:148)
    if (bt.width != 6 || bt.bits != n_rrr_blocks * 6 || btnrp.width == 0 || btnrp.width > 64 || btnrp.bits != n_super * btnrp.width || invert.bits != n_super ||
        rank_samples.width == 0)
        return load_fail(FMSI_HIP_EIO, "rrr_vector<63> image has inconsistent geometry");
    unsigned long long dollar = 0, counts[4] = {0, 0, 0, 0};
    int k = 0;
    {
        FILE* f = std::fopen((base + ".misc").c_str(), "r");
        if (!f) return load_fail(FMSI_HIP_EIO, "cannot read " + base + ".misc");
        const int got = std::fscanf(f, "%llu %llu %llu %llu %llu %d", &dollar, &counts[0], &counts[1], &counts[2], &counts[3], &k);
        std::fclose(f);
        if (got != 6) return load_fail(FMSI_HIP_EIO, "malformed " + base + ".misc");
    }
    if (n == 0 || acgt.bits != n || ac.bits + gt.bits != n) return load_fail(FMSI_HIP_EIO, "index parts disagree on the number of rows");
    // (the reference reads these blindly, fms_index.h:513-524; a kLCP vector of another index, a `$` row or C[] values outside the rows would send
    // -S queries and LF steps past the arrays)
    if (with_klcp && klcp.bits != n) return load_fail(FMSI_HIP_EIO, "index parts disagree on the number of rows (" + base + ".klcp)");
    if (dollar >= n || counts[0] != 1 || counts[0] > counts[1] || counts[1] > counts[2] || counts[2] > counts[3] || counts[3] > n ||
        counts[2] != ac.bits)   // #A (with the `$` slot) + #C = |ac| (fms_index.h:427-451)
        return load_fail(FMSI_HIP_EIO, "index not correctly loaded: " + base + ".misc does not belong to these bit-vectors");
    if (n >= (1ull << 32) - kBlockRows || std::getenv("FMSI_HIP_FORCE_ROWS64")) {
        // 64-bit rows go through the host path (fmsi_wide.hip's superblock table is built there)
        fmsi_host_index* host = nullptr;
        int rc = fmsi_host_index_load(prefix_c, use_klcp, &host);
        if (rc) return rc;
        fmsi_index_view view;
        fmsi_host_index_view(host, &view);
        view.prefix_len = prefix_len;
        rc = fmsi_hip_index_create(&view, device, out);
        fmsi_host_index_free(host);
        return rc;
    }
    clock.mark("files mapped, headers");
    int rc = check_device(device);
    if (rc) return rc;
    FMSI_HIP_ON_DEVICE(device);
    clock.mark("device selected (HIP init)");
    // FMSI_HIP_LOAD_ASYNC_TABLE: the stream the table's signatures will be written on is created beside the upload (creating a stream
    // takes 20 ms, 150 ms the first time on a box: tools/micro/upload_bench.hip)
    hipStream_t bg_stream = nullptr;
    std::thread bg_stream_maker;
    if (async_table && prefix_len > 0)
        bg_stream_maker = std::thread([&bg_stream, device] {
            if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&bg_stream, hipStreamNonBlocking) != hipSuccess) {
                (void)hipGetLastError();
                bg_stream = nullptr;
            }
        });
    struct JoinMaker {
        std::thread& t;
        hipStream_t& s;
        bool handed_over = false;
        ~JoinMaker() {
            if (t.joinable()) t.join();
            if (s && !handed_over) (void)hipStreamDestroy(s);
        }
    } join_maker{bg_stream_maker, bg_stream};

    // ---- payloads to the device (two zero words of padding behind each) -----------------------------------------------------
    const uint64_t n_words = (n + 63) >> 6, n_blocks = ((n + kBlockRows - 1) >> kBlockRowsLog2) + 1;
    DevBuf d_acgt, d_ac, d_gt, d_bt, d_btnr, d_btnrp, d_invert, d_binom, d_space, d_mask, d_gtoff, d_pc, d_corrupt, d_klcp;
    std::vector<UploadJob> jobs;
    jobs.reserve(8);   // (the upload's pieces point into it)
    auto upload = [&](DevBuf& b, const Span& s, const Mapped& f) -> int {
        int r = b.alloc((s.words + 2) * 8);
        if (r) return r;
        if (s.words) jobs.push_back({f.fd, static_cast<uint64_t>(s.p - f.data), static_cast<char*>(b.p), s.words * 8});
        return 0;
    };
    if ((rc = upload(d_bt, bt, f_mask)) || (rc = upload(d_btnr, btnr, f_mask)) || (rc = upload(d_btnrp, btnrp, f_mask)) || (rc = upload(d_invert, invert, f_mask)) ||
        (rc = upload(d_acgt, acgt, f_acgt)) || (rc = upload(d_ac, ac, f_ac)) || (rc = upload(d_gt, gt, f_gt)))
        return rc;
    if (with_klcp && (rc = upload(d_klcp, klcp, f_klcp))) return rc;
    clock.mark("device buffers");
    const bool serial_upload = std::getenv("FMSI_HIP_LOAD_SERIAL_UPLOAD") != nullptr;   // (A/B runs: one copy per part straight from the mapping, as rounds 2-4 did)
    UploadTeam upload_team;   // (declared behind the buffers it fills: its threads are gone before any of them is freed)
    if (!serial_upload) upload_team.start(jobs, device);
    // While the payloads travel, everything of the load that does not need them: the process's first kernel launch (the runtime loads
    // its own fill kernels then: 50 ms), this library's code objects (loaded at the first use of a kernel of theirs: 10-20 ms each), the
    // small tables and the work buffers.
    {
        struct Pad {
            DevBuf* b;
            const Span* s;
        };
        for (const Pad& pd : {Pad{&d_bt, &bt}, Pad{&d_btnr, &btnr}, Pad{&d_btnrp, &btnrp}, Pad{&d_invert, &invert}, Pad{&d_acgt, &acgt}, Pad{&d_ac, &ac}, Pad{&d_gt, &gt}})
            FMSI_HIP_TRY(hipMemsetAsync(static_cast<char*>(pd.b->p) + pd.s->words * 8, 0, 16, 0));
        if (with_klcp) FMSI_HIP_TRY(hipMemsetAsync(static_cast<char*>(d_klcp.p) + klcp.words * 8, 0, 16, 0));
    }
    std::vector<uint64_t> h_binom(64 * 64);
    uint8_t h_space[64];
    rrr_tables(h_binom.data(), h_space);
    if ((rc = d_binom.alloc(64 * 64 * 8)) || (rc = d_space.alloc(64)) || (rc = d_mask.alloc((n_words + 64) * 8)) || (rc = d_gtoff.alloc((n_words + 1) * 4)) ||
        (rc = d_pc.alloc((n_words + 1) * 4)) || (rc = d_corrupt.alloc(4)))
        return rc;
    FMSI_HIP_TRY(hipMemcpyAsync(d_binom.p, h_binom.data(), 64 * 64 * 8, hipMemcpyHostToDevice, 0));
    FMSI_HIP_TRY(hipMemcpyAsync(d_space.p, h_space, 64, hipMemcpyHostToDevice, 0));
    FMSI_HIP_TRY(hipMemsetAsync(d_corrupt.p, 0, 4, 0));
    FMSI_HIP_TRY(hipMemsetAsync(d_pc.p, 0, (n_words + 1) * 4, 0));
    {
        hipFuncAttributes attr;
        (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&rrr_decode_kernel));   // (this file's code object)
        if (prefix_len > 0) warm_table_module();
        if (with_klcp) warm_stream_module();
        (void)hipGetLastError();
    }
    if (serial_upload) {
        for (const UploadJob& j : jobs) {
            const Mapped* f = j.fd == f_mask.fd ? &f_mask : j.fd == f_acgt.fd ? &f_acgt : j.fd == f_ac.fd ? &f_ac : j.fd == f_gt.fd ? &f_gt : &f_klcp;
            FMSI_HIP_TRY(hipMemcpyAsync(j.dst, f->data + j.file_off, j.bytes, hipMemcpyHostToDevice, 0));
        }
    } else if ((rc = upload_team.finish())) {
        return rc;
    }
    clock.mark("payloads read and copied (beside: first fill, code objects)");
    // ---- mask: RRR-63 -> plain words ---------------------------------------------------------------------------------------------
    {
        const uint64_t waves = (n_rrr_blocks + 63) / 64;
        hipLaunchKernelGGL(rrr_decode_kernel, dim3(grid_for(waves * 64)), dim3(256), 0, 0, d_bt.as<uint64_t>(), d_btnr.as<uint64_t>(), btnr.bits,
                           d_btnrp.as<uint64_t>(), static_cast<uint32_t>(btnrp.width), d_invert.as<uint64_t>(), d_binom.as<uint64_t>(), d_space.as<uint8_t>(), n,
                           n_rrr_blocks, n_super, d_mask.as<uint64_t>(), n_words, d_corrupt.as<uint32_t>());
        FMSI_HIP_TRY(hipGetLastError());
    }
    clock.mark("mask decoded (first kernel of this module)");
    // ---- where every word's G/T bits start in `gt` --------------------------------------------------------------------------------
    hipLaunchKernelGGL(word_popcount_kernel, dim3(grid_for(n_words)), dim3(256), 0, 0, d_acgt.as<uint64_t>(), n_words, n, d_pc.as<uint32_t>());
    FMSI_HIP_TRY(hipGetLastError());
    DevBuf scan_temp;
    {
        size_t tb = 0;
        FMSI_HIP_TRY(rocprim::exclusive_scan(nullptr, tb, d_pc.as<uint32_t>(), d_gtoff.as<uint32_t>(), 0u, static_cast<size_t>(n_words + 1), rocprim::plus<uint32_t>()));
        if ((rc = scan_temp.alloc(tb))) return rc;
        FMSI_HIP_TRY(rocprim::exclusive_scan(scan_temp.p, tb, d_pc.as<uint32_t>(), d_gtoff.as<uint32_t>(), 0u, static_cast<size_t>(n_words + 1), rocprim::plus<uint32_t>()));
    }
    {   // popcount(ac_gt) must be |gt| (and the rest |ac|) BEFORE layout_words_kernel indexes gt[] / ac[] by ranks in ac_gt: a mismatched
        // pair of files is an error of the load, not a memory fault on the device
        uint32_t gt_total = 0;
        FMSI_HIP_TRY(hipMemcpy(&gt_total, d_gtoff.as<uint32_t>() + n_words, 4, hipMemcpyDeviceToHost));
        if (gt_total != gt.bits || n - gt_total != ac.bits)
            return load_fail(FMSI_HIP_EIO, "index not correctly loaded: popcount(ac_gt) != |gt| (" + base + ".ac_gt / .ac / .gt are not one index's)");
    }
    clock.mark("word offsets (scan)");
    // ---- the blocks -----------------------------------------------------------------------------------------------------------------
    DevBuf blocks, cnts, cnts_scanned, mcounts, mrank;
    const size_t blocks_bytes = n_blocks * kBlockBytes;
    if ((rc = blocks.alloc(blocks_bytes)) || (rc = cnts.alloc((n_blocks + 1) * sizeof(U4))) || (rc = cnts_scanned.alloc((n_blocks + 1) * sizeof(U4))) ||
        (rc = mcounts.alloc((n_blocks + 1) * 4)) || (rc = mrank.alloc((n_blocks + 2) * 4)))
        return rc;
    FMSI_HIP_TRY(hipMemsetAsync(blocks.p, 0, blocks_bytes, 0));
    FMSI_HIP_TRY(hipMemsetAsync(cnts.p, 0, (n_blocks + 1) * sizeof(U4), 0));
    FMSI_HIP_TRY(hipMemsetAsync(mcounts.p, 0, (n_blocks + 1) * 4, 0));
    hipLaunchKernelGGL(layout_words_kernel, dim3(static_cast<uint32_t>((n_words + 3) / 4)), dim3(256), 0, 0, d_acgt.as<uint64_t>(), d_ac.as<uint64_t>(), d_gt.as<uint64_t>(),
                       d_mask.as<uint64_t>(), d_gtoff.as<uint32_t>(), n, n_words, blocks.as<uint32_t>(), cnts.as<U4>(), mcounts.as<uint32_t>());
    FMSI_HIP_TRY(hipGetLastError());
    {
        size_t t1 = 0, t2 = 0;
        FMSI_HIP_TRY(rocprim::exclusive_scan(nullptr, t1, cnts.as<U4>(), cnts_scanned.as<U4>(), U4{0, 0, 0, 0}, static_cast<size_t>(n_blocks + 1), U4Plus()));
        FMSI_HIP_TRY(rocprim::exclusive_scan(nullptr, t2, mcounts.as<uint32_t>(), mrank.as<uint32_t>(), 0u, static_cast<size_t>(n_blocks + 1), rocprim::plus<uint32_t>()));
        DevBuf t;
        if ((rc = t.alloc(std::max(t1, t2)))) return rc;
        FMSI_HIP_TRY(rocprim::exclusive_scan(t.p, t1, cnts.as<U4>(), cnts_scanned.as<U4>(), U4{0, 0, 0, 0}, static_cast<size_t>(n_blocks + 1), U4Plus()));
        FMSI_HIP_TRY(rocprim::exclusive_scan(t.p, t2, mcounts.as<uint32_t>(), mrank.as<uint32_t>(), 0u, static_cast<size_t>(n_blocks + 1), rocprim::plus<uint32_t>()));
        hipLaunchKernelGGL(scatter_counts_kernel, dim3(grid_for(n_blocks)), dim3(256), 0, 0, cnts_scanned.as<U4>(), blocks.as<uint32_t>(), n_blocks);
        FMSI_HIP_TRY(hipGetLastError());
        U4 totals;
        uint32_t corrupt = 0;
        FMSI_HIP_TRY(hipMemcpy(&totals, cnts_scanned.as<U4>() + n_blocks, sizeof(U4), hipMemcpyDeviceToHost));
        FMSI_HIP_TRY(hipMemcpy(&corrupt, d_corrupt.p, 4, hipMemcpyDeviceToHost));
        FMSI_HIP_TRY(hipDeviceSynchronize());
        if (corrupt) return load_fail(FMSI_HIP_EIO, "rrr_vector<63> image: block offsets run past the end of the offset vector");
        // the symbol counts of the rows are what C[] says (fms_index.h:451: {1, #A+1, #A+#C+1, #A+#C+#G+1} with the `$` slot stored as 'A')
        if (static_cast<uint64_t>(totals.a) != counts[1] || static_cast<uint64_t>(totals.a) + totals.c != counts[2] ||
            static_cast<uint64_t>(totals.a) + totals.c + totals.g != counts[3])
            return load_fail(FMSI_HIP_EIO, "index not correctly loaded: " + base + ".misc's counts are not those of the bit-vectors");
    }
    clock.mark("blocks laid out");
    auto* idx = new fmsi_hip_index();
    idx->device = device;
    idx->n = n;
    idx->k = k;
    idx->n_blocks = n_blocks;
    idx->d_blocks = blocks.release();
    idx->d_mrank = mrank.release();
    idx->device_bytes = blocks_bytes + (n_blocks + 2) * 4;
    if (with_klcp) {
        idx->d_klcp = d_klcp.release();
        idx->has_klcp = 1;
        idx->device_bytes += (klcp.words + 2) * 8;
    }
    if (dev_malloc(reinterpret_cast<void**>(&idx->d_counters), FMSI_HIP_N_COUNTERS * 8) != hipSuccess) {
        fmsi_hip_index_destroy(idx);
        return load_fail(FMSI_HIP_ENOMEM, "hipMalloc failed");
    }
    idx->dev.blocks = static_cast<const uint4*>(idx->d_blocks);
    idx->dev.mrank = static_cast<const uint32_t*>(idx->d_mrank);
    idx->dev.ptable = nullptr;
    idx->dev.ptable_len = 0;
    idx->dev.ptable_packed = 0;
    idx->dev.n = static_cast<uint32_t>(n);
    idx->dev.dollar = static_cast<uint32_t>(dollar);
    idx->dev.C0 = static_cast<uint32_t>(counts[0]);
    idx->dev.C1 = static_cast<uint32_t>(counts[1]);
    idx->dev.C2 = static_cast<uint32_t>(counts[2]);
    idx->dev.C3 = static_cast<uint32_t>(counts[3]);
    for (int c = 0; c < 4; ++c) idx->counts64[c] = counts[c];
    idx->dollar64 = dollar;
    // the temporaries go before the table (a freed block is waited for, devpool.hip: behind the signature pass that would be the whole pass)
    for (DevBuf* b : {&d_acgt, &d_ac, &d_gt, &d_bt, &d_btnr, &d_btnrp, &d_invert, &d_binom, &d_space, &d_mask, &d_gtoff, &d_pc, &d_corrupt, &scan_temp, &cnts,
                      &cnts_scanned, &mcounts})
        b->drop();
    if (async_table && prefix_len > 0) {
        // ... and so does whatever else would wait for the device on the handle's first query: the streaming layout
        if (with_klcp) (void)fmsi_hip_index_enable_stream_layout(idx);
        if (bg_stream_maker.joinable()) bg_stream_maker.join();
        idx->bg_stream = bg_stream;
        join_maker.handed_over = true;
        clock.mark("temporaries freed, streaming layout");
    }
    if (prefix_len > 0 && (rc = build_prefix_table(idx, prefix_len, async_table && !clock.on))) {
        fmsi_hip_index_destroy(idx);
        return rc;
    }
    clock.mark("prefix table");
    *out = idx;
    return 0;
}

extern "C" int fmsi_hip_index_wait_background(fmsi_hip_index* idx) {
    if (!idx) return FMSI_HIP_EINVAL;
    return ::fmsi::wait_background(idx);
}
