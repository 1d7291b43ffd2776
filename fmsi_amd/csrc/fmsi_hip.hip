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
#include <vector>

#include "../../include/fmsi_hip.h"
#include "common.h"
#include "device_fm.h"
#include "index_priv.h"
#include "trace.h"
#include "index_writer.h"
#include "layout.h"

namespace fmsi {

struct LaneCounters {
    uint32_t lf_steps = 0, blocks = 0, probes = 0, searches = 0, lookups = 0;
};

__device__ __forceinline__ void flush_counters(const LaneCounters& cn, unsigned long long* counters, uint32_t windows_done) {
    // wave reduction, then one atomic per wave and counter
    uint32_t v[6] = {cn.lf_steps, cn.blocks, cn.probes, cn.searches, cn.lookups, windows_done};
    const int slot[6] = {0, 1, 2, 3, 4, 6};
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        uint32_t x = v[q];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(&counters[slot[q]], static_cast<unsigned long long>(x));
    }
}

// get_range_with_pattern + infer_presence for one strand, plain SIMT form (lockstep kernel).
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

// second strand needed?  src/fms_index.h:283-299 (first strand = forward, i.e. no predictor swap)
__device__ __forceinline__ bool need_second(uint32_t strict, uint32_t orders, uint32_t mode_all, int res, long long ord) {
    if (strict) return true;
    if (orders) return ord < 0;
    if (mode_all) return res == -1;
    return res != 1;
}

__device__ __forceinline__ void write_result(uint64_t q, uint32_t strict, uint32_t orders, bool second, int res_f, int res_r,
                                             long long ord_f, long long ord_r, uint8_t* __restrict__ out,
                                             long long* __restrict__ out_orders) {
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

// ------------------------------------------------------------------------------------------------
// Kernel 1 ("lockstep"): one lane per k-mer; the wave walks the k LF steps in lock-step and leaves the
// loop when its last lane is done.  Kept as the simple baseline for A/B runs.
// ------------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(256) void query_lockstep_kernel(DevIndex ix, const uint64_t* __restrict__ kmers, uint64_t n_windows,
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
        const uint64_t f = kmers[q];
        long long ord_f = -1, ord_r = -1;
        const int res_f = search_strand(ix, f, k, mode_all, orders, ord_f, cn);
        int res_r = -1;
        const bool second = need_second(strict, orders, mode_all, res_f, ord_f);
        if (second) res_r = search_strand(ix, revcomp_2bit(f, k), k, mode_all, orders, ord_r, cn);
        write_result(q, strict, orders, second, res_f, res_r, ord_f, ord_r, out, out_orders);
    }
    if (COUNT) flush_counters(cn, counters, valid ? 1u : 0u);
}

// ------------------------------------------------------------------------------------------------
// Kernel 2 ("refill"): persistent lanes.  A wave owns a contiguous chunk of k-mers; every lane runs a
// small state machine in which EVERY iteration issues exactly one round of independent memory reads
// (prefix-table line, or the interleaved block(s) of its interval, whose mask plane also serves the final
// probe).  A lane that finishes takes its prefetched next k-mer immediately; new k-mers are handed out
// with wavefront ballot + mbcnt prefix ranks from a wave-uniform cursor -- no atomics, no LDS.  So no
// lane idles while a neighbour still walks, and the k-mer fetch latency hides under the block reads.
// ------------------------------------------------------------------------------------------------
enum : uint32_t { ST_IDLE = 0, ST_LOOKUP = 1, ST_LF = 2, ST_PROBE = 3 };

template <bool COUNT, bool PTABLE>
__global__ __launch_bounds__(256) void query_refill_kernel(DevIndex ix, const uint64_t* __restrict__ kmers, uint64_t n_windows,
                                                           int k, uint32_t flags, uint32_t chunk, uint8_t* __restrict__ out,
                                                           long long* __restrict__ out_orders,
                                                           unsigned long long* __restrict__ counters) {
    // Per wave in LDS: a ring of 128 k-mers and the chunk's result bytes.  Every memory request -- cached or not --
    // counts against the chip's random-request ceiling, so k-mers arrive as one coalesced 512-byte read per 64 and
    // results leave as 16 bytes per lane at the end of the chunk instead of one request per k-mer each way.
    __shared__ uint64_t s_ring[4][128];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[4][4096];
    const uint32_t mode_all = flags & FMSI_HIP_MODE_ALL;
    const uint32_t strict = flags & FMSI_HIP_STRICT;
    const uint32_t orders = flags & FMSI_HIP_ORDERS;
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t* ring = s_ring[threadIdx.x >> 6];
    uint8_t* obuf = s_out[threadIdx.x >> 6];
    const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const uint64_t start = wave * chunk;                            // my chunk: [start, start + len), len <= 4096
    const uint32_t len = start < n_windows ? static_cast<uint32_t>(min(static_cast<uint64_t>(chunk), n_windows - start)) : 0u;
    uint32_t next = 0, loaded = 0;                                  // wave-uniform: handed out / landed in the ring
    uint64_t pend = 0;                                              // k-mer batch in flight from global memory
    bool pend_valid = false;
    const int T = PTABLE ? static_cast<int>(ix.ptable_len) : 0;

    uint64_t f = 0, cur = 0;
    uint32_t q = 0, i = 0, j = 0, st = ST_IDLE, phase = 0, done = 0;
    int t = 0, res_f = -1;
    long long ord_f = -1;
    LaneCounters cn;

    auto start_strand = [&](uint64_t kmer) {
        cur = kmer;
        if (COUNT) cn.searches++;
        if (PTABLE) {
            st = ST_LOOKUP;
        } else {
            i = 0;
            j = ix.n;
            t = k - 1;
            st = ST_LF;
        }
    };
    // result of the strand just searched: src/fms_index.h:263-331 with the forward strand first
    auto finalize = [&](int res, long long ord) {
        bool second = false;
        if (phase == 0) {
            if (need_second(strict, orders, mode_all, res, ord)) {
                res_f = res;
                ord_f = ord;
                phase = 1;
                start_strand(revcomp_2bit(f, k));
                return;
            }
        } else {
            second = true;
        }
        const int rf = second ? res_f : res, rr = second ? res : -1;
        if (orders) write_result(start + q, strict, orders, second, rf, rr, second ? ord_f : ord, second ? ord : -1, out, out_orders);
        else if (strict) obuf[q] = static_cast<uint8_t>((rf + 1) | ((rr + 1) << 2));
        else obuf[q] = static_cast<uint8_t>((second ? rr : rf) == 1);
        st = ST_IDLE;
        done++;
    };

    for (;;) {
        // (0) the batch of k-mers requested last iteration has arrived: land it in the ring
        if (pend_valid) {
            ring[(loaded + lane) & 127u] = pend;
            loaded = min(len, loaded + 64u);
            pend_valid = false;
        }
        // (1) idle lanes draw the next k-mers of the chunk: ballot + mbcnt prefix rank over a wave-uniform cursor
        {
            const bool want = st == ST_IDLE;
            const uint64_t m = __ballot(want);
            if (m) {
                const uint32_t rank = lanes_below(m);
                const uint32_t avail = loaded - next;
                if (want && rank < avail) {
                    q = next + rank;
                    f = ring[q & 127u];
                    phase = 0;
                    start_strand(f);
                }
                next += min(static_cast<uint32_t>(__popcll(m)), avail);
            }
        }
        // (2) keep the ring ahead of the cursor: consumed next iteration, so the latency hides under (3)
        if (!pend_valid && loaded < len && loaded - next <= 64u) {
            const uint32_t idx = loaded + lane;
            pend = idx < len ? kmers[start + idx] : 0ull;
            pend_valid = true;
        }
        if (!__any(st != ST_IDLE)) {
            if (next >= len && !pend_valid) break;
            continue;
        }
        // (3) one round of reads per active lane
        if (st != ST_IDLE) {
            const uint4* pi;
            const uint4* pj;
            uint32_t bi = 0, bj = 0, tidx = 0;
            if (PTABLE && st == ST_LOOKUP) {
                tidx = static_cast<uint32_t>(cur >> (2 * (k - T))) & static_cast<uint32_t>((1ull << (2 * T)) - 1ull);
                pi = reinterpret_cast<const uint4*>(ix.ptable + (tidx & ~7u));   // the 64-B line holding the entry
                pj = pi;
            } else {
                bi = i >> kBlockRowsLog2;
                bj = (j - 1u) >> kBlockRowsLog2;
                pi = ix.blocks + (static_cast<size_t>(bi) << 2);
                pj = ix.blocks + (static_cast<size_t>(bj) << 2);
            }
            // All reads of the round are issued before anything waits: the (predicated) second block first, then
            // the first block; only afterwards are lanes with a single block given copies.  Writing `b = a` before
            // the predicated loads makes hipcc wait for `a` in between, which serialises the two reads.
            const bool two = pj != pi;
            uint4 b0 = make_uint4(0, 0, 0, 0), b1 = b0, b2 = b0;
            if (two) {
                b0 = pj[0];
                b1 = pj[1];
                b2 = pj[2];
            }
            const uint4 a0 = pi[0], a1 = pi[1], a2 = pi[2], a3 = pi[3];
            if (!two) {
                b0 = a0;
                b1 = a1;
                b2 = a2;
            }
            if (COUNT) cn.blocks += 1u + (two ? 1u : 0u);
            if (st == ST_LF) {
                const uint32_t c = static_cast<uint32_t>(cur >> (2 * t)) & 3u;
                uint32_t ri = rank_in_block(a0, a1, a2, c, i & (kBlockRows - 1u));
                uint32_t rj = rank_in_block(b0, b1, b2, c, j - (bj << kBlockRowsLog2));
                if (c == 0u) {
                    ri -= (i > ix.dollar);
                    rj -= (j > ix.dollar);
                }
                const uint32_t base = sel4(c, ix.C0, ix.C1, ix.C2, ix.C3);
                i = base + ri;
                j = base + rj;
                --t;
                if (COUNT) cn.lf_steps++;
                if (i == j) finalize(-1, -1);
                else if (t < 0) st = ST_PROBE;
            } else if (st == ST_PROBE) {
                if (COUNT) cn.probes++;
                const uint32_t pi_off = i & (kBlockRows - 1u);
                int res;
                long long ord = -1;
                if (mode_all && !orders) {
                    const uint32_t w = sel4(pi_off >> 5, opaque(a3.x), opaque(a3.y), opaque(a3.z), opaque(a3.w));
                    res = static_cast<int>((w >> (pi_off & 31u)) & 1u);
                } else {
                    uint32_t ones;
                    const uint32_t pj_off = j - (bj << kBlockRowsLog2);
                    if (bj == bi) {
                        ones = popc_prefix128(a3, pj_off) - popc_prefix128(a3, pi_off);
                    } else {   // interval spans blocks (long repeats): rank difference through the side array
                        ones = mask_rank(ix, bj, pj_off) - (ix.mrank[bi] + popc_prefix128(a3, pi_off));
                    }
                    res = ones > 0u;
                    if (orders && res) ord = static_cast<long long>(ix.mrank[bi] + popc_prefix128(a3, pi_off));
                }
                finalize(res, ord);
            } else {   // ST_LOOKUP
                if (COUNT) cn.lookups++;
                const uint32_t e = (tidx & 7u) >> 1;
                const uint32_t x0 = sel4(e, opaque(a0.x), opaque(a1.x), opaque(a2.x), opaque(a3.x));
                const uint32_t x1 = sel4(e, opaque(a0.y), opaque(a1.y), opaque(a2.y), opaque(a3.y));
                const uint32_t x2 = sel4(e, opaque(a0.z), opaque(a1.z), opaque(a2.z), opaque(a3.z));
                const uint32_t x3 = sel4(e, opaque(a0.w), opaque(a1.w), opaque(a2.w), opaque(a3.w));
                t = k - 1 - T;
                decode_table_entry((tidx & 1u) ? x2 : x0, (tidx & 1u) ? x3 : x1, ix.ptable_packed, t >= 3,
                                   t >= 3 ? static_cast<uint32_t>(cur >> (2 * (t - 3))) & 0xFFu : 0u, i, j);
                if (i == j) finalize(-1, -1);
                else st = t < 0 ? ST_PROBE : ST_LF;
            }
        }
    }
    if (!orders && len) {
        uint8_t* dst = out + start;
        const bool aligned = (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;
        for (uint32_t off = lane * 16u; off < len; off += 64u * 16u) {
            if (aligned && off + 16u <= len) {
                *reinterpret_cast<uint4*>(dst + off) = *reinterpret_cast<const uint4*>(obuf + off);
            } else {
                for (uint32_t b = off; b < min(len, off + 16u); ++b) dst[b] = obuf[b];
            }
        }
    }
    if (COUNT) flush_counters(cn, counters, done);
}

// windows of a 2-bit packed sequence -> packed k-mers (streaming, 8 B out per window)
__global__ __launch_bounds__(256) void extract_kmers_kernel(const uint64_t* __restrict__ seq, const uint64_t* __restrict__ window_start,
                                                            uint64_t n_windows, int k, uint64_t* __restrict__ kmers) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t q = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; q < n_windows; q += stride)
        kmers[q] = window_kmer(seq, window_start ? window_start[q] : q, k);
}

// runs of consecutive windows -> per-window packed k-mers (KMERS) or window starts: window q belongs to the run r with
// run_off[r] <= q < run_off[r + 1] (binary search; neighbouring lanes walk the same path, so the probes are cache hits)
template <bool KMERS>
__global__ __launch_bounds__(256) void expand_runs_kernel(const uint64_t* __restrict__ seq, const uint64_t* __restrict__ run_start,
                                                          const uint64_t* __restrict__ run_off, uint64_t n_runs, uint64_t n_windows,
                                                          int k, uint64_t* __restrict__ dst) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t q = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; q < n_windows; q += stride) {
        uint64_t lo = 0, hi = n_runs;   // invariant: run_off[lo] <= q < run_off[hi]
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (run_off[mid] <= q) lo = mid;
            else hi = mid;
        }
        const uint64_t start = run_start[lo] + (q - run_off[lo]);
        dst[q] = KMERS ? window_kmer(seq, start, k) : start;
    }
}

// FMSI_HIP_OUT_BITS for the kernels that write result bytes: n bytes -> one bit per window (bit w & 63 of u64 word w >> 6), STRICT:
// one nibble (nibble w & 15 of word w >> 4).  One thread per output word; a streaming pass of n bytes in, n / 8 (n / 2) out.
template <bool STRICT>
__global__ __launch_bounds__(256) void pack_results_kernel(const uint8_t* __restrict__ bytes, uint64_t n, unsigned long long* __restrict__ out) {
    constexpr uint32_t per_word = STRICT ? 16u : 64u;
    const uint64_t words = (n + per_word - 1) / per_word;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    const bool aligned = (reinterpret_cast<uintptr_t>(bytes) & 15u) == 0;
    for (uint64_t w = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; w < words; w += stride) {
        const uint64_t first = w * per_word;
        unsigned long long v = 0;
        if (aligned && first + per_word <= n) {
            const uint4* src = reinterpret_cast<const uint4*>(bytes + first);
            if (STRICT) {
                const uint4 x = src[0];
                auto nib = [](uint32_t d) -> unsigned long long { return (d & 0xFu) | ((d >> 4) & 0xF0u) | ((d >> 8) & 0xF00u) | ((d >> 12) & 0xF000u); };
                v = nib(x.x) | (nib(x.y) << 16) | (nib(x.z) << 32) | (nib(x.w) << 48);
            } else {
#pragma unroll
                for (uint32_t q = 0; q < 4u; ++q) {
                    const uint4 x = src[q];
                    auto bits4 = [](uint32_t d) -> unsigned long long { return (((d & 0x01010101u) * 0x10204080u) >> 28) & 0xFu; };
                    v |= (bits4(x.x) | (bits4(x.y) << 4) | (bits4(x.z) << 8) | (bits4(x.w) << 12)) << (16u * q);
                }
            }
        } else {
            for (uint32_t b = 0; b < per_word && first + b < n; ++b)
                v |= static_cast<unsigned long long>(STRICT ? (bytes[first + b] & 0xFu) : (bytes[first + b] & 1u)) << ((STRICT ? 4u : 1u) * b);
        }
        out[w] = v;
    }
}

// one level of the prefix table: entry idx of level L = LF step with base (idx & 3) on entry idx >> 2 of level L-1.
// FINAL: the last level leaves as (i, length) -- the packed format's tag-0 form, decode_table_entry -- and the longest interval is
// taken on the way (one atomic per workgroup: a launch of 2^28 waves with an atomic each spent 48 ms on that one address)
template <bool FINAL>
__global__ __launch_bounds__(256) void ptable_level_kernel(DevIndex ix, const uint2* __restrict__ prev, uint2* __restrict__ cur, uint64_t count,
                                                           uint32_t* __restrict__ max_len) {
    __shared__ uint32_t s_max[4];
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;   // a launch holds < 2^32 threads; level 16 has 2^32 entries
    uint32_t m = 0;
    for (uint64_t idx = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < count; idx += stride) {
        uint2 e = prev ? prev[idx >> 2] : make_uint2(0u, ix.n);
        if (e.x != e.y) {
            uint32_t dummy = 0;
            lf_step(ix, static_cast<uint32_t>(idx & 3u), e.x, e.y, dummy);
        }
        if (FINAL) {
            e.y -= e.x;
            m = max(m, e.y);
        }
        cur[idx] = e;
    }
    if (FINAL) {
        for (int off = 32; off > 0; off >>= 1) m = max(m, static_cast<uint32_t>(__shfl_down(static_cast<int>(m), off, 64)));
        if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
            if (m) atomicMax(max_len, m);
        }
    }
}

// (i, length) -> (i, j): the unpacked format, for a table with an interval of 2^30 rows and more (a text that is mostly one letter)
__global__ __launch_bounds__(256) void ptable_unpack_kernel(uint2* __restrict__ tab, uint64_t count) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t idx = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < count; idx += stride) {
        uint2 e = tab[idx];
        e.y += e.x;
        tab[idx] = e;
    }
}

// (i, length) -> the entries of 1..5 rows get the signatures of their rows (decode_table_entry: four symbols each for 1..3 rows, three for
// 4 and 5), in place, each entry one 8-byte store: a kernel that reads the table meanwhile sees an entry in the one form or the other, and
// both say the same interval
__global__ __launch_bounds__(256) void ptable_sign_kernel(DevIndex ix, uint2* __restrict__ tab, uint64_t count) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t idx = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < count; idx += stride) {
        const uint2 e = tab[idx];
        const uint32_t len = e.y;
        if (len < 1u || len > 5u) continue;   // (a signed entry has bit 30 or 31 set)
        const bool wide = len <= 3u;          // four symbols per row, else three
        const int steps = wide ? 4 : 3;
        uint32_t y = wide ? ((1u << 30) | (len << 24)) : ((len - 2u) << 30);
        for (uint32_t m = 0; m < len; ++m) {
            uint32_t row = e.x + m, sig = 0;
            bool alive = true;
            for (int step = 0; step < steps; ++step) {
                uint32_t c = 0;
                if (alive && row != ix.dollar) {
                    const uint32_t* blk = reinterpret_cast<const uint32_t*>(ix.blocks) + (static_cast<size_t>(row >> kBlockRowsLog2) << 4);
                    const uint32_t o = row & (kBlockRows - 1u), w = o >> 5, sh = o & 31u;
                    c = (((blk[4 + w] >> sh) & 1u) << 1) | ((blk[8 + w] >> sh) & 1u);   // access(), src/fms_index.h:88-95
                    uint32_t a = row, b = row + 1u, dummy = 0;
                    lf_step(ix, c, a, b, dummy);
                    row = a;
                } else {
                    alive = false;   // the text starts here: no such k-mer; any signature will do
                }
                sig = (sig << 2) | c;
            }
            y |= sig << ((wide ? 8u : 6u) * m);
        }
        tab[idx] = make_uint2(e.x, y);
    }
}

void warm_table_module() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&ptable_level_kernel<true>));
    (void)hipGetLastError();
}

uint64_t sign_after_windows() {
    static const uint64_t v = [] {
        const char* e = std::getenv("FMSI_HIP_SIGN_AFTER_WINDOWS");
        return e ? static_cast<uint64_t>(std::max(0ll, atoll(e))) : 4000000000ull;
    }();
    return v;
}

void start_deferred_sign(fmsi_hip_index* idx) {
    if (!idx || !idx->sign_deferred) return;
    idx->sign_deferred = false;
    if (!idx->d_ptable || !idx->bg_stream) return;
    ::fmsi::DeviceGuard device_guard_(idx->device);
    const uint64_t final_count = 1ull << (2 * idx->prefix_len);
    const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>((final_count + 255) / 256, 1u << 20));
    hipLaunchKernelGGL(ptable_sign_kernel, dim3(blocks), dim3(256), 0, idx->bg_stream, idx->dev, static_cast<uint2*>(idx->d_ptable), final_count);
    if (hipGetLastError() == hipSuccess) idx->bg_pending = true;   // (a launch that fails leaves the intervals: every answer stays what it is)
}

int wait_background(fmsi_hip_index* idx) {
    if (!idx) return 0;
    start_deferred_sign(idx);   // (whoever waits wants the table in its final form)
    if (!idx->bg_pending) return 0;
    ::fmsi::DeviceGuard device_guard_(idx->device);
    const hipError_t e = hipStreamSynchronize(idx->bg_stream);
    idx->bg_pending = false;
    if (e != hipSuccess) {
        set_last_error(std::string("prefix table signatures: ") + hipGetErrorString(e));
        return FMSI_HIP_EHIP;
    }
    return 0;
}

int build_prefix_table(fmsi_hip_index* idx, int t, bool sign_in_background) {
    if (t <= 0) return 0;
    TraceRange trace_range("fmsi: build prefix table");
    if (t > 16) {
        set_last_error("prefix_len must be <= 16");
        return FMSI_HIP_EINVAL;
    }
    const uint64_t final_count = 1ull << (2 * t);
    uint2 *a = nullptr, *b = nullptr;
    uint32_t* d_max = nullptr;
    // +64 B so that reading the whole line of the last entry stays inside the allocation
    if (dev_malloc(reinterpret_cast<void**>(&a), final_count * sizeof(uint2) + 64) != hipSuccess) {
        (void)hipGetLastError();
        set_last_error("hipMalloc failed for the prefix table (" + std::to_string(final_count * sizeof(uint2) >> 20) + " MiB)");
        return FMSI_HIP_ENOMEM;
    }
    auto fail = [&](int code, const std::string& msg) {
        (void)dev_free(a);
        if (b) (void)dev_free(b);
        if (d_max) (void)dev_free(d_max);
        set_last_error(msg);
        return code;
    };
    if (t > 1 && dev_malloc(reinterpret_cast<void**>(&b), (final_count / 4) * sizeof(uint2) + 64) != hipSuccess) return fail(FMSI_HIP_ENOMEM, "hipMalloc failed for the prefix table");
    if (dev_malloc(reinterpret_cast<void**>(&d_max), 4) != hipSuccess || hipMemsetAsync(d_max, 0, 4, 0) != hipSuccess) return fail(FMSI_HIP_ENOMEM, "hipMalloc failed");
    // levels alternate between the two buffers so that the last level lands in `a`
    uint2* bufs[2] = {a, b};
    int cur = (t % 2 == 1) ? 0 : 1;
    const uint2* prev = nullptr;
    for (int level = 1; level <= t; ++level) {
        const uint64_t count = 1ull << (2 * level);
        if (level < t) {
            const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>((count + 255) / 256, 1u << 22));
            hipLaunchKernelGGL((ptable_level_kernel<false>), dim3(blocks), dim3(256), 0, 0, idx->dev, prev, bufs[cur], count, d_max);
        } else {
            const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>((count + 255) / 256, 1u << 14));
            hipLaunchKernelGGL((ptable_level_kernel<true>), dim3(blocks), dim3(256), 0, 0, idx->dev, prev, bufs[cur], count, d_max);
        }
        if (hipGetLastError() != hipSuccess) return fail(FMSI_HIP_EHIP, "prefix table build: kernel launch failed");
        prev = bufs[cur];
        cur ^= 1;
    }
    uint32_t h_max = 0;
    hipError_t e = hipMemcpy(&h_max, d_max, 4, hipMemcpyDeviceToHost);   // (waits for the levels)
    if (e != hipSuccess) return fail(FMSI_HIP_EHIP, std::string("prefix table build: ") + hipGetErrorString(e));
    if (b) (void)dev_free(b);
    b = nullptr;
    (void)dev_free(d_max);
    d_max = nullptr;
    // packed entries with next-base signatures, unless some interval is too long for the 30-bit length field
    const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>((final_count + 255) / 256, 1u << 20));
    idx->dev.ptable_packed = h_max < (1u << 30) ? 1u : 0u;
    if (!idx->dev.ptable_packed) {
        hipLaunchKernelGGL(ptable_unpack_kernel, dim3(blocks), dim3(256), 0, 0, a, final_count);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) return fail(FMSI_HIP_EHIP, std::string("prefix table: ") + hipGetErrorString(e));
    } else {
        if (sign_in_background && !idx->bg_stream && hipStreamCreateWithFlags(&idx->bg_stream, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            idx->bg_stream = nullptr;
        }
        const bool bg = sign_in_background && idx->bg_stream;
        // (the load's temporaries and this table's upper levels are parked blocks, and whoever is handed one waits for the device first: let that
        // wait be now, on a device with nothing to do, not the first batch's behind the pass)
        if (bg) dev_settle();
        static const bool sign_at_load = std::getenv("FMSI_HIP_SIGN_AT_LOAD") != nullptr;   // (A/B runs: round 5's start of the pass, right here)
        if (bg && !sign_at_load) {
            idx->sign_deferred = true;   // started by the first streaming call / after kSignAfterLookups k-mers: start_deferred_sign
            idx->lookups_unsigned = idx->streamed_unsigned = 0;
        } else {
            hipLaunchKernelGGL(ptable_sign_kernel, dim3(blocks), dim3(256), 0, bg ? idx->bg_stream : static_cast<hipStream_t>(0), idx->dev, a, final_count);
            e = hipGetLastError();
            if (e == hipSuccess && !bg) e = hipDeviceSynchronize();
            if (e != hipSuccess) return fail(FMSI_HIP_EHIP, std::string("prefix table signatures: ") + hipGetErrorString(e));
            idx->bg_pending = bg;
        }
    }
    idx->d_ptable = a;
    idx->dev.ptable = a;
    idx->dev.ptable_len = static_cast<uint32_t>(t);
    idx->prefix_len = t;
    idx->device_bytes += final_count * sizeof(uint2) + 64;
    return 0;
}

int launch_pack_results(const uint8_t* d_bytes, uint64_t n, bool strict, void* d_out, hipStream_t s) {
    if (reinterpret_cast<uintptr_t>(d_out) & 7u) {
        set_last_error("FMSI_HIP_OUT_BITS needs an 8-byte aligned output buffer");
        return FMSI_HIP_EINVAL;
    }
    const uint64_t words = (n + (strict ? 15 : 63)) / (strict ? 16 : 64);
    const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>((words + 255) / 256, 1u << 16));
    if (strict) hipLaunchKernelGGL((pack_results_kernel<true>), dim3(blocks), dim3(256), 0, s, d_bytes, n, static_cast<unsigned long long*>(d_out));
    else hipLaunchKernelGGL((pack_results_kernel<false>), dim3(blocks), dim3(256), 0, s, d_bytes, n, static_cast<unsigned long long*>(d_out));
    FMSI_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace fmsi

using namespace fmsi;

namespace {

template <bool COUNT, bool PTABLE>
void launch_refill(fmsi_hip_index* idx, const uint64_t* d_kmers, uint64_t n, int k, uint32_t flags, uint8_t* o8, long long* o64,
                   unsigned long long* d_counters, hipStream_t s) {
    // chunk per wave: enough waves to fill the chip several times over, chunks long enough to amortise the drain
    uint64_t chunk = n / (256ull * 32 * 8);
    chunk = std::min<uint64_t>(std::max<uint64_t>(chunk, 256), 4096);
    const uint64_t waves = (n + chunk - 1) / chunk;
    const uint32_t blocks = static_cast<uint32_t>((waves + 3) / 4);
    hipLaunchKernelGGL((query_refill_kernel<COUNT, PTABLE>), dim3(blocks), dim3(256), 0, s, idx->dev, d_kmers, n, k, flags,
                       static_cast<uint32_t>(chunk), o8, o64, d_counters);
}

int launch_query(fmsi_hip_index* idx, const uint64_t* d_kmers, uint64_t n_windows, int k, uint32_t flags, void* d_out,
                 unsigned long long* d_counters, hipStream_t s) {
    if (n_windows == 0) return 0;
    if (n_windows > (1ull << 38)) {
        set_last_error("too many windows for one launch");
        return FMSI_HIP_ERANGE;
    }
    if (idx->wide) {   // 64-bit rows: one kernel, whatever the selection flags say
        PackAfter pack;
        uint32_t kflags = flags;
        int rcw = pack.begin(idx, kflags, n_windows, d_out);
        if (rcw) return rcw;
        size_t slotw;
        if ((rcw = prof_begin(idx, s, &slotw))) return rcw;
        if ((rcw = launch_wide_query(idx, d_kmers, n_windows, k, kflags, pack.bytes, static_cast<long long*>(d_out), d_counters, s))) return rcw;
        if ((rcw = pack.end(n_windows, flags, d_out, s))) return rcw;
        return prof_end(idx, s, slotw);
    }
    uint8_t* o8 = static_cast<uint8_t*>(d_out);
    long long* o64 = static_cast<long long*>(d_out);
    uint32_t kernel = flags & FMSI_HIP_KERNEL_MASK;
    const bool use_table = idx->d_ptable && !(flags & FMSI_HIP_NO_PREFIX_TABLE) && k >= idx->prefix_len;
    int rc;
    // FMSI_HIP_OUT_BITS: the one-lane lookup kernel and its stepping kernel pack their results themselves; every other
    // kernel writes its bytes to a scratch buffer that one streaming pass packs (inside the profiled region)
    const bool dict = dict_applies(idx, k, flags);
    // a handle that keeps the super-k-mer dictionary ALONE answers isolated k-mers from it, too: the 8-byte k-mers are a packed sequence of
    // one-window reads at a stride of 32 bases (every window finds its own minimizer: a bucket read per k-mer, no LF step)
    const bool sk_iso = !dict && skdict_applies(idx, k, flags);
    if (kernel == FMSI_HIP_KERNEL_DICT && !dict) {   // (no dictionary for this call: AUTO)
        flags &= ~FMSI_HIP_KERNEL_MASK;
        kernel = FMSI_HIP_KERNEL_AUTO;
    }
    // AUTO on a handle with the pair layout and no usable table: a LARGE batch takes the batch-sorted walk (3.8 G k-mers/s at human scale
    // where the per-k-mer pair kernel does 2.1 G; its sort and its per-batch top levels do not pay below a few million k-mers); a handle
    // without the room for its scratch (about 66 bytes per k-mer of a sub-batch of 2^26: 4.5 GB, fmsi_sorted.hip) keeps the pair kernel
    const bool sorted_auto = !dict && !sk_iso && kernel == FMSI_HIP_KERNEL_AUTO && !use_table && idx->d_pblocks && n_windows >= (1ull << 22) &&
                             sorted_walk_applies(idx, k, (flags & ~FMSI_HIP_KERNEL_MASK) | FMSI_HIP_KERNEL_SORTED);
    const bool sorted = !dict && !sk_iso && (sorted_auto || sorted_walk_applies(idx, k, flags));
    if (kernel == FMSI_HIP_KERNEL_SORTED && !sorted) {   // (lookup, k > 32, no pair layout: AUTO)
        flags &= ~FMSI_HIP_KERNEL_MASK;
        kernel = FMSI_HIP_KERNEL_AUTO;
    }
    const bool phased = !dict && !sk_iso && (kernel == FMSI_HIP_KERNEL_PHASED || kernel == FMSI_HIP_KERNEL_AUTO) && idx->d_pblocks && use_table;
    const bool want_bits = (flags & FMSI_HIP_OUT_BITS) && !(flags & FMSI_HIP_ORDERS);
    const bool pack_after = want_bits && !dict && !sk_iso && !(phased && phased_packs_bits(idx, k, flags));
    if (!want_bits || pack_after) flags &= ~FMSI_HIP_OUT_BITS;
    if (want_bits && (reinterpret_cast<uintptr_t>(d_out) & 7u)) {
        set_last_error("FMSI_HIP_OUT_BITS needs an 8-byte aligned output buffer");
        return FMSI_HIP_EINVAL;
    }
    if (pack_after) {
        if ((rc = ensure_cap(&idx->d_bytes, &idx->d_bytes_cap, n_windows))) return rc;
        o8 = static_cast<uint8_t*>(idx->d_bytes);
    }
    size_t slot;
    if ((rc = prof_begin(idx, s, &slot))) return rc;
    if (dict) {
        if ((rc = launch_dict_query(idx, d_kmers, n_windows, k, flags, o8, d_counters, s))) return rc;
    } else if (sk_iso) {
        static const bool as_reads = std::getenv("FMSI_HIP_SK_ONE_WINDOW_READS") != nullptr;   // (A/B runs: round 3's form, sk_runs_kernel on one-window reads)
        const uint64_t one_window_reads[3] = {0, 32, 1};
        if (as_reads) rc = launch_skdict(idx, d_kmers, n_windows, n_windows, k, flags, o8, d_counters, s, one_window_reads);
        else rc = launch_skdict_kmers(idx, d_kmers, n_windows, flags, o8, d_counters, s);
        if (rc) return rc;
    } else if (sorted) {
        rc = launch_sorted_query(idx, d_kmers, n_windows, k, flags, o8, d_counters, s, sorted_auto);
        if (rc == FMSI_HIP_ENOMEM && sorted_auto) {   // (nothing was launched: the scratch is claimed first)
            (void)hipGetLastError();
            launch_pair_query(idx, d_kmers, n_windows, k, flags, false, o8, o64, d_counters, s);
            rc = 0;
        }
        if (rc) return rc;
    } else if (kernel == FMSI_HIP_KERNEL_LOCKSTEP) {
        const uint64_t blocks64 = (n_windows + 255) / 256;
        if (blocks64 > 0x7FFFFFFFull) {
            set_last_error("too many windows for one lockstep launch");
            return FMSI_HIP_ERANGE;
        }
        const dim3 grid(static_cast<uint32_t>(blocks64)), block(256);
        if (d_counters)
            hipLaunchKernelGGL((query_lockstep_kernel<true>), grid, block, 0, s, idx->dev, d_kmers, n_windows, k, flags, o8, o64, d_counters);
        else
            hipLaunchKernelGGL((query_lockstep_kernel<false>), grid, block, 0, s, idx->dev, d_kmers, n_windows, k, flags, o8, o64, d_counters);
    } else if (phased) {
        if ((rc = launch_phased_query(idx, d_kmers, n_windows, k, flags, o8, o64, d_counters, s))) return rc;
    } else if (kernel == FMSI_HIP_KERNEL_PAIR || kernel == FMSI_HIP_KERNEL_PHASED || (kernel == FMSI_HIP_KERNEL_AUTO && idx->d_pblocks)) {
        // (PHASED without a usable table is the one-kernel form)
        if (!idx->d_pblocks) {
            set_last_error("FMSI_HIP_KERNEL_PAIR needs fmsi_hip_index_enable_pair_layout() on this handle");
            return FMSI_HIP_EINVAL;
        }
        launch_pair_query(idx, d_kmers, n_windows, k, flags, use_table, o8, o64, d_counters, s);
    } else if (use_table) {
        if (d_counters) launch_refill<true, true>(idx, d_kmers, n_windows, k, flags, o8, o64, d_counters, s);
        else launch_refill<false, true>(idx, d_kmers, n_windows, k, flags, o8, o64, d_counters, s);
    } else {
        if (d_counters) launch_refill<true, false>(idx, d_kmers, n_windows, k, flags, o8, o64, d_counters, s);
        else launch_refill<false, false>(idx, d_kmers, n_windows, k, flags, o8, o64, d_counters, s);
    }
    FMSI_HIP_TRY(hipGetLastError());
    if (pack_after && (rc = launch_pack_results(o8, n_windows, flags & FMSI_HIP_STRICT, d_out, s))) return rc;
    return prof_end(idx, s, slot);
}

}  // namespace

namespace fmsi {
// bytes of the result buffer of a query call over n_windows windows
size_t out_buffer_bytes(uint64_t n_windows, uint32_t flags) {
    if (flags & FMSI_HIP_ORDERS) return n_windows * ((flags & FMSI_HIP_STRICT) ? 16 : 8);
    if (flags & FMSI_HIP_OUT_BITS) return ((flags & FMSI_HIP_STRICT) ? (n_windows + 15) / 16 : (n_windows + 63) / 64) * 8;
    return n_windows;
}
}  // namespace fmsi

namespace {

int check_query_args(const fmsi_hip_index* idx, const void* in, const void* out, int k, int k_max) {
    if (!idx || !in || !out) {
        set_last_error("null argument");
        return FMSI_HIP_EINVAL;
    }
    if (k < 1 || k > k_max) {
        set_last_error("k must be in [1, " + std::to_string(k_max) + "] for this entry point");
        return FMSI_HIP_ERANGE;
    }
    return 0;
}

// Common path of the query entry points.  input_kind 0: `in` = packed k-mers; 1: `in` = packed sequence (+ window starts)
int run_query(fmsi_hip_index* idx, int input_kind, const uint64_t* in, size_t in_bytes, const uint64_t* wstart, uint64_t n_windows,
              int k, uint32_t flags, void* out, uint64_t* counters, void* stream) {
    FMSI_HIP_ON_DEVICE(idx->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t out_bytes = out_buffer_bytes(n_windows, flags);
    const uint64_t* d_in = in;
    const uint64_t* d_ws = wstart;
    void* d_out = out;
    unsigned long long* d_cnt = reinterpret_cast<unsigned long long*>(counters);
    const bool dev_ptrs = flags & FMSI_HIP_DEVICE_PTRS;
    int rc;
    if (!dev_ptrs) {
        if ((rc = ensure_cap(&idx->d_in, &idx->d_in_cap, in_bytes))) return rc;
        if ((rc = ensure_cap(&idx->d_out, &idx->d_out_cap, out_bytes))) return rc;
        FMSI_HIP_TRY(hipMemcpyAsync(idx->d_in, in, in_bytes, hipMemcpyHostToDevice, s));
        d_in = static_cast<const uint64_t*>(idx->d_in);
        if (wstart) {
            if ((rc = ensure_cap(&idx->d_aux, &idx->d_aux_cap, n_windows * 8))) return rc;
            FMSI_HIP_TRY(hipMemcpyAsync(idx->d_aux, wstart, n_windows * 8, hipMemcpyHostToDevice, s));
            d_ws = static_cast<const uint64_t*>(idx->d_aux);
        }
        d_out = idx->d_out;
        if (counters) {
            FMSI_HIP_TRY(hipMemcpyAsync(idx->d_counters, counters, FMSI_HIP_N_COUNTERS * 8, hipMemcpyHostToDevice, s));
            d_cnt = idx->d_counters;
        }
    }
    const uint64_t* d_kmers = d_in;
    if (input_kind == 1 && k > 32) {
        // the k-mer does not fit a register pair: walk the window off the packed sequence (fmsi_longk.hip)
        if (n_windows == 0) return 0;
        PackAfter pack;
        if ((rc = pack.begin(idx, flags, n_windows, d_out))) return rc;
        size_t slot;
        if ((rc = prof_begin(idx, s, &slot))) return rc;
        launch_longk_query(idx, d_in, d_ws, n_windows, k, flags, pack.bytes, static_cast<long long*>(d_out), d_cnt, s);
        FMSI_HIP_TRY(hipGetLastError());
        if ((rc = pack.end(n_windows, flags, d_out, s))) return rc;
        if ((rc = prof_end(idx, s, slot))) return rc;
        if (!dev_ptrs) {
            if (out_bytes) FMSI_HIP_TRY(hipMemcpyAsync(out, idx->d_out, out_bytes, hipMemcpyDeviceToHost, s));
            if (counters) FMSI_HIP_TRY(hipMemcpyAsync(counters, idx->d_counters, FMSI_HIP_N_COUNTERS * 8, hipMemcpyDeviceToHost, s));
        }
        return 0;
    }
    if (input_kind == 1) {
        // windows -> packed k-mers in a library-owned buffer (8 B per window, streaming)
        if ((rc = ensure_cap(&idx->d_kmers, &idx->d_kmers_cap, n_windows * 8))) return rc;
        const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>((n_windows + 255) / 256, 256 * 16));
        hipLaunchKernelGGL(extract_kmers_kernel, dim3(blocks), dim3(256), 0, s, d_in, d_ws, n_windows, k,
                           static_cast<uint64_t*>(idx->d_kmers));
        FMSI_HIP_TRY(hipGetLastError());
        d_kmers = static_cast<const uint64_t*>(idx->d_kmers);
    }
    if ((rc = launch_query(idx, d_kmers, n_windows, k, flags, d_out, d_cnt, s))) return rc;
    if (!dev_ptrs) {
        if (out_bytes) FMSI_HIP_TRY(hipMemcpyAsync(out, idx->d_out, out_bytes, hipMemcpyDeviceToHost, s));
        if (counters) FMSI_HIP_TRY(hipMemcpyAsync(counters, idx->d_counters, FMSI_HIP_N_COUNTERS * 8, hipMemcpyDeviceToHost, s));
    }
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
    ::fmsi::ScratchScope scratch_scope_(device);   // (devpool.h: the outermost builder returns the block cache to the driver)
    if (!view || !out || !view->ac_gt || !view->ac || !view->gt || !view->mask) {
        set_last_error("null argument");
        return FMSI_HIP_EINVAL;
    }
    if (view->mask_bits == 0) {
        set_last_error("empty index");
        return FMSI_HIP_EINVAL;
    }
    // 2^32 rows and more (or FMSI_HIP_FORCE_ROWS64, for tests): the 64-bit row variant -- same blocks, plain search only (fmsi_wide.hip)
    const bool forced_wide = std::getenv("FMSI_HIP_FORCE_ROWS64") != nullptr;
    const bool wide = forced_wide || view->mask_bits >= (1ull << 32) - kBlockRows;
    int ndev = fmsi_hip_device_count();
    if (ndev <= 0 || device < 0 || device >= ndev) {
        set_last_error("no usable HIP device (libfmsi_hip has no CPU fallback)");
        return FMSI_HIP_ENODEV;
    }
    FMSI_HIP_ON_DEVICE(device);
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
    if (dev_malloc(&idx->d_blocks, blocks_bytes) != hipSuccess || dev_malloc(&idx->d_mrank, mrank_bytes) != hipSuccess ||
        dev_malloc(reinterpret_cast<void**>(&idx->d_counters), FMSI_HIP_N_COUNTERS * 8) != hipSuccess) {
        set_last_error("hipMalloc failed for the device layout");
        return fail(FMSI_HIP_ENOMEM);
    }
    if (hipMemcpy(idx->d_blocks, L.blocks.data(), blocks_bytes, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(idx->d_mrank, L.mrank.data(), mrank_bytes, hipMemcpyHostToDevice) != hipSuccess) {
        set_last_error("hipMemcpy of the device layout failed");
        return fail(FMSI_HIP_EHIP);
    }
    idx->device_bytes = blocks_bytes + mrank_bytes;
    for (int c = 0; c < 4; ++c) idx->counts64[c] = view->counts[c];
    idx->dollar64 = view->dollar_position;
    if (wide) {
        // superblocks of 2^31 rows; a forced small index gets superblocks of 8 blocks so that the table matters
        const uint32_t sb_log2 = forced_wide && view->mask_bits < (1ull << 32) ? 3u : 24u;
        std::vector<uint64_t> super;
        wide_super_table(L.blocks.data(), L.mrank.data(), L.n_blocks, sb_log2, super);
        if (dev_malloc(&idx->d_super, super.size() * 8) != hipSuccess ||
            hipMemcpy(idx->d_super, super.data(), super.size() * 8, hipMemcpyHostToDevice) != hipSuccess) {
            set_last_error("uploading the superblock table failed");
            return fail(FMSI_HIP_EHIP);
        }
        idx->device_bytes += super.size() * 8;
        idx->wide = true;
        idx->devw.blocks = static_cast<const uint4*>(idx->d_blocks);
        idx->devw.mrank = static_cast<const uint32_t*>(idx->d_mrank);
        idx->devw.super = static_cast<const uint64_t*>(idx->d_super);
        idx->devw.sb_log2 = sb_log2;
        idx->devw.n = idx->n;
        idx->devw.dollar = view->dollar_position;
        idx->devw.C0 = view->counts[0];
        idx->devw.C1 = view->counts[1];
        idx->devw.C2 = view->counts[2];
        idx->devw.C3 = view->counts[3];
        idx->devw.ptable = nullptr;
        idx->devw.ptable_len = 0;
        // (no kLCP kernel for such a handle: view->klcp is not used; the prefix table has 16-byte entries and is at most 14 deep)
        if (view->prefix_len > 0 && (rc = build_prefix_table_wide(idx, view->prefix_len))) return fail(rc);
        *out = idx;
        return 0;
    }
    idx->dev.blocks = static_cast<const uint4*>(idx->d_blocks);
    idx->dev.mrank = static_cast<const uint32_t*>(idx->d_mrank);
    idx->dev.ptable = nullptr;
    idx->dev.ptable_len = 0;
    idx->dev.ptable_packed = 0;
    idx->dev.n = static_cast<uint32_t>(idx->n);
    idx->dev.dollar = static_cast<uint32_t>(view->dollar_position);
    idx->dev.C0 = static_cast<uint32_t>(view->counts[0]);
    idx->dev.C1 = static_cast<uint32_t>(view->counts[1]);
    idx->dev.C2 = static_cast<uint32_t>(view->counts[2]);
    idx->dev.C3 = static_cast<uint32_t>(view->counts[3]);
    if (view->klcp && view->klcp_bits) {
        const size_t kb = ((view->klcp_bits + 63) / 64 + 1) * 8;
        if (dev_malloc(&idx->d_klcp, kb) != hipSuccess ||
            hipMemcpy(idx->d_klcp, view->klcp, ((view->klcp_bits + 63) / 64) * 8, hipMemcpyHostToDevice) != hipSuccess) {
            set_last_error("uploading the kLCP array failed");
            return fail(FMSI_HIP_EHIP);
        }
        idx->has_klcp = 1;
        idx->device_bytes += kb;
    }
    if (view->prefix_len > 0) {
        rc = build_prefix_table(idx, view->prefix_len);
        if (rc) return fail(rc);
    }
    *out = idx;
    return 0;
}

int fmsi_hip_trim_scratch(int device) {
    int ndev = fmsi_hip_device_count();
    if (device < 0 || device >= ndev) return FMSI_HIP_ENODEV;
    FMSI_HIP_ON_DEVICE(device);
    (void)dev_trim();
    return 0;
}

void fmsi_hip_keep_scratch(int on) { dev_keep_scratch(on != 0); }
uint64_t fmsi_hip_scratch_bytes(int device) {
    if (device < 0 || device >= fmsi_hip_device_count()) return 0;
    ::fmsi::DeviceGuard device_guard_(device);
    return dev_cached_bytes();
}

void fmsi_hip_index_destroy(fmsi_hip_index* idx) {
    if (!idx) return;
    ::fmsi::DeviceGuard device_guard_(idx->device);
    idx->sign_deferred = false;   // (a pass that never started is not started to be waited for)
    (void)wait_background(idx);
    if (idx->bg_stream) (void)hipStreamDestroy(idx->bg_stream);
    for (void* p : {idx->d_sk, idx->d_sk_aux[0], idx->d_sk_aux[1], idx->d_sk_aux[2], idx->d_sk_aux[3]})
        if (p) (void)dev_free(p);
    for (void* p : {idx->d_super, idx->d_blocks, idx->d_mrank, idx->d_ptable, idx->d_klcp, idx->d_sblocks, idx->d_text2, idx->d_stable, idx->d_vtable, idx->d_in, idx->d_aux, idx->d_out,
                    idx->d_kmers, idx->d_tasks, idx->d_bytes, idx->d_case, idx->d_sorted, idx->d_dict[0], idx->d_dict[1], idx->d_dict[2], idx->d_dict[3], idx->d_posans, idx->d_slot_pos, idx->d_pblocks, idx->d_sbt, idx->d_c2, static_cast<void*>(idx->d_counters)})
        if (p) (void)dev_free(p);
    for (auto& m : idx->meta_ring) {
        if (m.in_flight) (void)hipEventSynchronize(m.done);
        if (m.host) (void)hipHostFree(m.host);
        if (m.done) (void)hipEventDestroy(m.done);
    }
    for (auto& e : idx->events) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    delete idx;
    (void)dev_trim();   // (what this handle's construction left in the block cache goes back to the driver with it)
}

int fmsi_hip_index_release_scratch(fmsi_hip_index* idx) {
    if (!idx) return FMSI_HIP_EINVAL;
    FMSI_HIP_ON_DEVICE(idx->device);
    FMSI_HIP_TRY(hipDeviceSynchronize());   // (whatever still reads or writes them)
    struct Buf {
        void** p;
        size_t* cap;
    };
    for (const Buf& b : {Buf{&idx->d_in, &idx->d_in_cap}, Buf{&idx->d_aux, &idx->d_aux_cap}, Buf{&idx->d_out, &idx->d_out_cap}, Buf{&idx->d_kmers, &idx->d_kmers_cap},
                         Buf{&idx->d_tasks, &idx->d_tasks_cap}, Buf{&idx->d_bytes, &idx->d_bytes_cap}, Buf{&idx->d_case, &idx->d_case_cap},
                         Buf{&idx->d_sorted, &idx->d_sorted_cap}}) {
        if (*b.p) (void)dev_free(*b.p);
        *b.p = nullptr;
        *b.cap = 0;
    }
    (void)dev_trim();
    return 0;
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
    for (int c = 0; c < 4; ++c) info->counts[c] = idx->counts64[c];
    info->dollar_position = idx->dollar64;
    info->rows64 = idx->wide ? 1u : 0u;
    info->candidate_entry_bytes = idx->d_vtable ? (idx->vwide ? 32u : 16u) : 0u;
    info->dict_bytes = idx->dict_bytes;
    for (uint32_t lv = 0; lv < kDictMaxLevels; ++lv) info->dict_keys[lv] = idx->dict_keys[lv];
    info->dict_ambiguous = idx->dict_ambiguous;
    info->dict_levels = idx->dict.levels;
    info->dict_log2_buckets = idx->dict.levels ? 64u - idx->dict.lv[0].shift : 0u;
    info->dict_chain_bytes = idx->dict_chain_bytes;
    info->sk_bytes = idx->sk_bytes;
    info->sk_records = idx->sk_records;
    info->sk_heavy_minimizers = idx->sk_heavy;
    info->sk_overflowed_buckets = idx->sk_overflowed;
    info->sk_aux_keys = idx->sk_aux_keys;
    info->sk_log2_buckets = idx->sk.bits;
    info->sk_minimizer_len = idx->sk.bits ? static_cast<uint32_t>(idx->sk.m) : 0u;
    return 0;
}

int fmsi_hip_index_download_layout(const fmsi_hip_index* idx, uint32_t* blocks, uint32_t* mrank) {
    if (!idx || !blocks || !mrank) return FMSI_HIP_EINVAL;
    FMSI_HIP_ON_DEVICE(idx->device);
    FMSI_HIP_TRY(hipMemcpy(blocks, idx->d_blocks, idx->n_blocks * kBlockBytes, hipMemcpyDeviceToHost));
    FMSI_HIP_TRY(hipMemcpy(mrank, idx->d_mrank, (idx->n_blocks + 1) * 4, hipMemcpyDeviceToHost));
    return 0;
}

int fmsi_hip_index_write_files(const fmsi_hip_index* idx, const char* prefix) {
    ::fmsi::ScratchScope scratch_scope_(idx ? idx->device : -1);   // (devpool.h: the outermost builder returns the block cache to the driver)
    if (!idx || !prefix) return FMSI_HIP_EINVAL;
    FMSI_HIP_ON_DEVICE(idx->device);
    // (no clearing pass over buffers the copies fill: 1.4 GB of blocks at human scale)
    std::vector<uint32_t, NoInitAlloc<uint32_t>> blocks(idx->n_blocks * kBlockDwords);
    FMSI_HIP_TRY(hipMemcpy(blocks.data(), idx->d_blocks, blocks.size() * 4, hipMemcpyDeviceToHost));
    HostVectors hv;
    vectors_from_blocks(blocks.data(), idx->n, hv);
    blocks.clear();
    blocks.shrink_to_fit();
    std::vector<uint64_t, NoInitAlloc<uint64_t>> klcp;
    if (idx->d_klcp) {
        const uint64_t kw = (idx->n + 63) / 64;
        klcp.resize(kw + 2);
        klcp[kw] = klcp[kw + 1] = 0;
        FMSI_HIP_TRY(hipMemcpy(klcp.data(), idx->d_klcp, kw * 8, hipMemcpyDeviceToHost));
    }
    fmsi_index_view v;
    std::memset(&v, 0, sizeof v);
    v.ac_gt = hv.ac_gt.data(); v.ac_gt_bits = hv.n;
    v.ac = hv.ac.data(); v.ac_bits = hv.ac_bits;
    v.gt = hv.gt.data(); v.gt_bits = hv.gt_bits;
    v.mask = hv.mask.data(); v.mask_bits = hv.n;
    if (!klcp.empty()) {
        v.klcp = klcp.data();
        v.klcp_bits = idx->n;
    }
    for (int c = 0; c < 4; ++c) v.counts[c] = idx->counts64[c];
    v.dollar_position = idx->dollar64;
    v.k = idx->k;
    std::string err;
    int rc = write_index_files(prefix, v, err);
    if (rc) set_last_error(err);
    return rc;
}

int fmsi_hip_index_download_klcp(const fmsi_hip_index* idx, uint64_t* words) {
    if (!idx || !words || !idx->d_klcp) {
        set_last_error("the handle holds no kLCP array");
        return FMSI_HIP_EINVAL;
    }
    FMSI_HIP_ON_DEVICE(idx->device);
    FMSI_HIP_TRY(hipMemcpy(words, idx->d_klcp, ((idx->n + 63) / 64) * 8, hipMemcpyDeviceToHost));
    return 0;
}

int fmsi_hip_query_kmers(fmsi_hip_index* idx, const uint64_t* kmers, uint64_t n_kmers, int k, uint32_t flags, void* out,
                         uint64_t* counters, void* stream) {
    if (n_kmers == 0) return 0;
    int rc = check_query_args(idx, kmers, out, k, 32);
    if (rc) return rc;
    rc = run_query(idx, 0, kmers, n_kmers * 8, nullptr, n_kmers, k, flags, out, counters, stream);
    if (!rc) ::fmsi::note_lookups(idx, n_kmers);   // (a table still waiting for its signatures: index_priv.h)
    return rc;
}

static int query_runs_impl(fmsi_hip_index* idx, const uint64_t* packed_seq, uint64_t seq_bases, const uint64_t* run_start,
                           const uint32_t* run_windows, uint64_t n_runs, uint64_t n_windows, int k, uint32_t flags, void* out,
                           uint64_t* counters, void* stream);
int fmsi_hip_query_runs(fmsi_hip_index* idx, const uint64_t* packed_seq, uint64_t seq_bases, const uint64_t* run_start,
                        const uint32_t* run_windows, uint64_t n_runs, uint64_t n_windows, int k, uint32_t flags, void* out,
                        uint64_t* counters, void* stream) {
    const int rc = query_runs_impl(idx, packed_seq, seq_bases, run_start, run_windows, n_runs, n_windows, k, flags, out, counters, stream);
    if (!rc && idx) ::fmsi::note_lookups(idx, n_windows);
    return rc;
}
static int query_runs_impl(fmsi_hip_index* idx, const uint64_t* packed_seq, uint64_t seq_bases, const uint64_t* run_start,
                           const uint32_t* run_windows, uint64_t n_runs, uint64_t n_windows, int k, uint32_t flags, void* out,
                           uint64_t* counters, void* stream) {
    if (n_windows == 0 || n_runs == 0) return 0;
    int rc = check_query_args(idx, packed_seq, out, k, 1 << 20);
    if (rc) return rc;
    if (!run_start || !run_windows) {
        set_last_error("null argument");
        return FMSI_HIP_EINVAL;
    }
    FMSI_HIP_ON_DEVICE(idx->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the runs are checked first; UNIFORM runs -- equal lengths at one stride: reads of one length laid out back to back -- need no
    // table on the device (three numbers instead), which the dictionary's kernel takes as such
    uint64_t total = 0;
    const uint64_t wpr = run_windows[0], stride = n_runs > 1 ? run_start[1] - run_start[0] : wpr;
    bool uniform = n_runs == 1 || (run_start[1] > run_start[0] && stride >= wpr);
    for (uint64_t r = 0; r < n_runs; ++r) {
        if (run_windows[r] == 0 || run_start[r] + run_windows[r] + static_cast<uint64_t>(k) - 1 > seq_bases) {
            set_last_error("an empty run, or a run of windows that extends past the end of the sequence");
            return FMSI_HIP_EINVAL;
        }
        if (uniform && (run_windows[r] != wpr || (r > 0 && run_start[r] - run_start[r - 1] != stride))) uniform = false;
        total += run_windows[r];
    }
    if (total != n_windows) {
        set_last_error("n_windows does not equal the sum of run_windows");
        return FMSI_HIP_EINVAL;
    }
    const uint64_t uniform_runs[3] = {run_start[0], stride, wpr};
    const bool fn_mode = (flags & kModeFn) != 0;   // a demasking function: fmsi_hip_query_runs_fn (fmsi_fn.hip)
    flags &= ~kModeFn;
    const bool via_sk = !fn_mode && skdict_applies(idx, k, flags) && n_windows <= (1ull << 38);   // (reads of any lengths)
    const bool table_free = uniform && (via_sk || (!fn_mode && dict_applies(idx, k, flags) && n_windows <= (1ull << 38)));
    // otherwise: [0, n_runs): run starts, [n_runs, 2 n_runs + 1): exclusive prefix sums of the run lengths -- in a page-locked slot
    // of the handle's ring, so that nothing has to wait for the copy (the call stays asynchronous on `stream`)
    void* meta_host = nullptr;
    unsigned meta_slot = 0;
    uint64_t* meta = nullptr;
    const size_t meta_words = 2 * n_runs + 1;
    if (!table_free) {
        if ((rc = meta_stage(idx, meta_words * 8, &meta_host, &meta_slot))) return rc;
        meta = static_cast<uint64_t*>(meta_host);
        uint64_t off = 0;
        for (uint64_t r = 0; r < n_runs; ++r) {
            meta[r] = run_start[r];
            meta[n_runs + r] = off;
            off += run_windows[r];
        }
        meta[2 * n_runs] = off;
    }
    const size_t out_bytes = out_buffer_bytes(n_windows, flags);
    const size_t seq_bytes = (seq_bases / 32 + 2) * 8;
    const bool dev_ptrs = flags & FMSI_HIP_DEVICE_PTRS;
    const uint64_t* d_seq = packed_seq;
    void* d_out = out;
    unsigned long long* d_cnt = reinterpret_cast<unsigned long long*>(counters);
    if (!dev_ptrs) {
        if ((rc = ensure_cap(&idx->d_in, &idx->d_in_cap, seq_bytes))) return rc;
        if ((rc = ensure_cap(&idx->d_out, &idx->d_out_cap, out_bytes))) return rc;
        FMSI_HIP_TRY(hipMemcpyAsync(idx->d_in, packed_seq, seq_bytes, hipMemcpyHostToDevice, s));
        d_seq = static_cast<const uint64_t*>(idx->d_in);
        d_out = idx->d_out;
        if (counters) {
            FMSI_HIP_TRY(hipMemcpyAsync(idx->d_counters, counters, FMSI_HIP_N_COUNTERS * 8, hipMemcpyHostToDevice, s));
            d_cnt = idx->d_counters;
        }
    }
    uint32_t kernel = flags & FMSI_HIP_KERNEL_MASK;
    if (via_sk) {   // reads of one length through the super-k-mer dictionary: a bucket read per minimizer
        size_t slot;
        if ((rc = prof_begin(idx, s, &slot))) return rc;
        if ((rc = launch_skdict(idx, d_seq, (seq_bases + 31) / 32, n_windows, k, flags, static_cast<uint8_t*>(d_out), d_cnt, s, table_free ? uniform_runs : nullptr, meta, n_runs))) return rc;
        if (!table_free && (rc = meta_commit(idx, meta_slot, s))) return rc;
        if ((rc = prof_end(idx, s, slot))) return rc;
        if (!dev_ptrs) {
            if (out_bytes) FMSI_HIP_TRY(hipMemcpyAsync(out, idx->d_out, out_bytes, hipMemcpyDeviceToHost, s));
            if (counters) FMSI_HIP_TRY(hipMemcpyAsync(counters, idx->d_counters, FMSI_HIP_N_COUNTERS * 8, hipMemcpyDeviceToHost, s));
        }
        return 0;
    }
    if (!fn_mode && dict_applies(idx, k, flags) && n_windows <= (1ull << 38)) {
        // the dictionary's lookup kernel cuts the windows itself, too
        size_t slot;
        if ((rc = prof_begin(idx, s, &slot))) return rc;
        if (table_free && dict_chain_applies(idx, k, flags)) {   // reads of one length and the chain data: 32 windows per lane along the text
            if ((rc = launch_dict_chain(idx, d_seq, n_windows, k, flags, static_cast<uint8_t*>(d_out), d_cnt, s, uniform_runs))) return rc;
        } else if ((rc = launch_dict_query(idx, d_seq, n_windows, k, flags, static_cast<uint8_t*>(d_out), d_cnt, s, meta, n_runs,
                                           table_free ? uniform_runs : nullptr))) {
            return rc;
        }
        if (!table_free && (rc = meta_commit(idx, meta_slot, s))) return rc;
        if ((rc = prof_end(idx, s, slot))) return rc;
        if (!dev_ptrs) {
            if (out_bytes) FMSI_HIP_TRY(hipMemcpyAsync(out, idx->d_out, out_bytes, hipMemcpyDeviceToHost, s));
            if (counters) FMSI_HIP_TRY(hipMemcpyAsync(counters, idx->d_counters, FMSI_HIP_N_COUNTERS * 8, hipMemcpyDeviceToHost, s));
        }
        return 0;
    }
    if (kernel == FMSI_HIP_KERNEL_DICT) {
        flags &= ~FMSI_HIP_KERNEL_MASK;
        kernel = FMSI_HIP_KERNEL_AUTO;
    }
    if (!fn_mode && !idx->wide && k <= 32 && (kernel == FMSI_HIP_KERNEL_PHASED || kernel == FMSI_HIP_KERNEL_AUTO) && idx->d_pblocks && idx->d_ptable &&
        !(flags & FMSI_HIP_NO_PREFIX_TABLE) && k >= idx->prefix_len && n_windows <= (1ull << 38)) {
        // the phased kernels cut the windows themselves: no pass that writes 8 bytes per window and reads them back
        PackAfter pack;   // (the one-lane lookup kernel packs FMSI_HIP_OUT_BITS output itself)
        uint32_t kflags = flags;
        if (!phased_packs_bits(idx, k, flags) && (rc = pack.begin(idx, kflags, n_windows, d_out))) return rc;
        size_t slot;
        if ((rc = prof_begin(idx, s, &slot))) return rc;
        if ((rc = launch_phased_query(idx, d_seq, n_windows, k, kflags, pack.on ? pack.bytes : static_cast<uint8_t*>(d_out),
                                      static_cast<long long*>(d_out), d_cnt, s, meta, n_runs)))
            return rc;
        if ((rc = meta_commit(idx, meta_slot, s))) return rc;
        if ((rc = pack.end(n_windows, flags, d_out, s))) return rc;
        if ((rc = prof_end(idx, s, slot))) return rc;
        if (!dev_ptrs) {
            if (out_bytes) FMSI_HIP_TRY(hipMemcpyAsync(out, idx->d_out, out_bytes, hipMemcpyDeviceToHost, s));
            if (counters) FMSI_HIP_TRY(hipMemcpyAsync(counters, idx->d_counters, FMSI_HIP_N_COUNTERS * 8, hipMemcpyDeviceToHost, s));
        }
        return 0;
    }
    if ((rc = ensure_cap(&idx->d_tasks, &idx->d_tasks_cap, meta_words * 8))) return rc;
    FMSI_HIP_TRY(hipMemcpyAsync(idx->d_tasks, meta, meta_words * 8, hipMemcpyHostToDevice, s));
    if ((rc = meta_commit(idx, meta_slot, s))) return rc;
    const uint64_t* d_rs = static_cast<const uint64_t*>(idx->d_tasks);
    const uint64_t* d_ro = d_rs + n_runs;
    const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>((n_windows + 255) / 256, 256 * 16));
    if (fn_mode) {
        // f(ones, total) per window, both strands: the walk off the packed sequence, any k (fmsi_fn.hip)
        if ((rc = ensure_cap(&idx->d_aux, &idx->d_aux_cap, n_windows * 8))) return rc;
        hipLaunchKernelGGL((expand_runs_kernel<false>), dim3(blocks), dim3(256), 0, s, d_seq, d_rs, d_ro, n_runs, n_windows, k,
                           static_cast<uint64_t*>(idx->d_aux));
        FMSI_HIP_TRY(hipGetLastError());
        PackAfter pack;
        uint32_t kflags = flags;
        if ((rc = pack.begin(idx, kflags, n_windows, d_out))) return rc;
        size_t slot;
        if ((rc = prof_begin(idx, s, &slot))) return rc;
        launch_fn_query(idx, d_seq, static_cast<const uint64_t*>(idx->d_aux), n_windows, k, !(flags & FMSI_HIP_NO_PREFIX_TABLE), pack.bytes, d_cnt, s);
        FMSI_HIP_TRY(hipGetLastError());
        if ((rc = pack.end(n_windows, flags, d_out, s))) return rc;
        if ((rc = prof_end(idx, s, slot))) return rc;
    } else if (k > 32) {
        if ((rc = ensure_cap(&idx->d_aux, &idx->d_aux_cap, n_windows * 8))) return rc;
        hipLaunchKernelGGL((expand_runs_kernel<false>), dim3(blocks), dim3(256), 0, s, d_seq, d_rs, d_ro, n_runs, n_windows, k,
                           static_cast<uint64_t*>(idx->d_aux));
        FMSI_HIP_TRY(hipGetLastError());
        PackAfter pack;
        uint32_t kflags = flags;
        if ((rc = pack.begin(idx, kflags, n_windows, d_out))) return rc;
        size_t slot;
        if ((rc = prof_begin(idx, s, &slot))) return rc;
        launch_longk_query(idx, d_seq, static_cast<const uint64_t*>(idx->d_aux), n_windows, k, kflags, pack.bytes,
                           static_cast<long long*>(d_out), d_cnt, s);
        FMSI_HIP_TRY(hipGetLastError());
        if ((rc = pack.end(n_windows, flags, d_out, s))) return rc;
        if ((rc = prof_end(idx, s, slot))) return rc;
    } else {
        if ((rc = ensure_cap(&idx->d_kmers, &idx->d_kmers_cap, n_windows * 8))) return rc;
        hipLaunchKernelGGL((expand_runs_kernel<true>), dim3(blocks), dim3(256), 0, s, d_seq, d_rs, d_ro, n_runs, n_windows, k,
                           static_cast<uint64_t*>(idx->d_kmers));
        FMSI_HIP_TRY(hipGetLastError());
        if ((rc = launch_query(idx, static_cast<const uint64_t*>(idx->d_kmers), n_windows, k, flags, d_out, d_cnt, s))) return rc;
    }
    if (!dev_ptrs) {
        if (out_bytes) FMSI_HIP_TRY(hipMemcpyAsync(out, idx->d_out, out_bytes, hipMemcpyDeviceToHost, s));
        if (counters) FMSI_HIP_TRY(hipMemcpyAsync(counters, idx->d_counters, FMSI_HIP_N_COUNTERS * 8, hipMemcpyDeviceToHost, s));
    }
    return 0;
}

int fmsi_hip_query_windows(fmsi_hip_index* idx, const uint64_t* packed_seq, uint64_t seq_bases, const uint64_t* window_start,
                           uint64_t n_windows, int k, uint32_t flags, void* out, uint64_t* counters, void* stream) {
    if (n_windows == 0) return 0;
    int rc = check_query_args(idx, packed_seq, out, k, 1 << 20);
    if (rc) return rc;
    if (!window_start && n_windows + static_cast<uint64_t>(k) - 1 > seq_bases) {
        set_last_error("implicit windows run past the end of the sequence");
        return FMSI_HIP_EINVAL;
    }
    return run_query(idx, 1, packed_seq, (seq_bases / 32 + 2) * 8, window_start, n_windows, k, flags, out, counters, stream);
}

// fmsi_hip_query_stream: fmsi_stream.hip

int fmsi_hip_sync(fmsi_hip_index* idx, void* stream) {
    if (!idx) return FMSI_HIP_EINVAL;
    FMSI_HIP_ON_DEVICE(idx->device);
    FMSI_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return 0;
}

void* fmsi_hip_host_alloc(uint64_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        set_last_error("hipHostMalloc failed");
        return nullptr;
    }
    return p;
}

void fmsi_hip_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int fmsi_hip_profile_enable(fmsi_hip_index* idx, int on) {
    if (!idx) return FMSI_HIP_EINVAL;
    idx->profiling = on != 0;
    return 0;
}

int fmsi_hip_profile_read(fmsi_hip_index* idx, uint64_t* n_launches, double* total_ms, int reset) {
    if (!idx) return FMSI_HIP_EINVAL;
    FMSI_HIP_ON_DEVICE(idx->device);
    for (size_t e = 0; e < idx->events_used; ++e) {
        FMSI_HIP_TRY(hipEventSynchronize(idx->events[e].second));
        float ms = 0;
        FMSI_HIP_TRY(hipEventElapsedTime(&ms, idx->events[e].first, idx->events[e].second));
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
