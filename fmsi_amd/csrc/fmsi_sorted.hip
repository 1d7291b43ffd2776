// The table-less LF walk for a BATCH of isolated k-mers: strand searches sorted by the bases the backward search consumes first, walked
// in lock-step -- one wavefront lane per strand search -- with two-base steps over the pair layout (FMSI_HIP_KERNEL_SORTED).
//
// What it replaces in the reference: get_range_with_pattern() + infer_presence() per strand (src/fms_index.h:117-144) under the strand
// schedule of query_kmers_single<>() (src/fms_index.h:263-331); every interval is the one update_range() (src/fms_index.h:98-103) produces,
// so the answers are the reference's.  No table of any kind is read: what makes it fast is the ORDER of the searches.
//
// Why (measured, profiles/r04b_lf_sorted_probe.txt): the walk is bound by random line requests.  The first ~log4(n) steps of
// every search touch blocks that other searches of the batch touch as well; in the order the k-mers arrive those reads are spread over
// the launch and most of them miss the caches (16.8 lines per k-mer from the memory side where ~7 are private to a search).  Sorted by
// their last bases -- the ones the backward search consumes first -- neighbouring lanes walk the SAME intervals for their first
// log4(batch) steps: in a lock-step round their loads fall on one line and the texture unit merges them into one request, and what follows
// (the private part of a search) is the only traffic that reaches memory.  The plain lock-step kernel on a sorted batch already beats the
// persistent-lane pair kernel on the unsorted one (59 vs 65 ms per 2^27 31-mers at n = 2.8 x 10^9); this file adds what the order makes
// possible on top:
//   * both strands of every k-mer are search ITEMS of their own (key = the strand's bases left-aligned, value = 2 q + strand), sorted
//     together by rocPRIM's radix sort on the first 16 bases: the reverse-complement searches share their first steps like the forward
//     ones (a second strand that the schedule would not have asked for costs a few shared steps and dies; its ternary is simply unused);
//   * two-base steps over the 128-byte pair blocks, the lines fetched COOPERATIVELY -- eight lanes load 16 bytes each of one lane's line
//     in one instruction (lanes of a sorted wave that want the same line share the request), staged through LDS at a 144-byte pitch,
//     the owner reads its planes with conflict-free b128 reads and ranks both boundaries itself (no lane pairs, no shuffles);
//   * the walk is cut in two launches where the absent k-mers have died out (after ~log4(n) + 2 bases): the survivors of three waves
//     continue in ONE wave of the second launch, so that the long private tails of the present k-mers run with full lanes;
//   * the last pair step answers infer_presence() from the pre-mapped mask plane of the block(s) at hand (as query_pair_kernel does).
// Results: one ternary per item, scattered to tern[2 q + strand] (absent = the array's zero fill: three quarters of the items write
// nothing), folded per k-mer by a last pass into the call's output encoding.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cmath>
#include <string>

#include "../../include/fmsi_hip.h"
#include "common.h"
#include "device_fm.h"
#include "index_priv.h"
#include "pair_layout.h"
#include "trace.h"

namespace fmsi {
namespace {

constexpr uint32_t kLinePitch = 9;      // uint4 per staged line: 144 bytes
constexpr uint32_t kJoin = 3;           // waves of the first launch whose survivors one wave of the second launch takes
// rocPRIM's radix sort without its merge-sort path (the one it takes between 1 K and 1 M items): on ROCm 7.2 / gfx950 that path returned
// pairs in an order that changed from run to run for a (begin_bit, end_bit) = (32, 64) sort over double buffers -- 2 000 ... 40 000 items
// came back wrong, 600 and 2^28 right (tools/sorted_debug.py) -- so every size above one block goes through the onesweep passes
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;

#define SW_TRY(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) {                                                                            \
            ::fmsi::set_last_error(std::string(#expr) + ": " + hipGetErrorString(e_));                    \
            return FMSI_HIP_EHIP;                                                                          \
        }                                                                                                  \
    } while (0)

// ---- (1) the items: both strands of every k-mer, bases left-aligned (the base consumed first in bits 63..62) ------------------------
__global__ __launch_bounds__(256) void sw_items_kernel(const uint64_t* __restrict__ kmers, uint64_t n, int k, ulonglong2* __restrict__ keys,
                                                       uint2* __restrict__ vals) {
    const uint64_t q = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    const uint64_t f = __builtin_nontemporal_load(kmers + q) & kmask;
    const uint32_t sh = 64u - 2u * static_cast<uint32_t>(k);
    keys[q] = make_ulonglong2(f << sh, revcomp_2bit(f, k) << sh);
    vals[q] = make_uint2(static_cast<uint32_t>(2u * q), static_cast<uint32_t>(2u * q + 1u));
}

// ---- (3) the walk ----------------------------------------------------------------------------------------------------------------------
struct WalkArgs {
    const uint64_t* keys;     // first launch: the sorted items
    const uint32_t* vals;
    uint64_t n_items;
    // hand-over between the two launches: wave w of the first launch leaves its survivors in slots [64 w, 64 w + count[w])
    uint64_t* h_key;          // the bases still to go, left-aligned
    uint2* h_ij;
    uint32_t* h_val;
    uint32_t* h_count;
    uint64_t n_waves_a;       // waves of the first launch
    uint8_t* tern;            // result + 1 per item (0 = absent: pre-zeroed)
    uint32_t left0;           // bases to go when this launch starts
    uint32_t stop_left;       // hand over when this many bases are left (0: walk to the end)
    uint32_t second;          // 1: this is the second launch (items come from the hand-over slots)
    uint32_t mode_all;
};

__device__ __forceinline__ uint64_t below64s(int s) { return s >= 64 ? ~0ull : (s <= 0 ? 0ull : ((1ull << s) - 1ull)); }

// rows of a staged pair block that match pair p (single: symbol c1 = p >> 2 only): halves 0 (rows 0-63) and 1 (rows 64-127)
__device__ __forceinline__ void block_match(const uint4* L4, uint32_t p, bool single, uint64_t& m_lo, uint64_t& m_hi) {
    const uint32_t x1h = (p & 8u) ? 0u : 0xFFFFFFFFu, x1l = (p & 4u) ? 0u : 0xFFFFFFFFu;
    const uint32_t x2h = (p & 2u) ? 0u : 0xFFFFFFFFu, x2l = (p & 1u) ? 0u : 0xFFFFFFFFu;
    const uint32_t all = single ? 0xFFFFFFFFu : 0u;
    const uint4 a0 = L4[0], a1 = L4[1], b0 = L4[4], b1 = L4[5];
    const uint32_t m0 = (a0.x ^ x1h) & (a0.z ^ x1l) & ((a1.x ^ x2h) | all) & ((a1.z ^ x2l) | all);
    const uint32_t m1 = (a0.y ^ x1h) & (a0.w ^ x1l) & ((a1.y ^ x2h) | all) & ((a1.w ^ x2l) | all);
    const uint32_t m2 = (b0.x ^ x1h) & (b0.z ^ x1l) & ((b1.x ^ x2h) | all) & ((b1.z ^ x2l) | all);
    const uint32_t m3 = (b0.y ^ x1h) & (b0.w ^ x1l) & ((b1.y ^ x2h) | all) & ((b1.w ^ x2l) | all);
    m_lo = (static_cast<uint64_t>(m1) << 32) | m0;
    m_hi = (static_cast<uint64_t>(m3) << 32) | m2;
}
// cumulative count of pair p before the block (24 bits, relative to its superblock); single: of the four pairs with c1 = p >> 2
__device__ __forceinline__ uint32_t block_count(const uint32_t* L, uint32_t p, bool single) {
    const uint32_t h = p >> 3;                      // the half that keeps the counts of c1 in {2h, 2h+1}
    const uint32_t* H = L + 16u * h;
    if (single) {
        const uint32_t g = (p >> 2) & 1u;
        const uint32_t l0 = H[10u + 2u * g], l1 = H[11u + 2u * g], hb = H[14u + g];
        return (l0 & 0xFFFFu) + (l0 >> 16) + (l1 & 0xFFFFu) + (l1 >> 16) +
               (((hb & 0xFFu) + ((hb >> 8) & 0xFFu) + ((hb >> 16) & 0xFFu) + (hb >> 24)) << 16);
    }
    const uint32_t m = p & 7u;
    const uint32_t lw = H[10u + (m >> 1)], hw = H[14u + (m >> 2)];
    return ((lw >> (16u * (m & 1u))) & 0xFFFFu) | (((hw >> (8u * (m & 3u))) & 0xFFu) << 16);
}
__device__ __forceinline__ uint32_t rows_below(uint64_t m_lo, uint64_t m_hi, int o) {
    return static_cast<uint32_t>(__popcll(m_lo & below64s(o)) + __popcll(m_hi & below64s(o - 64)));
}
// a row that is stored as 'A' but stands for no symbol (suffix starts 0 and 1) is not a row of any interval
__device__ __forceinline__ void drop_row128(uint32_t row, uint32_t blk, uint64_t& m_lo, uint64_t& m_hi) {
    if ((row >> kBlockRowsLog2) == blk) {
        const uint32_t b = row & 127u;
        if (b < 64u) m_lo &= ~(1ull << b);
        else m_hi &= ~(1ull << (b - 64u));
    }
}

template <bool COUNT>
__global__ __launch_bounds__(256) void sorted_walk_kernel(PairIndex ix, WalkArgs a, unsigned long long* __restrict__ counters) {
    __shared__ uint4 s_line[4][64 * kLinePitch];
    __shared__ uint32_t s_c2[16];
    if (threadIdx.x < 16) s_c2[threadIdx.x] = ix.c2[threadIdx.x];
    __syncthreads();
    const uint32_t ln = threadIdx.x & 63u;
    uint4* lines = s_line[threadIdx.x >> 6];
    const uint4* L4 = lines + ln * kLinePitch;
    const uint32_t* L = reinterpret_cast<const uint32_t*>(L4);
    const uint4* pb4 = reinterpret_cast<const uint4*>(ix.pb);
    const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const uint32_t piece = ln & 7u, grp = ln >> 3;
    uint32_t n_steps = 0, n_blocks = 0, n_probes = 0, n_items = 0;

    // the second launch takes the survivors of kJoin waves of the first, 64 at a time
    uint32_t c0 = 0, c1n = 0, total = 0;
    if (a.second) {
        const uint64_t w0 = wave * kJoin;
        if (w0 >= a.n_waves_a) return;
        c0 = a.h_count[w0];
        c1n = w0 + 1 < a.n_waves_a ? a.h_count[w0 + 1] : 0u;
        const uint32_t c2n = w0 + 2 < a.n_waves_a ? a.h_count[w0 + 2] : 0u;
        total = c0 + c1n + c2n;
    } else {
        if (wave * 64u >= a.n_items) return;
        total = static_cast<uint32_t>(min(static_cast<uint64_t>(64), a.n_items - wave * 64u));
    }

    for (uint32_t at = 0; at < total; at += 64u) {
        // ---- this pass's items ----
        uint64_t key = 0;
        uint32_t i = 0, j = 0, val = 0;
        bool alive = false;
        const uint32_t me = at + ln;
        if (me < total) {
            alive = true;
            if (a.second) {
                const uint32_t r = me < c0 ? 0u : (me < c0 + c1n ? 1u : 2u);
                const uint64_t slot = (wave * kJoin + r) * 64u + (me - (r == 0u ? 0u : (r == 1u ? c0 : c0 + c1n)));
                key = a.h_key[slot];
                const uint2 ij = a.h_ij[slot];
                i = ij.x;
                j = ij.y;
                val = a.h_val[slot];
            } else {
                const uint64_t it = wave * 64u + me;
                key = a.keys[it];
                val = a.vals[it];
                i = 0;
                j = ix.n;
            }
            if (COUNT && !a.second) n_items++;
        }
        int res = -1;          // the item's ternary once it is known
        bool answered = false;
        uint32_t left = a.left0;
        while (left > a.stop_left) {
            if (!__any(alive)) break;
            const bool single = (left & 1u) != 0u;   // an odd number of bases to go: one single-base step first
            const uint32_t c1 = static_cast<uint32_t>(key >> 62);
            const uint32_t c2 = single ? 0u : (static_cast<uint32_t>(key >> 60) & 3u);
            const uint32_t p = c2 | (c1 << 2);
            const uint32_t bi = i >> kBlockRowsLog2, bj = (j - 1u) >> kBlockRowsLog2;
            const bool two = alive && bj != bi;
            const uint64_t line_mask = __ballot(alive);
            const uint64_t two_mask = __ballot(two);
            // superblock counts and C2 (small tables: cache resident)
            uint32_t sbi = 0, sbj = 0, cbase = 0;
            if (alive) {
                if (single) {
                    const uint4 si = *reinterpret_cast<const uint4*>(ix.sbt + ((static_cast<size_t>(i >> kSuperLog2)) << 4) + (p & 12u));
                    sbi = si.x + si.y + si.z + si.w;
                    sbj = sbi;
                    if (two) {
                        const uint4 sj = *reinterpret_cast<const uint4*>(ix.sbt + ((static_cast<size_t>((j - 1u) >> kSuperLog2)) << 4) + (p & 12u));
                        sbj = sj.x + sj.y + sj.z + sj.w;
                    }
                    cbase = sel4(c1, ix.C0, ix.C1, ix.C2_, ix.C3);
                } else {
                    sbi = ix.sbt[((static_cast<size_t>(i >> kSuperLog2)) << 4) + p];
                    sbj = two ? ix.sbt[((static_cast<size_t>((j - 1u) >> kSuperLog2)) << 4) + p] : sbi;
                    cbase = s_c2[p];
                }
            }
            // ---- the lines: eight lanes per line, eight lines per instruction ----
            uint4 va[8], vb[8];
#pragma unroll
            for (uint32_t g = 0; g < 8; ++g) {
                const uint32_t owner = 8u * g + grp;
                const uint32_t bo = static_cast<uint32_t>(__shfl(static_cast<int>(bi), static_cast<int>(owner), 64));
                va[g] = make_uint4(0, 0, 0, 0);
                if ((line_mask >> owner) & 1ull) va[g] = pb4[(static_cast<size_t>(bo) << 3) + piece];
            }
            if (two_mask) {
#pragma unroll
                for (uint32_t g = 0; g < 8; ++g) {
                    const uint32_t owner = 8u * g + grp;
                    const uint32_t bo = static_cast<uint32_t>(__shfl(static_cast<int>(bj), static_cast<int>(owner), 64));
                    vb[g] = make_uint4(0, 0, 0, 0);
                    if ((two_mask >> owner) & 1ull) vb[g] = pb4[(static_cast<size_t>(bo) << 3) + piece];
                }
            }
#pragma unroll
            for (uint32_t g = 0; g < 8; ++g) lines[(8u * g + grp) * kLinePitch + piece] = va[g];
            __builtin_amdgcn_wave_barrier();
            const int oi = static_cast<int>(i & (kBlockRows - 1u)), oj = static_cast<int>(j - (bj << kBlockRowsLog2));   // 0..127, 1..128
            uint64_t ma_lo = 0, ma_hi = 0, mb_lo = 0, mb_hi = 0;
            uint64_t ka_lo = 0, ka_hi = 0, kb_lo = 0, kb_hi = 0;   // pre-mapped mask planes (last step only)
            const bool last = !single && left == 2u;
            uint32_t ri = 0, rj = 0;
            if (alive) {
                block_match(L4, p, single, ma_lo, ma_hi);
                const uint32_t cnt = block_count(L, p, single);
                const uint32_t below_i = rows_below(ma_lo, ma_hi, oi);
                ri = cnt + below_i;
                if (!two) rj = cnt + rows_below(ma_lo, ma_hi, oj);
                if (last) {
                    ka_lo = *reinterpret_cast<const uint64_t*>(L + 8);
                    ka_hi = *reinterpret_cast<const uint64_t*>(L + 24);
                }
            }
            if (two_mask) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (uint32_t g = 0; g < 8; ++g) lines[(8u * g + grp) * kLinePitch + piece] = vb[g];
                __builtin_amdgcn_wave_barrier();
                if (two) {
                    block_match(L4, p, single, mb_lo, mb_hi);
                    rj = block_count(L, p, single) + rows_below(mb_lo, mb_hi, oj);
                    if (last) {
                        kb_lo = *reinterpret_cast<const uint64_t*>(L + 8);
                        kb_hi = *reinterpret_cast<const uint64_t*>(L + 24);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();   // (the next round's staging must not overtake this round's reads)
            if (alive) {
                if (COUNT) {
                    n_steps += single ? 1u : 2u;
                    n_blocks += two ? 2u : 1u;
                }
                // the last step also answers the search when the block(s) at hand cover the interval: the rows of [i, j) that match
                // the pair ARE the new interval, in order, and the mask plane holds mask[LF2(row)] (src/fms_index.h:126-144)
                int folded = -2;
                if (last && bj - bi <= 1u) {
                    uint64_t sa_lo = ma_lo & ~below64s(oi), sa_hi = ma_hi & ~below64s(oi - 64), sb_lo = 0, sb_hi = 0;
                    if (two) {
                        sb_lo = mb_lo & below64s(oj);
                        sb_hi = mb_hi & below64s(oj - 64);
                    } else {
                        sa_lo &= below64s(oj);
                        sa_hi &= below64s(oj - 64);
                    }
                    drop_row128(ix.r0, bi, sa_lo, sa_hi);
                    drop_row128(ix.r1, bi, sa_lo, sa_hi);
                    drop_row128(ix.r0, bj, sb_lo, sb_hi);
                    drop_row128(ix.r1, bj, sb_lo, sb_hi);
                    if (a.mode_all) {   // the mask bit of the interval's first row
                        if (sa_lo) folded = static_cast<int>((ka_lo >> (__ffsll(static_cast<unsigned long long>(sa_lo)) - 1)) & 1ull);
                        else if (sa_hi) folded = static_cast<int>((ka_hi >> (__ffsll(static_cast<unsigned long long>(sa_hi)) - 1)) & 1ull);
                        else if (sb_lo) folded = static_cast<int>((kb_lo >> (__ffsll(static_cast<unsigned long long>(sb_lo)) - 1)) & 1ull);
                        else if (sb_hi) folded = static_cast<int>((kb_hi >> (__ffsll(static_cast<unsigned long long>(sb_hi)) - 1)) & 1ull);
                    } else if (sa_lo | sa_hi | sb_lo | sb_hi) {   // any ON row in the interval
                        folded = ((sa_lo & ka_lo) | (sa_hi & ka_hi) | (sb_lo & kb_lo) | (sb_hi & kb_hi)) ? 1 : 0;
                    }
                }
                ri += sbi;
                rj += sbj;
                if (single) {
                    if (c1 == 0u) {   // the '$' slot is stored as c1 = 'A' (src/fms_index.h:81)
                        ri -= (i > ix.r0);
                        rj -= (j > ix.r0);
                    }
                    key <<= 2;
                } else {
                    if (p == 0u) {    // '$' slot stored as (A, A)
                        ri -= (i > ix.r0);
                        rj -= (j > ix.r0);
                    }
                    if (p == ix.p1) {   // the row of suffix 1 has no c2: stored as (A, T[0])
                        ri -= (i > ix.r1);
                        rj -= (j > ix.r1);
                    }
                    key <<= 4;
                }
                i = cbase + ri;
                j = cbase + rj;
                if (i == j) {
                    alive = false;          // absent: the array's zero fill says so
                    answered = true;
                } else if (folded >= 0) {
                    res = folded;
                    alive = false;
                    answered = true;
                }
            }
            left -= single ? 1u : 2u;
        }
        // ---- what is left of this pass ----
        if (a.stop_left) {
            // hand the survivors over: dense in this wave's slots (none when the wave's last search died before the cut)
            const uint64_t m = __ballot(alive);
            if (alive) {
                const uint64_t slot = wave * 64u + lanes_below(m);
                a.h_key[slot] = key;
                a.h_ij[slot] = make_uint2(i, j);
                a.h_val[slot] = val;
            }
            if (ln == 0u) a.h_count[wave] = static_cast<uint32_t>(__popcll(m));
        } else {
            if (alive) {
                // a non-empty interval that the last step could not answer (more than two blocks, or k == 1): the single-step layout's
                // mask plane / rank directory (infer_presence, src/fms_index.h:126-144)
                const uint32_t bi = i >> kBlockRowsLog2, o = i & (kBlockRows - 1u);
                const uint4 mi = ix.sblocks[(static_cast<size_t>(bi) << 2) + 3u];
                if (a.mode_all) {
                    const uint32_t mw = sel4(o >> 5, opaque(mi.x), opaque(mi.y), opaque(mi.z), opaque(mi.w));
                    res = static_cast<int>((mw >> (o & 31u)) & 1u);
                } else {
                    const uint32_t bj = (j - 1u) >> kBlockRowsLog2, oj = j - (bj << kBlockRowsLog2);
                    const uint4 mj = ix.sblocks[(static_cast<size_t>(bj) << 2) + 3u];
                    res = (ix.mrank[bj] + popc_prefix128(mj, oj)) > (ix.mrank[bi] + popc_prefix128(mi, o));
                }
                answered = true;
            }
        }
        if (answered && res >= 0) {
            a.tern[val] = static_cast<uint8_t>(res + 1);
            if (COUNT) n_probes++;
        }
    }
    if (COUNT) {
        uint32_t v[4] = {n_steps, n_blocks, n_probes, n_items};
        const int slot[4] = {0, 1, 2, 3};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t x = v[q];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if (ln == 0 && x) atomicAdd(&counters[slot[q]], static_cast<unsigned long long>(x));
        }
    }
}

// ---- (4) both strands' ternaries -> the call's output bytes (src/fms_index.h:283-299 with the forward strand first) ----------------------
template <bool COUNT>
__global__ __launch_bounds__(256) void sw_combine_kernel(const uint16_t* __restrict__ tern2, uint64_t n, uint32_t flags, uint8_t* __restrict__ out,
                                                         unsigned long long* __restrict__ counters) {
    const uint64_t q = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const uint32_t t2 = tern2[q];
    const int rf = static_cast<int>(t2 & 0xFFu) - 1, rr = static_cast<int>(t2 >> 8) - 1;
    uint8_t byte;
    if (flags & FMSI_HIP_STRICT) byte = static_cast<uint8_t>((rf + 1) | ((rr + 1) << 2));
    else if (flags & FMSI_HIP_MODE_ALL) byte = static_cast<uint8_t>((rf != -1 ? rf : rr) == 1);
    else byte = static_cast<uint8_t>(rf == 1 || rr == 1);
    out[q] = byte;
    if (COUNT && (q & 63u) == 0u) atomicAdd(&counters[6], static_cast<unsigned long long>(min(static_cast<uint64_t>(64), n - q)));
}

template <bool COUNT>
int launch_sorted_t(fmsi_hip_index* idx, const uint64_t* d_kmers, uint64_t n, int k, uint32_t flags, uint8_t* o8, unsigned long long* d_counters, hipStream_t s) {
    const uint64_t items = 2 * n;
    const uint64_t waves_a = (items + 63) / 64;
    // scratch: two key / value buffers (the sort's double buffers; the hand-over reuses the spare ones), (i, j) per hand-over slot, the
    // ternaries, the sort's own temporary storage
    const int sort_bases = std::min(k, 16);
    const unsigned begin_bit = 64u - 2u * static_cast<unsigned>(sort_bases), end_bit = 64u;
    auto round_up = [](size_t x) { return (x + 255) & ~size_t(255); };
    const size_t key_b = round_up(items * 8), val_b = round_up(items * 4), ij_b = round_up(waves_a * 64 * 8), tern_b = round_up(items), cnt_b = round_up(waves_a * 4);
    size_t temp_b = 0;
    {
        rocprim::double_buffer<uint64_t> dk(nullptr, nullptr);
        rocprim::double_buffer<uint32_t> dv(nullptr, nullptr);
        SW_TRY(rocprim::radix_sort_pairs<SortConfig>(nullptr, temp_b, dk, dv, static_cast<size_t>(items), begin_bit, end_bit, s));
    }
    temp_b = round_up(temp_b);
    const size_t need = 2 * round_up(waves_a * 64 * 8) + 2 * round_up(waves_a * 64 * 4) + ij_b + tern_b + cnt_b + temp_b;
    (void)key_b;
    (void)val_b;
    int rc = ensure_cap(&idx->d_sorted, &idx->d_sorted_cap, need);
    if (rc) return rc;
    char* base = static_cast<char*>(idx->d_sorted);
    uint64_t* keys0 = reinterpret_cast<uint64_t*>(base);
    base += round_up(waves_a * 64 * 8);
    uint64_t* keys1 = reinterpret_cast<uint64_t*>(base);
    base += round_up(waves_a * 64 * 8);
    uint32_t* vals0 = reinterpret_cast<uint32_t*>(base);
    base += round_up(waves_a * 64 * 4);
    uint32_t* vals1 = reinterpret_cast<uint32_t*>(base);
    base += round_up(waves_a * 64 * 4);
    uint2* ij = reinterpret_cast<uint2*>(base);
    base += ij_b;
    uint8_t* tern = reinterpret_cast<uint8_t*>(base);
    base += tern_b;
    uint32_t* h_count = reinterpret_cast<uint32_t*>(base);
    base += cnt_b;
    void* temp = base;

    hipLaunchKernelGGL(sw_items_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, s, d_kmers, n, k, reinterpret_cast<ulonglong2*>(keys0),
                       reinterpret_cast<uint2*>(vals0));
    SW_TRY(hipGetLastError());
    SW_TRY(hipMemsetAsync(tern, 0, items, s));
    rocprim::double_buffer<uint64_t> dk(keys0, keys1);
    rocprim::double_buffer<uint32_t> dv(vals0, vals1);
    static const bool no_sort = std::getenv("FMSI_HIP_SORTED_NOSORT") != nullptr;   // (measurements: what the order is worth)
    if (!no_sort) SW_TRY(rocprim::radix_sort_pairs<SortConfig>(temp, temp_b, dk, dv, static_cast<size_t>(items), begin_bit, end_bit, s));

    const PairIndex pix = make_pair_index(idx);
    // where to cut: the absent searches have died out a couple of bases past log4(rows)
    uint32_t bases_a = static_cast<uint32_t>(std::ceil(std::log(static_cast<double>(std::max<uint64_t>(idx->n, 4))) / std::log(4.0))) + 2u;
    static const char* cut_env = std::getenv("FMSI_HIP_SORTED_CUT");   // (measurements: bases walked by the first launch; 0 = one launch)
    if (cut_env) bases_a = static_cast<uint32_t>(std::atoi(cut_env));
    uint32_t stop_left = 0;
    if (bases_a != 0 && static_cast<uint32_t>(k) > bases_a + 3u) {
        stop_left = static_cast<uint32_t>(k) - bases_a;
        stop_left &= ~1u;                // pair steps from there on
    }
    WalkArgs a{};
    a.keys = dk.current();
    a.vals = dv.current();
    a.n_items = items;
    a.h_key = dk.alternate();
    a.h_ij = ij;
    a.h_val = dv.alternate();
    a.h_count = h_count;
    a.n_waves_a = waves_a;
    a.tern = tern;
    a.left0 = static_cast<uint32_t>(k);
    a.stop_left = stop_left;
    a.second = 0;
    a.mode_all = (flags & FMSI_HIP_MODE_ALL) ? 1u : 0u;
    hipLaunchKernelGGL((sorted_walk_kernel<COUNT>), dim3(static_cast<uint32_t>((waves_a + 3) / 4)), dim3(256), 0, s, pix, a, d_counters);
    SW_TRY(hipGetLastError());
    if (stop_left) {
        WalkArgs b = a;
        b.left0 = stop_left;
        b.stop_left = 0;
        b.second = 1;
        const uint64_t waves_b = (waves_a + kJoin - 1) / kJoin;
        hipLaunchKernelGGL((sorted_walk_kernel<COUNT>), dim3(static_cast<uint32_t>((waves_b + 3) / 4)), dim3(256), 0, s, pix, b, d_counters);
        SW_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL((sw_combine_kernel<COUNT>), dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const uint16_t*>(tern), n, flags, o8,
                       d_counters);
    SW_TRY(hipGetLastError());
    return 0;
}

}  // namespace

bool sorted_walk_applies(const fmsi_hip_index* idx, int k, uint32_t flags) {
    return (flags & FMSI_HIP_KERNEL_MASK) == FMSI_HIP_KERNEL_SORTED && idx->d_pblocks && !idx->wide && k >= 1 && k <= 32 && !(flags & FMSI_HIP_ORDERS);
}

int launch_sorted_query(fmsi_hip_index* idx, const uint64_t* d_kmers, uint64_t n, int k, uint32_t flags, uint8_t* o8, unsigned long long* d_counters, hipStream_t s) {
    // sub-batches: item values are 32 bits (2 q + strand) and one radix sort takes < 2^32 items
    constexpr uint64_t kSub = 1ull << 28;
    for (uint64_t at = 0; at < n; at += kSub) {
        const uint64_t m = std::min(kSub, n - at);
        const int rc = d_counters ? launch_sorted_t<true>(idx, d_kmers + at, m, k, flags, o8 + at, d_counters, s)
                                  : launch_sorted_t<false>(idx, d_kmers + at, m, k, flags, o8 + at, d_counters, s);
        if (rc) return rc;
    }
    return 0;
}

}  // namespace fmsi
