// The table-less LF walk for a BATCH of isolated k-mers: strand searches sorted by the bases the backward search consumes first, walked
// in lock-step -- one wavefront lane per strand search -- with two-base steps over the pair layout (FMSI_HIP_KERNEL_SORTED).
//
// What it replaces in the reference: get_range_with_pattern() + infer_presence() per strand (src/fms_index.h:117-144) under the strand
// schedule of query_kmers_single<>() (src/fms_index.h:263-331); every interval is the one update_range() (src/fms_index.h:98-103) produces,
// so the answers are the reference's.  No table of the index is read: what makes it fast is the ORDER of the searches (and that the top of
// the search tree, which every search of a large batch shares, is walked once per batch).
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
//     together by rocPRIM's radix sort on the first 12 bases (three passes): the reverse-complement searches share their first steps like
//     the forward ones (a second strand that the schedule would not have asked for costs a few shared steps and dies; its ternary is unused);
//   * the top of the search tree ONCE per batch: the intervals of all 4^D1, then all 4^D2 first-base strings (D2 <= 13, about one search
//     per entry; scratch memory of the call, made anew for every batch) by this kernel with a lane per entry -- the items start from
//     their D2 entry;
//   * two-base steps over the 128-byte pair blocks, the lines fetched COOPERATIVELY -- eight lanes load 16 bytes each of one lane's line
//     in one instruction (lanes of a sorted wave that want the same line share the request), staged through LDS at a 144-byte pitch,
//     the owner reads its planes with conflict-free b128 reads and ranks both boundaries itself (no lane pairs, no shuffles);
//   * the items' walk is cut in two launches where the absent k-mers have died out (after ~log4(n) + 3 bases): the survivors of three
//     waves continue in ONE wave of the second launch, so that the long private tails of the present k-mers run with most lanes live;
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
// rocPRIM's radix sort without its merge-sort path (the one it takes between 1 K and 1 M items).  On ROCm 7.2 (rocPRIM 4.2) / gfx950 that path
// is wrong for a sort over double buffers whose bit range does not start at bit 0: tools/micro/rocprim_sort_repro.hip, plain rocPRIM and a host
// check -- (begin_bit, end_bit) = (32, 64) and (40, 64) over 2 000 ... 1 000 000 pairs come back out of order, differently from run to run, below
// 300 000 with keys and values torn apart; 600 and 3 000 000 pairs are right, every range from bit 0 is right ((0, 44), (0, 60), (0, 64): what the
// index builder and the dictionaries ask for), and with the limit at 0 -- onesweep passes for every size above one block -- all of it is right
// (profiles/r05q_rocprim_sort_repro.txt).  This file is the only one that sorts from a bit above 0.
#include <rocprim/rocprim_version.hpp>
static_assert(ROCPRIM_VERSION_MAJOR >= 4, "radix_sort_config<SingleSortConfig, MergeSortConfig, OnesweepConfig, MergeSortLimit> as rocPRIM 4 (ROCm 7) has it; check the alias below against another release");
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
// what a launch of the walk takes its searches from and where it leaves them
enum : uint32_t {
    kSrcItems = 0,     // the sorted items, from the root interval [0, n) or -- top != nullptr -- from the top table's entry of their first top_bases bases
    kSrcHandover = 1,  // the survivors an earlier launch left in the hand-over slots (kJoin waves' slots per wave)
    kSrcTable = 2,     // lane e = entry e of a top-of-tree level: the out_bases-mer e, from the root or from its parent entry in `top`
};
enum : uint32_t {
    kSinkResults = 0,  // walks to the k-mer's end: ternaries
    kSinkHandover = 1, // stops when stop_left bases are left: survivors to the hand-over slots
    kSinkTable = 2,    // stops at the level's depth: the interval (or (0, 0)) to top_out[e]
};
struct WalkArgs {
    const uint64_t* keys;     // kSrcItems: the sorted items
    const uint32_t* vals;
    uint64_t n_items;         // items, or entries of the level being made
    // hand-over between launches: wave w leaves its survivors in slots [64 w, 64 w + count[w])
    uint64_t* h_key;          // the bases still to go, left-aligned
    uint2* h_ij;
    uint32_t* h_val;
    uint32_t* h_count;
    uint64_t n_waves_a;       // waves of the launch that filled the slots
    uint8_t* tern;            // result + 1 per item (0 = absent: pre-zeroed)
    // the top of the search tree, made once per batch (transient: scratch memory of this call): entry e of a level of D bases = the
    // interval of the D-mer e (first-consumed base most significant), (0, 0) when it is empty
    const uint2* top;
    uint2* top_out;
    uint32_t top_bases, out_bases;
    uint32_t left0;           // bases to go when this launch starts
    uint32_t stop_left;       // stop when this many bases are left (0: walk to the end)
    uint32_t src, sink;
    uint32_t mode_all;
};

// ---- rank arithmetic on a staged pair block, in 32-bit words (the kernel is bound by vector instructions: DESIGN.md 5) ----------------
struct Match {
    uint32_t m0, m1, m2, m3;   // rows 0-31, 32-63, 64-95, 96-127 of the block that match the step's pattern
};
// rows of the staged block that match pair p = c2 | c1 << 2 (SINGLE: symbol c1 = p >> 2 alone)
template <bool SINGLE>
__device__ __forceinline__ Match block_match(const uint4* L4, uint32_t p) {
    const uint32_t x1h = (p & 8u) ? 0u : 0xFFFFFFFFu, x1l = (p & 4u) ? 0u : 0xFFFFFFFFu;
    const uint4 a0 = L4[0], b0 = L4[4];   // {c1hi 0-31, c1hi 32-63, c1lo 0-31, c1lo 32-63} of rows 0-63 / 64-127
    Match m;
    m.m0 = (a0.x ^ x1h) & (a0.z ^ x1l);
    m.m1 = (a0.y ^ x1h) & (a0.w ^ x1l);
    m.m2 = (b0.x ^ x1h) & (b0.z ^ x1l);
    m.m3 = (b0.y ^ x1h) & (b0.w ^ x1l);
    if (!SINGLE) {
        const uint32_t x2h = (p & 2u) ? 0u : 0xFFFFFFFFu, x2l = (p & 1u) ? 0u : 0xFFFFFFFFu;
        const uint4 a1 = L4[1], b1 = L4[5];   // the same for c2
        m.m0 &= (a1.x ^ x2h) & (a1.z ^ x2l);
        m.m1 &= (a1.y ^ x2h) & (a1.w ^ x2l);
        m.m2 &= (b1.x ^ x2h) & (b1.z ^ x2l);
        m.m3 &= (b1.y ^ x2h) & (b1.w ^ x2l);
    }
    return m;
}
// cumulative count of pair p before the block (24 bits, relative to its superblock); SINGLE: of the four pairs with c1 = p >> 2
template <bool SINGLE>
__device__ __forceinline__ uint32_t block_count(const uint32_t* L, uint32_t p) {
    const uint32_t* H = L + 16u * (p >> 3);         // the half that keeps the counts of c1 in {2h, 2h+1}
    if (SINGLE) {
        const uint32_t g = (p >> 2) & 1u;
        const uint32_t l0 = H[10u + 2u * g], l1 = H[11u + 2u * g], hb = H[14u + g];
        return (l0 & 0xFFFFu) + (l0 >> 16) + (l1 & 0xFFFFu) + (l1 >> 16) +
               (((hb & 0xFFu) + ((hb >> 8) & 0xFFu) + ((hb >> 16) & 0xFFu) + (hb >> 24)) << 16);
    }
    const uint32_t m = p & 7u;
    const uint32_t lw = H[10u + (m >> 1)], hw = H[14u + (m >> 2)];
    return ((lw >> (16u * (m & 1u))) & 0xFFFFu) | (((hw >> (8u * (m & 3u))) & 0xFFu) << 16);
}
// matching rows among the block's first o rows, o in [0, 128]: whole words through the running sums, one masked popcount
struct MatchSums {
    uint32_t p1, p2, p3, p4;
};
__device__ __forceinline__ MatchSums match_sums(const Match& m) {
    MatchSums s;
    s.p1 = __popc(m.m0);
    s.p2 = s.p1 + __popc(m.m1);
    s.p3 = s.p2 + __popc(m.m2);
    s.p4 = s.p3 + __popc(m.m3);
    return s;
}
__device__ __forceinline__ uint32_t rows_below(const Match& m, const MatchSums& s, uint32_t o) {
    const uint32_t wq = o >> 5, r = o & 31u;
    uint32_t word = wq > 0u ? m.m1 : m.m0, full = wq > 0u ? s.p1 : 0u;
    word = wq > 1u ? m.m2 : word;
    full = wq > 1u ? s.p2 : full;
    word = wq > 2u ? m.m3 : word;
    full = wq > 2u ? s.p3 : full;
    full = wq > 3u ? s.p4 : full;                   // (o == 128: r == 0, the masked word counts nothing)
    return full + __popc(word & ((1u << r) - 1u));
}
// the last step's rows: bits [lo, hi) of a 128-row block as four words
__device__ __forceinline__ Match row_range(int lo, int hi) {
    Match r;
    r.m0 = prefix32(hi) & ~prefix32(lo);
    r.m1 = prefix32(hi - 32) & ~prefix32(lo - 32);
    r.m2 = prefix32(hi - 64) & ~prefix32(lo - 64);
    r.m3 = prefix32(hi - 96) & ~prefix32(lo - 96);
    return r;
}
// a row that is stored as 'A' but stands for no symbol (suffix starts 0 and 1) is not a row of any interval
__device__ __forceinline__ void drop_row128(uint32_t row, uint32_t blk, Match& m) {
    if ((row >> kBlockRowsLog2) == blk) {
        const uint32_t b = row & 127u, bit = ~(1u << (b & 31u));
        if (b < 32u) m.m0 &= bit;
        else if (b < 64u) m.m1 &= bit;
        else if (b < 96u) m.m2 &= bit;
        else m.m3 &= bit;
    }
}
// first selected row's bit of the pre-mapped mask plane (2 | bit), or 0 when no row is selected
__device__ __forceinline__ uint32_t first_row_bit(const Match& sel, const Match& K) {
    if (sel.m0) return 2u | ((K.m0 >> (__ffs(static_cast<int>(sel.m0)) - 1)) & 1u);
    if (sel.m1) return 2u | ((K.m1 >> (__ffs(static_cast<int>(sel.m1)) - 1)) & 1u);
    if (sel.m2) return 2u | ((K.m2 >> (__ffs(static_cast<int>(sel.m2)) - 1)) & 1u);
    if (sel.m3) return 2u | ((K.m3 >> (__ffs(static_cast<int>(sel.m3)) - 1)) & 1u);
    return 0u;
}

// what a lane carries from round to round
struct Search {
    uint64_t key;     // the bases still to go, left-aligned
    uint32_t i, j;    // the interval (i == j == 0 for a lane without a search)
    bool alive;
    int res;          // the ternary once the last step has folded it
};

// One lock-step round: a pair step (SINGLE: the one single-base step of an odd k) for every live lane of the wave.  LAST: the k-mer's
// last step, which also answers the search when the block(s) at hand cover the interval.
template <bool SINGLE, bool LAST, bool COUNT>
__device__ __forceinline__ void walk_round(const PairIndex& ix, const uint32_t* s_c2, uint4* lines, const uint4* L4, uint32_t ln, uint32_t mode_all, Search& q,
                                           uint32_t& n_steps, uint32_t& n_blocks) {
    const uint32_t* L = reinterpret_cast<const uint32_t*>(L4);
    const char* pbase = reinterpret_cast<const char*>(ix.pb);
    const uint32_t piece16 = (ln & 7u) * 16u, grp = ln >> 3, piece = ln & 7u;
    const uint32_t p = SINGLE ? (static_cast<uint32_t>(q.key >> 62) << 2) : static_cast<uint32_t>(q.key >> 60);
    const uint32_t jm1 = q.alive ? q.j - 1u : 0u;   // (a lane without a search computes on row 0: every address it forms is valid)
    const uint32_t bi = q.i >> kBlockRowsLog2, bj = jm1 >> kBlockRowsLog2;
    const bool two = q.alive && bj != bi;
    const bool any_two = __any(two);
    // byte offsets of the lines (a lane without a search, or without a second line, asks for a line that is at hand anyway: no predicates)
    const uint32_t off_a = q.alive ? (bi << 7) : 0u;
    const uint32_t off_b = two ? (bj << 7) : off_a;
    // superblock counts and C2 (small tables: cache resident)
    uint32_t sbi, sbj, cbase;
    if (SINGLE) {
        const uint4 si = *reinterpret_cast<const uint4*>(ix.sbt + ((q.i >> kSuperLog2) << 4) + (p & 12u));
        const uint4 sj = *reinterpret_cast<const uint4*>(ix.sbt + ((jm1 >> kSuperLog2) << 4) + (p & 12u));
        sbi = si.x + si.y + si.z + si.w;
        sbj = sj.x + sj.y + sj.z + sj.w;
        cbase = sel4(p >> 2, ix.C0, ix.C1, ix.C2_, ix.C3);
    } else {
        sbi = ix.sbt[((q.i >> kSuperLog2) << 4) + p];
        sbj = any_two ? ix.sbt[((jm1 >> kSuperLog2) << 4) + p] : sbi;
        cbase = s_c2[p];
    }
    // ---- the lines: eight lanes per line, eight lines per instruction ----
    uint4 va[8];
#pragma unroll
    for (uint32_t g = 0; g < 8; ++g) {
        const uint32_t o = static_cast<uint32_t>(__shfl(static_cast<int>(off_a), static_cast<int>(8u * g + grp), 64));
        va[g] = *reinterpret_cast<const uint4*>(pbase + static_cast<uint64_t>(o + piece16));
    }
#pragma unroll
    for (uint32_t g = 0; g < 8; ++g) lines[(8u * g + grp) * kLinePitch + piece] = va[g];
    __builtin_amdgcn_wave_barrier();
    const uint32_t oi = q.i & (kBlockRows - 1u), oj = q.j - (bj << kBlockRowsLog2);   // 0..127, 1..128
    const Match ma = block_match<SINGLE>(L4, p);
    const MatchSums sa = match_sums(ma);
    const uint32_t cnt_a = block_count<SINGLE>(L, p);
    uint32_t ri = cnt_a + rows_below(ma, sa, oi);
    uint32_t rj;
    Match mb = ma;
    Match ka{0, 0, 0, 0}, kb{0, 0, 0, 0};   // the pre-mapped mask planes of the two blocks (last step only)
    if (LAST) {
        ka.m0 = L[8]; ka.m1 = L[9]; ka.m2 = L[24]; ka.m3 = L[25];
    }
    if (any_two) {   // (the second lines in a pass of their own: rare once the walk is below the top of the tree, and no registers held for it)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (uint32_t g = 0; g < 8; ++g) {
            const uint32_t o = static_cast<uint32_t>(__shfl(static_cast<int>(off_b), static_cast<int>(8u * g + grp), 64));
            va[g] = *reinterpret_cast<const uint4*>(pbase + static_cast<uint64_t>(o + piece16));
        }
#pragma unroll
        for (uint32_t g = 0; g < 8; ++g) lines[(8u * g + grp) * kLinePitch + piece] = va[g];
        __builtin_amdgcn_wave_barrier();
        mb = block_match<SINGLE>(L4, p);
        rj = block_count<SINGLE>(L, p) + rows_below(mb, match_sums(mb), oj);
        if (LAST) {
            kb.m0 = L[8]; kb.m1 = L[9]; kb.m2 = L[24]; kb.m3 = L[25];
        }
    } else {
        rj = cnt_a + rows_below(ma, sa, oj);
    }
    __builtin_amdgcn_wave_barrier();   // (the next round's staging must not overtake this round's reads)
    if (COUNT && q.alive) {
        n_steps += SINGLE ? 1u : 2u;
        n_blocks += two ? 2u : 1u;
    }
    // the last step also answers the search when the block(s) at hand cover the interval: the rows of [i, j) that match the pair ARE the
    // new interval, in order, and the mask plane holds mask[LF2(row)] (src/fms_index.h:126-144)
    int folded = -2;
    if (LAST && q.alive && bj - bi <= 1u) {
        Match ra = row_range(static_cast<int>(oi), two ? 128 : static_cast<int>(oj));
        ra.m0 &= ma.m0; ra.m1 &= ma.m1; ra.m2 &= ma.m2; ra.m3 &= ma.m3;
        Match rb = row_range(0, two ? static_cast<int>(oj) : 0);
        rb.m0 &= mb.m0; rb.m1 &= mb.m1; rb.m2 &= mb.m2; rb.m3 &= mb.m3;
        drop_row128(ix.r0, bi, ra);
        drop_row128(ix.r1, bi, ra);
        drop_row128(ix.r0, bj, rb);
        drop_row128(ix.r1, bj, rb);
        if (mode_all) {   // the mask bit of the interval's first row
            uint32_t h = first_row_bit(ra, ka);
            if (!h) h = first_row_bit(rb, kb);
            if (h) folded = static_cast<int>(h & 1u);
        } else if (ra.m0 | ra.m1 | ra.m2 | ra.m3 | rb.m0 | rb.m1 | rb.m2 | rb.m3) {   // any ON row in the interval
            folded = ((ra.m0 & ka.m0) | (ra.m1 & ka.m1) | (ra.m2 & ka.m2) | (ra.m3 & ka.m3) | (rb.m0 & kb.m0) | (rb.m1 & kb.m1) | (rb.m2 & kb.m2) |
                      (rb.m3 & kb.m3))
                         ? 1
                         : 0;
        }
    }
    ri += sbi;
    rj += sbj;
    if (SINGLE) {
        if (p == 0u) {   // the '$' slot is stored as c1 = 'A' (src/fms_index.h:81)
            ri -= (q.i > ix.r0);
            rj -= (q.j > ix.r0);
        }
        q.key <<= 2;
    } else {
        if (p == 0u) {   // '$' slot stored as (A, A)
            ri -= (q.i > ix.r0);
            rj -= (q.j > ix.r0);
        }
        if (p == ix.p1) {   // the row of suffix 1 has no c2: stored as (A, T[0])
            ri -= (q.i > ix.r1);
            rj -= (q.j > ix.r1);
        }
        q.key <<= 4;
    }
    if (q.alive) {
        q.i = cbase + ri;
        q.j = cbase + rj;
        if (q.i == q.j) {
            q.alive = false;          // absent: the ternary array's zero fill says so
            q.i = q.j = 0;
        } else if (LAST && folded >= 0) {
            q.res = folded;
            q.alive = false;
            q.i = q.j = 0;
        }
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
    const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    uint32_t n_steps = 0, n_blocks = 0, n_probes = 0, n_items = 0;

    // a launch over the hand-over slots takes the survivors of kJoin waves of the launch before, 64 at a time
    uint32_t c0 = 0, c1n = 0, total = 0;
    if (a.src == kSrcHandover) {
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
        // ---- this pass's searches ----
        Search q;
        q.key = 0;
        q.i = q.j = 0;
        q.alive = false;
        q.res = -1;
        uint32_t val = 0;
        const uint32_t me = at + ln;
        if (me < total) {
            if (a.src == kSrcHandover) {
                const uint32_t r = me < c0 ? 0u : (me < c0 + c1n ? 1u : 2u);
                const uint64_t slot = (wave * kJoin + r) * 64u + (me - (r == 0u ? 0u : (r == 1u ? c0 : c0 + c1n)));
                q.key = a.h_key[slot];
                const uint2 ij = a.h_ij[slot];
                q.i = ij.x;
                q.j = ij.y;
                val = a.h_val[slot];
            } else {
                const uint64_t it = wave * 64u + me;
                if (a.src == kSrcTable) {
                    q.key = it << (64u - 2u * a.out_bases);
                    val = static_cast<uint32_t>(it);
                } else {
                    q.key = a.keys[it];
                    val = a.vals[it];
                }
                q.i = 0;
                q.j = ix.n;
                if (a.top) {   // the first top_bases bases were walked once for the whole batch
                    const uint2 ij = a.top[q.key >> (64u - 2u * a.top_bases)];
                    q.i = ij.x;
                    q.j = ij.y;
                    q.key <<= 2u * a.top_bases;
                }
            }
            q.alive = q.i != q.j;
            if (COUNT && a.src == kSrcItems) n_items++;
        }
        uint32_t left = a.left0;
        if ((left & 1u) && left > a.stop_left && __any(q.alive)) {   // an odd number of bases to go: one single-base step first
            walk_round<true, false, COUNT>(ix, s_c2, lines, L4, ln, a.mode_all, q, n_steps, n_blocks);
            left -= 1u;
        }
        const uint32_t fold_at = a.sink == kSinkResults ? 2u : 0u;      // the k-mer's last pair step folds the answer
        while (left > a.stop_left + fold_at && __any(q.alive)) {
            walk_round<false, false, COUNT>(ix, s_c2, lines, L4, ln, a.mode_all, q, n_steps, n_blocks);
            left -= 2u;
        }
        if (fold_at && left == 2u && __any(q.alive)) {
            walk_round<false, true, COUNT>(ix, s_c2, lines, L4, ln, a.mode_all, q, n_steps, n_blocks);
            left = 0;
        }
        // ---- what is left of this pass ----
        if (a.sink == kSinkHandover) {
            // the survivors: dense in this wave's slots (none when the wave's last search died before the cut)
            const uint64_t m = __ballot(q.alive);
            if (q.alive) {
                const uint64_t slot = wave * 64u + lanes_below(m);
                a.h_key[slot] = q.key;
                a.h_ij[slot] = make_uint2(q.i, q.j);
                a.h_val[slot] = val;
            }
            if (ln == 0u) a.h_count[wave] = static_cast<uint32_t>(__popcll(m));
        } else if (a.sink == kSinkTable) {
            if (me < total) a.top_out[val] = make_uint2(q.i, q.j);   // ((0, 0) when the entry's interval is empty)
        } else if (q.alive) {
            // a non-empty interval that the last step could not answer (more than two blocks, or k == 1): the single-step layout's
            // mask plane / rank directory (infer_presence, src/fms_index.h:126-144)
            const uint32_t bi = q.i >> kBlockRowsLog2, o = q.i & (kBlockRows - 1u);
            const uint4 mi = ix.sblocks[(static_cast<size_t>(bi) << 2) + 3u];
            if (a.mode_all) {
                const uint32_t mw = sel4(o >> 5, opaque(mi.x), opaque(mi.y), opaque(mi.z), opaque(mi.w));
                q.res = static_cast<int>((mw >> (o & 31u)) & 1u);
            } else {
                const uint32_t bj = (q.j - 1u) >> kBlockRowsLog2, oj = q.j - (bj << kBlockRowsLog2);
                const uint4 mj = ix.sblocks[(static_cast<size_t>(bj) << 2) + 3u];
                q.res = (ix.mrank[bj] + popc_prefix128(mj, oj)) > (ix.mrank[bi] + popc_prefix128(mi, o));
            }
        }
        if (q.res >= 0) {
            a.tern[val] = static_cast<uint8_t>(q.res + 1);
            if (COUNT) n_probes++;
        }
    }
    if (COUNT) {
        uint32_t v[4] = {n_steps, n_blocks, n_probes, n_items};
        const int slot[4] = {0, 1, 2, 3};
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            uint32_t x = v[qq];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if (ln == 0 && x) atomicAdd(&counters[slot[qq]], static_cast<unsigned long long>(x));
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

// depths of the batch's transient top-of-tree levels (0 = none): D2 = the deepest level with at least one search per entry on average
// (at most 13 bases: 4^13 intervals are 537 MB), D1 four bases above it; both of k's parity, so that the walk's one single-base step --
// taken first when k is odd -- falls into the first level and everything below it is pair steps
void top_depths(uint64_t items, int k, uint32_t& d1, uint32_t& d2) {
    d1 = d2 = 0;
    static const bool off = std::getenv("FMSI_HIP_SORTED_NO_TOP") != nullptr;   // (measurements)
    if (off) return;
    uint32_t d = 0;
    while (d + 2 <= 14 && (1ull << (2 * (d + 2))) <= items) d += 2;
    if ((static_cast<uint32_t>(k) & 1u) && d > 0) d -= 1;       // odd k: odd depths
    d = std::min(d, 13u - ((static_cast<uint32_t>(k) & 1u) ? 0u : 1u));
    if (d < 3 || static_cast<uint32_t>(k) < d + 4) return;
    d2 = d;
    if (d >= 7) d1 = d - 4;
}

template <bool COUNT>
int launch_sorted_t(fmsi_hip_index* idx, const uint64_t* d_kmers, uint64_t n, int k, uint32_t flags, uint8_t* o8, unsigned long long* d_counters, hipStream_t s) {
    const uint64_t items = 2 * n;
    const uint64_t waves_a = (items + 63) / 64;
    uint32_t d1 = 0, d2 = 0;
    top_depths(items, k, d1, d2);
    // scratch: two key / value buffers (the sort's double buffers; the hand-over reuses the spare ones), (i, j) per hand-over slot, the
    // ternaries, the top-of-tree levels, the sort's own temporary storage
    // sorted by their first 12 bases: three radix passes.  What the order has to deliver is (a) neighbouring lanes under one entry of the
    // deepest top level (13 bases at most) and (b) the lines of the first step below it shared inside a wave; 16 bases (a fourth pass,
    // +2.0 ms per 2^28 items) bought 0.3 ms of walk at human scale, 8 bases lost 0.8 ms (profiles/r04f_*)
    static const char* sort_env = std::getenv("FMSI_HIP_SORTED_BASES");   // (measurements)
    const int sort_bases = std::min(k, sort_env ? std::max(1, std::atoi(sort_env)) : 12);
    const unsigned begin_bit = 64u - 2u * static_cast<unsigned>(sort_bases), end_bit = 64u;
    auto round_up = [](size_t x) { return (x + 255) & ~size_t(255); };
    const size_t key_b = round_up(waves_a * 64 * 8), val_b = round_up(waves_a * 64 * 4), ij_b = round_up(waves_a * 64 * 8), tern_b = round_up(items),
                 cnt_b = round_up(waves_a * 4);
    const size_t top1_b = d1 ? round_up((size_t(1) << (2 * d1)) * 8) : 0, top2_b = d2 ? round_up((size_t(1) << (2 * d2)) * 8) : 0;
    size_t temp_b = 0;
    {
        rocprim::double_buffer<uint64_t> dk(nullptr, nullptr);
        rocprim::double_buffer<uint32_t> dv(nullptr, nullptr);
        SW_TRY(rocprim::radix_sort_pairs<SortConfig>(nullptr, temp_b, dk, dv, static_cast<size_t>(items), begin_bit, end_bit, s));
    }
    temp_b = round_up(temp_b);
    const size_t need = 2 * key_b + 2 * val_b + ij_b + tern_b + cnt_b + top1_b + top2_b + temp_b;
    if (ensure_cap(&idx->d_sorted, &idx->d_sorted_cap, need)) {   // (before anything is launched: a caller may fall back to another kernel)
        (void)hipGetLastError();
        set_last_error("no device memory for the sorted walk's scratch (" + std::to_string(need >> 20) + " MiB)");
        return FMSI_HIP_ENOMEM;
    }
    char* base = static_cast<char*>(idx->d_sorted);
    auto carve = [&](size_t bytes) {
        char* p = base;
        base += bytes;
        return p;
    };
    uint64_t* keys0 = reinterpret_cast<uint64_t*>(carve(key_b));
    uint64_t* keys1 = reinterpret_cast<uint64_t*>(carve(key_b));
    uint32_t* vals0 = reinterpret_cast<uint32_t*>(carve(val_b));
    uint32_t* vals1 = reinterpret_cast<uint32_t*>(carve(val_b));
    uint2* ij = reinterpret_cast<uint2*>(carve(ij_b));
    uint8_t* tern = reinterpret_cast<uint8_t*>(carve(tern_b));
    uint32_t* h_count = reinterpret_cast<uint32_t*>(carve(cnt_b));
    uint2* top1 = reinterpret_cast<uint2*>(carve(top1_b));
    uint2* top2 = reinterpret_cast<uint2*>(carve(top2_b));
    void* temp = carve(temp_b);

    hipLaunchKernelGGL(sw_items_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, s, d_kmers, n, k, reinterpret_cast<ulonglong2*>(keys0),
                       reinterpret_cast<uint2*>(vals0));
    SW_TRY(hipGetLastError());
    SW_TRY(hipMemsetAsync(tern, 0, items, s));
    rocprim::double_buffer<uint64_t> dk(keys0, keys1);
    rocprim::double_buffer<uint32_t> dv(vals0, vals1);
    static const bool no_sort = std::getenv("FMSI_HIP_SORTED_NOSORT") != nullptr;   // (measurements: what the order is worth)
    if (!no_sort) SW_TRY(rocprim::radix_sort_pairs<SortConfig>(temp, temp_b, dk, dv, static_cast<size_t>(items), begin_bit, end_bit, s));

    const PairIndex pix = make_pair_index(idx);
    auto launch = [&](const WalkArgs& w, uint64_t waves) {
        hipLaunchKernelGGL((sorted_walk_kernel<COUNT>), dim3(static_cast<uint32_t>((waves + 3) / 4)), dim3(256), 0, s, pix, w, d_counters);
    };
    WalkArgs a{};
    a.h_key = dk.alternate();
    a.h_ij = ij;
    a.h_val = dv.alternate();
    a.h_count = h_count;
    a.tern = tern;
    a.mode_all = (flags & FMSI_HIP_MODE_ALL) ? 1u : 0u;
    // ---- the top of the search tree, once for the batch ----
    const uint2* top = nullptr;
    uint32_t top_bases = 0;
    for (int lv = 0; lv < 2; ++lv) {
        const uint32_t d = lv == 0 ? d1 : d2;
        if (!d) continue;
        WalkArgs t = a;
        t.src = kSrcTable;
        t.sink = kSinkTable;
        t.n_items = 1ull << (2 * d);
        t.top = top;
        t.top_bases = top_bases;
        t.top_out = lv == 0 ? top1 : top2;
        t.out_bases = d;
        t.left0 = d - top_bases;
        t.stop_left = 0;
        launch(t, (t.n_items + 63) / 64);
        SW_TRY(hipGetLastError());
        top = t.top_out;
        top_bases = d;
    }
    // ---- the items: to where the absent searches have died out (a few bases past log4(rows)), the survivors to the end ----
    uint32_t bases_a = static_cast<uint32_t>(std::ceil(std::log(static_cast<double>(std::max<uint64_t>(idx->n, 4))) / std::log(4.0))) + 3u;
    static const char* cut_env = std::getenv("FMSI_HIP_SORTED_CUT");   // (measurements: bases walked before the cut; 0 = no cut)
    if (cut_env) bases_a = static_cast<uint32_t>(std::atoi(cut_env));
    uint32_t stop_left = 0;
    if (bases_a != 0 && static_cast<uint32_t>(k) > bases_a + 3u && bases_a > top_bases) {
        stop_left = static_cast<uint32_t>(k) - bases_a;
        stop_left &= ~1u;                // pair steps from there on
    }
    a.keys = dk.current();
    a.vals = dv.current();
    a.n_items = items;
    a.n_waves_a = waves_a;
    a.top = top;
    a.top_bases = top_bases;
    a.left0 = static_cast<uint32_t>(k) - top_bases;
    a.stop_left = stop_left;
    a.src = kSrcItems;
    a.sink = stop_left ? kSinkHandover : kSinkResults;
    launch(a, waves_a);
    SW_TRY(hipGetLastError());
    if (stop_left) {
        WalkArgs b = a;
        b.top = nullptr;
        b.left0 = stop_left;
        b.stop_left = 0;
        b.src = kSrcHandover;
        b.sink = kSinkResults;
        launch(b, (waves_a + kJoin - 1) / kJoin);
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

int launch_sorted_query(fmsi_hip_index* idx, const uint64_t* d_kmers, uint64_t n, int k, uint32_t flags, uint8_t* o8, unsigned long long* d_counters, hipStream_t s,
                        bool chosen_by_auto) {
    // sub-batches: item values are 32 bits (2 q + strand) and one radix sort takes < 2^32 items.  The scratch stays on the handle (d_sorted):
    // about 66 bytes per k-mer of a sub-batch -- two items per k-mer, each 2 keys x 8 + 2 values x 4 + 8 hand-over bytes + a ternary -- plus the
    // per-batch top levels (up to 537 MB) and rocPRIM's temporary: 17-18 GB for 2^28 k-mers.  A caller that asked for this kernel gets
    // that sub-batch; one that left the choice to the library (AUTO) is not surprised by it: 2^26 k-mers, 4.5 GB,
    // released again by fmsi_hip_index_release_scratch.
    static const char* sub_env = std::getenv("FMSI_HIP_SORTED_SUB");   // (tests: k-mers per sub-batch)
    const uint64_t kSub = sub_env ? std::max<uint64_t>(1, std::strtoull(sub_env, nullptr, 10)) : (chosen_by_auto ? 1ull << 26 : 1ull << 28);
    for (uint64_t at = 0; at < n; at += kSub) {
        const uint64_t m = std::min(kSub, n - at);
        const int rc = d_counters ? launch_sorted_t<true>(idx, d_kmers + at, m, k, flags, o8 + at, d_counters, s)
                                  : launch_sorted_t<false>(idx, d_kmers + at, m, k, flags, o8 + at, d_counters, s);
        if (rc) return rc;
    }
    return 0;
}

}  // namespace fmsi
