// Two-base LF steps over 128-byte blocks ("pair layout"), read cooperatively by PAIRS of lanes -- the
// request-count optimisation of the query path.
//
// Why: the chip serves ~50 G random block reads/s whether a block is 64 B or 128 B, as long as adjacent lanes share
// lines and a wave keeps <= ~8 dwordx4 loads in flight per lane (profiles/r01_gather_ceiling.md).  The single-step kernels run at that request ceiling, so the only way up is
// fewer requests per k-mer.  A 128-byte block can carry what TWO consecutive backward-search steps need: for every
// row r (suffix start s = SA[r]) both preceding symbols c1 = T[s-1] (the BWT symbol) and c2 = T[s-2], plus
// cumulative counts of the 16 (c2, c1) pairs.  Then
//     [i, j)  --(prepend "c2 c1")-->  [C2[c2,c1] + rank_pair(i), C2[c2,c1] + rank_pair(j))
// is ONE block read per boundary -- exactly the interval two update_range() calls (reference
// src/fms_index.h:98-103) produce, so the final SA interval, the early exit on an empty interval and therefore
// every result are unchanged.  c2 is derived on the device from the single-step layout alone (c2(r) = BWT[LF(r)]),
// so the layout can be added to any handle, whether it was loaded from the reference's files or built on the device.
//
// Block = 128 B = 128 rows = two 64-byte HALVES; half h covers rows 64h .. 64h+63 and is read by lane h of a lane
// pair with four dwordx4 loads (the pair's loads coalesce into one 128-B request):
//   dw 0,1  c1 hi plane   dw 2,3  c1 lo plane   dw 4,5  c2 hi plane   dw 6,7  c2 lo plane      (64 rows each)
//   dw 8,9  pre-mapped mask plane: bit r = mask[LF2(r)], the SA-transformed mask bit of the row a pair step through row r
//           lands on -- the rows of [i, j) that match the pair ARE the new interval, so the last step of a search reads
//           infer_presence() off the block it already holds (explicit probes use the single-step layout's mask plane)
//   dw 10..13  low 16 bits, dw 14,15 high 8 bits of the eight cumulative counts of the pairs with c1 in {2h, 2h+1}
//              (pair index p = c2 | c1 << 2 = 8h .. 8h+7): 24-bit counts relative to the superblock start
// Superblock = 2^23 rows; `sbt` holds the 16 absolute counts at each superblock start (21 KB at human scale: cache
// resident).  Rows without a c1 / c2 (suffix starts 0 and 1) are stored as 'A' and subtracted in the rank, like the
// '$' slot of the reference (src/fms_index.h:81).
//
// Kernel: the persistent-lane state machine of query_refill_kernel with a lane PAIR per k-mer: both lanes hold the
// same state, each ranks its 64 rows, the partial popcounts are added with one lane-swap and the count / mask bit /
// table entry owned by one lane is broadcast with one wave shuffle.  32 k-mers per wave and <= 8 dwordx4 in flight
// per lane: measured against the alternatives (one lane per k-mer: too many load instructions in flight, TA-bound at
// ~36 G requests/s; four lanes per k-mer: the state machine's VALU work per k-mer quadruples, VALU-bound at ~36 G/s).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/fmsi_hip.h"
#include "common.h"
#include "device_fm.h"
#include "index_priv.h"
#include "pair_layout.h"
#include "trace.h"

namespace fmsi {

namespace {

struct U16 {
    uint32_t v[16];
};
struct U16Plus {
    __host__ __device__ U16 operator()(const U16& a, const U16& b) const {
        U16 r;
#pragma unroll
        for (int q = 0; q < 16; ++q) r.v[q] = a.v[q] + b.v[q];
        return r;
    }
};

// ---- layout construction -----------------------------------------------------------------------------------
// One thread per row (256 threads = 4 waves = 2 blocks; one wave = one half): c1 from the single-step block,
// c2 = BWT[LF(row)].
__global__ __launch_bounds__(256) void pair_emit_kernel(DevIndex ix, uint32_t* __restrict__ pb, U16* __restrict__ counts,
                                                        uint32_t* __restrict__ specials /* [0] = r1, [1] = c1(r1) */) {
    __shared__ uint32_t s_cnt[4][16];
    const uint64_t row64 = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const bool valid = row64 < ix.n;
    const uint32_t row = static_cast<uint32_t>(row64);
    uint32_t c1 = 0, c2 = 0, mbit = 0;   // mbit = mask[LF2(row)]: the mask bit of the row a pair step through this row lands on
    if (valid) {
        const uint32_t* blk = reinterpret_cast<const uint32_t*>(ix.blocks) + (static_cast<size_t>(row >> kBlockRowsLog2) << 4);
        const uint32_t o = row & (kBlockRows - 1u), w = o >> 5, sh = o & 31u;
        c1 = (((blk[4 + w] >> sh) & 1u) << 1) | ((blk[8 + w] >> sh) & 1u);
        if (row != ix.dollar) {
            uint32_t a = row, b = row + 1u, dummy = 0;
            lf_step(ix, c1, a, b, dummy);                 // a = LF(row)
            if (a == ix.dollar) {                         // suffix 1: its predecessor is the '$' slot -> no c2
                specials[0] = row;
                specials[1] = c1;
            } else {
                const uint32_t* tb = reinterpret_cast<const uint32_t*>(ix.blocks) + (static_cast<size_t>(a >> kBlockRowsLog2) << 4);
                const uint32_t o2 = a & (kBlockRows - 1u), w2 = o2 >> 5, sh2 = o2 & 31u;
                c2 = (((tb[4 + w2] >> sh2) & 1u) << 1) | ((tb[8 + w2] >> sh2) & 1u);
                uint32_t a2 = a, b2 = a + 1u;
                lf_step(ix, c2, a2, b2, dummy);               // a2 = LF(LF(row))
                mbit = mask_bit(ix, a2);
            }
        }
    }
    const uint64_t h1 = __ballot(valid && (c1 & 2u)), l1 = __ballot(valid && (c1 & 1u));
    const uint64_t h2 = __ballot(valid && (c2 & 2u)), l2 = __ballot(valid && (c2 & 1u));
    const uint64_t mm = __ballot(valid && mbit), vv = __ballot(valid);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0 && vv) {
        const uint64_t w = row64 >> 6;                                   // 64-row word of the index = one half
        uint64_t* hd = reinterpret_cast<uint64_t*>(pb + (w >> 1) * kPairBlockDwords + static_cast<uint32_t>(w & 1) * 16u);
        hd[0] = h1;
        hd[1] = l1;
        hd[2] = h2;
        hd[3] = l2;
        hd[4] = mm;
    }
    if (lane < 16) {   // lane p counts pair p = c2 | c1 << 2 of its wave
        const uint64_t m = vv & ((lane & 8u) ? h1 : ~h1) & ((lane & 4u) ? l1 : ~l1) & ((lane & 2u) ? h2 : ~h2) & ((lane & 1u) ? l2 : ~l2);
        s_cnt[wave][lane] = static_cast<uint32_t>(__popcll(m));
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        const uint32_t which = threadIdx.x >> 4, p = threadIdx.x & 15u;
        counts[static_cast<uint64_t>(blockIdx.x) * 2 + which].v[p] = s_cnt[2 * which][p] + s_cnt[2 * which + 1][p];
    }
}

// per block: 24-bit counts relative to the superblock start; absolute counts of each superblock start
__global__ void pair_counts_kernel(const U16* __restrict__ scanned, uint64_t n_blocks, uint32_t* __restrict__ pb, uint32_t* __restrict__ sbt) {
    const uint64_t b = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const U16 a = scanned[b];
    const uint64_t sb = b >> kSuperBlocksLog2;
    const U16 base = scanned[sb << kSuperBlocksLog2];
    uint32_t* blk = pb + b * kPairBlockDwords;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint32_t lo[4] = {0, 0, 0, 0}, hi[2] = {0, 0};
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const uint32_t rel = a.v[8 * h + m] - base.v[8 * h + m];
            lo[m >> 1] |= (rel & 0xFFFFu) << (16 * (m & 1));
            hi[m >> 2] |= ((rel >> 16) & 0xFFu) << (8 * (m & 3));
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) blk[16 * h + 10 + d] = lo[d];
        blk[16 * h + 14] = hi[0];
        blk[16 * h + 15] = hi[1];
    }
    if ((b & ((1ull << kSuperBlocksLog2) - 1)) == 0) {
#pragma unroll
        for (int p = 0; p < 16; ++p) sbt[sb * 16 + p] = a.v[p];
    }
}

constexpr uint32_t kFlagOverlay = 0x40000000u;   // kernel-internal flag bit (never part of the ABI's flags)

struct alignas(8) WorkItem {   // a search the lookup kernel leaves to the stepping kernel
    uint64_t f;        // the k-mer (forward orientation)
    uint32_t x, y;     // (i, j): forward strand alive after the table; (j, i): the reverse complement is, the forward strand is absent
    uint32_t ql, pad;  // index of the k-mer in its wave's chunk
};
static_assert(sizeof(WorkItem) == 24, "WorkItem layout");

// ---- phase 1 of the phased path: the table lookups ------------------------------------------------------------------
// With the next-base signatures in the table most absent k-mers are settled by their two lookups.  Doing the lookups
// in a kernel of their own -- dense, no state machine -- and handing only the searches that are still alive to the
// stepping kernel keeps that kernel's lanes in one state (measured: the one-kernel form spends its time serialising
// lookup / step / finalize branches and reaches 40 G requests/s on a half-absent batch, against 50 G/s on all-present
// and 44 G/s on all-absent ones).
// A lookup is done by a LANE PAIR, each lane loading its 32 bytes of the entry's 64-byte line: measured on this table
// (34 GB, random lines) 46 G lookups/s, against 17 G/s when every lane fetches the whole line itself (four loads) -- the
// memory path pays per load instruction and distinct line (tools/gather_bench entry), and two adjacent
// lanes on one line are one request.  (Where 16 bytes per k-mer are enough -- the candidate tables of k - t <= 15 --
// lookup_lane_kernel below does the lookups with one lane and one load each.)
// Result per k-mer: one 8-byte record, (i, j) of the strand that is alive after the table -- stored as (j, i) when that
// is the reverse complement (the forward strand died in the table) -- or (0, 0) when both strands died.
// A wave owns a contiguous chunk of k-mers (the same chunks as the stepping kernel's waves).  K-mers that are settled here
// get their result written at once; the others go to the wave's PRIVATE work list -- list[chunk start + count++], no
// atomics -- as (k-mer, interval, index in the chunk), the interval (i, j) with i < j when the forward strand is alive
// after the table and (j, i) when that is the reverse complement (forward strand absent).
// Text verification (ix.stable): an interval of one to three rows leaves that many candidate occurrences; their text
// positions are stored per table entry, so the remaining k - t bases are compared with the text itself -- the candidates
// that match are the rows of the interval get_range_with_pattern() would end with, their mask bits are kept in the entry
// (bits 26-28) -- instead of (k - t) / 2 dependent block reads: infer_presence() (src/fms_index.h:117-144) from 1 + rows
// reads.
// RUNS: `kmers` is a 2-bit packed sequence and window q is base run_start[r] + q - run_off[r] of it, r the run with
// run_off[r] <= q < run_off[r + 1] (fmsi_hip_query_runs): the windows are cut here instead of by a pass of their own.
template <bool COUNT, bool RUNS>
__global__ __launch_bounds__(256) void pair_lookup_kernel(PairIndex ix, const uint64_t* __restrict__ kmers, uint64_t n_windows, int k, uint32_t flags,
                                                          uint32_t chunk, uint8_t* __restrict__ out, long long* __restrict__ out_orders,
                                                          WorkItem* __restrict__ list, uint32_t* __restrict__ list_count,
                                                          unsigned long long* __restrict__ counters, const uint64_t* __restrict__ run_start,
                                                          const uint64_t* __restrict__ run_off, uint64_t n_runs) {
    const uint32_t strict = flags & FMSI_HIP_STRICT;
    const uint32_t orders = flags & FMSI_HIP_ORDERS;
    const uint32_t mode_all = flags & FMSI_HIP_MODE_ALL;
    const int T = static_cast<int>(ix.ptable_len);
    const int t = k - 1 - T;
    const uint32_t R = static_cast<uint32_t>(k - T);                       // bases left after the table
    const uint64_t rmask = R >= 32u ? ~0ull : ((1ull << (2u * R)) - 1ull);
    const bool verify = ix.vtable != nullptr && !ix.vwide && ix.ptable_packed && !orders && k == ix.verify_k && R >= 1u && R <= 32u;
    const uint32_t tmask = static_cast<uint32_t>((1ull << (2 * T)) - 1ull);
    const uint32_t lane = threadIdx.x & 63u, sub = lane & 1u;
    const int pbase = static_cast<int>(lane & ~1u);
    const uint64_t pairs_below = (1ull << (lane & ~1u)) - 1ull;
    uint32_t n_lookups = 0, n_final = 0, n_vreads = 0;
    const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const uint64_t start = wave * chunk;
    const uint32_t len = start < n_windows ? static_cast<uint32_t>(min(static_cast<uint64_t>(chunk), n_windows - start)) : 0u;
    uint32_t count = 0;   // wave-uniform: work items written so far
    // RUNS: the run of this lane's current window -- windows only move forward, so the search restarts from the last run
    uint64_t run = 0, run_last = 0, run_end = 0, run_delta = 0;   // run_end = run_off[run + 1], run_delta = run_start[run] - run_off[run]
    auto find_run = [&](uint64_t lo, uint64_t hi, uint64_t q) -> uint64_t {   // run_off[lo] <= q < run_off[hi]
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (run_off[mid] <= q) lo = mid;
            else hi = mid;
        }
        return lo;
    };
    if (RUNS && len) {
        run = find_run(0, n_runs, start);
        run_last = find_run(run, n_runs, start + len - 1u);
        run_end = run_off[run + 1];
        run_delta = run_start[run] - run_off[run];
    }

    // presence from candidate slots [lo, hi): the matching ones are the rows of the final interval in their order --
    // ALL takes the mask bit of the first, OR any mask bit (src/fms_index.h:126-144); none = the empty interval
    auto slots_result = [&](uint32_t lo, uint32_t hi, bool m0, bool m1, bool m2, uint32_t ey) -> int {
        const bool a0 = m0 && lo == 0u && hi > 0u, a1 = m1 && lo <= 1u && hi > 1u, a2 = m2 && lo <= 2u && hi > 2u;
        const uint32_t k0 = (ey >> 26) & 1u, k1 = (ey >> 27) & 1u, k2 = (ey >> 28) & 1u;
        if (!(a0 || a1 || a2)) return -1;
        if (mode_all) return static_cast<int>(a0 ? k0 : (a1 ? k1 : k2));
        return static_cast<int>((a0 && k0) || (a1 && k1) || (a2 && k2));
    };
    // one strand: table lookup (+ verification).  state 0 = absent, 1 = alive (needs the stepping kernel), 2 = settled (res),
    // 3 = this entry settled BOTH strands (res, res_other) -- only asked for on the forward strand's lookup
    auto strand = [&](uint64_t cur, bool go, bool want_both, uint32_t& i, uint32_t& j, int& res, int& res_other) -> uint32_t {
        const uint32_t tidx = static_cast<uint32_t>(cur >> (2 * (k - T))) & tmask;
        uint4 u0 = make_uint4(0, 0, 0, 0), u1 = u0;
        uint32_t ex, ey, cand0 = 0, cand1 = 0;
        if (verify) {
            // the lookup kernel's 16-byte entries: four to a line, entry e in lane e >> 1 of the pair, register e & 1
            const uint4* p4 = ix.vtable + (tidx & ~3u) + sub * 2u;
            if (go) {
                u0 = p4[0];
                u1 = p4[1];
            }
            const uint32_t e = tidx & 3u;
            const uint4 v = (e & 1u) ? u1 : u0;
            const int src = pbase | static_cast<int>(e >> 1);
            ex = static_cast<uint32_t>(__shfl(static_cast<int>(v.x), src, 64));
            ey = static_cast<uint32_t>(__shfl(static_cast<int>(v.y), src, 64));
            cand0 = static_cast<uint32_t>(__shfl(static_cast<int>(v.z), src, 64));
            cand1 = static_cast<uint32_t>(__shfl(static_cast<int>(v.w), src, 64));
        } else {
            const uint4* p4 = reinterpret_cast<const uint4*>(ix.ptable + (tidx & ~7u)) + sub * 2u;
            if (go) {
                u0 = p4[0];
                u1 = p4[1];
            }
            // entry e of the 8-entry line sits in lane e >> 2 of the pair, slot e & 3 of its 32 bytes
            const uint32_t e = tidx & 7u;
            const uint4 v = (e & 2u) ? u1 : u0;
            const uint32_t xi = (e & 1u) ? v.z : v.x, xj = (e & 1u) ? v.w : v.y;
            const int src = pbase | static_cast<int>(e >> 2);
            ex = static_cast<uint32_t>(__shfl(static_cast<int>(xi), src, 64));
            ey = static_cast<uint32_t>(__shfl(static_cast<int>(xj), src, 64));
        }
        if (!go) return 0u;   // (results of a lane that sits this lookup out stay as they are)
        res = -1;
        res_other = -1;
        i = j = 0;
        if (verify && (ey >> 30) == 2u) {
            // double-stranded entry: slots [0, nf) are the occurrences of this key in the text (this strand), slots
            // [nf, nf + nr) its occurrences in the text's reverse complement -- i.e. the OTHER strand's candidates, stored as
            // the bases that precede the key there.  Both are compared with the same remaining bases of the query.
            if (sub == 0u) n_lookups++;
            // (such an entry never needs its interval, so its first word is the third candidate)
            const uint32_t nf = (ey >> 24) & 3u, nr = (ey >> 22) & 3u, total = nf + nr;
            const uint32_t want = static_cast<uint32_t>(cur & rmask);
            const bool m0 = cand0 == want, m1 = cand1 == want, m2 = ex == want;
            res = slots_result(0u, nf, m0, m1, m2, ey);
            if (want_both) {
                res_other = slots_result(nf, total, m0, m1, m2, ey);
                return 3u;
            }
            return nf == 0u ? 0u : 2u;
        }
        // (a candidate-table copy of an entry of 4 or 5 rows is a plain interval whose three-symbol signatures ride in the third word)
        const uint32_t ysig = (verify && (ey >> 30) == 0u && (cand0 >> 31)) ? cand0 : ey;
        decode_table_entry(ex, ysig, ix.ptable_packed, t >= 3, t >= 3 ? static_cast<uint32_t>(cur >> (2 * (t - 3))) & 0xFFu : 0u, i, j);
        if (sub == 0u) n_lookups++;
        if (i == j) return 0u;
        if (!(verify && (ey >> 30) == 1u)) return 1u;
        // one to three rows: candidates 0 and 1 came with the entry, the third is read on demand.  Candidates that match are
        // the rows of the final interval, in the same order (an equal prefix does not change the order of two suffixes):
        // ALL takes the mask bit of the first, OR any mask bit (src/fms_index.h:126-144).
        const uint32_t rows = (ey >> 24) & 3u;
        const uint32_t pos0 = cand0, pos1 = cand1, pos2 = rows > 2u ? ix.stable[tidx] : 0u;
        auto text_matches = [&](uint32_t pos) -> bool {
            if (pos < R) return false;
            const uint64_t p0 = static_cast<uint64_t>(pos - R), w = p0 >> 5;
            const uint32_t sh = static_cast<uint32_t>(p0 & 31u) * 2u;
            uint64_t tx = ix.text2[w] >> sh;
            if (sh) tx |= ix.text2[w + 1] << (64u - sh);
            return ((tx ^ cur) & rmask) == 0ull;
        };
        bool m0, m1, m2;
        if (ix.verify_bases) {   // the words hold the bases themselves: no text read
            const uint32_t want = static_cast<uint32_t>(cur & rmask);
            m0 = pos0 == want;
            m1 = rows > 1u && pos1 == want;
            m2 = rows > 2u && pos2 == want;
            if (sub == 0u && rows > 2u) n_vreads += 1u;
        } else {
            m0 = text_matches(pos0);
            m1 = rows > 1u && text_matches(pos1);
            m2 = rows > 2u && text_matches(pos2);
            if (sub == 0u) n_vreads += rows + (rows > 2u ? 1u : 0u);
        }
        res = slots_result(0u, 3u, m0, m1, m2, ey);
        return 2u;
    };

    for (uint32_t base = 0; base < len; base += 32u) {   // 32 k-mers per round, one per lane pair; whole waves stay in the loop
        const uint32_t ql = base + (lane >> 1);
        const bool valid = ql < len;
        const uint64_t q = start + ql;
        uint64_t f = 0ull;
        if (RUNS) {
            if (valid) {
                if (q >= run_end) {
                    run = find_run(run + 1, run_last + 1, q);
                    run_end = run_off[run + 1];
                    run_delta = run_start[run] - run_off[run];
                }
                f = window_kmer(kmers, q + run_delta, k);
            }
        } else if (valid) {
            f = kmers[q];
        }
        uint32_t i_f = 0, j_f = 0, i_r = 0, j_r = 0;
        int res_f = -1, res_r = -1, unused = -1;
        const uint32_t sf = strand(f, valid, true, i_f, j_f, res_f, res_r);
        // the reverse complement is searched when the forward strand does not decide (src/fms_index.h:283-299); a forward
        // strand that still needs steps (state 1) leaves that decision to the stepping kernel; state 3 brought both results
        const bool need_rc = valid && sf != 1u && sf != 3u && (strict || (mode_all && !orders ? res_f == -1 : res_f != 1));
        uint32_t sr = 0;
        if (__any(need_rc)) sr = strand(revcomp_2bit(f, k), need_rc, false, i_r, j_r, res_r, unused);
        const bool alive = valid && (sf == 1u || sr == 1u);
        if (valid && !alive && sub == 0u) {   // settled: the result leaves here
            n_final++;
            if (orders) {   // (verification is off for lookup: both strands absent)
                if (strict) {
                    out_orders[2 * q] = -1;
                    out_orders[2 * q + 1] = -1;
                } else {
                    out_orders[q] = -1;
                }
            } else if (strict) {
                out[q] = static_cast<uint8_t>(static_cast<uint32_t>(res_f + 1) | (static_cast<uint32_t>(res_r + 1) << 2));
            } else {
                out[q] = static_cast<uint8_t>(mode_all ? (res_f != -1 ? res_f : res_r) == 1 : (res_f == 1 || res_r == 1));
            }
        }
        const uint64_t m = __ballot(alive) & 0x5555555555555555ull;   // one bit per lane pair
        if (m) {
            if (alive && sub == 0u) {
                WorkItem w;
                w.f = f;
                // a settled forward 0 / 1 cannot ride along with a reverse complement that needs steps: the forward strand is redone
                const bool fwd_item = sf == 1u || res_f != -1;
                w.x = fwd_item ? i_f : j_r;
                w.y = fwd_item ? j_f : i_r;
                w.ql = ql;
                w.pad = 0;
                list[start + count + static_cast<uint32_t>(__popcll(m & pairs_below))] = w;
            }
            count += static_cast<uint32_t>(__popcll(m));
        }
    }
    if (lane == 0u && start < n_windows) list_count[wave] = count;
    if (COUNT) {
        uint32_t v[4] = {n_lookups, n_lookups, n_final, n_vreads};   // strand searches started = table lookups = lines read
        const int slot[4] = {3, 4, 6, 7};
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            uint32_t x = v[qq];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if ((threadIdx.x & 63u) == 0 && x) {
                atomicAdd(&counters[slot[qq]], static_cast<unsigned long long>(x));
                if (qq == 0) atomicAdd(&counters[1], static_cast<unsigned long long>(x));   // [1] counts the table lines with the blocks
            }
        }
    }
}

// 16 bytes that will not be wanted again (a table entry of one k-mer): a non-temporal load leaves the caches alone
// (tools/gather_bench entry: 51.5 against 49.5 G random lines/s)
__device__ __forceinline__ uint4 load16_once(const uint4* p) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(t.x, t.y, t.z, t.w);
}

// ---- phase 1 with candidate tables: ONE LANE PER K-MER ----------------------------------------------------------------
// With the 16-byte candidate entries a k-mer needs 16 bytes of the table, and a lane that issues ONE 16-byte load per
// lookup costs the memory path one request and one translation per line whether or not a neighbour shares the line
// (tools/gather_bench entry 69: 48 G lookups/s for 1, 2 or 4 lanes per entry -- the 17 G/s of one lane per line that
// pair_lookup_kernel's comment quotes was a lane reading its line with FOUR loads).  So here a wave looks 64 k-mers up
// per round instead of 32, which doubles the lines it keeps in flight, and everything that would make the whole wave
// wait a second time inside a round -- the reverse complement's lookup (6 % of the k-mers: every round has one), the
// third candidate's read -- is deferred: such k-mers go to a per-wave queue in LDS and are done 64 at a time in dense
// "slow" rounds.  A fast round is: k-mers in (one coalesced 512-B read), one table load per lane, decide.
//   fast round, per entry: double-stranded -> both strands settled; forward alive (needs steps) -> work item; forward
//   absent and the reverse complement wanted -> queue (RC); anything else that needs more (third candidate, text read, a
//   settled forward result that may have to be redone) -> queue (FULL: the slow round starts from the forward lookup).
// Only for candidate words that are the bases themselves (k - t <= 15; the launcher checks).
// WIDE: 32-byte entries whose double-stranded form lists up to SEVEN occurrences (fmsi_hip_index_enable_verify makes them when
// the HBM is there: 137 GB at t = 16) -- nearly every key of a random text is then settled by its one line.  A lane must not
// fetch its 32 bytes with two loads of its own (tools/gather_bench: the memory path pays per instruction and distinct line,
// 38 G/s instead of 48 G/s), so the two lanes of a pair fetch the even lane's entry together, then the odd lane's, 16 bytes
// each, and swap the halves they hold for each other: two instructions of 32 lines per 64 k-mers.
// The chunk's result bytes are collected in LDS and leave as 16 bytes per lane.
// Same contract as pair_lookup_kernel (work items, result bytes of settled k-mers, counters) but for the work lists: so few
// searches are left that the four waves of a workgroup share ONE list (list[first k-mer of the workgroup ..], its length
// in list_count[workgroup], items appended through a counter in LDS, ql relative to the workgroup's first k-mer), and the
// stepping kernel is launched with chunks four times as long -- a quarter of its workgroups, whose fixed cost (the
// tables into LDS) is most of what it has left to do.
template <bool COUNT, bool RUNS, bool WIDE>
__global__ __launch_bounds__(256) void lookup_lane_kernel(PairIndex ix, const uint64_t* __restrict__ kmers, uint64_t n_windows, int k, uint32_t flags,
                                                          uint32_t chunk, uint8_t* __restrict__ out, WorkItem* __restrict__ list,
                                                          uint32_t* __restrict__ list_count, unsigned long long* __restrict__ counters,
                                                          const uint64_t* __restrict__ run_start, const uint64_t* __restrict__ run_off, uint64_t n_runs) {
    constexpr uint32_t kMaxChunk = 4096u, kQueue = 128u;
    __shared__ uint64_t s_qf[4][kQueue];
    __shared__ uint32_t s_qi[4][kQueue];
    __shared__ __attribute__((aligned(16))) uint8_t s_res[4][kMaxChunk];
    __shared__ uint32_t s_items;   // work items of the workgroup so far
    if (threadIdx.x == 0) s_items = 0;
    __syncthreads();
    const uint32_t strict = flags & FMSI_HIP_STRICT;
    const uint32_t mode_all = flags & FMSI_HIP_MODE_ALL;
    const int T = static_cast<int>(ix.ptable_len);
    const int t = k - 1 - T;
    const uint32_t R = static_cast<uint32_t>(k - T);                       // bases left after the table
    const uint64_t rmask = R >= 32u ? ~0ull : ((1ull << (2u * R)) - 1ull);
    const uint32_t tmask = static_cast<uint32_t>((1ull << (2 * T)) - 1ull);
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6, sub = lane & 1u;
    const int peven = static_cast<int>(lane & ~1u);
    uint64_t* qf = s_qf[wib];
    uint32_t* qi = s_qi[wib];
    uint8_t* res = s_res[wib];
    // FMSI_HIP_OUT_BITS: the chunk's results leave bit-packed (1 bit per window, STRICT: 4).  The result bytes of the k-mers
    // that go on to the stepping kernel must then read as zero -- it ORs its answers into the packed words
    const bool out_bits = (flags & FMSI_HIP_OUT_BITS) != 0u;
    if (out_bits)
        for (uint32_t off = lane * 16u; off < kMaxChunk; off += 64u * 16u) *reinterpret_cast<uint4*>(res + off) = make_uint4(0, 0, 0, 0);
    uint32_t n_lookups = 0, n_final = 0, n_vreads = 0;
    const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const uint64_t start = wave * chunk;
    const uint32_t len = start < n_windows ? static_cast<uint32_t>(min(static_cast<uint64_t>(chunk), n_windows - start)) : 0u;
    const uint64_t group_start = static_cast<uint64_t>(blockIdx.x) * 4u * chunk;   // first k-mer of the workgroup
    const uint32_t ql_base = wib * chunk;
    uint32_t qhead = 0, qtail = 0; // wave-uniform: the queue holds [qhead, qtail), positions mod kQueue
    uint64_t run = 0, run_last = 0, run_end = 0, run_delta = 0;
    auto find_run = [&](uint64_t lo, uint64_t hi, uint64_t q) -> uint64_t {   // run_off[lo] <= q < run_off[hi]
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (run_off[mid] <= q) lo = mid;
            else hi = mid;
        }
        return lo;
    };
    if (RUNS && len) {
        run = find_run(0, n_runs, start);
        run_last = find_run(run, n_runs, start + len - 1u);
        run_end = run_off[run + 1];
        run_delta = run_start[run] - run_off[run];
    }
    // presence from the candidate slots [lo, hi): mm = slots equal to the query's remaining bases, kb = their mask bits.  The
    // matching ones are the rows of the final interval in their order -- ALL takes the mask bit of the first, OR any mask bit
    // (src/fms_index.h:126-144); none = the empty interval
    auto pick = [&](uint32_t mm, uint32_t kb, uint32_t lo, uint32_t hi) -> int {
        const uint32_t sel = mm & ((1u << hi) - 1u) & ~((1u << lo) - 1u);
        if (!sel) return -1;
        if (mode_all) return static_cast<int>((kb >> (__ffs(static_cast<int>(sel)) - 1)) & 1u);
        return static_cast<int>((sel & kb) != 0u);
    };
    // the table entries of 64 k-mers: one 16-byte load per lane -- WIDE: two, shared with the neighbour (see above); every lane
    // of the wave comes here, `go` says whose entry is wanted
    auto load_entry = [&](uint32_t tidx, bool go, uint4& lo, uint4& hi) {
        lo = hi = make_uint4(0, 0, 0, 0);
        if (!WIDE) {
            if (go) lo = load16_once(ix.vtable + tidx);
            return;
        }
        const uint32_t t_even = static_cast<uint32_t>(__shfl(static_cast<int>(tidx), peven, 64));
        const uint32_t t_odd = static_cast<uint32_t>(__shfl(static_cast<int>(tidx), peven | 1, 64));
        const bool g_even = __shfl(static_cast<int>(go), peven, 64) != 0, g_odd = __shfl(static_cast<int>(go), peven | 1, 64) != 0;
        uint4 a = make_uint4(0, 0, 0, 0), b = a;
        if (g_even) a = load16_once(ix.vtable + 2ull * t_even + sub);
        if (g_odd) b = load16_once(ix.vtable + 2ull * t_odd + sub);
        const uint4 give = sub ? a : b;   // the odd lane holds the even entry's second half, the even lane the odd entry's first
        uint4 got;
        got.x = static_cast<uint32_t>(__shfl_xor(static_cast<int>(give.x), 1, 64));
        got.y = static_cast<uint32_t>(__shfl_xor(static_cast<int>(give.y), 1, 64));
        got.z = static_cast<uint32_t>(__shfl_xor(static_cast<int>(give.z), 1, 64));
        got.w = static_cast<uint32_t>(__shfl_xor(static_cast<int>(give.w), 1, 64));
        lo = sub ? got : a;
        hi = sub ? b : got;
    };
    // what a strand's table entry says.  0 = absent, 1 = alive (needs the stepping kernel), 2 = settled (res), 3 = the
    // entry settled BOTH strands (res, res_other; only when want_both), 4 = needs a further read and `full` is false
    auto eval = [&](uint64_t cur, uint32_t tidx, const uint4& v, const uint4& vh, bool want_both, bool full, uint32_t& i, uint32_t& j, int& r,
                    int& r_other) -> uint32_t {
        const uint32_t ex = v.x, ey = v.y, cand0 = v.z, cand1 = v.w;
        r = -1;
        r_other = -1;
        i = j = 0;
        if ((ey >> 30) == 2u) {   // double-stranded entry (see pair_lookup_kernel)
            const uint32_t want = static_cast<uint32_t>(cur & rmask);
            uint32_t mm = (cand0 == want ? 1u : 0u) | (cand1 == want ? 2u : 0u) | (ex == want ? 4u : 0u);
            uint32_t nf, nr, kb;
            if (WIDE) {   // (tag, nf in bits 0-2, nr in bits 3-5, the seven mask bits from bit 8)
                nf = ey & 7u;
                nr = (ey >> 3) & 7u;
                kb = (ey >> 8) & 0x7Fu;
                mm |= (vh.x == want ? 8u : 0u) | (vh.y == want ? 16u : 0u) | (vh.z == want ? 32u : 0u) | (vh.w == want ? 64u : 0u);
            } else {
                nf = (ey >> 24) & 3u;
                nr = (ey >> 22) & 3u;
                kb = (ey >> 26) & 7u;
            }
            r = pick(mm, kb, 0u, nf);
            if (want_both) {
                r_other = pick(mm, kb, nf, nf + nr);
                return 3u;
            }
            return nf == 0u ? 0u : 2u;
        }
        decode_table_entry(ex, ((ey >> 30) == 0u && (cand0 >> 31)) ? cand0 : ey, 1u, t >= 3, t >= 3 ? static_cast<uint32_t>(cur >> (2 * (t - 3))) & 0xFFu : 0u, i, j);
        if (i == j) return 0u;
        if ((ey >> 30) != 1u) return 1u;
        const uint32_t rows = (ey >> 24) & 3u;
        if (!WIDE && !full && rows > 2u) return 4u;
        // (the launcher sends only k - t <= 15 here: the candidate words are the bases themselves)
        const uint32_t want = static_cast<uint32_t>(cur & rmask);
        uint32_t mm = (cand0 == want ? 1u : 0u) | (rows > 1u && cand1 == want ? 2u : 0u);
        if (rows > 2u) {
            if (WIDE) {   // the third candidate rides in the entry's second half
                mm |= vh.x == want ? 4u : 0u;
            } else {
                mm |= ix.stable[tidx] == want ? 4u : 0u;
                n_vreads += 1u;
            }
        }
        r = pick(mm, (ey >> 26) & 7u, 0u, 3u);
        return 2u;
    };
    auto encode = [&](int res_f, int res_r) -> uint8_t {
        if (strict) return static_cast<uint8_t>(static_cast<uint32_t>(res_f + 1) | (static_cast<uint32_t>(res_r + 1) << 2));
        return static_cast<uint8_t>(mode_all ? (res_f != -1 ? res_f : res_r) == 1 : (res_f == 1 || res_r == 1));
    };
    auto emit_items = [&](bool alive, uint64_t f, uint32_t x, uint32_t y, uint32_t ql) {
        const uint64_t m = __ballot(alive);
        if (m) {
            uint32_t at = 0;
            if (lane == 0u) at = atomicAdd(&s_items, static_cast<uint32_t>(__popcll(m)));
            at = static_cast<uint32_t>(__shfl(static_cast<int>(at), 0, 64));
            if (alive) {
                WorkItem w;
                w.f = f;
                w.x = x;
                w.y = y;
                w.ql = ql_base + ql;
                w.pad = 0;
                list[group_start + at + lanes_below(m)] = w;
            }
        }
    };
    // a dense round over the first `items` queue entries
    auto slow_round = [&](uint32_t items) {
        const bool active = lane < items;
        const uint32_t pos = (qhead + lane) & (kQueue - 1u);
        const uint64_t f = active ? qf[pos] : 0ull;
        const uint32_t info = active ? qi[pos] : 0u;
        const uint32_t ql = info & 0xFFFFu;
        const bool redo = (info >> 16) & 1u;
        uint32_t i_f = 0, j_f = 0, i_r = 0, j_r = 0, sf = 0;
        int res_f = -1, res_r = -1, unused = -1;
        if (__any(active && redo)) {
            const uint32_t tidx = static_cast<uint32_t>(f >> (2u * R)) & tmask;
            uint4 v, vh;
            load_entry(tidx, active && redo, v, vh);
            if (active && redo) {
                n_lookups++;
                sf = eval(f, tidx, v, vh, false, true, i_f, j_f, res_f, unused);
            }
        }
        const bool need_rc = active && sf != 1u && (strict || (mode_all ? res_f == -1 : res_f != 1));
        uint32_t sr = 0;
        if (__any(need_rc)) {
            const uint64_t rc = revcomp_2bit(f, k);
            const uint32_t tidx = static_cast<uint32_t>(rc >> (2u * R)) & tmask;
            uint4 v, vh;
            load_entry(tidx, need_rc, v, vh);
            if (need_rc) {
                n_lookups++;
                sr = eval(rc, tidx, v, vh, false, true, i_r, j_r, res_r, unused);
            }
        }
        const bool alive = active && (sf == 1u || sr == 1u);
        if (active && !alive) {
            n_final++;
            res[ql] = encode(res_f, res_r);
        }
        // a settled forward 0 / 1 cannot ride along with a reverse complement that needs steps: the forward strand is redone
        const bool fwd_item = sf == 1u || res_f != -1;
        emit_items(alive, f, fwd_item ? i_f : j_r, fwd_item ? j_f : i_r, ql);
        qhead += items;
    };

    for (uint32_t base = 0; base < len; base += 64u) {
        const uint32_t ql = base + lane;
        const bool valid = ql < len;
        const uint64_t q = start + ql;
        uint64_t f = 0ull;
        if (RUNS) {
            if (valid) {
                if (q >= run_end) {
                    run = find_run(run + 1, run_last + 1, q);
                    run_end = run_off[run + 1];
                    run_delta = run_start[run] - run_off[run];
                }
                f = window_kmer(kmers, q + run_delta, k);
            }
        } else if (valid) {
            f = __builtin_nontemporal_load(kmers + q);   // (read once)
        }
        const uint32_t tidx = static_cast<uint32_t>(f >> (2u * R)) & tmask;
        uint32_t i_f = 0, j_f = 0, sf = 0;
        int res_f = -1, res_r = -1;
        bool fin = false, alive = false, slow = false, redo = false;
        uint4 v, vh;
        load_entry(tidx, valid, v, vh);
        if (valid) {
            n_lookups++;
            sf = eval(f, tidx, v, vh, true, false, i_f, j_f, res_f, res_r);
            if (sf == 3u) {
                fin = true;
            } else if (sf == 1u) {
                alive = true;
            } else if (sf == 4u) {
                slow = redo = true;
            } else if (strict || (mode_all ? res_f == -1 : res_f != 1)) {   // the reverse complement is wanted (src/fms_index.h:283-299)
                slow = true;
                redo = res_f != -1;   // (the slow round needs the forward interval then)
            } else {
                fin = true;
            }
        }
        if (fin) {
            n_final++;
            res[ql] = encode(res_f, res_r);
        }
        emit_items(alive, f, i_f, j_f, ql);
        const uint64_t m = __ballot(slow);
        if (m) {
            if (slow) {
                const uint32_t pos = (qtail + lanes_below(m)) & (kQueue - 1u);
                qf[pos] = f;
                qi[pos] = ql | (redo ? 0x10000u : 0u);
            }
            qtail += static_cast<uint32_t>(__popcll(m));
            while (qtail - qhead >= 64u) slow_round(64u);
        }
    }
    if (qtail != qhead) slow_round(qtail - qhead);
    __syncthreads();   // (every wave of the workgroup comes here)
    if (threadIdx.x == 0 && group_start < n_windows) list_count[blockIdx.x] = s_items;
    if (len && out_bits) {
        flush_packed_results(res, len, strict != 0u, out, start, lane);
    } else if (len) {   // the chunk's result bytes leave in 16-byte pieces; those of the work items are the stepping kernel's to overlay
        uint8_t* dst = out + start;
        const bool aligned = (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;
        for (uint32_t off = lane * 16u; off < len; off += 64u * 16u) {
            if (aligned && off + 16u <= len) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(res + off), reinterpret_cast<u32x4*>(dst + off));
            } else {
                for (uint32_t b = off; b < min(len, off + 16u); ++b) dst[b] = res[b];
            }
        }
    }
    if (COUNT) {
        uint32_t v[4] = {n_lookups, n_lookups, n_final, n_vreads};   // strand searches started = table lookups = lines read
        const int slot[4] = {3, 4, 6, 7};
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            uint32_t x = v[qq];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if (lane == 0 && x) {
                atomicAdd(&counters[slot[qq]], static_cast<unsigned long long>(x));
                if (qq == 0) atomicAdd(&counters[1], static_cast<unsigned long long>(x));   // [1] counts the table lines with the blocks
            }
        }
    }
}

// ---- query kernel -----------------------------------------------------------------------------------------
enum : uint32_t { PS_IDLE = 0, PS_LOOKUP = 1, PS_STEP = 2, PS_PROBE = 3 };

__device__ __forceinline__ uint32_t pair_sum(uint32_t x) { return x + static_cast<uint32_t>(__shfl_xor(static_cast<int>(x), 1, 64)); }
__device__ __forceinline__ uint32_t bytes_sum(uint32_t x) { return (x & 0xFFu) + ((x >> 8) & 0xFFu) + ((x >> 16) & 0xFFu) + (x >> 24); }

// one half as loaded by a lane
struct Half {
    uint4 u0, u1, u2, u3;   // {c1hi0 c1hi1 c1lo0 c1lo1} {c2hi0 c2hi1 c2lo0 c2lo1} {mask0 mask1 lo01 lo23} {lo45 lo67 hi0123 hi4567}
};

// count m (0..7) of this half, or (sum) the four counts m = 4g .. 4g+3 of group g = m >> 2
__device__ __forceinline__ uint32_t half_count(const Half& H, uint32_t m, bool sum) {
    if (sum) {
        const uint32_t l0 = (m & 4u) ? H.u3.x : H.u2.z, l1 = (m & 4u) ? H.u3.y : H.u2.w;
        const uint32_t hb = (m & 4u) ? H.u3.w : H.u3.z;
        return (l0 & 0xFFFFu) + (l0 >> 16) + (l1 & 0xFFFFu) + (l1 >> 16) + (bytes_sum(hb) << 16);
    }
    const uint32_t lw = sel4(m >> 1, opaque(H.u2.z), opaque(H.u2.w), opaque(H.u3.x), opaque(H.u3.y));
    const uint32_t hw = (m & 4u) ? H.u3.w : H.u3.z;
    return ((lw >> (16u * (m & 1u))) & 0xFFFFu) | (((hw >> (8u * (m & 3u))) & 0xFFu) << 16);
}

// rows of this half matching pair p (or symbol c1 = p >> 2 when single): one bit per row
__device__ __forceinline__ void half_bits(const Half& H, uint32_t p, bool single, uint32_t& m0, uint32_t& m1) {
    const uint32_t x1h = (p & 8u) ? 0u : 0xFFFFFFFFu, x1l = (p & 4u) ? 0u : 0xFFFFFFFFu;
    const uint32_t x2h = (p & 2u) ? 0u : 0xFFFFFFFFu, x2l = (p & 1u) ? 0u : 0xFFFFFFFFu;
    const uint32_t all = single ? 0xFFFFFFFFu : 0u;   // single step: the c2 planes do not restrict
    m0 = (H.u0.x ^ x1h) & (H.u0.z ^ x1l) & ((H.u1.x ^ x2h) | all) & ((H.u1.z ^ x2l) | all);
    m1 = (H.u0.y ^ x1h) & (H.u0.w ^ x1l) & ((H.u1.y ^ x2h) | all) & ((H.u1.w ^ x2l) | all);
}
// ... among the half's first `o` rows (o may be <= 0 or >= 64)
__device__ __forceinline__ uint32_t bits_below(uint32_t m0, uint32_t m1, int o) { return __popc(m0 & prefix32(o)) + __popc(m1 & prefix32(o - 32)); }
// drops row `row` from my half's bits of block `blk` (rows that are stored as 'A' but stand for no symbol)
__device__ __forceinline__ void drop_row(uint32_t row, uint32_t blk, uint32_t sub, uint32_t& m0, uint32_t& m1) {
    if ((row >> kBlockRowsLog2) == blk && ((row >> 6) & 1u) == sub) {
        const uint32_t b = row & 63u;
        if (b < 32u) m0 &= ~(1u << b);
        else m1 &= ~(1u << (b - 32u));
    }
}
// first set bit of (r0, r1) -> 2 | mask bit at that row; 0 if none
__device__ __forceinline__ uint32_t first_hit(uint32_t r0, uint32_t r1, uint32_t k0, uint32_t k1) {
    if (r0) return 2u | ((k0 >> (__ffs(static_cast<int>(r0)) - 1)) & 1u);
    if (r1) return 2u | ((k1 >> (__ffs(static_cast<int>(r1)) - 1)) & 1u);
    return 0u;
}

struct PairCounters {
    uint32_t lf_steps = 0, blocks = 0, probes = 0, searches = 0, lookups = 0;
};

// LIST: phase 2 of the phased path -- the wave's work items are the searches pair_lookup_kernel left alive in the same
// chunk (list[chunk start ..], list_count[wave] of them); they enter the state machine after their table lookup and
// their results go straight to their k-mers' places.
template <bool COUNT, bool PTABLE, bool LIST>
__global__ __launch_bounds__(256) void query_pair_kernel(PairIndex ix, const uint64_t* __restrict__ kmers, uint64_t n_windows, int k,
                                                         uint32_t flags, uint32_t chunk, uint32_t n_super, uint8_t* __restrict__ out,
                                                         long long* __restrict__ out_orders, unsigned long long* __restrict__ counters,
                                                         const WorkItem* __restrict__ list, const uint32_t* __restrict__ list_count) {
    // LDS: C2[16] and the superblock table (every step needs three of their words; as global loads -- even cache hits --
    // they compete with the block reads for the memory pipeline), and per wave a ring of 128 k-mers and the chunk's
    // result bytes, so that k-mers arrive and results leave in whole cache lines instead of one request each.
    extern __shared__ uint32_t lds[];   // [0,16) = C2, [16, 16 + 16 * n_super) = sbt, then 4 x (ring 1 KB + results 2 KB)
    const uint32_t tbl_words = (16u + 16u * n_super + 1u + 1u) & ~1u;
    for (uint32_t w = threadIdx.x; w < tbl_words; w += blockDim.x)
        lds[w] = w < 16u ? ix.c2[w] : (w < 16u + 16u * n_super ? ix.sbt[w - 16u] : 0u);
    __syncthreads();
    const uint32_t mode_all = flags & FMSI_HIP_MODE_ALL;
    const uint32_t strict = flags & FMSI_HIP_STRICT;
    const uint32_t orders = flags & FMSI_HIP_ORDERS;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wib = threadIdx.x >> 6;          // wave in block
    constexpr uint32_t kRingWords = LIST ? 640u : 256u;   // per wave: 128 k-mers (+ LIST: their (i, j) and chunk-relative index)
    uint64_t* ring = reinterpret_cast<uint64_t*>(lds + tbl_words) + wib * (kRingWords / 2u);
    // the chunk's result bytes live here.  LIST: only when the launcher asks for it (overlay) -- with the candidate tables a few
    // per cent of the chunk are left, and their own byte stores cost less than the occupancy the overlay's LDS takes
    const bool overlay = LIST && (flags & kFlagOverlay);
    const uint32_t out_bytes = LIST ? (overlay ? ((chunk + 127u) & ~127u) : 0u) : 2048u;
    uint8_t* obuf = reinterpret_cast<uint8_t*>(lds + tbl_words + 4u * kRingWords) + wib * out_bytes;
    uint32_t* ring_ql = reinterpret_cast<uint32_t*>(ring + 256);   // LIST only
    const uint32_t sub = lane & 1u;                 // my half
    const uint32_t pbase = lane & ~1u;              // first lane of my pair
    const uint64_t pairs_below = (1ull << pbase) - 1ull;
    const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const uint64_t start = wave * chunk;                                               // my chunk: [start, start + len)
    const uint32_t len = start >= n_windows ? 0u
                         : (LIST ? list_count[wave] : static_cast<uint32_t>(min(static_cast<uint64_t>(chunk), n_windows - start)));
    // LIST with overlay: the chunk's result bytes -- the lookup kernel wrote the settled ones -- are taken into LDS, overlaid with
    // this wave's results and written back as whole lines (a chunk with nothing left to search is not touched)
    const uint32_t clen = start < n_windows ? static_cast<uint32_t>(min(static_cast<uint64_t>(chunk), n_windows - start)) : 0u;
    if (overlay && len && !orders) {
        const uint8_t* src = out + start;
        const bool aligned = (reinterpret_cast<uintptr_t>(src) & 15u) == 0;
        for (uint32_t off = lane * 16u; off < clen; off += 64u * 16u) {
            if (aligned && off + 16u <= clen) {
                *reinterpret_cast<uint4*>(obuf + off) = *reinterpret_cast<const uint4*>(src + off);
            } else {
                for (uint32_t b = off; b < min(clen, off + 16u); ++b) obuf[b] = src[b];
            }
        }
    }
    uint32_t next = 0, loaded = 0;                  // wave-uniform: handed out / landed in the ring (relative to start)
    uint64_t pend = 0, pend_ij = 0;                 // k-mer (+ LIST: record) batch in flight from global memory
    bool pend_valid = false;
    uint32_t pend_ql = 0;
    const int T = PTABLE ? static_cast<int>(ix.ptable_len) : 0;
    const int shift = static_cast<int>(sub * 64u);  // first row of my half within the block

    // both lanes of a pair carry the same state
    uint64_t f = 0, cur = 0;
    uint32_t q = 0, i = 0, j = 0, st = PS_IDLE, phase = 0, done = 0;
    int t = 0, res_f = -1;
    long long ord_f = -1;
    PairCounters cn;

    auto start_strand = [&](uint64_t kmer) {
        cur = kmer;
        if (COUNT) cn.searches++;
        if (PTABLE) {
            st = PS_LOOKUP;
        } else {
            i = 0;
            j = ix.n;
            t = k - 1;
            st = PS_STEP;
        }
    };
    auto finalize = [&](int res, long long ord) {
        bool second;
        if (strict) second = true;
        else if (orders) second = ord < 0;
        else if (mode_all) second = res == -1;
        else second = res != 1;
        if (phase == 0 && second) {
            res_f = res;
            ord_f = ord;
            phase = 1;
            start_strand(revcomp_2bit(f, k));
            return;
        }
        const bool was_second = phase == 1;
        const int rf = was_second ? res_f : res, rr = was_second ? res : -1;
        const long long of = was_second ? ord_f : ord, orr = was_second ? ord : -1;
        if (sub == 0) {
            if (orders) {
                const uint64_t qa = start + q;
                if (strict) {
                    out_orders[2 * qa] = of;
                    out_orders[2 * qa + 1] = orr;
                } else {
                    out_orders[qa] = was_second ? orr : of;
                }
            } else {
                const uint8_t byte = strict ? static_cast<uint8_t>((rf + 1) | ((rr + 1) << 2)) : static_cast<uint8_t>((was_second ? rr : rf) == 1);
                if (LIST && !overlay) {
                    const uint64_t w = start + q;
                    if (!(flags & FMSI_HIP_OUT_BITS)) {
                        out[w] = byte;
                    } else if (byte) {   // packed output: the lookup kernel wrote this window's bits as zero (flush_packed)
                        uint32_t* o32 = reinterpret_cast<uint32_t*>(out);
                        if (strict) atomicOr(o32 + (w >> 3), static_cast<uint32_t>(byte) << (4u * static_cast<uint32_t>(w & 7u)));
                        else atomicOr(o32 + (w >> 5), 1u << static_cast<uint32_t>(w & 31u));
                    }
                } else {
                    obuf[q] = byte;
                }
            }
        }
        st = PS_IDLE;
        done++;
    };

    for (;;) {
        // (0) the batch of k-mers requested last iteration has arrived: land it in the ring
        if (pend_valid) {
            ring[(loaded + lane) & 127u] = pend;
            if (LIST) {
                ring[128u + ((loaded + lane) & 127u)] = pend_ij;
                ring_ql[(loaded + lane) & 127u] = pend_ql;
            }
            loaded = min(len, loaded + 64u);
            pend_valid = false;
        }
        // (1) idle lane pairs draw the next k-mers of the chunk: ballot + prefix rank over a wave-uniform cursor
        {
            const bool want = st == PS_IDLE;
            const uint64_t m = __ballot(want) & 0x5555555555555555ull;   // one bit per lane pair
            if (m) {
                const uint32_t rank = static_cast<uint32_t>(__popcll(m & pairs_below));
                const uint32_t avail = loaded - next;
                if (want && rank < avail) {
                    const uint32_t slot = (next + rank) & 127u;
                    f = ring[slot];
                    phase = 0;
                    if (LIST) {   // the strand's table lookup is done: straight to the steps
                        const uint64_t ij = ring[128u + slot];
                        q = ring_ql[slot];
                        const uint32_t x = static_cast<uint32_t>(ij), y = static_cast<uint32_t>(ij >> 32);
                        phase = x > y ? 1u : 0u;       // (j, i): the forward strand died in the table, this is the reverse complement
                        i = min(x, y);
                        j = max(x, y);
                        cur = phase ? revcomp_2bit(f, k) : f;
                        res_f = -1;
                        ord_f = -1;
                        t = k - 1 - T;
                        st = t < 0 ? PS_PROBE : PS_STEP;
                    } else {
                        q = next + rank;
                        start_strand(f);
                    }
                }
                next += min(static_cast<uint32_t>(__popcll(m)), avail);
            }
        }
        // (2) keep the ring ahead of the cursor: one coalesced 512-byte read per 64 k-mers, consumed next iteration
        if (!pend_valid && loaded < len && loaded - next <= 64u) {
            const uint32_t idx = loaded + lane;
            if (LIST) {
                const uint64_t* w = reinterpret_cast<const uint64_t*>(list + start + (idx < len ? idx : 0u));
                pend = w[0];
                pend_ij = w[1];
                pend_ql = static_cast<uint32_t>(w[2]);
            } else {
                pend = idx < len ? kmers[start + idx] : 0ull;
            }
            pend_valid = true;
        }
        if (!__any(st != PS_IDLE)) {
            if (next >= len && !pend_valid) break;
            continue;
        }
        if (st != PS_IDLE) {
            const bool single = st == PS_STEP && ((t + 1) & 1);          // odd number of bases left: one single-base step first
            uint32_t p = 0;                                              // pair index c2 | c1 << 2 (single: c1 << 2)
            if (st == PS_STEP) {
                const uint32_t c1 = static_cast<uint32_t>(cur >> (2 * t)) & 3u;
                const uint32_t c2 = single ? 0u : (static_cast<uint32_t>(cur >> (2 * (t - 1))) & 3u);
                p = c2 | (c1 << 2);
            }
            const uint32_t bi = i >> kBlockRowsLog2, bj = (j - 1u) >> kBlockRowsLog2;
            // my 64-byte half of each boundary's block -- or my 32 bytes of the prefix-table line
            const uint4* pi4 = reinterpret_cast<const uint4*>(ix.pb + static_cast<size_t>(bi) * kPairBlockDwords) + sub * 4u;
            const uint4* pj4 = reinterpret_cast<const uint4*>(ix.pb + static_cast<size_t>(bj) * kPairBlockDwords) + sub * 4u;
            uint32_t tidx = 0;
            const bool lookup = PTABLE && st == PS_LOOKUP;
            if (lookup) {
                tidx = static_cast<uint32_t>(cur >> (2 * (k - T))) & static_cast<uint32_t>((1ull << (2 * T)) - 1ull);
                pi4 = reinterpret_cast<const uint4*>(ix.ptable + (tidx & ~7u)) + sub * 2u;
            }
            const bool two = st == PS_STEP && bj != bi;
            const uint4 z = make_uint4(0, 0, 0, 0);
            Half A, B;
            A.u2 = A.u3 = z;
            B.u0 = B.u1 = B.u2 = B.u3 = z;
            uint32_t sbi = 0, sbj = 0, c2base = 0;
            if (two) {
                B.u0 = pj4[0];
                B.u1 = pj4[1];
                B.u2 = pj4[2];
                B.u3 = pj4[3];
            }
            if (st == PS_PROBE) {   // explicit probe: the mask plane (128 rows) of the single-step block of row i
                A.u0 = A.u1 = z;
                A.u2 = ix.sblocks[(static_cast<size_t>(bi) << 2) + 3u];
            } else {
                A.u0 = pi4[0];
                A.u1 = pi4[1];
                if (!lookup) {
                    A.u2 = pi4[2];
                    A.u3 = pi4[3];
                }
            }
            if (st == PS_STEP) {   // C2 and the superblock table come from LDS
                // pair step -> count p; single step -> my half's share of the four counts with c1 = p >> 2 (summed below)
                const uint32_t se = single ? ((p & 12u) + 2u * sub) : p;
                const uint32_t* si = lds + 16u + ((i >> kSuperLog2) << 4) + se;
                sbi = si[0];
                if (single) sbi += si[1];
                sbj = sbi;
                c2base = lds[p];
            }
            // After the table an interval is a few rows: both boundaries almost always sit in ONE block, and then the
            // upper rank is the lower one plus the matching rows in between.  The second block's share of the work (its
            // superblock counts, match bits, cumulative count and the copy A -> B) is done only in the iterations in
            // which some lane of the wave really has two blocks.
            const bool wave_two = __any(two);
            if (wave_two) {
                if (st == PS_STEP) {
                    const uint32_t se = single ? ((p & 12u) + 2u * sub) : p;
                    const uint32_t* sj = lds + 16u + (((j - 1u) >> kSuperLog2) << 4) + se;
                    sbj = sj[0];
                    if (single) sbj += sj[1];
                }
                if (!two) B = A;
            }
            if (COUNT && sub == 0) cn.blocks += 1u + (two ? 1u : 0u);

            if (st == PS_STEP) {
                const int oi = static_cast<int>(i & (kBlockRows - 1u)) - shift, oj = static_cast<int>(j - (bj << kBlockRowsLog2)) - shift;
                uint32_t ma0, ma1, mb0, mb1;
                half_bits(A, p, single, ma0, ma1);
                const uint32_t below_i = bits_below(ma0, ma1, oi);
                uint32_t ri = pair_sum(below_i), rj;
                // cumulative counts live in the half of c1 = p >> 2: half (p >> 3), slot p & 7
                const int owner = static_cast<int>(pbase | (p >> 3));
                const uint32_t cnt_i = static_cast<uint32_t>(__shfl(static_cast<int>(half_count(A, p & 7u, single)), owner, 64));
                uint32_t cnt_j = cnt_i;
                if (wave_two) {
                    half_bits(B, p, single, mb0, mb1);
                    rj = pair_sum(bits_below(mb0, mb1, oj));
                    cnt_j = static_cast<uint32_t>(__shfl(static_cast<int>(half_count(B, p & 7u, single)), owner, 64));
                } else {
                    mb0 = ma0;
                    mb1 = ma1;
                    rj = ri + pair_sum(bits_below(ma0, ma1, oj) - below_i);
                }
                // The last step of a search also answers it when the block(s) at hand cover the whole interval: the
                // rows of [i, j) that match the pair are exactly the rows of the new interval, in order, and the block's
                // mask plane holds mask[LF2(row)] -- infer_presence() (src/fms_index.h:126-144) without another read.
                int folded = -2;   // -2: not folded
                if (!single && t == 1 && !orders && bj - bi <= 1u) {
                    const uint32_t same = two ? 0u : 0xFFFFFFFFu;
                    uint32_t ra0 = ma0 & ~prefix32(oi) & (prefix32(oj) | ~same), ra1 = ma1 & ~prefix32(oi - 32) & (prefix32(oj - 32) | ~same);
                    uint32_t rb0 = two ? (mb0 & prefix32(oj)) : 0u, rb1 = two ? (mb1 & prefix32(oj - 32)) : 0u;
                    drop_row(ix.r0, bi, sub, ra0, ra1);
                    drop_row(ix.r1, bi, sub, ra0, ra1);
                    drop_row(ix.r0, bj, sub, rb0, rb1);
                    drop_row(ix.r1, bj, sub, rb0, rb1);
                    if (mode_all) {   // mask bit of the interval's first row
                        const uint32_t mine = first_hit(ra0, ra1, A.u2.x, A.u2.y), mineb = first_hit(rb0, rb1, B.u2.x, B.u2.y);
                        const uint32_t packed = mine | (mineb << 2);
                        const uint32_t other = static_cast<uint32_t>(__shfl_xor(static_cast<int>(packed), 1, 64));
                        const uint32_t lo = sub == 0u ? packed : other, hi = sub == 0u ? other : packed;   // rows 0-63, rows 64-127
                        const uint32_t h = (lo & 2u) ? lo : ((hi & 2u) ? hi : (((lo >> 2) & 2u) ? (lo >> 2) : (hi >> 2)));
                        if (h & 2u) folded = static_cast<int>(h & 1u);
                    } else {          // any ON row in the interval
                        const uint32_t any = (ra0 & A.u2.x) | (ra1 & A.u2.y) | (rb0 & B.u2.x) | (rb1 & B.u2.y);
                        const uint32_t have = ra0 | ra1 | rb0 | rb1;
                        const uint32_t packed = (any ? 1u : 0u) | (have ? 2u : 0u);
                        const uint32_t both = packed | static_cast<uint32_t>(__shfl_xor(static_cast<int>(packed), 1, 64));
                        if (both & 2u) folded = static_cast<int>(both & 1u);
                    }
                }
                ri += cnt_i;
                rj += cnt_j;
                if (single) {
                    ri += pair_sum(sbi);
                    rj += pair_sum(sbj);
                    const uint32_t y = p >> 2;
                    if (y == 0u) {   // the '$' slot is stored as c1 = 'A'
                        ri -= (i > ix.r0);
                        rj -= (j > ix.r0);
                    }
                    const uint32_t base = sel4(y, ix.C0, ix.C1, ix.C2_, ix.C3);
                    i = base + ri;
                    j = base + rj;
                    t -= 1;
                    if (COUNT && sub == 0) cn.lf_steps += 1;
                } else {
                    ri += sbi;
                    rj += sbj;
                    if (p == 0u) {   // '$' slot stored as (A, A)
                        ri -= (i > ix.r0);
                        rj -= (j > ix.r0);
                    }
                    if (p == ix.p1) {   // the row of suffix 1 has no c2: stored as (A, T[0])
                        ri -= (i > ix.r1);
                        rj -= (j > ix.r1);
                    }
                    i = c2base + ri;
                    j = c2base + rj;
                    t -= 2;
                    if (COUNT && sub == 0) cn.lf_steps += 2;
                }
                if (i == j) {
                    finalize(-1, -1);
                } else if (t < 0) {
                    if (folded >= 0) {
                        if (COUNT && sub == 0) cn.probes++;
                        finalize(folded, -1);
                    } else {
                        st = PS_PROBE;
                    }
                }
            } else if (st == PS_PROBE) {
                if (COUNT && sub == 0) cn.probes++;
                // Only what the last step could not answer itself comes here: lookup (the order needs the landing row's
                // rank), intervals spanning more than two blocks, searches that end on a single-base step or in the table.
                const uint32_t o = i & (kBlockRows - 1u);
                int res;
                long long ord = -1;
                if (mode_all && !orders) {
                    const uint32_t mw = sel4(o >> 5, opaque(A.u2.x), opaque(A.u2.y), opaque(A.u2.z), opaque(A.u2.w));
                    res = static_cast<int>((mw >> (o & 31u)) & 1u);
                } else {
                    const uint32_t oj = j - (bj << kBlockRowsLog2);
                    const uint32_t below_i = popc_prefix128(A.u2, o);
                    uint32_t ones;
                    if (bj == bi) {
                        ones = popc_prefix128(A.u2, oj) - below_i;
                    } else {   // interval spans blocks (long repeats): rank difference through the side array
                        const uint4 mj = ix.sblocks[(static_cast<size_t>(bj) << 2) + 3u];
                        ones = (ix.mrank[bj] + popc_prefix128(mj, oj)) - (ix.mrank[bi] + below_i);
                    }
                    res = ones > 0u;
                    if (orders && res) ord = static_cast<long long>(ix.mrank[bi] + below_i);
                }
                finalize(res, ord);
            } else {   // PS_LOOKUP: entry e of the 8-entry line sits in lane e >> 2 of the pair, slot e & 3 of its 32 bytes
                if (COUNT && sub == 0) cn.lookups++;
                const uint32_t e = tidx & 7u;
                const uint4 v = (e & 2u) ? A.u1 : A.u0;
                const uint32_t xi = (e & 1u) ? v.z : v.x, xj = (e & 1u) ? v.w : v.y;
                const int src = static_cast<int>(pbase | (e >> 2));
                t = k - 1 - T;
                decode_table_entry(static_cast<uint32_t>(__shfl(static_cast<int>(xi), src, 64)), static_cast<uint32_t>(__shfl(static_cast<int>(xj), src, 64)),
                                   ix.ptable_packed, t >= 3, t >= 3 ? static_cast<uint32_t>(cur >> (2 * (t - 3))) & 0xFFu : 0u, i, j);
                if (i == j) finalize(-1, -1);
                else st = t < 0 ? PS_PROBE : PS_STEP;
            }
        }
    }
    if ((!LIST || overlay) && !orders && len) {   // the chunk's result bytes leave in 16-byte pieces (whole lines), not one request per k-mer
        uint8_t* dst = out + start;
        const uint32_t olen = LIST ? clen : len;
        const bool aligned = (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;
        for (uint32_t off = lane * 16u; off < olen; off += 64u * 16u) {
            if (aligned && off + 16u <= olen) {
                *reinterpret_cast<uint4*>(dst + off) = *reinterpret_cast<const uint4*>(obuf + off);
            } else {
                for (uint32_t b = off; b < min(olen, off + 16u); ++b) dst[b] = obuf[b];
            }
        }
    }
    if (COUNT) {
        uint32_t v[6] = {cn.lf_steps, cn.blocks, cn.probes, sub == 0 ? cn.searches : 0u, cn.lookups, sub == 0 ? done : 0u};
        const int slot[6] = {0, 1, 2, 3, 4, 6};
#pragma unroll
        for (int qq = 0; qq < 6; ++qq) {
            uint32_t x = v[qq];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if (lane == 0 && x) atomicAdd(&counters[slot[qq]], static_cast<unsigned long long>(x));
        }
    }
}

}  // namespace

PairIndex make_pair_index(const fmsi_hip_index* idx) {
    PairIndex p;
    p.pb = static_cast<const uint32_t*>(idx->d_pblocks);
    p.sbt = static_cast<const uint32_t*>(idx->d_sbt);
    p.c2 = static_cast<const uint32_t*>(idx->d_c2);
    p.mrank = idx->dev.mrank;
    p.sblocks = idx->dev.blocks;
    p.ptable = idx->dev.ptable;
    p.ptable_len = idx->dev.ptable_len;
    p.ptable_packed = idx->dev.ptable_packed;
    p.vtable = static_cast<const uint4*>(idx->d_vtable);
    p.stable = static_cast<const uint32_t*>(idx->d_stable);
    p.vwide = idx->vwide ? 1u : 0u;
    p.verify_k = idx->k;
    p.verify_bases = (idx->k - idx->prefix_len) <= 15 ? 1u : 0u;
    p.text2 = static_cast<const uint64_t*>(idx->d_text2);
    p.n = idx->dev.n;
    p.r0 = idx->dev.dollar;
    p.r1 = idx->pair_r1;
    p.p1 = idx->pair_p1;
    p.C0 = idx->dev.C0; p.C1 = idx->dev.C1; p.C2_ = idx->dev.C2; p.C3 = idx->dev.C3;
    return p;
}

namespace {

// the phased launch that writes FMSI_HIP_OUT_BITS output itself: the one-lane lookup kernel + the stepping kernel's ORs
bool lane_lookup_applies(const fmsi_hip_index* idx, int k, uint32_t flags) {
    static const bool force_pairs = std::getenv("FMSI_HIP_LOOKUP_PAIRS") != nullptr;   // A/B switch for measurements
    const int rest = k - idx->prefix_len;
    // candidate tables in play (the condition pair_lookup_kernel calls `verify`) whose words are the bases themselves, k - t <= 15:
    // with text positions every candidate is a read of its own, which the one-lane kernel defers, and the lane pairs are the
    // faster form -- measured at n = 4.7 M, t = 12: 26.3 vs 24.6 G k-mers/s
    return idx->d_vtable && idx->dev.ptable_packed && !(flags & FMSI_HIP_ORDERS) && rest >= 1 && rest <= 15 && (!force_pairs || idx->vwide);
}

template <bool COUNT, bool RUNS>
int launch_phased_t(fmsi_hip_index* idx, const uint64_t* d_kmers, uint64_t n, int k, uint32_t flags, uint8_t* o8, long long* o64,
                    unsigned long long* d_counters, hipStream_t s, const uint64_t* h_runs, uint64_t n_runs) {
    // both kernels cut the batch into the same chunks, one per wave.  Long chunks: the stepping kernel has little left to
    // do per chunk, and its fixed cost per workgroup (the tables into LDS) should be paid seldom
    // (4096 with the candidate tables, which leave it a few per cent of the searches; 1024 -- more workgroups per CU --
    // when it has to do the bulk of them)
    uint64_t chunk = n / (256ull * 4 * 2);
    chunk = std::min<uint64_t>(std::max<uint64_t>(chunk, 128), idx->d_vtable && !(flags & FMSI_HIP_ORDERS) ? 4096 : 1024) & ~127ull;
    const uint64_t waves = (n + chunk - 1) / chunk;
    const uint32_t blocks = static_cast<uint32_t>((waves + 3) / 4);
    // per wave a private work list with room for its whole chunk, and its length
    const size_t list_bytes = (n * sizeof(WorkItem) + 63) / 64 * 64;
    // (RUNS: and behind them the run table -- n_runs starts, n_runs + 1 offsets)
    const size_t count_bytes = (waves * 4 + 63) / 64 * 64;
    int rc = ensure_cap(&idx->d_tasks, &idx->d_tasks_cap, list_bytes + count_bytes + (RUNS ? (2 * n_runs + 1) * 8 : 0));
    if (rc) return rc;
    auto* d_list = static_cast<WorkItem*>(idx->d_tasks);
    auto* d_count = reinterpret_cast<uint32_t*>(static_cast<char*>(idx->d_tasks) + list_bytes);
    auto* d_runs = reinterpret_cast<uint64_t*>(static_cast<char*>(idx->d_tasks) + list_bytes + count_bytes);
    if (RUNS) {
        FMSI_HIP_TRY(hipMemcpyAsync(d_runs, h_runs, (2 * n_runs + 1) * 8, hipMemcpyHostToDevice, s));   // (h_runs: a slot of the handle's page-locked ring)
    }
    const PairIndex pix = make_pair_index(idx);
    // candidate tables in play: one lane per k-mer
    const bool lane_lookup = k == pix.verify_k && lane_lookup_applies(idx, k, flags);
    if (lane_lookup && pix.vwide)
        hipLaunchKernelGGL((lookup_lane_kernel<COUNT, RUNS, true>), dim3(blocks), dim3(256), 0, s, pix, d_kmers, n, k, flags, static_cast<uint32_t>(chunk), o8,
                           d_list, d_count, d_counters, d_runs, d_runs + n_runs, n_runs);
    else if (lane_lookup)
        hipLaunchKernelGGL((lookup_lane_kernel<COUNT, RUNS, false>), dim3(blocks), dim3(256), 0, s, pix, d_kmers, n, k, flags, static_cast<uint32_t>(chunk), o8,
                           d_list, d_count, d_counters, d_runs, d_runs + n_runs, n_runs);
    else
        hipLaunchKernelGGL((pair_lookup_kernel<COUNT, RUNS>), dim3(blocks), dim3(256), 0, s, pix, d_kmers, n, k, flags, static_cast<uint32_t>(chunk), o8,
                           o64, d_list, d_count, d_counters, d_runs, d_runs + n_runs, n_runs);
    FMSI_HIP_TRY(hipGetLastError());
    const uint32_t n_super = static_cast<uint32_t>(((idx->n_blocks - 1) >> kSuperBlocksLog2) + 1);
    // (the result overlay in LDS only when the stepping kernel has the bulk of the chunk to answer)
    const bool overlay = !lane_lookup && !(flags & FMSI_HIP_ORDERS);
    const size_t lds_bytes = (((16 + 16 * static_cast<size_t>(n_super) + 2) & ~size_t(1)) + 4 * 640 + (overlay ? static_cast<size_t>(chunk) : 0)) * 4;
    if (lds_bytes > 64 * 1024)   // 160 KB of LDS per CU on gfx950; more than 64 KB per workgroup has to be asked for
        FMSI_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&query_pair_kernel<COUNT, true, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes)));
    // (after lookup_lane_kernel the work lists are per lookup workgroup: four chunks each)
    const uint32_t step_chunk = static_cast<uint32_t>(lane_lookup ? 4 * chunk : chunk);
    const uint32_t step_blocks = lane_lookup ? (blocks + 3) / 4 : blocks;
    hipLaunchKernelGGL((query_pair_kernel<COUNT, true, true>), dim3(step_blocks), dim3(256), lds_bytes, s, pix, d_kmers, n, k,
                       flags | (overlay ? kFlagOverlay : 0u), step_chunk, n_super, o8, o64, d_counters, d_list, d_count);
    FMSI_HIP_TRY(hipGetLastError());
    return 0;
}

template <bool COUNT, bool PTABLE>
void launch_pair_t(fmsi_hip_index* idx, const uint64_t* d_kmers, uint64_t n, int k, uint32_t flags, uint8_t* o8, long long* o64,
                   unsigned long long* d_counters, hipStream_t s) {
    // 32 k-mers in flight per wave: chunks are half of the single-lane kernels'
    uint64_t chunk = n / (256ull * 32 * 8 * 2);
    chunk = std::min<uint64_t>(std::max<uint64_t>(chunk, 128), 2048);
    const uint64_t waves = (n + chunk - 1) / chunk;
    const uint32_t blocks = static_cast<uint32_t>((waves + 3) / 4);
    const uint32_t n_super = static_cast<uint32_t>(((idx->n_blocks - 1) >> kSuperBlocksLog2) + 1);
    // tables (<= 33 KB for any n < 2^32) + per wave: 1 KB k-mer ring + 2 KB result bytes (chunk <= 2048)
    const size_t lds_bytes = (((16 + 16 * static_cast<size_t>(n_super) + 2) & ~size_t(1)) + 4 * 256 + 4 * 512) * 4;
    hipLaunchKernelGGL((query_pair_kernel<COUNT, PTABLE, false>), dim3(blocks), dim3(256), lds_bytes, s, make_pair_index(idx), d_kmers, n, k, flags,
                       static_cast<uint32_t>(chunk), n_super, o8, o64, d_counters, nullptr, nullptr);
}

}  // namespace

bool phased_packs_bits(const fmsi_hip_index* idx, int k, uint32_t flags) {
    return idx->d_pblocks && idx->d_ptable && k == idx->k && lane_lookup_applies(idx, k, flags);
}

int launch_phased_query(fmsi_hip_index* idx, const uint64_t* d_kmers, uint64_t n, int k, uint32_t flags, uint8_t* o8, long long* o64,
                        unsigned long long* d_counters, hipStream_t s, const uint64_t* h_runs, uint64_t n_runs) {
    if (h_runs)
        return d_counters ? launch_phased_t<true, true>(idx, d_kmers, n, k, flags, o8, o64, d_counters, s, h_runs, n_runs)
                          : launch_phased_t<false, true>(idx, d_kmers, n, k, flags, o8, o64, d_counters, s, h_runs, n_runs);
    return d_counters ? launch_phased_t<true, false>(idx, d_kmers, n, k, flags, o8, o64, d_counters, s, nullptr, 0)
                      : launch_phased_t<false, false>(idx, d_kmers, n, k, flags, o8, o64, d_counters, s, nullptr, 0);
}

void launch_pair_query(fmsi_hip_index* idx, const uint64_t* d_kmers, uint64_t n, int k, uint32_t flags, bool use_table, uint8_t* o8,
                       long long* o64, unsigned long long* d_counters, hipStream_t s) {
    if (use_table) {
        if (d_counters) launch_pair_t<true, true>(idx, d_kmers, n, k, flags, o8, o64, d_counters, s);
        else launch_pair_t<false, true>(idx, d_kmers, n, k, flags, o8, o64, d_counters, s);
    } else {
        if (d_counters) launch_pair_t<true, false>(idx, d_kmers, n, k, flags, o8, o64, d_counters, s);
        else launch_pair_t<false, false>(idx, d_kmers, n, k, flags, o8, o64, d_counters, s);
    }
}

}  // namespace fmsi

using namespace fmsi;

extern "C" int fmsi_hip_index_enable_pair_layout(fmsi_hip_index* idx) {
    ::fmsi::ScratchScope scratch_scope_(idx ? idx->device : -1);   // (devpool.h: the outermost builder returns the block cache to the driver)
    if (!idx) return FMSI_HIP_EINVAL;
    FMSI_HIP_NARROW_ONLY(idx, "this accelerator");
    if (idx->d_pblocks) return 0;
    TraceRange trace_range("fmsi: build pair layout");
    FMSI_HIP_ON_DEVICE(idx->device);
    PhaseClock clock{"pair"};
    const uint64_t n_blocks = idx->n_blocks;
    const uint64_t n_super = ((n_blocks - 1) >> kSuperBlocksLog2) + 1;
    void *pb = nullptr, *counts = nullptr, *scanned = nullptr, *sbt = nullptr, *c2 = nullptr, *spec = nullptr, *temp = nullptr;
    auto cleanup = [&]() {
        for (void* p : {counts, scanned, spec, temp})
            if (p) (void)dev_free(p);
    };
    auto fail = [&](int code, const char* msg) {
        cleanup();
        for (void* p : {pb, sbt, c2})
            if (p) (void)dev_free(p);
        set_last_error(msg);
        return code;
    };
    const size_t pb_bytes = n_blocks * kPairBlockDwords * 4;
    if (dev_malloc(&pb, pb_bytes) != hipSuccess || dev_malloc(&counts, (n_blocks + 1) * sizeof(U16)) != hipSuccess ||
        dev_malloc(&scanned, (n_blocks + 1) * sizeof(U16)) != hipSuccess || dev_malloc(&sbt, n_super * 64 + 64) != hipSuccess ||
        dev_malloc(&c2, 64) != hipSuccess || dev_malloc(&spec, 8) != hipSuccess)
        return fail(FMSI_HIP_ENOMEM, "hipMalloc failed for the pair layout");
    if (hipMemset(pb, 0, pb_bytes) != hipSuccess || hipMemset(counts, 0, (n_blocks + 1) * sizeof(U16)) != hipSuccess ||
        hipMemset(spec, 0xFF, 8) != hipSuccess || hipMemset(sbt, 0, n_super * 64 + 64) != hipSuccess)
        return fail(FMSI_HIP_EHIP, "hipMemset failed for the pair layout");
    const uint32_t grid = static_cast<uint32_t>((idx->n + 255) / 256);
    hipLaunchKernelGGL(pair_emit_kernel, dim3(grid), dim3(256), 0, 0, idx->dev, static_cast<uint32_t*>(pb), static_cast<U16*>(counts),
                       static_cast<uint32_t*>(spec));
    size_t tb = 0;
    U16 zero{};
    hipError_t e1 = hipGetLastError();   // pair_emit_kernel's launch
    if (e1 == hipSuccess)
        e1 = rocprim::exclusive_scan(nullptr, tb, static_cast<U16*>(counts), static_cast<U16*>(scanned), zero, static_cast<size_t>(n_blocks + 1),
                                     U16Plus());
    if (e1 == hipSuccess) e1 = dev_malloc(&temp, tb);
    if (e1 == hipSuccess)
        e1 = rocprim::exclusive_scan(temp, tb, static_cast<U16*>(counts), static_cast<U16*>(scanned), zero, static_cast<size_t>(n_blocks + 1),
                                     U16Plus());
    if (e1 != hipSuccess) {
        const std::string msg = std::string("pair layout: prefix sums failed: ") + hipGetErrorString(e1);
        return fail(FMSI_HIP_EHIP, msg.c_str());
    }
    hipLaunchKernelGGL(pair_counts_kernel, dim3(static_cast<uint32_t>((n_blocks + 255) / 256)), dim3(256), 0, 0, static_cast<const U16*>(scanned),
                       n_blocks, static_cast<uint32_t*>(pb), static_cast<uint32_t*>(sbt));
    U16 totals;
    uint32_t specials[2] = {0, 0}, block0[16];
    if (hipMemcpy(&totals, static_cast<U16*>(scanned) + n_blocks, sizeof(U16), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(specials, spec, 8, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(block0, idx->d_blocks, 64, hipMemcpyDeviceToHost) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
        return fail(FMSI_HIP_EHIP, "pair layout: device work failed");
    // C2[c2 | c1 << 2] = rows whose suffix is smaller than any suffix starting with "c2 c1":
    //   C[c2] + [last text symbol == c2] + sum over y' < c1 of #(text positions with T[p] = c2, T[p+1] = y')
    uint32_t pairs[16];
    for (int p = 0; p < 16; ++p) pairs[p] = totals.v[p];
    pairs[0] -= 1;                                                // the '$' slot, stored as (A, A)
    uint32_t r1 = 0xFFFFFFFFu, p1 = 0xFFFFFFFFu;
    if (specials[0] != 0xFFFFFFFFu) {
        r1 = specials[0];
        p1 = specials[1] << 2;                                    // stored as (A, T[0])
        pairs[p1] -= 1;
    }
    const uint32_t last_sym = ((block0[4] & 1u) << 1) | (block0[8] & 1u);   // c1 of row 0 (the empty suffix) = T[N-1]
    const uint32_t C[4] = {idx->dev.C0, idx->dev.C1, idx->dev.C2, idx->dev.C3};
    uint32_t c2h[16];
    for (uint32_t a = 0; a < 4; ++a) {
        uint32_t run = C[a] + (idx->n > 1 && last_sym == a ? 1u : 0u);
        for (uint32_t y = 0; y < 4; ++y) {
            c2h[a | (y << 2)] = run;
            run += pairs[a | (y << 2)];
        }
    }
    if (hipMemcpy(c2, c2h, 64, hipMemcpyHostToDevice) != hipSuccess) return fail(FMSI_HIP_EHIP, "pair layout: upload failed");
    cleanup();
    idx->d_pblocks = pb;
    idx->d_sbt = sbt;
    idx->d_c2 = c2;
    idx->pair_r1 = r1;
    idx->pair_p1 = p1;
    idx->device_bytes += pb_bytes + n_super * 64 + 64;
    clock.mark("pair layout");
    return 0;
}

// ---- text verification (fmsi_hip_index_enable_verify) ----------------------------------------------------------------
namespace fmsi {
namespace {

// letters (any case) -> 2-bit packed text, one thread per 32-base word; zero past the end
__global__ __launch_bounds__(256) void chars_to_text2_kernel(const char* __restrict__ chars, uint64_t len, uint64_t n_words, uint64_t* __restrict__ text2) {
    const uint64_t w = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint64_t v = 0;
    for (uint32_t b = 0; b < 32u; ++b) {
        const uint64_t p = w * 32u + b;
        if (p >= len) break;
        const uint32_t ch = static_cast<uint8_t>(chars[p]) | 0x20u;   // lower case
        const uint64_t c = ch == 'a' ? 0u : (ch == 'c' ? 1u : (ch == 'g' ? 2u : 3u));
        v |= c << (2u * b);
    }
    text2[w] = v;
}

// The lookup kernel's table (16-byte entries + a third candidate per entry), from the 8-byte table, the suffix array of
// the LF walk and the text.  Entry K (a t-mer):
//   * double-stranded (tag 2), bases mode only: when K occurs nf times in the text and its reverse complement nr times,
//     nf + nr <= 3, the entry lists all of them -- forward candidates = the R bases in front of each occurrence of K,
//     reverse candidates = the reverse complement of the R bases BEHIND each occurrence of rc(K), i.e. the bases in front
//     of K on the other strand -- with the mask bit of the k-mer each would complete.  One lookup then settles a k-mer
//     ending in K on both strands (nf = nr = 0: absent on both).
//   * otherwise a copy of the 8-byte entry, with the forward candidates and mask bits when it has 1..3 rows.
__global__ __launch_bounds__(256) void verify_fill_kernel(const uint2* __restrict__ tab, uint64_t count, const uint32_t* __restrict__ sa,
                                                          const char* __restrict__ chars, uint64_t text_len, uint32_t T, uint32_t R,
                                                          uint32_t bases_mode, uint32_t wide, uint4* __restrict__ vtable, uint32_t* __restrict__ stable) {
    const uint32_t max_total = wide ? 7u : 3u;   // occurrences a double-stranded entry can list
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    auto code_of = [](char c) -> uint32_t {
        const uint32_t ch = static_cast<uint8_t>(c) | 0x20u;
        return ch == 'a' ? 0u : (ch == 'c' ? 1u : (ch == 'g' ? 2u : 3u));
    };
    auto rows_of = [](uint2 e) -> uint32_t { return (e.y >> 30) == 1u ? ((e.y >> 24) & 3u) : e.y; };   // 8-byte format: tag 0 = length
    for (uint64_t idx = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < count; idx += stride) {
        uint2 e = tab[idx];
        // (the 8-byte table's entries of 4 and 5 rows -- tags 2 and 3, three-symbol signatures -- are plain intervals here: in this table tag 2 is
        // the double-stranded entry)
        auto plain = [](uint2 t) -> uint2 { return (t.y >> 30) >= 2u ? make_uint2(t.x, 2u + (t.y >> 30)) : t; };
        const uint32_t signed_y = e.y;   // (... and their signatures ride where such an entry has no candidate: the third word, bit 31 set)
        e = plain(e);
        uint32_t cand[7] = {0, 0, 0, 0, 0, 0, 0};
        const uint32_t nf = rows_of(e);
        bool done = false;
        if (bases_mode && nf <= max_total) {
            const uint2 er = plain(tab[revcomp_2bit(idx, static_cast<int>(T))]);
            const uint32_t nr = rows_of(er);
            if (nf + nr <= max_total) {
                // narrow: nf in bits 24-25, nr in bits 22-23, mask bits from bit 26; wide: nf bits 0-2, nr bits 3-5, mask bits from bit 8
                const uint32_t mbit = wide ? 8u : 26u;
                uint32_t y = (2u << 30) | (wide ? (nf | (nr << 3)) : ((nf << 24) | (nr << 22))), slot = 0;
                for (uint32_t m = 0; m < nf; ++m, ++slot) {            // this strand: the R bases in front of the occurrence
                    const uint32_t pos = sa[e.x + m];
                    uint32_t v = 0x80000000u;                          // no k-mer ends here (the text starts too late)
                    if (pos >= R) {
                        v = 0;
                        for (uint32_t b = 0; b < R; ++b) v |= code_of(chars[pos - R + b]) << (2u * b);
                        if (chars[pos - R] >= 'A' && chars[pos - R] <= 'Z') y |= 1u << (mbit + slot);   // is_upper = mask 1, src/kmers.h:71-73
                    }
                    cand[slot] = v;
                }
                for (uint32_t m = 0; m < nr; ++m, ++slot) {            // the other strand: rc of the R bases behind the occurrence of rc(K)
                    const uint64_t p = sa[er.x + m];
                    uint32_t v = 0x80000000u;
                    if (p + T + R <= text_len) {
                        uint64_t w = 0;
                        for (uint32_t b = 0; b < R; ++b) w |= static_cast<uint64_t>(code_of(chars[p + T + b])) << (2u * b);
                        v = static_cast<uint32_t>(revcomp_2bit(w, static_cast<int>(R)));
                        if (chars[p] >= 'A' && chars[p] <= 'Z') y |= 1u << (mbit + slot);               // the k-mer rc(query) starts at p
                    }
                    cand[slot] = v;
                }
                e.x = cand[2];   // no interval is ever needed from a double-stranded entry: the third candidate rides in its place
                e.y = y;
                done = true;
            }
        }
        if (!done && (e.y >> 30) == 1u) {
            const uint32_t rows = (e.y >> 24) & 3u;
            uint32_t y = e.y;
            for (uint32_t m = 0; m < rows; ++m) {
                const uint32_t pos = sa[e.x + m];
                cand[m] = pos;
                if (pos >= R && chars[pos - R] >= 'A' && chars[pos - R] <= 'Z') y |= 1u << (26u + m);
                if (bases_mode) {   // R <= 15: the R bases in front of the row -- or bit 31 when the text starts too late
                    uint32_t v = 0x80000000u;
                    if (pos >= R) {
                        v = 0;
                        for (uint32_t b = 0; b < R; ++b) v |= code_of(chars[pos - R + b]) << (2u * b);
                    }
                    cand[m] = v;
                }
            }
            e.y = y;   // (the 8-byte table stays as it is: the mask bits live in the lookup kernel's copy)
        }
        if (!done && (signed_y >> 31)) cand[0] = signed_y;
        if (wide) {   // (no side array: an interval entry's third candidate sits where a double-stranded entry has its fourth)
            vtable[2 * idx] = make_uint4(e.x, e.y, cand[0], cand[1]);
            vtable[2 * idx + 1] = done ? make_uint4(cand[3], cand[4], cand[5], cand[6]) : make_uint4(cand[2], 0, 0, 0);
        } else {
            vtable[idx] = make_uint4(e.x, e.y, cand[0], cand[1]);
            stable[idx] = cand[2];
        }
    }
}

}  // namespace
}  // namespace fmsi

extern "C" int fmsi_hip_index_enable_verify(fmsi_hip_index* idx) {
    if (idx) (void)::fmsi::wait_background(idx);   // (the prefix table's signature pass, when it runs beside the first queries)
    ::fmsi::ScratchScope scratch_scope_(idx ? idx->device : -1);   // (devpool.h: the outermost builder returns the block cache to the driver)
    if (!idx) return FMSI_HIP_EINVAL;
    FMSI_HIP_NARROW_ONLY(idx, "this accelerator");
    if (idx->d_vtable) return 0;
    TraceRange trace_range("fmsi: build candidate tables");
    const int T = idx->prefix_len, R = idx->k - T;
    if (!idx->d_ptable || !idx->dev.ptable_packed || idx->k > 32 || R < 1) {
        set_last_error("text verification needs a packed prefix table with t < k <= 32");
        return FMSI_HIP_EINVAL;
    }
    FMSI_HIP_ON_DEVICE(idx->device);
    const uint64_t n = idx->n, len = n - 1, count = 1ull << (2 * T);
    char* d_chars = nullptr;
    uint32_t *d_sa = nullptr, *d_stable = nullptr;
    uint4* d_vtable = nullptr;
    uint64_t* d_text2 = nullptr;
    auto fail = [&](int code, const char* msg) {
        for (void* p : {static_cast<void*>(d_chars), static_cast<void*>(d_sa), static_cast<void*>(d_stable), static_cast<void*>(d_text2),
                        static_cast<void*>(d_vtable)})
            if (p) (void)dev_free(p);
        if (msg) set_last_error(msg);
        return code;
    };
    // 32-byte entries (double-stranded entries of up to seven occurrences, lookup_lane_kernel<.., WIDE>) when the candidate words
    // are bases and the device has the room -- tables, the walk's temporaries and 16 GB to spare for the caller's batches
    bool wide = R <= 15 && !std::getenv("FMSI_HIP_NARROW_CANDIDATES") && !std::getenv("FMSI_HIP_LOOKUP_PAIRS");
    if (wide) {
        size_t free_b = 0, total_b = 0;
        const uint64_t need = count * 32 + (len + 1) + n * 4 + ((len >> 5) + 2) * 8 + (16ull << 30);
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b + dev_cached_bytes() < need) wide = false;   // (cached blocks are given back on demand)
    }
    const uint64_t entry_bytes = wide ? 32 : 16;
    if (dev_malloc(reinterpret_cast<void**>(&d_chars), len + 1) != hipSuccess || dev_malloc(reinterpret_cast<void**>(&d_sa), n * 4) != hipSuccess ||
        (!wide && dev_malloc(reinterpret_cast<void**>(&d_stable), count * 4) != hipSuccess) ||
        dev_malloc(reinterpret_cast<void**>(&d_vtable), count * entry_bytes + 64) != hipSuccess)
        return fail(FMSI_HIP_ENOMEM, "hipMalloc failed for the text verification tables");
    int rc = export_walk(idx, d_chars, d_sa);
    if (rc) return fail(rc, nullptr);
    const uint64_t n_words = (len >> 5) + 2;
    if (!idx->d_text2) {
        if (dev_malloc(reinterpret_cast<void**>(&d_text2), n_words * 8) != hipSuccess) return fail(FMSI_HIP_ENOMEM, "hipMalloc failed for the text");
        hipLaunchKernelGGL(chars_to_text2_kernel, dim3(static_cast<uint32_t>((n_words + 255) / 256)), dim3(256), 0, 0, d_chars, len, n_words, d_text2);
    }
    const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>((count + 255) / 256, 1u << 20));
    hipLaunchKernelGGL(verify_fill_kernel, dim3(blocks), dim3(256), 0, 0, static_cast<const uint2*>(idx->d_ptable), count, d_sa, d_chars, len,
                       static_cast<uint32_t>(T), static_cast<uint32_t>(R), R <= 15 ? 1u : 0u, wide ? 1u : 0u, d_vtable, d_stable);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(FMSI_HIP_EHIP, "text verification: device work failed");
    (void)dev_free(d_chars);
    (void)dev_free(d_sa);
    if (d_text2) {
        idx->d_text2 = d_text2;
        idx->text_len = len;
        idx->device_bytes += n_words * 8;
    }
    idx->d_stable = d_stable;
    idx->d_vtable = d_vtable;
    idx->vwide = wide;
    idx->device_bytes += count * (entry_bytes + (wide ? 0 : 4)) + 64;
    return 0;
}
