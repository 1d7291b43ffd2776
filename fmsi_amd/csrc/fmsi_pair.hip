// Two-base LF steps over 128-byte blocks ("pair layout"), read cooperatively by quads of lanes -- the
// request-count optimisation of the query path.
//
// Why: the chip serves ~52 G random block reads/s whether a block is 64 B or 128 B, as long as the randomly
// accessed footprint stays below ~3.2 GB and a wave keeps <= ~8 dwordx4 loads in flight per lane
// (profiles/r01_gather_ceiling.md).  The single-step kernels already run at that request ceiling, so the only way
// up is fewer requests per k-mer.  A 128-byte block can carry what TWO consecutive backward-search steps need: for
// every row r (suffix start s = SA[r]) both preceding symbols c1 = T[s-1] (the BWT symbol) and c2 = T[s-2], plus
// cumulative counts of the 16 (c2, c1) pairs.  Then
//     [i, j)  --(prepend "c2 c1")-->  [C2[c2,c1] + rank_pair(i), C2[c2,c1] + rank_pair(j))
// is ONE block read per boundary -- exactly the interval two update_range() calls (reference
// src/fms_index.h:98-103) produce, so the final SA interval, the early exit on an empty interval and therefore
// every result are unchanged.  c2 is derived on the device from the single-step layout alone (c2(r) = BWT[LF(r)]),
// so the layout can be added to any handle, whether it was loaded from the reference's files or built on the device.
//
// Block = 128 B = 128 rows = four 32-byte QUARTERS; quarter s covers rows 32s .. 32s+31 and is read by lane s of a
// quad with two dwordx4 loads (the quad's four loads coalesce into one 128-B request):
//   dw 0  c1 hi plane   dw 1  c1 lo plane   dw 2  c2 hi plane   dw 3  c2 lo plane      (32 rows each)
//   dw 4  mask plane (SA-transformed mask, 32 rows)
//   dw 5..7  the four cumulative counts of the pairs with c1 = s (pair index p = c2 | c1 << 2 = 4s .. 4s+3),
//            24 bits each, relative to the superblock start
// Superblock = 2^23 rows; `sbt` holds the 16 absolute counts at each superblock start (21 KB at human scale: cache
// resident).  Rows without a c1 / c2 (suffix starts 0 and 1) are stored as 'A' and subtracted in the rank, like the
// '$' slot of the reference (src/fms_index.h:81).
//
// Kernel: the persistent-lane state machine of query_refill_kernel with a QUAD per k-mer: the four lanes hold the
// same state, each ranks its 32 rows, partial popcounts are summed with two quad shuffles and the count / mask bit /
// table entry owned by one lane is broadcast with one wave shuffle.  16 k-mers per wave, <= 4 dwordx4 in flight per
// lane, ~60 VGPRs.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/fmsi_hip.h"
#include "common.h"
#include "device_fm.h"
#include "index_priv.h"

namespace fmsi {

constexpr uint32_t kPairBlockDwords = 32;
constexpr uint32_t kSuperLog2 = 23;                       // rows per superblock
constexpr uint32_t kSuperBlocksLog2 = kSuperLog2 - kBlockRowsLog2;

struct PairIndex {
    const uint32_t* pb;       // n_blocks x 32 dwords
    const uint32_t* sbt;      // n_super x 16 absolute pair counts
    const uint32_t* c2;       // 16 x C2[p]
    const uint32_t* mrank;    // n_blocks + 1 cumulative mask popcounts (shared with the single-step layout)
    const uint2* ptable;
    uint32_t ptable_len;
    uint32_t n, r0, r1, p1;   // rows; row of suffix 0 ('$' slot); row of suffix 1; pair index stored for r1
    uint32_t C0, C1, C2_, C3;
};

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
// One thread per row (256 threads = 2 blocks): c1 from the single-step block, c2 = BWT[LF(row)].
__global__ __launch_bounds__(256) void pair_emit_kernel(DevIndex ix, uint32_t* __restrict__ pb, U16* __restrict__ counts,
                                                        uint32_t* __restrict__ specials /* [0] = r1, [1] = c1(r1) */) {
    __shared__ uint32_t s_cnt[4][16];
    const uint64_t row64 = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const bool valid = row64 < ix.n;
    const uint32_t row = static_cast<uint32_t>(row64);
    uint32_t c1 = 0, c2 = 0, mbit = 0;
    if (valid) {
        const uint32_t* blk = reinterpret_cast<const uint32_t*>(ix.blocks) + (static_cast<size_t>(row >> kBlockRowsLog2) << 4);
        const uint32_t o = row & (kBlockRows - 1u), w = o >> 5, sh = o & 31u;
        c1 = (((blk[4 + w] >> sh) & 1u) << 1) | ((blk[8 + w] >> sh) & 1u);
        mbit = (blk[12 + w] >> sh) & 1u;
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
            }
        }
    }
    const uint64_t h1 = __ballot(valid && (c1 & 2u)), l1 = __ballot(valid && (c1 & 1u));
    const uint64_t h2 = __ballot(valid && (c2 & 2u)), l2 = __ballot(valid && (c2 & 1u));
    const uint64_t mm = __ballot(valid && mbit), vv = __ballot(valid);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane < 2 && vv) {   // lane 0: rows 0..31 of the wave, lane 1: rows 32..63 -> one quarter each
        const uint64_t w = row64 >> 6;                                   // 64-row word of the index
        const uint32_t quarter = static_cast<uint32_t>(w & 1) * 2u + lane;
        uint32_t* qd = pb + (w >> 1) * kPairBlockDwords + quarter * 8u;
        const uint32_t sh = lane * 32u;
        qd[0] = static_cast<uint32_t>(h1 >> sh);
        qd[1] = static_cast<uint32_t>(l1 >> sh);
        qd[2] = static_cast<uint32_t>(h2 >> sh);
        qd[3] = static_cast<uint32_t>(l2 >> sh);
        qd[4] = static_cast<uint32_t>(mm >> sh);
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
    for (int s = 0; s < 4; ++s) {
        const uint32_t r0 = a.v[4 * s] - base.v[4 * s], r1 = a.v[4 * s + 1] - base.v[4 * s + 1];
        const uint32_t r2 = a.v[4 * s + 2] - base.v[4 * s + 2], r3 = a.v[4 * s + 3] - base.v[4 * s + 3];
        blk[8 * s + 5] = (r0 & 0xFFFFFFu) | (r1 << 24);
        blk[8 * s + 6] = ((r1 >> 8) & 0xFFFFu) | (r2 << 16);
        blk[8 * s + 7] = ((r2 >> 16) & 0xFFu) | (r3 << 8);
    }
    if ((b & ((1ull << kSuperBlocksLog2) - 1)) == 0) {
#pragma unroll
        for (int p = 0; p < 16; ++p) sbt[sb * 16 + p] = a.v[p];
    }
}

// ---- query kernel -----------------------------------------------------------------------------------------
enum : uint32_t { PS_IDLE = 0, PS_LOOKUP = 1, PS_STEP = 2, PS_PROBE = 3 };

__device__ __forceinline__ uint32_t quad_sum(uint32_t x) {
    x += __shfl_xor(x, 1, 64);
    x += __shfl_xor(x, 2, 64);
    return x;
}

// count m (0..3) of a quarter's packed 24-bit counts, or their sum
__device__ __forceinline__ uint32_t quarter_count(uint32_t a, uint32_t b, uint32_t c, uint32_t m, bool sum) {
    const uint32_t c0 = a & 0xFFFFFFu, c1 = (a >> 24) | ((b & 0xFFFFu) << 8), c2 = (b >> 16) | ((c & 0xFFu) << 16), c3 = c >> 8;
    return sum ? c0 + c1 + c2 + c3 : sel4(m, c0, c1, c2, c3);
}

struct PairCounters {
    uint32_t lf_steps = 0, blocks = 0, probes = 0, searches = 0, lookups = 0;
};

template <bool COUNT, bool PTABLE>
__global__ __launch_bounds__(256) void query_pair_kernel(PairIndex ix, const uint64_t* __restrict__ kmers, uint64_t n_windows, int k,
                                                         uint32_t flags, uint32_t chunk, uint8_t* __restrict__ out,
                                                         long long* __restrict__ out_orders, unsigned long long* __restrict__ counters) {
    const uint32_t mode_all = flags & FMSI_HIP_MODE_ALL;
    const uint32_t strict = flags & FMSI_HIP_STRICT;
    const uint32_t orders = flags & FMSI_HIP_ORDERS;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t sub = lane & 3u;                 // my quarter
    const uint32_t qbase = lane & ~3u;              // first lane of my quad
    const uint64_t quads_below = (1ull << qbase) - 1ull;
    const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    uint64_t next = wave * chunk;
    const uint64_t end = min(n_windows, next + static_cast<uint64_t>(chunk));
    const int T = PTABLE ? static_cast<int>(ix.ptable_len) : 0;

    // all four lanes of a quad carry the same state
    uint64_t f = 0, cur = 0, f_next = 0, q = 0, q_next = 0;
    uint32_t i = 0, j = 0, st = PS_IDLE, phase = 0, done = 0;
    int t = 0, res_f = -1;
    long long ord_f = -1;
    bool has_next = false;
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
                if (strict) {
                    out_orders[2 * q] = of;
                    out_orders[2 * q + 1] = orr;
                } else {
                    out_orders[q] = was_second ? orr : of;
                }
            } else if (strict) {
                out[q] = static_cast<uint8_t>((rf + 1) | ((rr + 1) << 2));
            } else {
                out[q] = static_cast<uint8_t>((was_second ? rr : rf) == 1);
            }
        }
        st = PS_IDLE;
        done++;
    };

    for (;;) {
        if (st == PS_IDLE && has_next) {
            f = f_next;
            q = q_next;
            has_next = false;
            phase = 0;
            start_strand(f);
        }
        {
            const bool want = !has_next;
            const uint64_t m = __ballot(want) & 0x1111111111111111ull;   // one bit per quad
            if (m) {
                const uint64_t cand = next + static_cast<uint64_t>(__popcll(m & quads_below));
                if (want && cand < end) {
                    q_next = cand;
                    f_next = kmers[cand];   // the quad reads one address: a single request
                    has_next = true;
                }
                next += static_cast<uint64_t>(__popcll(m));
            }
        }
        if (!__any(st != PS_IDLE)) {
            if (!__any(has_next)) break;
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
            // my 32-byte quarter of each boundary's block -- or my 16 bytes of the prefix-table line
            const uint4* pi4 = reinterpret_cast<const uint4*>(ix.pb + static_cast<size_t>(bi) * kPairBlockDwords) + sub * 2u;
            const uint4* pj4 = reinterpret_cast<const uint4*>(ix.pb + static_cast<size_t>(bj) * kPairBlockDwords) + sub * 2u;
            uint32_t tidx = 0;
            const bool lookup = PTABLE && st == PS_LOOKUP;
            if (lookup) {
                tidx = static_cast<uint32_t>(cur >> (2 * (k - T))) & ((1u << (2 * T)) - 1u);
                pi4 = reinterpret_cast<const uint4*>(ix.ptable + (tidx & ~7u)) + sub;
            }
            const bool two = st == PS_STEP && bj != bi;
            uint4 b0 = make_uint4(0, 0, 0, 0), b1 = b0, a1 = b0;
            uint32_t sbi = 0, sbj = 0, c2base = 0;
            if (two) {
                b0 = pj4[0];
                b1 = pj4[1];
            }
            const uint4 a0 = pi4[0];
            if (!lookup) a1 = pi4[1];
            if (st == PS_STEP) {   // tiny cache-resident tables: not random HBM requests.  One unconditional load each:
                // pair step -> count p; single step -> my quarter's share of the four counts with c1 = p >> 2 (summed below)
                const uint32_t se = single ? ((p & 12u) + sub) : p;
                sbi = ix.sbt[(static_cast<size_t>(i >> kSuperLog2) << 4) + se];
                sbj = ix.sbt[(static_cast<size_t>((j - 1u) >> kSuperLog2) << 4) + se];
                c2base = ix.c2[p];
            }
            if (!two) {
                b0 = a0;
                b1 = a1;
            }
            if (COUNT && sub == 0) cn.blocks += 1u + (two ? 1u : 0u);

            if (st == PS_STEP) {
                const uint32_t oi = i & (kBlockRows - 1u), oj = j - (bj << kBlockRowsLog2);
                const uint32_t x1h = (p & 8u) ? 0u : 0xFFFFFFFFu, x1l = (p & 4u) ? 0u : 0xFFFFFFFFu;
                const uint32_t x2h = (p & 2u) ? 0u : 0xFFFFFFFFu, x2l = (p & 1u) ? 0u : 0xFFFFFFFFu;
                const uint32_t all = single ? 0xFFFFFFFFu : 0u;   // single step: the c2 planes do not restrict
                const uint32_t mi = (a0.x ^ x1h) & (a0.y ^ x1l) & ((a0.z ^ x2h) | all) & ((a0.w ^ x2l) | all);
                const uint32_t mj = (b0.x ^ x1h) & (b0.y ^ x1l) & ((b0.z ^ x2h) | all) & ((b0.w ^ x2l) | all);
                const int shift = static_cast<int>(sub * 32u);
                uint32_t ri = quad_sum(__popc(mi & prefix32(static_cast<int>(oi) - shift)));
                uint32_t rj = quad_sum(__popc(mj & prefix32(static_cast<int>(oj) - shift)));
                // cumulative counts live in the quarter of c1 = p >> 2
                const uint32_t owner = qbase | (p >> 2);
                const uint32_t ci = quarter_count(a1.y, a1.z, a1.w, p & 3u, single);
                const uint32_t cj = quarter_count(b1.y, b1.z, b1.w, p & 3u, single);
                ri += static_cast<uint32_t>(__shfl(static_cast<int>(ci), static_cast<int>(owner), 64));
                rj += static_cast<uint32_t>(__shfl(static_cast<int>(cj), static_cast<int>(owner), 64));
                if (single) {
                    ri += quad_sum(sbi);
                    rj += quad_sum(sbj);
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
                if (i == j) finalize(-1, -1);
                else if (t < 0) st = PS_PROBE;
            } else if (st == PS_PROBE) {
                if (COUNT && sub == 0) cn.probes++;
                const uint32_t oi = i & (kBlockRows - 1u);
                const int shift = static_cast<int>(sub * 32u);
                int res;
                long long ord = -1;
                if (mode_all && !orders) {
                    const uint32_t bit = (a1.x >> (oi & 31u)) & 1u;            // valid in the lane that owns row i
                    res = __shfl(static_cast<int>(bit), static_cast<int>(qbase | (oi >> 5)), 64);
                } else {
                    const uint32_t oj = j - (bj << kBlockRowsLog2);
                    const uint32_t below_i = quad_sum(__popc(a1.x & prefix32(static_cast<int>(oi) - shift)));
                    uint32_t ones;
                    if (bj == bi) {
                        ones = quad_sum(__popc(a1.x & prefix32(static_cast<int>(oj) - shift))) - below_i;
                    } else {   // interval spans blocks (long repeats): rank difference through the side array
                        const uint32_t mjq = ix.pb[static_cast<size_t>(bj) * kPairBlockDwords + sub * 8u + 4u];
                        ones = (ix.mrank[bj] + quad_sum(__popc(mjq & prefix32(static_cast<int>(oj) - shift)))) - (ix.mrank[bi] + below_i);
                    }
                    res = ones > 0u;
                    if (orders && res) ord = static_cast<long long>(ix.mrank[bi] + below_i);
                }
                finalize(res, ord);
            } else {   // PS_LOOKUP: entry e of the 8-entry line sits in lane e >> 1 of the quad
                if (COUNT && sub == 0) cn.lookups++;
                const uint32_t e = tidx & 7u;
                const uint32_t xi = (e & 1u) ? a0.z : a0.x, xj = (e & 1u) ? a0.w : a0.y;
                const int src = static_cast<int>(qbase | (e >> 1));
                i = static_cast<uint32_t>(__shfl(static_cast<int>(xi), src, 64));
                j = static_cast<uint32_t>(__shfl(static_cast<int>(xj), src, 64));
                t = k - 1 - T;
                if (i == j) finalize(-1, -1);
                else st = t < 0 ? PS_PROBE : PS_STEP;
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

PairIndex make_pair_index(const fmsi_hip_index* idx) {
    PairIndex p;
    p.pb = static_cast<const uint32_t*>(idx->d_pblocks);
    p.sbt = static_cast<const uint32_t*>(idx->d_sbt);
    p.c2 = static_cast<const uint32_t*>(idx->d_c2);
    p.mrank = idx->dev.mrank;
    p.ptable = idx->dev.ptable;
    p.ptable_len = idx->dev.ptable_len;
    p.n = idx->dev.n;
    p.r0 = idx->dev.dollar;
    p.r1 = idx->pair_r1;
    p.p1 = idx->pair_p1;
    p.C0 = idx->dev.C0; p.C1 = idx->dev.C1; p.C2_ = idx->dev.C2; p.C3 = idx->dev.C3;
    return p;
}

template <bool COUNT, bool PTABLE>
void launch_pair_t(fmsi_hip_index* idx, const uint64_t* d_kmers, uint64_t n, int k, uint32_t flags, uint8_t* o8, long long* o64,
                   unsigned long long* d_counters, hipStream_t s) {
    // 16 k-mers in flight per wave: chunks are a quarter of the single-lane kernels'
    uint64_t chunk = n / (256ull * 32 * 8 * 4);
    chunk = std::min<uint64_t>(std::max<uint64_t>(chunk, 128), 2048);
    const uint64_t waves = (n + chunk - 1) / chunk;
    const uint32_t blocks = static_cast<uint32_t>((waves + 3) / 4);
    hipLaunchKernelGGL((query_pair_kernel<COUNT, PTABLE>), dim3(blocks), dim3(256), 0, s, make_pair_index(idx), d_kmers, n, k, flags,
                       static_cast<uint32_t>(chunk), o8, o64, d_counters);
}

}  // namespace

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
    if (!idx) return FMSI_HIP_EINVAL;
    if (idx->d_pblocks) return 0;
    FMSI_HIP_TRY(hipSetDevice(idx->device));
    const uint64_t n_blocks = idx->n_blocks;
    const uint64_t n_super = ((n_blocks - 1) >> kSuperBlocksLog2) + 1;
    void *pb = nullptr, *counts = nullptr, *scanned = nullptr, *sbt = nullptr, *c2 = nullptr, *spec = nullptr, *temp = nullptr;
    auto cleanup = [&]() {
        for (void* p : {counts, scanned, spec, temp})
            if (p) (void)hipFree(p);
    };
    auto fail = [&](int code, const char* msg) {
        cleanup();
        for (void* p : {pb, sbt, c2})
            if (p) (void)hipFree(p);
        set_last_error(msg);
        return code;
    };
    const size_t pb_bytes = n_blocks * kPairBlockDwords * 4;
    if (hipMalloc(&pb, pb_bytes) != hipSuccess || hipMalloc(&counts, (n_blocks + 1) * sizeof(U16)) != hipSuccess ||
        hipMalloc(&scanned, (n_blocks + 1) * sizeof(U16)) != hipSuccess || hipMalloc(&sbt, n_super * 64) != hipSuccess ||
        hipMalloc(&c2, 64) != hipSuccess || hipMalloc(&spec, 8) != hipSuccess)
        return fail(FMSI_HIP_ENOMEM, "hipMalloc failed for the pair layout");
    if (hipMemset(pb, 0, pb_bytes) != hipSuccess || hipMemset(counts, 0, (n_blocks + 1) * sizeof(U16)) != hipSuccess ||
        hipMemset(spec, 0xFF, 8) != hipSuccess)
        return fail(FMSI_HIP_EHIP, "hipMemset failed for the pair layout");
    const uint32_t grid = static_cast<uint32_t>((idx->n + 255) / 256);
    hipLaunchKernelGGL(pair_emit_kernel, dim3(grid), dim3(256), 0, 0, idx->dev, static_cast<uint32_t*>(pb), static_cast<U16*>(counts),
                       static_cast<uint32_t*>(spec));
    size_t tb = 0;
    U16 zero{};
    if (rocprim::exclusive_scan(nullptr, tb, static_cast<U16*>(counts), static_cast<U16*>(scanned), zero, static_cast<size_t>(n_blocks + 1),
                                U16Plus()) != hipSuccess ||
        hipMalloc(&temp, tb) != hipSuccess ||
        rocprim::exclusive_scan(temp, tb, static_cast<U16*>(counts), static_cast<U16*>(scanned), zero, static_cast<size_t>(n_blocks + 1),
                                U16Plus()) != hipSuccess)
        return fail(FMSI_HIP_EHIP, "pair layout: prefix sums failed");
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
    return 0;
}
