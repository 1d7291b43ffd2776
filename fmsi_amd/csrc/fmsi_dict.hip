// The k-mer dictionary: one 128-byte random read per query k-mer, whatever the text looks like.
//
// Why.  With the prefix table + candidate tables (fmsi_pair.hip) a k-mer costs one table line as long as the table key -- its
// last t bases -- has few occurrences; a real genome's repeats break that (17 G k-mers/s instead of 43 G on a
// text with 2 %-diverged repeat families), the tables take 137 GB at t = 16 for ANY index, and `-S` batches with substitutions
// fall back to steps.  What decides a query is the full k-mer, so this private accelerator is keyed by it: every distinct
// canonical k-mer of the superstring -- c = min(q, rc(q)) -- is stored with the answers get_range_with_pattern() +
// infer_presence() (src/fms_index.h:117-144) give for c and for rc(c): absent / present with `-O` and default-mode bits, 2 bits
// per strand.  A query is then: canonical form, hash, ONE bucket read, compare -- both strands settled, in both modes, for present
// and absent k-mers alike; a k-mer that is not in the dictionary occurs on neither strand.  The answers in the slots are computed
// from the index itself (below), so results are the reference's by construction; tests/ pin dictionary == LF walk == oracle.
//
// Layout.  Level 0: 2^b buckets of 128 bytes = 16 slots of 8 bytes, b the smallest with <= 10.5 keys per bucket on average
// (human scale: 2^28 buckets = 34 GB for 2.8 G k-mers; the footprint follows n, an E. coli index takes 64 MB).  h = mix64(c) is a
// bijection on 64 bits; bucket = h >> (64 - b), and a slot holds the other 64 - b bits ("quotient", <= 60 bits) << 4 | answer
// nibble, so a match is exact -- no fingerprints, no false positives.  Answer nibble = code(c) | code(rc(c)) << 2 with code 0
// absent, 1 present (all occurrences OFF), 2 some occurrence ON but not the first row of the interval, 3 first row ON; it is never 0
// for a stored key (the k-mer occurs on one strand at least), so an all-zero slot is empty.  A bucket that would hold more than
// 16 keys keeps 15 and a marker in its last slot; its other keys go to level 1 (its own hash and size, <= 3 keys per bucket), and
// so on -- 1.1 % of the keys and 3.8 % of the buckets at the densest level-0 load, so 1.04 reads per k-mer at worst.
//
// The memory path serves ~45-48 G random 128-byte requests per second when EIGHT lanes fetch 16 bytes each of one line in one
// instruction (tools/gather_bench entry, coop 8: profiles/r01k_gather_entry.txt), the same rate as for 64-byte or 16-byte
// requests: the bucket size is free, and 128 bytes keep the overflow rare at 12 bytes of HBM per k-mer.  So a wave looks 64
// k-mers up per round with eight load instructions of eight buckets each; lane l owns k-mer l, lanes 8j .. 8j+7 serve k-mer
// 8g + j in round g.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/fmsi_hip.h"
#include "common.h"
#include "device_fm.h"
#include "index_priv.h"
#include "trace.h"

namespace fmsi {

namespace {

constexpr uint32_t kDictSlots = 16;
constexpr uint32_t kMarkerLo = 0xFFFFFFF0u, kMarkerHi = 0xFFFFFFFFu;   // slot 15 of an overflowed bucket (answer nibble 0: never a key)
constexpr uint64_t kLevelSalt[kDictMaxLevels] = {0ull, 0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull, 0xA24BAED4963EE407ull};

// murmur3's 64-bit finalizer: a bijection on u64 (each step is invertible)
__host__ __device__ inline uint64_t dict_mix(uint64_t x) {
    x ^= x >> 33;
    x *= 0xFF51AFD7ED558CCDull;
    x ^= x >> 33;
    x *= 0xC4CEB9FE1A85EC53ull;
    x ^= x >> 33;
    return x;
}
__host__ __device__ inline uint64_t dict_unmix(uint64_t x) {
    x ^= x >> 33;
    x *= 0x9CB4B2F8129337DBull;   // inverse of 0xC4CEB9FE1A85EC53 mod 2^64
    x ^= x >> 33;
    x *= 0x4F74430C22A54005ull;   // inverse of 0xFF51AFD7ED558CCD mod 2^64
    x ^= x >> 33;
    return x;
}
static_assert(0xFF51AFD7ED558CCDull * 0x4F74430C22A54005ull == 1ull && 0xC4CEB9FE1A85EC53ull * 0x9CB4B2F8129337DBull == 1ull, "hash inverses");

__device__ __forceinline__ uint4 load16_nt(const uint4* p) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(t.x, t.y, t.z, t.w);
}

constexpr uint32_t kFlagSelfRight = 0x20000000u;   // ... its second pass (skdict.inc: ties to the rightmost minimal m-mer)
constexpr uint32_t kFlagSelf = 0x40000000u;   // kernel-internal (never part of the ABI's flags): the construction's pass over the text's own k-mers

// 64 hashes, one per lane (`active` lanes only), against one level: eight rounds of eight buckets.  Returns at the OWNING lane the
// answer nibble of its k-mer (0: not in this bucket) | 16 when the bucket had overflowed (the key may be a level further) | the slot
// it sits in << 8.  `tmp`: 64 words of LDS of this wave.
__device__ __forceinline__ uint32_t dict_probe(const DictLevel L, uint64_t h, bool active, uint32_t lane, uint32_t* tmp) {
    const uint32_t grp = lane >> 3, sub8 = lane & 7u;
    const uint32_t bidx = static_cast<uint32_t>(h >> L.shift);
    const uint64_t key = (h << (64u - L.shift)) >> (60u - L.shift);   // quotient << 4 (shift <= 60)
    const uint32_t klo = static_cast<uint32_t>(key) | (active ? 1u : 0u), khi = static_cast<uint32_t>(key >> 32);
    uint4 piece[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {   // all loads first: eight lines per instruction, 64 in flight per wave
        const int src = 8 * g + static_cast<int>(grp);
        const uint32_t b = static_cast<uint32_t>(__shfl(static_cast<int>(bidx), src, 64));
        const uint32_t a = static_cast<uint32_t>(__shfl(static_cast<int>(klo), src, 64)) & 1u;
        piece[g] = make_uint4(0, 0, 0, 0);
        if (a) piece[g] = load16_nt(L.buckets + (static_cast<size_t>(b) << 3) + sub8);
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int src = 8 * g + static_cast<int>(grp);
        const uint32_t wl = static_cast<uint32_t>(__shfl(static_cast<int>(klo), src, 64)) & ~15u;
        const uint32_t wh = static_cast<uint32_t>(__shfl(static_cast<int>(khi), src, 64));
        const uint4 p = piece[g];
        uint32_t v = 0;
        if (((p.x ^ wl) & ~15u) == 0u && p.y == wh && (p.x & 15u)) v = (p.x & 15u) | ((2u * sub8) << 8);     // (an empty or marker slot has nibble 0)
        if (((p.z ^ wl) & ~15u) == 0u && p.w == wh && (p.z & 15u)) v = (p.z & 15u) | ((2u * sub8 + 1u) << 8);
        if (sub8 == 7u && p.z == kMarkerLo && p.w == kMarkerHi) v |= 16u;
        v |= static_cast<uint32_t>(__shfl_xor(static_cast<int>(v), 1, 64));
        v |= static_cast<uint32_t>(__shfl_xor(static_cast<int>(v), 2, 64));
        v |= static_cast<uint32_t>(__shfl_xor(static_cast<int>(v), 4, 64));
        if (sub8 == 0u) tmp[src] = v;
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t mine = tmp[lane];
    __builtin_amdgcn_wave_barrier();
    return mine;
}

// answer nibble (0 = the k-mer is on neither strand) -> a window's result byte
__device__ __forceinline__ uint8_t dict_encode(uint32_t ans, bool swapped, bool pal, uint32_t flags) {
    if (flags & kFlagSelf) return static_cast<uint8_t>(ans);                // (construction: the nibble itself, per text position)
    const uint32_t mode_all = flags & FMSI_HIP_MODE_ALL;
    const uint32_t cc = ans & 3u, cr = pal ? cc : (ans >> 2);           // codes of the canonical k-mer and of its reverse complement
    const uint32_t cf = swapped ? cr : cc, cb = swapped ? cc : cr;      // ... of the query as given and of its reverse complement
    const int rf = cf == 0u ? -1 : static_cast<int>(mode_all ? cf == 3u : cf >= 2u);
    const int rr = cb == 0u ? -1 : static_cast<int>(mode_all ? cb == 3u : cb >= 2u);
    if (flags & FMSI_HIP_STRICT) return static_cast<uint8_t>(static_cast<uint32_t>(rf + 1) | (static_cast<uint32_t>(rr + 1) << 2));
    // forward strand first, the reverse complement when it does not decide (src/fms_index.h:283-299)
    return static_cast<uint8_t>(mode_all ? (rf != -1 ? rf : rr) == 1 : (rf == 1 || rr == 1));
}

// ---- the lookup kernel --------------------------------------------------------------------------------------------------
// A wave owns a contiguous chunk of windows (<= 4096, a multiple of 128), as the other query kernels do; result bytes collect in
// LDS and leave as 16 bytes per lane -- or bit-packed (FMSI_HIP_OUT_BITS).  RUNS: `kmers` is a 2-bit packed sequence and the
// windows are cut here (fmsi_hip_query_runs; see pair_lookup_kernel).
template <bool COUNT, bool RUNS>
__global__ __launch_bounds__(256) void dict_lookup_kernel(DictIndex dx, const uint64_t* __restrict__ kmers, uint64_t n_windows, int k, uint32_t flags,
                                                          uint32_t chunk, uint8_t* __restrict__ out, unsigned long long* __restrict__ counters,
                                                          const uint64_t* __restrict__ run_start, const uint64_t* __restrict__ run_off, uint64_t n_runs,
                                                          uint64_t u_start0, uint64_t u_stride, uint64_t u_wpr, uint32_t* __restrict__ slot_pos) {
    // (RUNS with u_wpr != 0: uniform runs -- run r is the u_wpr windows from base u_start0 + r * u_stride, reads of one length laid out
    // back to back -- and there is no run table to fetch or to search)
    constexpr uint32_t kMaxChunk = 4096u, kQueue = 128u;
    __shared__ uint64_t s_qh[4][kQueue];     // k-mers whose level-0 bucket had overflowed: their hash ...
    __shared__ uint32_t s_qi[4][kQueue];     // ... and index in the chunk | orientation flags
    __shared__ uint32_t s_tmp[4][64];        // probe results on their way from the serving lanes to the owning lane
    __shared__ __attribute__((aligned(16))) uint8_t s_res[4][kMaxChunk];
    const uint32_t strict = flags & FMSI_HIP_STRICT;
    const bool out_bits = (flags & FMSI_HIP_OUT_BITS) != 0u;
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
    uint64_t* qh = s_qh[wib];
    uint32_t* qi = s_qi[wib];
    uint32_t* tmp = s_tmp[wib];
    uint8_t* res = s_res[wib];
    const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const uint64_t start = wave * chunk;
    const uint32_t len = start < n_windows ? static_cast<uint32_t>(min(static_cast<uint64_t>(chunk), n_windows - start)) : 0u;
    uint32_t qhead = 0, qtail = 0;   // wave-uniform: the queue holds [qhead, qtail), positions mod kQueue
    uint32_t n_reads = 0, n_final = 0;
    uint64_t run = 0, run_last = 0, run_end = 0, run_delta = 0;
    auto find_run = [&](uint64_t lo, uint64_t hi, uint64_t q) -> uint64_t {   // run_off[lo] <= q < run_off[hi]
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (run_off[mid] <= q) lo = mid;
            else hi = mid;
        }
        return lo;
    };
    auto enter_run = [&](uint64_t r) {   // run_end = first window behind run r, run_delta = its first base - its first window
        run = r;
        if (u_wpr) {
            run_end = (r + 1u) * u_wpr;
            run_delta = u_start0 + r * (u_stride - u_wpr);
        } else {
            run_end = run_off[r + 1];
            run_delta = run_start[r] - run_off[r];
        }
    };
    if (RUNS && len) {
        if (u_wpr) {
            enter_run(start / u_wpr);
        } else {
            run_last = find_run(0, n_runs, start + len - 1u);
            enter_run(find_run(0, run_last + 1, start));
        }
    }

    auto probe = [&](uint32_t lv, uint64_t h, bool active) -> uint32_t { return dict_probe(dx.lv[lv], h, active, lane, tmp); };
    auto encode = [&](uint32_t ans, bool swapped, bool pal) -> uint8_t { return dict_encode(ans, swapped, pal, flags); };
    // construction's pass over the text (kFlagSelf): the slot of window q's k-mer learns q as its text position
    auto note_position = [&](uint32_t lv, uint64_t h, uint32_t v, uint64_t q) {
        if (slot_pos && (v & 15u)) slot_pos[dx.lv[lv].slot_base + ((h >> dx.lv[lv].shift) << 4) + ((v >> 8) & 15u)] = static_cast<uint32_t>(q);
    };
    // the queue's first `items` entries through the further levels
    auto slow_round = [&](uint32_t items) {
        bool active = lane < items;
        const uint32_t pos = (qhead + lane) & (kQueue - 1u);
        uint64_t h = active ? qh[pos] : 0ull;
        const uint32_t info = active ? qi[pos] : 0u;
        uint32_t ans = 0;
        for (uint32_t lv = 1; lv < dx.levels && __any(active); ++lv) {
            h = dict_mix(h + kLevelSalt[lv]);
            const uint32_t v = probe(lv, h, active);
            if (active) {
                n_reads++;
                if ((v & 15u) || !(v & 16u)) {
                    ans = v & 15u;
                    note_position(lv, h, v, start + (info & 0xFFFFu));
                    active = false;
                }
            }
        }
        if (lane < items) {
            n_final++;
            res[info & 0xFFFFu] = encode(ans, (info >> 16) & 1u, (info >> 17) & 1u);
        }
        qhead += items;
    };

    for (uint32_t base = 0; base < len; base += 64u) {
        const uint32_t ql = base + lane;
        const bool valid = ql < len;
        const uint64_t q = start + ql;
        uint64_t f = 0ull;
        if (RUNS) {
            if (valid) {
                if (q >= run_end) enter_run(u_wpr ? q / u_wpr : find_run(run + 1, run_last + 1, q));
                f = window_kmer(kmers, q + run_delta, k);
            }
        } else if (valid) {
            f = __builtin_nontemporal_load(kmers + q);   // (read once)
        }
        const uint64_t rc = revcomp_2bit(f, k);
        const bool swapped = rc < f, pal = rc == f;
        const uint64_t h = dict_mix(swapped ? rc : f);
        const uint32_t v = probe(0u, h, valid);
        bool slow = false;
        if (valid) {
            n_reads++;
            if ((v & 15u) || !(v & 16u)) {
                n_final++;
                note_position(0u, h, v, q);
                res[ql] = encode(v & 15u, swapped, pal);
            } else {
                slow = true;
            }
        }
        const uint64_t m = __ballot(slow);
        if (m) {
            if (slow) {
                const uint32_t pos = (qtail + lanes_below(m)) & (kQueue - 1u);
                qh[pos] = h;
                qi[pos] = ql | (swapped ? 0x10000u : 0u) | (pal ? 0x20000u : 0u);
            }
            qtail += static_cast<uint32_t>(__popcll(m));
            while (qtail - qhead >= 64u) slow_round(64u);
        }
    }
    if (qtail != qhead) slow_round(qtail - qhead);
    if (len && out_bits) {
        flush_packed_results(res, len, strict != 0u, out, start, lane);
    } else if (len) {
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
        uint32_t v[3] = {n_reads, n_final, n_final};   // bucket reads; windows answered; searches (one settles both strands)
        const int slot[3] = {4, 6, 3};
#pragma unroll
        for (int qq = 0; qq < 3; ++qq) {
            uint32_t x = v[qq];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if (lane == 0 && x) {
                atomicAdd(&counters[slot[qq]], static_cast<unsigned long long>(x));
                if (qq == 0) atomicAdd(&counters[1], static_cast<unsigned long long>(x));   // [1] counts the bucket reads with the blocks
            }
        }
    }
}


// ---- chained lookups of reads ---------------------------------------------------------------------------------------------------
// Consecutive windows of a read share k - 1 bases, and when the read comes from the superstring they sit at consecutive text
// positions -- yet dict_lookup_kernel pays a bucket read for each.  With the construction's extras (FMSI_HIP_DICT_CHAIN: the 2-bit
// text, the answer nibble of the k-mer at every text position, and per dictionary slot the position of one occurrence) a lane takes
// 32 consecutive windows: it looks the first one up; if it is there it reads the occurrence's position, checks which way the read
// runs along the text (the window equals the text's k-mer there, or its reverse complement), and from then on a window costs one
// base compared with the text and one nibble of the per-position answers -- both cached in registers for the whole task (the answer
// of a window is a function of its k-mer, and the k-mer is known to be the text's by induction from the exact match of the first).
// A window that leaves the text (a substitution, the read's end, a read that never was in it) goes to the wave's queue of ordinary
// lookups, done 64 at a time like everywhere else; so a random read costs what it cost, and a read from the text 6 requests per 32
// windows instead of 32.  Uniform runs only (reads of one length; fmsi_hip_query_runs falls back to dict_lookup_kernel otherwise).
// The wave's chunk = 64 tasks = 2048 windows of the OUTPUT order, so results leave exactly as from the other kernels.
template <bool COUNT>
__global__ __launch_bounds__(256) void dict_chain_kernel(DictIndex dx, const uint64_t* __restrict__ seq, uint64_t n_windows, int k, uint32_t flags,
                                                         uint8_t* __restrict__ out, unsigned long long* __restrict__ counters, uint64_t u_start0,
                                                         uint64_t u_stride, uint64_t u_wpr) {
    constexpr uint32_t kTask = 32u, kChunk = 64u * kTask, kQueue = 128u;
    __shared__ uint64_t s_qh[4][kQueue];
    __shared__ uint32_t s_qi[4][kQueue];
    __shared__ uint32_t s_tmp[4][64];
    __shared__ __attribute__((aligned(16))) uint8_t s_res[4][kChunk];
    const uint32_t strict = flags & FMSI_HIP_STRICT;
    const bool out_bits = (flags & FMSI_HIP_OUT_BITS) != 0u;
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
    uint64_t* qh = s_qh[wib];
    uint32_t* qi = s_qi[wib];
    uint32_t* tmp = s_tmp[wib];
    uint8_t* res = s_res[wib];
    const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const uint64_t start = wave * kChunk;
    const uint32_t len = start < n_windows ? static_cast<uint32_t>(min(static_cast<uint64_t>(kChunk), n_windows - start)) : 0u;
    uint32_t qhead = 0, qtail = 0;
    uint32_t n_reads = 0, n_final = 0, n_side = 0;
    const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    // the queue's first `items` entries: ordinary lookups from level 0 on
    auto lookup_round = [&](uint32_t items) __attribute__((always_inline)) {
        bool active = lane < items;
        const uint32_t pos = (qhead + lane) & (kQueue - 1u);
        uint64_t h = active ? qh[pos] : 0ull;
        const uint32_t info = active ? qi[pos] : 0u;
        uint32_t ans = 0;
#pragma unroll   // (constant level numbers: the levels' descriptors stay in scalar registers instead of a scratch copy of `dx`)
        for (uint32_t lv = 0; lv < kDictMaxLevels; ++lv) {
            if (lv >= dx.levels || !__any(active)) break;
            if (lv) h = dict_mix(h + kLevelSalt[lv]);
            const uint32_t v = dict_probe(dx.lv[lv], h, active, lane, tmp);
            if (active) {
                n_reads++;
                if ((v & 15u) || !(v & 16u)) {
                    ans = v & 15u;
                    active = false;
                }
            }
        }
        if (lane < items) {
            n_final++;
            res[info & 0xFFFFu] = dict_encode(ans, (info >> 16) & 1u, (info >> 17) & 1u, flags);
        }
        qhead += items;
    };
    auto push_lookup = [&](bool want, uint64_t f, uint64_t rc, uint32_t ql) __attribute__((always_inline)) {   // (every lane of the wave comes here)
        const uint64_t m = __ballot(want);
        if (!m) return;
        if (want) {
            const bool swapped = rc < f;
            const uint32_t pos = (qtail + lanes_below(m)) & (kQueue - 1u);
            qh[pos] = dict_mix(swapped ? rc : f);
            qi[pos] = ql | (swapped ? 0x10000u : 0u) | (rc == f ? 0x20000u : 0u);
        }
        qtail += static_cast<uint32_t>(__popcll(m));
        while (qtail - qhead >= 64u) lookup_round(64u);
    };
    // base i (0..95) of three consecutive 32-base words
    auto base3 = [](uint64_t w0, uint64_t w1, uint64_t w2, uint32_t i) __attribute__((always_inline)) -> uint32_t {
        const uint64_t w = i < 32u ? w0 : (i < 64u ? w1 : w2);
        return static_cast<uint32_t>(w >> (2u * (i & 31u))) & 3u;
    };

    // ---- my task: windows [t0, t0 + mine) of the chunk; the first n1 of them lie in the run of the first
    const uint32_t t0 = lane * kTask;
    const uint32_t mine = t0 < len ? min(kTask, len - t0) : 0u;
    const uint64_t q0 = start + t0;
    uint64_t f = 0, rc = 0, sw0 = 0, sw1 = 0, sw2 = 0, b0 = 0;
    uint32_t n1 = 0;
    if (mine) {
        const uint64_t r = q0 / u_wpr;
        n1 = static_cast<uint32_t>(min(static_cast<uint64_t>(mine), (r + 1u) * u_wpr - q0));
        b0 = u_start0 + r * u_stride + (q0 - r * u_wpr);
        const uint64_t* sp = seq + (b0 >> 5);           // bases b0 .. b0 + k + 30 lie in these three words
        sw0 = sp[0];
        sw1 = sp[1];
        sw2 = (static_cast<uint32_t>(b0 & 31u) + static_cast<uint32_t>(k) + 30u >= 64u) ? sp[2] : 0ull;
        const uint32_t sh = static_cast<uint32_t>(b0 & 31u) * 2u;
        f = sw0 >> sh;
        if (sh) f |= sw1 << (64u - sh);
        f &= kmask;
        rc = revcomp_2bit(f, k);
    }
    // ---- window 0 of every task: a dense lookup; those that are there become anchors
    bool anchored = false;
    int dir = 0;
    uint64_t p = 0;                 // text position of the CURRENT window's k-mer (dir +1: it starts there; -1: its reverse complement does)
    uint64_t tw0 = 0, tw1 = 0, tbase = 0, aw0 = 0, aw1 = 0, aw2 = 0, abase = 0;
    {
        const bool swapped = rc < f, pal = rc == f;
        const uint64_t h = dict_mix(swapped ? rc : f);
        const uint32_t v = dict_probe(dx.lv[0], h, mine != 0u, lane, tmp);
        // (a first window whose bucket had overflowed and does not hold it takes the ordinary route: the 1 % of the keys that live in
        // the further levels are not anchors)
        const bool undecided = mine && !(v & 15u) && (v & 16u);
        if (mine && !undecided) {
            n_reads++;
            n_final++;
            res[t0] = dict_encode(v & 15u, swapped, pal, flags);
            if ((v & 15u) && n1 > 1u) {   // where in the text, and which way the read runs along it
                p = dx.slot_pos[dx.lv[0].slot_base + ((h >> dx.lv[0].shift) << 4) + ((v >> 8) & 15u)];
                const uint64_t tk = window_kmer(dx.text2, p, k);
                dir = tk == f ? 1 : (tk == rc ? -1 : 0);
                n_side += 3u;
                if (dir > 0) {   // text bases p + k .. p + k + 30, answers of positions p + 1 .. p + 31
                    tbase = (p + static_cast<uint64_t>(k)) >> 5;
                    abase = (p + 1u) >> 4;
                } else if (dir < 0) {   // text bases p - 31 .. p - 1, answers of the same positions
                    const uint64_t lo = p >= 31u ? p - 31u : 0ull;
                    tbase = lo >> 5;
                    abase = lo >> 4;
                }
                if (dir) {
                    anchored = true;
                    tw0 = dx.text2[tbase];
                    tw1 = dx.text2[tbase + 1];
                    aw0 = dx.posans[abase];
                    aw1 = dx.posans[abase + 1];
                    aw2 = dx.posans[abase + 2];
                    n_side += 3u;
                }
            }
        }
        push_lookup(undecided, f, rc, t0);
    }
    // ---- nobody found an occurrence to follow (a read that is not from the text: all 64 tasks of a wave are usually one read's): the
    // rest of the chunk the plain way -- consecutive windows on consecutive lanes, one bucket read each, 35.9 -> 46 G windows/s on such reads
    const bool plain_rest = !__any(anchored);
    if (plain_rest) {
        uint64_t run_end = 0, run_delta = 0;
        for (uint32_t base = 0; base < len; base += 64u) {
            const uint32_t ql = base + lane;
            const bool valid = ql < len && (ql & (kTask - 1u)) != 0u;   // (the tasks' first windows are answered or queued already)
            const uint64_t q = start + ql;
            uint64_t wf = 0, wrc = 0;
            if (valid) {
                if (q >= run_end) {
                    const uint64_t r = q / u_wpr;
                    run_end = (r + 1u) * u_wpr;
                    run_delta = u_start0 + r * (u_stride - u_wpr);
                }
                wf = window_kmer(seq, q + run_delta, k);
                wrc = revcomp_2bit(wf, k);
            }
            const bool swapped = wrc < wf;
            const uint32_t v = dict_probe(dx.lv[0], dict_mix(swapped ? wrc : wf), valid, lane, tmp);
            const bool undecided = valid && !(v & 15u) && (v & 16u);
            if (valid && !undecided) {
                n_reads++;
                n_final++;
                res[ql] = dict_encode(v & 15u, swapped, wrc == wf, flags);
            }
            push_lookup(undecided, wf, wrc, ql);
        }
    }
    // ---- windows 1 .. 31 of every task, the wave in lock-step over the window number
    for (uint32_t j = 1; j < kTask && !plain_rest; ++j) {
        bool want = false;
        if (j < mine) {
            if (j < n1) {
                const uint32_t nb = base3(sw0, sw1, sw2, static_cast<uint32_t>(b0 & 31u) + static_cast<uint32_t>(k) - 1u + j);
                f = (f >> 2) | (static_cast<uint64_t>(nb) << (2 * (k - 1)));
                rc = ((rc << 2) | (3u - nb)) & kmask;
                if (anchored) {
                    bool ok;
                    uint64_t tp;   // the text position whose k-mer this window must equal (or mirror)
                    if (dir > 0) {
                        tp = p + j;
                        ok = tp + static_cast<uint64_t>(k) <= dx.text_len;
                        if (ok) {
                            const uint64_t at = tp + static_cast<uint64_t>(k) - 1u - (tbase << 5);   // the window's last base in the text
                            ok = (static_cast<uint32_t>((at < 32u ? tw0 : tw1) >> (2u * (at & 31u))) & 3u) == nb;
                        }
                    } else {
                        ok = p >= j;
                        tp = p - j;
                        if (ok) {
                            const uint64_t at = tp - (tbase << 5);                                  // the mirrored window's first base
                            ok = (static_cast<uint32_t>((at < 32u ? tw0 : tw1) >> (2u * (at & 31u))) & 3u) == 3u - nb;
                        }
                    }
                    if (ok) {
                        const uint64_t at = tp - (abase << 4);
                        const uint64_t aw = at < 16u ? aw0 : (at < 32u ? aw1 : aw2);
                        n_final++;
                        res[t0 + j] = dict_encode(static_cast<uint32_t>(aw >> (4u * (at & 15u))) & 15u, rc < f, rc == f, flags);
                    } else {
                        anchored = false;
                        want = true;
                    }
                } else {
                    want = true;
                }
            } else {   // (the task runs over the end of its read: the windows behind it are another read's)
                const uint64_t q = q0 + j, r = q / u_wpr;
                f = window_kmer(seq, u_start0 + r * u_stride + (q - r * u_wpr), k);
                rc = revcomp_2bit(f, k);
                want = true;
            }
        }
        push_lookup(want, f, rc, t0 + j);
    }
    if (qtail != qhead) lookup_round(qtail - qhead);
    if (len && out_bits) {
        flush_packed_results(res, len, strict != 0u, out, start, lane);
    } else if (len) {
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
        uint32_t v[4] = {n_reads, n_final, n_final, n_side};   // bucket reads; windows answered; searches; position / text / answer reads
        const int slot[4] = {4, 6, 3, 7};
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            uint32_t x = v[qq];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if (lane == 0 && x) {
                atomicAdd(&counters[slot[qq]], static_cast<unsigned long long>(x));
                if (qq == 0 || qq == 3) atomicAdd(&counters[1], static_cast<unsigned long long>(x));   // [1]: every read the kernel makes
            }
        }
    }
}

// ---- construction ---------------------------------------------------------------------------------------------------------
struct DevMem {
    void* p = nullptr;
    ~DevMem() { reset(); }
    void reset() {
        if (p) (void)dev_free(p);
        p = nullptr;
    }
    int alloc(size_t bytes, const char* what) {
        reset();
        if (dev_malloc(&p, bytes ? bytes : 8) != hipSuccess) {
            (void)hipGetLastError();
            set_last_error(std::string("hipMalloc failed for the k-mer dictionary (") + what + ", " + std::to_string(bytes >> 20) + " MiB)");
            return FMSI_HIP_ENOMEM;
        }
        return 0;
    }
    template <class T>
    T* as() const { return static_cast<T*>(p); }
    void* release() {
        void* q = p;
        p = nullptr;
        return q;
    }
};

uint32_t grid_of(uint64_t items) { return static_cast<uint32_t>(std::min<uint64_t>((items + 255) / 256, 0x7FFFFFFFull)); }

// the walk's letters (case = mask) -> 2-bit text words and mask words; one thread per 64 positions
__global__ __launch_bounds__(256) void dict_planes_kernel(const char* __restrict__ chars, uint64_t len, uint64_t n_words64, uint64_t* __restrict__ text2,
                                                          uint64_t* __restrict__ maskw) {
    const uint64_t w = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (w >= n_words64) return;
    uint64_t t0 = 0, t1 = 0, m = 0;
    for (uint32_t b = 0; b < 64u; ++b) {
        const uint64_t p = w * 64u + b;
        if (p >= len) break;
        const uint32_t ch = static_cast<uint8_t>(chars[p]);
        const uint32_t lc = ch | 0x20u;
        const uint64_t c = lc == 'a' ? 0u : (lc == 'c' ? 1u : (lc == 'g' ? 2u : 3u));
        if (b < 32u) t0 |= c << (2u * b);
        else t1 |= c << (2u * (b - 32u));
        if (ch >= 'A' && ch <= 'Z') m |= 1ull << b;   // is_upper = mask 1, src/kmers.h:71-73
    }
    text2[2 * w] = t0;
    text2[2 * w + 1] = t1;
    maskw[w] = m;
}

// text window p -> (hash of its canonical k-mer, mask bit | orientation << 1)
__global__ __launch_bounds__(256) void dict_keys_kernel(const uint64_t* __restrict__ text2, const uint64_t* __restrict__ maskw, uint64_t n_windows, int k,
                                                        uint64_t* __restrict__ keys, uint8_t* __restrict__ vals) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t p = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; p < n_windows; p += stride) {
        const uint64_t f = window_kmer(text2, p, k), rc = revcomp_2bit(f, k);
        const bool swapped = rc < f;
        keys[p] = dict_mix(swapped ? rc : f);
        vals[p] = static_cast<uint8_t>(((maskw[p >> 6] >> (p & 63u)) & 1ull) | (swapped ? 2u : 0u));
    }
}

struct HeadFlag {
    const uint64_t* keys;
    __host__ __device__ uint32_t operator()(uint64_t i) const { return (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u; }
};

// One thread per occurrence of the sorted (hash, mask bit | orientation) pairs.  A key with ONE occurrence (nearly all of them) gets its
// code at once.  The occurrences of a repeated key OR their (orientation, ON / OFF) bit into the key's flag byte -- atomics on the
// containing dword, so a key with a million occurrences (poly-A) costs a million atomics, not a million iterations of one thread --
// and dict_codes_kernel turns the flags into codes: per orientation (0: the occurrence reads the canonical k-mer, 1: its reverse
// complement) code 0 absent, 1 all occurrences OFF, 3 all ON; when ON and OFF occurrences mix, the default mode's answer is 1 and
// `-O`'s is the mask bit of the interval's FIRST row, which only the suffix order knows -- such keys get code 2 for now and go to the
// list the index itself resolves (dict_patch_kernel).
__global__ __launch_bounds__(256) void dict_flags_kernel(const uint64_t* __restrict__ keys, const uint8_t* __restrict__ vals, const uint32_t* __restrict__ uidx,
                                                         uint64_t n, uint64_t* __restrict__ ukeys, uint8_t* __restrict__ ucodes, uint32_t* __restrict__ kflags32) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t key = keys[i];
        const bool head = i == 0 || keys[i - 1] != key, last = i + 1 == n || keys[i + 1] != key;
        const uint32_t u = uidx[i] - (head ? 0u : 1u), v = vals[i];   // (uidx counts the heads before i)
        if (head) ukeys[u] = key;
        if (head && last) ucodes[u] = static_cast<uint8_t>(((v & 1u) ? 3u : 1u) << (2u * (v >> 1)));
        else atomicOr(kflags32 + (u >> 2), (1u << ((v >> 1) * 2u + (v & 1u))) << (8u * (u & 3u)));
    }
}
__global__ __launch_bounds__(256) void dict_codes_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ uidx, uint64_t n,
                                                         const uint8_t* __restrict__ kflags, uint8_t* __restrict__ ucodes, uint64_t* __restrict__ amb_kmers,
                                                         uint32_t* __restrict__ amb_where, uint32_t amb_cap, uint32_t* __restrict__ amb_count) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t key = keys[i];
        if ((i != 0 && keys[i - 1] == key) || i + 1 == n || keys[i + 1] != key) continue;   // heads of repeated keys only
        const uint32_t u = uidx[i], fl = kflags[u];
        uint32_t code = 0, amb = 0;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const uint32_t off = (fl >> (2 * o)) & 1u, on = (fl >> (2 * o + 1)) & 1u;
            code |= (on ? (off ? 2u : 3u) : (off ? 1u : 0u)) << (2 * o);
            if (on && off) amb |= 1u << o;
        }
        ucodes[u] = static_cast<uint8_t>(code);
        if (amb) {
            const uint32_t at = atomicAdd(amb_count, 1u);
            if (at < amb_cap) {
                amb_kmers[at] = dict_unmix(key);   // the canonical k-mer: its forward strand is orientation 0
                amb_where[2 * at] = u;
                amb_where[2 * at + 1] = amb;
            }
        }
    }
}

// `-O` answers of the listed keys from the index's own search (STRICT bytes of query mode ALL: (fwd + 1) | (rc + 1) << 2)
__global__ __launch_bounds__(256) void dict_patch_kernel(const uint8_t* __restrict__ strict_bytes, const uint32_t* __restrict__ amb_where, uint32_t count,
                                                         uint8_t* __restrict__ ucodes) {
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= count) return;
    const uint32_t u = amb_where[2 * a], amb = amb_where[2 * a + 1], byte = strict_bytes[a];
    uint32_t code = ucodes[u];
    if (amb & 1u) code = (code & ~3u) | (2u + ((byte & 3u) == 2u ? 1u : 0u));
    if (amb & 2u) code = (code & ~12u) | ((2u + (((byte >> 2) & 3u) == 2u ? 1u : 0u)) << 2);
    ucodes[u] = static_cast<uint8_t>(code);
}

// sorted distinct hashes -> first / last index of every non-empty bucket
__global__ __launch_bounds__(256) void dict_bounds_kernel(const uint64_t* __restrict__ ukeys, uint64_t count, uint32_t shift, uint32_t* __restrict__ first,
                                                          uint32_t* __restrict__ last) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += stride) {
        const uint64_t b = ukeys[i] >> shift;
        if (i == 0 || (ukeys[i - 1] >> shift) != b) first[b] = static_cast<uint32_t>(i);
        if (i + 1 == count || (ukeys[i + 1] >> shift) != b) last[b] = static_cast<uint32_t>(i);
    }
}

// ... -> slots; the keys a bucket cannot hold go to the next level's list
__global__ __launch_bounds__(256) void dict_fill_kernel(const uint64_t* __restrict__ ukeys, const uint8_t* __restrict__ ucodes, uint64_t count, uint32_t shift,
                                                        const uint32_t* __restrict__ first, const uint32_t* __restrict__ last, uint64_t* __restrict__ slots,
                                                        uint64_t* __restrict__ spill_keys, uint8_t* __restrict__ spill_codes, uint32_t spill_cap,
                                                        uint32_t* __restrict__ spill_count) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += stride) {
        const uint64_t key = ukeys[i], b = key >> shift;
        const uint32_t r = static_cast<uint32_t>(i) - first[b], held = last[b] - first[b] + 1u;
        const bool over = held > kDictSlots;
        if (!over || r < kDictSlots - 1u) {
            slots[b * kDictSlots + r] = (((key << (64u - shift)) >> (60u - shift)) & ~15ull) | ucodes[i];
        } else {
            if (r == kDictSlots - 1u) slots[b * kDictSlots + r] = (static_cast<uint64_t>(kMarkerHi) << 32) | kMarkerLo;
            const uint32_t at = atomicAdd(spill_count, 1u);
            if (at < spill_cap) {
                spill_keys[at] = key;
                spill_codes[at] = ucodes[i];
            }
        }
    }
}

__global__ __launch_bounds__(256) void dict_rehash_kernel(uint64_t* __restrict__ keys, uint64_t count, uint64_t salt) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < count) keys[i] = dict_mix(keys[i] + salt);
}

uint32_t level_bits(uint64_t count, double per_bucket) {
    uint32_t b = 4;
    while (b < 32u && static_cast<double>(count) / static_cast<double>(1ull << b) > per_bucket) ++b;
    return b;
}

#define DICT_TRY(expr)                                                                                    \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            set_last_error(std::string("k-mer dictionary: ") + #expr + ": " + hipGetErrorString(e_));    \
            return FMSI_HIP_EHIP;                                                                         \
        }                                                                                                 \
    } while (0)

// sorted distinct hashes + codes of one level -> its buckets; what does not fit comes back in spill_* (count in *n_spill)
int fill_level(const uint64_t* d_ukeys, const uint8_t* d_ucodes, uint64_t count, uint32_t bits, DevMem& buckets, DevMem& spill_keys, DevMem& spill_codes,
               uint32_t* n_spill) {
    const uint32_t shift = 64u - bits;
    const uint64_t n_buckets = 1ull << bits;
    int rc;
    DevMem first, last, d_spill;
    if ((rc = buckets.alloc(n_buckets * 128, "buckets")) || (rc = first.alloc(n_buckets * 4, "bucket bounds")) || (rc = last.alloc(n_buckets * 4, "bucket bounds")) ||
        (rc = d_spill.alloc(4, "counter")))
        return rc;
    const uint32_t cap = static_cast<uint32_t>(std::min<uint64_t>(count / 8 + 4096, 0xFFFFFFFFull));
    if ((rc = spill_keys.alloc(static_cast<size_t>(cap) * 8, "overflow keys")) || (rc = spill_codes.alloc(cap, "overflow keys"))) return rc;
    DICT_TRY(hipMemset(buckets.p, 0, n_buckets * 128));
    DICT_TRY(hipMemset(d_spill.p, 0, 4));
    if (count) {
        hipLaunchKernelGGL(dict_bounds_kernel, dim3(grid_of(count)), dim3(256), 0, 0, d_ukeys, count, shift, first.as<uint32_t>(), last.as<uint32_t>());
        hipLaunchKernelGGL(dict_fill_kernel, dim3(grid_of(count)), dim3(256), 0, 0, d_ukeys, d_ucodes, count, shift, first.as<uint32_t>(), last.as<uint32_t>(),
                           buckets.as<uint64_t>(), spill_keys.as<uint64_t>(), spill_codes.as<uint8_t>(), cap, d_spill.as<uint32_t>());
        DICT_TRY(hipGetLastError());
    }
    DICT_TRY(hipMemcpy(n_spill, d_spill.p, 4, hipMemcpyDeviceToHost));
    if (*n_spill > cap) {
        set_last_error("k-mer dictionary: a level's overflow exceeds its list (degenerate hash distribution)");
        return FMSI_HIP_ERANGE;
    }
    return 0;
}

template <bool COUNT, bool RUNS>
void launch_dict_t(const fmsi_hip_index* idx, const uint64_t* d_in, uint64_t n, int k, uint32_t flags, uint8_t* o8, unsigned long long* d_counters, hipStream_t s,
                   const uint64_t* d_runs, uint64_t n_runs, const uint64_t* uniform, uint32_t* slot_pos = nullptr) {
    uint64_t chunk = n / (256ull * 4 * 2);
    chunk = std::min<uint64_t>(std::max<uint64_t>(chunk, 128), 4096) & ~127ull;
    const uint64_t waves = (n + chunk - 1) / chunk;
    hipLaunchKernelGGL((dict_lookup_kernel<COUNT, RUNS>), dim3(static_cast<uint32_t>((waves + 3) / 4)), dim3(256), 0, s, idx->dict, d_in, n, k, flags,
                       static_cast<uint32_t>(chunk), o8, d_counters, d_runs, d_runs + n_runs, n_runs, uniform ? uniform[0] : 0ull, uniform ? uniform[1] : 0ull,
                       uniform ? uniform[2] : 0ull, slot_pos);
}

}  // namespace

bool dict_applies(const fmsi_hip_index* idx, int k, uint32_t flags) {
    const uint32_t kernel = flags & FMSI_HIP_KERNEL_MASK;
    return idx->dict.levels != 0 && k == idx->k && !(flags & (FMSI_HIP_ORDERS | FMSI_HIP_NO_PREFIX_TABLE)) &&
           (kernel == FMSI_HIP_KERNEL_AUTO || kernel == FMSI_HIP_KERNEL_DICT);
}

int launch_dict_query(fmsi_hip_index* idx, const uint64_t* d_in, uint64_t n, int k, uint32_t flags, uint8_t* o8, unsigned long long* d_counters, hipStream_t s,
                      const uint64_t* h_runs, uint64_t n_runs, const uint64_t* uniform_runs) {
    const uint64_t* d_runs = nullptr;
    const bool runs = h_runs != nullptr || uniform_runs != nullptr;
    if (h_runs && !uniform_runs) {   // the run table: n_runs starts, n_runs + 1 offsets (a slot of the handle's page-locked ring)
        int rc = ensure_cap(&idx->d_tasks, &idx->d_tasks_cap, (2 * n_runs + 1) * 8);
        if (rc) return rc;
        FMSI_HIP_TRY(hipMemcpyAsync(idx->d_tasks, h_runs, (2 * n_runs + 1) * 8, hipMemcpyHostToDevice, s));
        d_runs = static_cast<const uint64_t*>(idx->d_tasks);
    }
    if (runs) {
        if (d_counters) launch_dict_t<true, true>(idx, d_in, n, k, flags, o8, d_counters, s, d_runs, n_runs, uniform_runs);
        else launch_dict_t<false, true>(idx, d_in, n, k, flags, o8, d_counters, s, d_runs, n_runs, uniform_runs);
    } else {
        if (d_counters) launch_dict_t<true, false>(idx, d_in, n, k, flags, o8, d_counters, s, nullptr, 0, nullptr);
        else launch_dict_t<false, false>(idx, d_in, n, k, flags, o8, d_counters, s, nullptr, 0, nullptr);
    }
    FMSI_HIP_TRY(hipGetLastError());
    return 0;
}

bool dict_chain_applies(const fmsi_hip_index* idx, int k, uint32_t flags) {
    static const bool off = std::getenv("FMSI_HIP_NO_CHAIN") != nullptr;   // A/B switch for measurements
    return !off && !(flags & FMSI_HIP_NO_DICT_CHAIN) && idx->dict.slot_pos != nullptr && dict_applies(idx, k, flags);
}

int launch_dict_chain(fmsi_hip_index* idx, const uint64_t* d_seq, uint64_t n, int k, uint32_t flags, uint8_t* o8, unsigned long long* d_counters, hipStream_t s,
                      const uint64_t* u) {
    const uint64_t waves = (n + 2047) / 2048;
    const dim3 grid(static_cast<uint32_t>((waves + 3) / 4)), block(256);
    if (d_counters) hipLaunchKernelGGL((dict_chain_kernel<true>), grid, block, 0, s, idx->dict, d_seq, n, k, flags, o8, d_counters, u[0], u[1], u[2]);
    else hipLaunchKernelGGL((dict_chain_kernel<false>), grid, block, 0, s, idx->dict, d_seq, n, k, flags, o8, d_counters, u[0], u[1], u[2]);
    FMSI_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace fmsi

using namespace fmsi;

namespace {
// The chained form's extras, by a pass of the lookup kernel over the text's own windows: the answer nibble of every text position
// (the kernel's STRICT | OUT_BITS packing of the raw nibbles) and, per slot, the position of an occurrence.  `text2`: the 2-bit text
// the construction made (taken over when the handle has none of its own).
#include "skdict.inc"

// sorted distinct hashes + codes -> the levels of a dictionary (level 0 at <= load0 keys per bucket, the overflow levels at <= 3); the
// level buffers are handed over in owned[] (freed here on failure), the key arrays are consumed
int build_dict_levels(DevMem& ukeys, DevMem& ucodes, uint64_t n_unique, double load0, DictIndex& dx, void* owned[kDictMaxLevels], uint64_t& bytes_total,
                      uint64_t* keys_per_level, uint64_t& total_slots) {
    int rc;
    auto fail = [&](int code) {
        for (uint32_t lv = 0; lv < kDictMaxLevels; ++lv) {
            if (owned[lv]) (void)dev_free(owned[lv]);
            owned[lv] = nullptr;
        }
        return code;
    };
    total_slots = 0;
    DevMem cur_keys, cur_codes;
    cur_keys.p = ukeys.release();
    cur_codes.p = ucodes.release();
    uint64_t count = n_unique;
    for (uint32_t lv = 0; lv < kDictMaxLevels; ++lv) {
        const uint32_t bits = level_bits(count, lv == 0 ? load0 : 3.0);
        DevMem buckets, spill_keys, spill_codes;
        uint32_t n_spill = 0;
        if ((rc = fill_level(cur_keys.as<uint64_t>(), cur_codes.as<uint8_t>(), count, bits, buckets, spill_keys, spill_codes, &n_spill))) return fail(rc);
        dx.lv[lv].shift = 64u - bits;
        dx.lv[lv].slot_base = total_slots;
        total_slots += (1ull << bits) * kDictSlots;
        dx.lv[lv].buckets = static_cast<const uint4*>(buckets.p);
        owned[lv] = buckets.release();
        bytes_total += (1ull << bits) * 128;
        dx.levels = lv + 1;
        if (keys_per_level) keys_per_level[lv] = count;
        if (n_spill == 0) break;
        if (lv + 1 == kDictMaxLevels) {
            set_last_error("k-mer dictionary: keys left over after the last level");
            return fail(FMSI_HIP_ERANGE);
        }
        // the next level: its own hash of the spilled hashes, sorted
        DevMem sk2, sc2, temp;
        if ((rc = sk2.alloc(static_cast<size_t>(n_spill) * 8, "overflow keys")) || (rc = sc2.alloc(n_spill, "overflow keys"))) return fail(rc);
        hipLaunchKernelGGL(dict_rehash_kernel, dim3((n_spill + 255) / 256), dim3(256), 0, 0, spill_keys.as<uint64_t>(), static_cast<uint64_t>(n_spill), kLevelSalt[lv + 1]);
        rocprim::double_buffer<uint64_t> dk(spill_keys.as<uint64_t>(), sk2.as<uint64_t>());
        rocprim::double_buffer<uint8_t> dv(spill_codes.as<uint8_t>(), sc2.as<uint8_t>());
        size_t tb = 0;
        if (rocprim::radix_sort_pairs(nullptr, tb, dk, dv, static_cast<size_t>(n_spill), 0, 64, 0) != hipSuccess || (rc = temp.alloc(tb, "sort scratch")) ||
            rocprim::radix_sort_pairs(temp.p, tb, dk, dv, static_cast<size_t>(n_spill), 0, 64, 0) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            if (!rc) set_last_error("k-mer dictionary: sorting an overflow level failed");
            return fail(rc ? rc : FMSI_HIP_EHIP);
        }
        const bool in_first = dk.current() == spill_keys.as<uint64_t>();
        cur_keys.reset();
        cur_codes.reset();
        cur_keys.p = in_first ? spill_keys.release() : sk2.release();
        cur_codes.p = (dv.current() == spill_codes.as<uint8_t>()) ? spill_codes.release() : sc2.release();
        count = n_spill;
    }
    if (hipDeviceSynchronize() != hipSuccess) {
        set_last_error("k-mer dictionary: device work failed");
        return fail(FMSI_HIP_EHIP);
    }
    return 0;
}

int add_chain_extras(fmsi_hip_index* idx, DevMem& text2, uint64_t total_slots) {
    const int k = idx->k;
    const uint64_t len = idx->n - 1, windows = len - static_cast<uint64_t>(k) + 1;
    int rc;
    DevMem posans, slot_pos;
    const uint64_t pa_words = (windows + 15) / 16 + 4;
    if ((rc = posans.alloc(pa_words * 8, "answers per text position")) || (rc = slot_pos.alloc(total_slots * 4, "positions per slot"))) return rc;
    DICT_TRY(hipMemset(posans.p, 0, pa_words * 8));
    DICT_TRY(hipMemset(slot_pos.p, 0xFF, total_slots * 4));
    const uint64_t* d_text = idx->d_text2 ? static_cast<const uint64_t*>(idx->d_text2) : text2.as<uint64_t>();
    const uint64_t whole_text[3] = {0, windows, windows};   // one run: every window of the superstring
    launch_dict_t<false, true>(idx, d_text, windows, k, FMSI_HIP_STRICT | FMSI_HIP_OUT_BITS | kFlagSelf, posans.as<uint8_t>(), nullptr, 0, nullptr, 0, whole_text,
                               slot_pos.as<uint32_t>());
    DICT_TRY(hipGetLastError());
    DICT_TRY(hipDeviceSynchronize());
    uint64_t extra = pa_words * 8 + total_slots * 4;
    if (!idx->d_text2) {   // (a handle built on the device already holds the same text)
        idx->d_text2 = text2.release();
        idx->text_len = len;
        idx->dict_owns_text = true;
        extra += ((len >> 6) + 2) * 16;
    }
    idx->d_posans = posans.release();
    idx->d_slot_pos = slot_pos.release();
    idx->dict.text2 = static_cast<const uint64_t*>(idx->d_text2);
    idx->dict.posans = static_cast<const uint64_t*>(idx->d_posans);
    idx->dict.slot_pos = static_cast<const uint32_t*>(idx->d_slot_pos);
    idx->dict.text_len = len;
    idx->dict_chain_bytes = extra;
    idx->dict_bytes += extra;
    idx->device_bytes += extra;
    return 0;
}
// The extras are an accelerator of an accelerator: when they cannot be had (no room for posans / slot_pos) the handle keeps the plain
// dictionary -- installed, consistent, used by every later presence query -- and the call SUCCEEDS; fmsi_hip_index_info.dict_chain_bytes
// stays 0 and fmsi_hip_last_error() says why.  (An error here used to leave caller and library in disagreement about the handle.)
int chain_extras_or_warn(fmsi_hip_index* idx, DevMem& text2, uint64_t total_slots) {
    const int rc = add_chain_extras(idx, text2, total_slots);
    if (rc) {
        (void)hipGetLastError();
        set_last_error(std::string("k-mer dictionary: chained lookups are off (") + fmsi_hip_last_error() + "); the plain dictionary is in use");
    }
    return 0;
}
}  // namespace

static int enable_dict_impl(fmsi_hip_index* idx, uint32_t flags) {
    if (!idx) return FMSI_HIP_EINVAL;
    FMSI_HIP_NARROW_ONLY(idx, "the k-mer dictionary");
    if (idx->dict.levels) {
        if (!(flags & FMSI_HIP_DICT_CHAIN) || idx->dict.slot_pos) return 0;
        // the dictionary is there, the chained lookups' extras are asked for now: the text (again from the LF walk when the handle keeps
        // none) and the pass over it
        FMSI_HIP_ON_DEVICE(idx->device);
        DevMem text2;
        if (!idx->d_text2) {
            const uint64_t len = idx->n - 1, n64 = (len >> 6) + 2;
            DevMem chars, maskw;
            int rc;
            if ((rc = chars.alloc(len + 1, "text")) || (rc = export_walk(idx, chars.as<char>(), nullptr)) || (rc = text2.alloc(n64 * 16, "text")) ||
                (rc = maskw.alloc(n64 * 8, "mask")))
                return rc;
            hipLaunchKernelGGL(dict_planes_kernel, dim3(grid_of(n64)), dim3(256), 0, 0, chars.as<char>(), len, n64, text2.as<uint64_t>(), maskw.as<uint64_t>());
            DICT_TRY(hipGetLastError());
            DICT_TRY(hipDeviceSynchronize());
        }
        uint64_t total_slots = 0;
        for (uint32_t lv = 0; lv < idx->dict.levels; ++lv) total_slots = idx->dict.lv[lv].slot_base + (16ull << (64u - idx->dict.lv[lv].shift));
        return chain_extras_or_warn(idx, text2, total_slots);
    }
    PhaseClock clock{"dict"};
    TraceRange trace_range("fmsi: build k-mer dictionary");
    const int k = idx->k;
    const uint64_t n = idx->n, len = n - 1;
    if (k < 1 || k > 32 || len < static_cast<uint64_t>(k)) {
        set_last_error("the k-mer dictionary needs 1 <= k <= 32 and a superstring of at least k bases");
        return FMSI_HIP_EINVAL;
    }
    FMSI_HIP_ON_DEVICE(idx->device);
    const uint64_t windows = len - static_cast<uint64_t>(k) + 1;
    if (windows > kMaxSortItems) {
        set_last_error("the k-mer dictionary is not built for superstrings of more than 2^32 - 8192 k-mers (the sort's limit)");
        return FMSI_HIP_ERANGE;
    }
    int rc;
    // 1. the superstring with its mask, from the index itself (the LF walk of `fmsi export`): any handle can add the dictionary
    DevMem text2, maskw;
    {
        DevMem chars;
        if ((rc = chars.alloc(len + 1, "text"))) return rc;
        if ((rc = export_walk(idx, chars.as<char>(), nullptr))) return rc;
        clock.mark("LF walk (text + mask)");
        const uint64_t n64 = (len >> 6) + 2;
        if ((rc = text2.alloc(n64 * 16, "text")) || (rc = maskw.alloc(n64 * 8, "mask"))) return rc;
        hipLaunchKernelGGL(dict_planes_kernel, dim3(grid_of(n64)), dim3(256), 0, 0, chars.as<char>(), len, n64, text2.as<uint64_t>(), maskw.as<uint64_t>());
        DICT_TRY(hipGetLastError());
        DICT_TRY(hipDeviceSynchronize());
        clock.mark("2-bit text + mask words");
    }
    // 2. one (hash, mask bit | orientation) pair per window of the superstring, sorted by hash
    DevMem ukeys, ucodes;
    uint64_t n_unique = 0;
    {
        DevMem keys_a, keys_b, vals_a, vals_b, temp, uidx;
        if ((rc = keys_a.alloc(windows * 8, "sort keys")) || (rc = keys_b.alloc(windows * 8, "sort keys")) || (rc = vals_a.alloc(windows, "sort values")) ||
            (rc = vals_b.alloc(windows, "sort values")))
            return rc;
        hipLaunchKernelGGL(dict_keys_kernel, dim3(grid_of(windows)), dim3(256), 0, 0, text2.as<uint64_t>(), maskw.as<uint64_t>(), windows, k, keys_a.as<uint64_t>(),
                           vals_a.as<uint8_t>());
        DICT_TRY(hipGetLastError());
        clock.mark("hashes of the text's k-mers");
        rocprim::double_buffer<uint64_t> dk(keys_a.as<uint64_t>(), keys_b.as<uint64_t>());
        rocprim::double_buffer<uint8_t> dv(vals_a.as<uint8_t>(), vals_b.as<uint8_t>());
        size_t tb = 0;
        DICT_TRY(rocprim::radix_sort_pairs(nullptr, tb, dk, dv, static_cast<size_t>(windows), 0, 64, 0));
        if ((rc = temp.alloc(tb, "sort scratch"))) return rc;
        DICT_TRY(rocprim::radix_sort_pairs(temp.p, tb, dk, dv, static_cast<size_t>(windows), 0, 64, 0));
        clock.mark("radix sort");
        if (!(flags & FMSI_HIP_DICT_CHAIN)) text2.reset();   // (the chained form keeps the text)
        maskw.reset();
        // 3. distinct hashes: index of every occurrence's key among them
        const uint64_t* sk = dk.current();
        const uint8_t* sv = dv.current();
        if ((rc = uidx.alloc((windows + 1) * 4, "key indices"))) return rc;
        auto heads = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), HeadFlag{sk});
        size_t sb = 0;
        DICT_TRY(rocprim::exclusive_scan(nullptr, sb, heads, uidx.as<uint32_t>(), 0u, static_cast<size_t>(windows), rocprim::plus<uint32_t>()));
        if (sb > tb && (rc = temp.alloc(sb, "scan scratch"))) return rc;
        DICT_TRY(rocprim::exclusive_scan(temp.p, sb, heads, uidx.as<uint32_t>(), 0u, static_cast<size_t>(windows), rocprim::plus<uint32_t>()));
        uint32_t last_idx = 0;   // distinct keys before the last occurrence's position; + 1 when that occurrence starts a key of its own
        uint64_t tail[2] = {0, 1};
        DICT_TRY(hipMemcpy(&last_idx, uidx.as<uint32_t>() + (windows - 1), 4, hipMemcpyDeviceToHost));
        if (windows >= 2) DICT_TRY(hipMemcpy(tail, sk + (windows - 2), 16, hipMemcpyDeviceToHost));
        n_unique = static_cast<uint64_t>(last_idx) + (tail[0] != tail[1] ? 1u : 0u);
        clock.mark("distinct keys (scan)");
        // 4. fold the occurrences into answer codes; keys whose `-O` answer depends on the suffix order are listed
        DevMem amb_kmers, amb_where, amb_count;
        const uint32_t amb_cap = static_cast<uint32_t>(std::min<uint64_t>(n_unique, 0x7FFFFFFFull));
        if ((rc = ukeys.alloc(n_unique * 8, "distinct keys")) || (rc = ucodes.alloc(n_unique, "answer codes")) || (rc = amb_count.alloc(4, "counter"))) return rc;
        DICT_TRY(hipMemset(amb_count.p, 0, 4));
        DevMem kflags;
        if ((rc = kflags.alloc(n_unique + 8, "key flags"))) return rc;
        DICT_TRY(hipMemset(kflags.p, 0, n_unique + 8));
        hipLaunchKernelGGL(dict_flags_kernel, dim3(grid_of(windows)), dim3(256), 0, 0, sk, sv, uidx.as<uint32_t>(), windows, ukeys.as<uint64_t>(), ucodes.as<uint8_t>(),
                           kflags.as<uint32_t>());
        DICT_TRY(hipGetLastError());
        // (the list is sized by a first counting pass: mixed masks on repeats only -- usually empty)
        for (int pass = 0; pass < 2; ++pass) {
            uint32_t cap = 0;
            if (pass == 1) {
                DICT_TRY(hipMemcpy(&cap, amb_count.p, 4, hipMemcpyDeviceToHost));
                if (cap == 0) break;
                cap = std::min(cap, amb_cap);
                if ((rc = amb_kmers.alloc(static_cast<size_t>(cap) * 8, "ambiguous keys")) || (rc = amb_where.alloc(static_cast<size_t>(cap) * 8, "ambiguous keys"))) return rc;
                DICT_TRY(hipMemset(amb_count.p, 0, 4));
            }
            hipLaunchKernelGGL(dict_codes_kernel, dim3(grid_of(windows)), dim3(256), 0, 0, sk, uidx.as<uint32_t>(), windows, kflags.as<uint8_t>(), ucodes.as<uint8_t>(),
                               amb_kmers.as<uint64_t>(), amb_where.as<uint32_t>(), cap, amb_count.as<uint32_t>());
            DICT_TRY(hipGetLastError());
            DICT_TRY(hipDeviceSynchronize());
        }
        uint32_t n_amb = 0;
        DICT_TRY(hipMemcpy(&n_amb, amb_count.p, 4, hipMemcpyDeviceToHost));
        clock.mark("answer codes (fold)");
        // the sort's buffers are no longer needed
        keys_a.reset();
        keys_b.reset();
        vals_a.reset();
        vals_b.reset();
        uidx.reset();
        temp.reset();
        if (n_amb) {
            // the index answers for them: first row of the interval, src/fms_index.h:129-131 (the dictionary is not installed yet,
            // so this is the ordinary search)
            DevMem bytes;
            if ((rc = bytes.alloc(n_amb, "ambiguous keys"))) return rc;
            if ((rc = fmsi_hip_query_kmers(idx, amb_kmers.as<uint64_t>(), n_amb, k, FMSI_HIP_MODE_ALL | FMSI_HIP_STRICT | FMSI_HIP_DEVICE_PTRS, bytes.p, nullptr, nullptr)))
                return rc;
            hipLaunchKernelGGL(dict_patch_kernel, dim3((n_amb + 255) / 256), dim3(256), 0, 0, bytes.as<uint8_t>(), amb_where.as<uint32_t>(), n_amb, ucodes.as<uint8_t>());
            DICT_TRY(hipGetLastError());
            DICT_TRY(hipDeviceSynchronize());
        }
        idx->dict_ambiguous = n_amb;
        clock.mark("suffix-order answers");
    }
    // 5. the levels: level 0 at <= 10.5 keys per bucket (FMSI_HIP_DICT_SPARSE: half that), the overflow levels at <= 3
    DictIndex dx{};
    dx.k = k;
    void* owned[kDictMaxLevels] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t bytes_total = 0, total_slots = 0;
    if ((rc = build_dict_levels(ukeys, ucodes, n_unique, (flags & FMSI_HIP_DICT_SPARSE) ? 5.25 : 10.5, dx, owned, bytes_total, idx->dict_keys, total_slots))) return rc;
    clock.mark("levels (fill)");
    idx->dict = dx;
    for (uint32_t lv = 0; lv < kDictMaxLevels; ++lv) idx->d_dict[lv] = owned[lv];
    idx->dict_bytes = bytes_total;
    idx->device_bytes += bytes_total;
    if (flags & FMSI_HIP_DICT_CHAIN) {
        (void)chain_extras_or_warn(idx, text2, total_slots);
        clock.mark("chain data (self pass)");
    }
    return 0;
}


// ---- the super-k-mer dictionary: construction (skdict.inc has the kernels) -------------------------------------------------------------
namespace {
#define SK_TRY(expr)                                                                                            \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) {                                                                                 \
            set_last_error(std::string("super-k-mer dictionary: ") + #expr + ": " + hipGetErrorString(e_));    \
            return FMSI_HIP_EHIP;                                                                               \
        }                                                                                                       \
    } while (0)

struct U32ToU64 {
    const uint32_t* p;
    __host__ __device__ uint64_t operator()(uint64_t i) const { return p[i]; }
};

// needs the k-mer dictionary with its chained-lookup extras (the 2-bit text and the answer nibble of every text position)
int build_skdict(fmsi_hip_index* idx) {
    PhaseClock clock{"skdict"};
    TraceRange trace_range("fmsi: build super-k-mer dictionary");
    const int k = idx->k;
    if (k < 23 || k > 32) {
        set_last_error("the super-k-mer dictionary is built for 23 <= k <= 32 (minimizers of 21 or 23 bases)");
        return FMSI_HIP_ERANGE;
    }
    const int m = k <= 31 ? 21 : 23;   // odd (an m-mer is never its own reverse complement), k - m <= 10
    const uint32_t mbits = 2u * static_cast<uint32_t>(m);
    const uint64_t len = idx->n - 1, windows = len - static_cast<uint64_t>(k) + 1;
    const uint64_t* text2 = idx->dict.text2;
    const uint64_t* posans = idx->dict.posans;
    if (!text2 || !posans) {
        set_last_error("the super-k-mer dictionary needs the k-mer dictionary's chained-lookup extras");
        return FMSI_HIP_EINVAL;
    }
    int rc;
    // 1. every window's minimizer occurrence(s); records per window; their places
    DevMem tl, tr, cnt, off;
    if ((rc = tl.alloc(windows * 4, "minimizer positions")) || (rc = tr.alloc(windows * 4, "minimizer positions")) || (rc = cnt.alloc(windows * 4, "record counts")) ||
        (rc = off.alloc((windows + 1) * 8, "record offsets")))
        return rc;
    hipLaunchKernelGGL(sk_minimizer_kernel, dim3(grid_of(windows)), dim3(256), 0, 0, text2, windows, k, m, tl.as<uint32_t>(), tr.as<uint32_t>());
    hipLaunchKernelGGL(sk_count_kernel, dim3(grid_of(windows)), dim3(256), 0, 0, tl.as<uint32_t>(), tr.as<uint32_t>(), windows, cnt.as<uint32_t>());
    SK_TRY(hipGetLastError());
    uint64_t n_rec = 0;
    {
        auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), U32ToU64{cnt.as<uint32_t>()});
        size_t tb = 0;
        DevMem temp;
        SK_TRY(rocprim::exclusive_scan(nullptr, tb, in, off.as<uint64_t>(), 0ull, static_cast<size_t>(windows), rocprim::plus<uint64_t>()));
        if ((rc = temp.alloc(tb, "scan scratch"))) return rc;
        SK_TRY(rocprim::exclusive_scan(temp.p, tb, in, off.as<uint64_t>(), 0ull, static_cast<size_t>(windows), rocprim::plus<uint64_t>()));
        uint64_t last_off = 0;
        uint32_t last_cnt = 0;
        SK_TRY(hipMemcpy(&last_off, off.as<uint64_t>() + (windows - 1), 8, hipMemcpyDeviceToHost));
        SK_TRY(hipMemcpy(&last_cnt, cnt.as<uint32_t>() + (windows - 1), 4, hipMemcpyDeviceToHost));
        n_rec = last_off + last_cnt;
    }
    cnt.reset();
    clock.mark("minimizers of the text's windows");
    if (n_rec > (3ull << 29) || n_rec >= kMaxSortItems) {   // (a low-complexity text: every window a record of its own)
        set_last_error("the super-k-mer dictionary is not built for this text (" + std::to_string(n_rec) + " super-k-mer records before deduplication)");
        return FMSI_HIP_ERANGE;
    }
    // 2. the records, sorted by (minimizer, neighbourhood); duplicates (repeats) collapse
    DevMem keys_a, keys_b, vals_a, vals_b;
    if ((rc = keys_a.alloc(n_rec * 16, "records")) || (rc = keys_b.alloc(n_rec * 16, "records")) || (rc = vals_a.alloc(n_rec * 8, "records")) ||
        (rc = vals_b.alloc(n_rec * 8, "records")))
        return rc;
    hipLaunchKernelGGL(sk_emit_kernel, dim3(grid_of(windows)), dim3(256), 0, 0, text2, posans, tl.as<uint32_t>(), tr.as<uint32_t>(), off.as<uint64_t>(), windows, k, m,
                       keys_a.as<__uint128_t>(), vals_a.as<uint64_t>());
    SK_TRY(hipGetLastError());
    SK_TRY(hipDeviceSynchronize());
    tl.reset();
    tr.reset();
    off.reset();
    clock.mark("records");
    uint64_t n_uniq = 0;
    DevMem ukeys, uvals;
    {
        rocprim::double_buffer<__uint128_t> dk(keys_a.as<__uint128_t>(), keys_b.as<__uint128_t>());
        rocprim::double_buffer<uint64_t> dv(vals_a.as<uint64_t>(), vals_b.as<uint64_t>());
        size_t tb = 0;
        DevMem temp;
        SK_TRY(rocprim::radix_sort_pairs(nullptr, tb, dk, dv, static_cast<size_t>(n_rec), 0, mbits + 48, 0));
        if ((rc = temp.alloc(tb, "sort scratch"))) return rc;
        SK_TRY(rocprim::radix_sort_pairs(temp.p, tb, dk, dv, static_cast<size_t>(n_rec), 0, mbits + 48, 0));
        clock.mark("radix sort");
        const __uint128_t* sk_ = dk.current();
        const uint64_t* sv = dv.current();
        DevMem flag, d_n;
        if ((rc = flag.alloc(n_rec, "flags")) || (rc = d_n.alloc(8, "counter"))) return rc;
        hipLaunchKernelGGL(sk_unique_flags_kernel, dim3(grid_of(n_rec)), dim3(256), 0, 0, sk_, n_rec, flag.as<uint8_t>());
        SK_TRY(hipGetLastError());
        __uint128_t* other_k = sk_ == keys_a.as<__uint128_t>() ? keys_b.as<__uint128_t>() : keys_a.as<__uint128_t>();
        uint64_t* other_v = sv == vals_a.as<uint64_t>() ? vals_b.as<uint64_t>() : vals_a.as<uint64_t>();
        size_t t1 = 0, t2 = 0;
        SK_TRY(rocprim::select(nullptr, t1, sk_, flag.as<uint8_t>(), other_k, d_n.as<size_t>(), static_cast<size_t>(n_rec)));
        SK_TRY(rocprim::select(nullptr, t2, sv, flag.as<uint8_t>(), other_v, d_n.as<size_t>(), static_cast<size_t>(n_rec)));
        if ((rc = temp.alloc(std::max(t1, t2), "select scratch"))) return rc;
        SK_TRY(rocprim::select(temp.p, t1, sk_, flag.as<uint8_t>(), other_k, d_n.as<size_t>(), static_cast<size_t>(n_rec)));
        SK_TRY(rocprim::select(temp.p, t2, sv, flag.as<uint8_t>(), other_v, d_n.as<size_t>(), static_cast<size_t>(n_rec)));
        size_t got = 0;
        SK_TRY(hipMemcpy(&got, d_n.p, sizeof(size_t), hipMemcpyDeviceToHost));
        n_uniq = got;
        // keep the compacted copies, free the sorted ones
        if (other_k == keys_a.as<__uint128_t>()) { ukeys.p = keys_a.release(); keys_b.reset(); } else { ukeys.p = keys_b.release(); keys_a.reset(); }
        if (other_v == vals_a.as<uint64_t>()) { uvals.p = vals_a.release(); vals_b.reset(); } else { uvals.p = vals_b.release(); vals_a.reset(); }
    }
    clock.mark("distinct records");
    // 3. the buckets: about 3.5 records each
    // (a record keeps 36 bits of the hashed minimizer beside the bucket number -- 16 in word A, 20 in word B: at least mbits - 36 of its
    // bits must select the bucket, however few records there are; k = 32 on a text of a few kb used to start below that)
    uint32_t bits = mbits > 42u ? mbits - 36u : 6u;
    while (bits < 31u && bits + 6u < mbits && static_cast<double>(n_uniq) / static_cast<double>(1ull << bits) > 3.5) ++bits;
    const uint64_t n_buckets = 1ull << bits;
    const uint32_t shift = mbits - bits;
    DevMem buckets, first, last, stats;
    if ((rc = buckets.alloc(n_buckets * 128, "buckets")) || (rc = first.alloc(n_buckets * 4, "bucket bounds")) || (rc = last.alloc(n_buckets * 4, "bucket bounds")) ||
        (rc = stats.alloc(32, "counters")))
        return rc;
    SK_TRY(hipMemset(buckets.p, 0, n_buckets * 128));
    SK_TRY(hipMemset(first.p, 0xFF, n_buckets * 4));
    SK_TRY(hipMemset(stats.p, 0, 32));
    if (n_uniq) {
        hipLaunchKernelGGL(sk_bounds_kernel, dim3(grid_of(n_uniq)), dim3(256), 0, 0, ukeys.as<__uint128_t>(), n_uniq, shift, first.as<uint32_t>(), last.as<uint32_t>());
        hipLaunchKernelGGL(sk_fill_kernel, dim3(grid_of(n_buckets)), dim3(256), 0, 0, ukeys.as<__uint128_t>(), uvals.as<uint64_t>(), first.as<uint32_t>(), last.as<uint32_t>(),
                           n_buckets, shift, buckets.as<uint64_t>(), stats.as<unsigned long long>());
        SK_TRY(hipGetLastError());
    }
    unsigned long long h_stats[4] = {0, 0, 0, 0};
    SK_TRY(hipMemcpy(h_stats, stats.p, 32, hipMemcpyDeviceToHost));
    ukeys.reset();
    uvals.reset();
    first.reset();
    last.reset();
    clock.mark("buckets");
    SkIndex sk{};
    sk.buckets = static_cast<const uint4*>(buckets.p);
    sk.bits = bits;
    sk.k = k;
    sk.m = m;
    // 4. the text through the structure itself: what the records do not hold goes to the aux dictionary; every record's nibble is checked
    DevMem aux_keys, aux_codes, aux_cnt;
    if ((rc = aux_cnt.alloc(16, "counter"))) return rc;
    uint64_t n_aux = 0;
    const uint64_t waves = (windows + kSkChunk - 1) / kSkChunk;
    for (int pass = 0; pass < 2; ++pass) {
        uint64_t cap = 0;
        if (pass == 1) {
            if (n_aux == 0) break;
            cap = n_aux;
            if ((rc = aux_keys.alloc(cap * 8, "aux keys")) || (rc = aux_codes.alloc(cap, "aux codes"))) return rc;
        }
        SK_TRY(hipMemset(aux_cnt.p, 0, 16));
        // twice: a window whose smallest score sits at several m-mers (a repeated m-mer, or two m-mers the 19-bit score does not tell apart) is
        // looked up under the LEFTMOST of them when it is asked as the text has it and under the RIGHTMOST when the other strand is asked --
        // two different minimizers, two different buckets, and either may be heavy or overflowed on its own.  (The lists may repeat k-mers:
        // they are sorted and made unique below.)
        for (uint32_t side : {0u, kFlagSelfRight}) {
            hipLaunchKernelGGL((sk_runs_kernel<false, true, false>), dim3(static_cast<uint32_t>((waves + 1) / 2)), dim3(128), 0, 0, sk, DictIndex{}, text2, windows,
                               static_cast<uint32_t>(kFlagSelf | side), nullptr, nullptr, 0ull, windows, windows, posans, aux_keys.as<uint64_t>(),
                               aux_codes.as<uint8_t>(), aux_cnt.as<unsigned long long>(), cap, nullptr, nullptr, 0ull, (windows + static_cast<uint64_t>(k) - 2ull) >> 5);
            SK_TRY(hipGetLastError());
        }
        unsigned long long got[2] = {0, 0};
        SK_TRY(hipMemcpy(got, aux_cnt.p, 16, hipMemcpyDeviceToHost));
        if (got[1] != 0) {
            set_last_error("super-k-mer dictionary: " + std::to_string(got[1]) + " of the text's own windows came back with another answer (construction bug)");
            return FMSI_HIP_EHIP;
        }
        n_aux = got[0];
    }
    clock.mark("text through the records (aux list)");
    DictIndex aux{};
    aux.k = k;
    void* owned[kDictMaxLevels] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t aux_bytes = 0, aux_unique = 0;
    if (n_aux) {
        if (n_aux >= kMaxSortItems) {
            set_last_error("super-k-mer dictionary: too many k-mers outside the records");
            return FMSI_HIP_ERANGE;
        }
        DevMem k2, c2, temp, uk, uc, flag, d_n;
        if ((rc = k2.alloc(n_aux * 8, "aux keys")) || (rc = c2.alloc(n_aux, "aux codes"))) return rc;
        rocprim::double_buffer<uint64_t> dk(aux_keys.as<uint64_t>(), k2.as<uint64_t>());
        rocprim::double_buffer<uint8_t> dv(aux_codes.as<uint8_t>(), c2.as<uint8_t>());
        size_t tb = 0;
        SK_TRY(rocprim::radix_sort_pairs(nullptr, tb, dk, dv, static_cast<size_t>(n_aux), 0, 64, 0));
        if ((rc = temp.alloc(tb, "sort scratch"))) return rc;
        SK_TRY(rocprim::radix_sort_pairs(temp.p, tb, dk, dv, static_cast<size_t>(n_aux), 0, 64, 0));
        if ((rc = uk.alloc(n_aux * 8, "aux keys")) || (rc = uc.alloc(n_aux, "aux codes")) || (rc = d_n.alloc(8, "counter"))) return rc;
        auto heads = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), HeadFlag{dk.current()});
        size_t t1 = 0, t2 = 0;
        SK_TRY(rocprim::select(nullptr, t1, dk.current(), heads, uk.as<uint64_t>(), d_n.as<size_t>(), static_cast<size_t>(n_aux)));
        SK_TRY(rocprim::select(nullptr, t2, dv.current(), heads, uc.as<uint8_t>(), d_n.as<size_t>(), static_cast<size_t>(n_aux)));
        if ((rc = temp.alloc(std::max(t1, t2), "select scratch"))) return rc;
        SK_TRY(rocprim::select(temp.p, t1, dk.current(), heads, uk.as<uint64_t>(), d_n.as<size_t>(), static_cast<size_t>(n_aux)));
        SK_TRY(rocprim::select(temp.p, t2, dv.current(), heads, uc.as<uint8_t>(), d_n.as<size_t>(), static_cast<size_t>(n_aux)));
        size_t got = 0;
        SK_TRY(hipMemcpy(&got, d_n.p, sizeof(size_t), hipMemcpyDeviceToHost));
        aux_unique = got;
        aux_keys.reset();
        aux_codes.reset();
        k2.reset();
        c2.reset();
        uint64_t slots = 0;
        if ((rc = build_dict_levels(uk, uc, aux_unique, 10.5, aux, owned, aux_bytes, nullptr, slots))) return rc;
    }
    clock.mark("aux dictionary");
    idx->sk = sk;
    idx->d_sk = buckets.release();
    idx->sk_aux = aux;
    for (uint32_t lv = 0; lv < kDictMaxLevels; ++lv) idx->d_sk_aux[lv] = owned[lv];
    idx->sk_bytes = n_buckets * 128 + aux_bytes;
    idx->sk_records = h_stats[2];
    idx->sk_heavy = h_stats[0];
    idx->sk_overflowed = h_stats[1];
    idx->sk_aux_keys = aux_unique;
    idx->device_bytes += idx->sk_bytes;
    return 0;
}

// the k-mer dictionary and the chained lookups' extras go (FMSI_HIP_DICT_SUPERKMER_ONLY): presence queries over runs take the super-k-mer
// dictionary, everything else the index's own search
void drop_kmer_dict(fmsi_hip_index* idx) {
    for (uint32_t lv = 0; lv < kDictMaxLevels; ++lv) {
        if (idx->d_dict[lv]) (void)dev_free(idx->d_dict[lv]);
        idx->d_dict[lv] = nullptr;
    }
    if (idx->d_posans) (void)dev_free(idx->d_posans);
    if (idx->d_slot_pos) (void)dev_free(idx->d_slot_pos);
    idx->d_posans = idx->d_slot_pos = nullptr;
    if (idx->dict_owns_text && idx->d_text2) {
        (void)dev_free(idx->d_text2);
        idx->d_text2 = nullptr;
        idx->dict_owns_text = false;
    }
    idx->device_bytes -= idx->dict_bytes;
    idx->dict_bytes = idx->dict_chain_bytes = 0;
    idx->dict = DictIndex{};
}
}  // namespace

extern "C" int fmsi_hip_index_enable_dict(fmsi_hip_index* idx, uint32_t flags) {
    if (idx) (void)::fmsi::wait_background(idx);   // (the prefix table's signature pass, when it runs beside the first queries)
    ::fmsi::ScratchScope scratch_scope_(idx ? idx->device : -1);   // (devpool.h: the outermost builder returns the block cache to the driver)
    if (!idx) return FMSI_HIP_EINVAL;
    const bool want_sk = (flags & (FMSI_HIP_DICT_SUPERKMER | FMSI_HIP_DICT_SUPERKMER_ONLY)) != 0u;
    if (!want_sk) return enable_dict_impl(idx, flags & (FMSI_HIP_DICT_SPARSE | FMSI_HIP_DICT_CHAIN));
    FMSI_HIP_NARROW_ONLY(idx, "the super-k-mer dictionary");
    if (idx->sk.bits) {   // (the records stand: _ONLY on such a handle still lets the k-mer dictionary go)
        if ((flags & FMSI_HIP_DICT_SUPERKMER_ONLY) && idx->dict.levels) {
            FMSI_HIP_ON_DEVICE(idx->device);
            drop_kmer_dict(idx);
        }
        return 0;
    }
    int rc = enable_dict_impl(idx, (flags & FMSI_HIP_DICT_SPARSE) | FMSI_HIP_DICT_CHAIN);   // (its per-position answers are what the records are made of)
    if (rc) return rc;
    if (!idx->dict.posans) {
        set_last_error(std::string("the super-k-mer dictionary needs the chained lookups' extras: ") + fmsi_hip_last_error());
        return FMSI_HIP_ENOMEM;
    }
    FMSI_HIP_ON_DEVICE(idx->device);
    if ((rc = build_skdict(idx))) return rc;
    if (flags & FMSI_HIP_DICT_SUPERKMER_ONLY) drop_kmer_dict(idx);
    return 0;
}

namespace fmsi {
bool skdict_applies(const fmsi_hip_index* idx, int k, uint32_t flags) {
    const uint32_t kernel = flags & FMSI_HIP_KERNEL_MASK;
    return idx->sk.bits != 0 && k == idx->k && !(flags & (FMSI_HIP_ORDERS | FMSI_HIP_NO_PREFIX_TABLE | FMSI_HIP_NO_SUPERKMER)) &&
           (kernel == FMSI_HIP_KERNEL_AUTO || kernel == FMSI_HIP_KERNEL_DICT);
}

// ---- isolated k-mers on the super-k-mer dictionary (round 4) ---------------------------------------------------------------------
// fmsi_hip_query_kmers on a handle that keeps the records ALONE used to send every k-mer through sk_runs_kernel as a one-window read;
// that kernel is built around what consecutive windows share, and an isolated k-mer shares nothing.  Here a lane owns a k-mer from the
// 8-byte load to the result byte, in the shape of dict_lookup_kernel: a wave takes 64 k-mers per round; each lane finds its k-mer's
// minimizer (the k - m + 1 m-mers of the k-mer and of its reverse complement side by side: one score each, the smallest key -- score, then
// position -- wins, exactly sk_runs_kernel's order), turns the k-mer into the minimizer's orientation, hashes the canonical m-mer to bucket +
// remainder; the 64 buckets arrive eight lanes x 16 bytes at a time and EACH OF THE EIGHT LANES matches the one record it holds against
// the owner's remainder and flank string (sk_match: all bases); three xor-shuffles fold the answer nibble / heavy / overflow flags to the
// owner.  A k-mer of a heavy minimizer or of an overflowed bucket that no record holds queues up for the aux dictionary's ordinary
// lookups, 64 at a time.  Results collect in LDS and leave as 16 bytes per lane or bit-packed.
template <bool COUNT>
__global__ __launch_bounds__(256) void sk_kmers_kernel(SkIndex sk, DictIndex aux, const uint64_t* __restrict__ kmers, uint64_t n_windows, uint32_t flags, uint32_t chunk,
                                                       uint8_t* __restrict__ out, unsigned long long* __restrict__ counters) {
    constexpr uint32_t kMaxChunk = 4096u, kQueue = 128u;
    __shared__ uint64_t s_qh[4][kQueue];     // k-mers waiting for the aux dictionary: canonical k-mer's hash ...
    __shared__ uint32_t s_qi[4][kQueue];     // ... and index in the chunk | (rc < f) << 16 | (rc == f) << 17
    __shared__ uint32_t s_tmp[4][64];
    __shared__ __attribute__((aligned(16))) uint8_t s_res[4][kMaxChunk];
    const int k = sk.k, m = sk.m;
    const uint32_t strict = flags & FMSI_HIP_STRICT;
    const bool out_bits = (flags & FMSI_HIP_OUT_BITS) != 0u;
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
    uint64_t* qh = s_qh[wib];
    uint32_t* qi = s_qi[wib];
    uint32_t* tmp = s_tmp[wib];
    uint8_t* res = s_res[wib];
    const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const uint64_t start = wave * chunk;
    const uint32_t len = start < n_windows ? static_cast<uint32_t>(min(static_cast<uint64_t>(chunk), n_windows - start)) : 0u;
    const uint32_t nflank = static_cast<uint32_t>(k - m), mbits = 2u * static_cast<uint32_t>(m), rshift = mbits - sk.bits;
    const uint64_t mmask = (1ull << mbits) - 1ull;
    const uint32_t grp = lane >> 3, piece = lane & 7u;
    uint32_t qhead = 0, qtail = 0;
    uint32_t n_reads = 0, n_final = 0, n_aux = 0;

    auto aux_round = [&](uint32_t items) {
        bool active = lane < items;
        const uint32_t pos = (qhead + lane) & (kQueue - 1u);
        uint64_t h = active ? qh[pos] : 0ull;
        const uint32_t info = active ? qi[pos] : 0u;
        uint32_t ans = 0;
        for (uint32_t lv = 0; lv < aux.levels && __any(active); ++lv) {
            if (lv) h = dict_mix(h + kLevelSalt[lv]);
            const uint32_t v = dict_probe(aux.lv[lv], h, active, lane, tmp);
            if (active) {
                n_aux++;
                if ((v & 15u) || !(v & 16u)) {
                    ans = v & 15u;
                    active = false;
                }
            }
        }
        if (lane < items) {
            n_final++;
            res[info & 0xFFFFu] = dict_encode(ans, (info >> 16) & 1u, (info >> 17) & 1u, flags);
        }
        qhead += items;
    };

    for (uint32_t base = 0; base < len; base += 64u) {
        const uint32_t ql = base + lane;
        const bool valid = ql < len;
        const uint64_t f = valid ? __builtin_nontemporal_load(kmers + start + ql) : 0ull;
        const uint64_t rc = revcomp_2bit(f, k);
        const bool swapped = rc < f, pal = rc == f;
        // ---- the minimizer: m-mer p of f is x, its reverse complement is m-mer (k - m - p) of rc
        uint32_t best = ~0u;
#pragma unroll 1
        for (uint32_t p = 0; p <= nflank; ++p) {
            const uint64_t x = (f >> (2u * p)) & mmask, y = (rc >> (2u * (nflank - p))) & mmask;
            const bool flip = y < x;
            const uint32_t key = (sk_score(flip ? y : x, mbits) << 5) | (p << 1) | (flip ? 1u : 0u);   // leftmost of equal scores: the smaller p
            best = min(best, key);
        }
        const uint32_t p = (best >> 1) & 15u;
        const bool flip = (best & 1u) != 0u;
        const uint64_t g = flip ? rc : f;                       // the k-mer in its minimizer's orientation
        const uint32_t pc = flip ? nflank - p : p;              // ... and the minimizer's offset in it
        const uint64_t xc = (g >> (2u * pc)) & mmask;           // canonical as it stands there
        const uint64_t G = sk_mix(sk_mix(xc, mbits, kSkScoreSalt), mbits, kSkBucketSalt);
        const uint32_t bidx = valid ? static_cast<uint32_t>(G >> rshift) : 0u;
        const uint64_t rem = G & ((1ull << rshift) - 1ull);
        const uint32_t rem_a = static_cast<uint32_t>(rem >> 20), rem_b = static_cast<uint32_t>(rem) & 0xFFFFFu;
        const uint32_t wf = sk_flanks(g, k, m, pc);
        // ---- the buckets: eight lanes per bucket, eight buckets per instruction (a lane without a k-mer asks for bucket 0)
        uint4 v[8];
#pragma unroll
        for (uint32_t gq = 0; gq < 8; ++gq) {
            const uint32_t bo = static_cast<uint32_t>(__shfl(static_cast<int>(bidx), static_cast<int>(8u * gq + grp), 64));
            v[gq] = load16_nt(sk.buckets + (static_cast<size_t>(bo) << 3) + piece);
        }
#pragma unroll
        for (uint32_t gq = 0; gq < 8; ++gq) {   // lane l holds record (l & 7) of owner 8 gq + (l >> 3)
            const int owner = static_cast<int>(8u * gq + grp);
            const uint32_t oa = static_cast<uint32_t>(__shfl(static_cast<int>(rem_a), owner, 64));
            const uint32_t ob = static_cast<uint32_t>(__shfl(static_cast<int>(rem_b), owner, 64));
            const uint32_t owf = static_cast<uint32_t>(__shfl(static_cast<int>(wf), owner, 64));
            const uint32_t opc = static_cast<uint32_t>(__shfl(static_cast<int>(pc), owner, 64));
            const uint4 r = v[gq];
            uint32_t got = 0;   // answer nibble | heavy << 4 | overflow << 5
            const bool mine = (r.z & 15u) != 0u && (r.y >> 16) == oa && (r.w >> 12) == ob;
            if (mine) {
                if (((r.y >> 8) & 15u) == 15u) got = 16u;
                else got = sk_match((static_cast<uint64_t>(r.y) << 32) | r.x, (static_cast<uint64_t>(r.w) << 32) | r.z, owf, opc, nflank);
            }
            if ((r.x & r.y) == ~0u && (r.z | r.w) == 0u) got |= 32u;
            got |= static_cast<uint32_t>(__shfl_xor(static_cast<int>(got), 1, 64));
            got |= static_cast<uint32_t>(__shfl_xor(static_cast<int>(got), 2, 64));
            got |= static_cast<uint32_t>(__shfl_xor(static_cast<int>(got), 4, 64));
            if (piece == 0u) tmp[owner] = got;
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t mine = tmp[lane];
        __builtin_amdgcn_wave_barrier();
        bool slow = false;
        if (valid) {
            n_reads++;
            const uint32_t ans = mine & 15u;
            // a heavy minimizer's k-mers live in the aux dictionary; so may a k-mer that an overflowed bucket does not hold
            if ((mine & 16u) || (!ans && (mine & 32u))) {
                slow = true;
            } else {
                n_final++;
                res[ql] = dict_encode(ans, swapped, pal, flags);
            }
        }
        const uint64_t mk = __ballot(slow);
        if (mk) {
            if (slow) {
                const uint32_t pos = (qtail + lanes_below(mk)) & (kQueue - 1u);
                qh[pos] = dict_mix(swapped ? rc : f);
                qi[pos] = ql | (swapped ? 0x10000u : 0u) | (pal ? 0x20000u : 0u);
            }
            qtail += static_cast<uint32_t>(__popcll(mk));
            while (qtail - qhead >= 64u) aux_round(64u);
        }
    }
    if (qtail != qhead) aux_round(qtail - qhead);
    if (len && out_bits) {
        flush_packed_results(res, len, strict != 0u, out, start, lane);
    } else if (len) {
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
        uint32_t vv[3] = {n_reads + n_aux, n_final, n_final};   // bucket reads (records + aux); windows answered; searches (one settles both strands)
        const int slot[3] = {4, 6, 3};
#pragma unroll
        for (int qq = 0; qq < 3; ++qq) {
            uint32_t x = vv[qq];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if (lane == 0 && x) {
                atomicAdd(&counters[slot[qq]], static_cast<unsigned long long>(x));
                if (qq == 0) atomicAdd(&counters[1], static_cast<unsigned long long>(x));
            }
        }
    }
}

// isolated k-mers (8 bytes each) through the records: fmsi_hip_query_kmers on a handle without the k-mer dictionary
int launch_skdict_kmers(fmsi_hip_index* idx, const uint64_t* d_kmers, uint64_t n, uint32_t flags, uint8_t* o8, unsigned long long* d_counters, hipStream_t s) {
    uint64_t chunk = n / (256ull * 8 * 8);
    static const uint64_t chunk_max = [] {   // (FMSI_HIP_SK_KMERS_CHUNK: measurements)
        const char* e = std::getenv("FMSI_HIP_SK_KMERS_CHUNK");
        return e ? std::min<uint64_t>(std::max<uint64_t>(static_cast<uint64_t>(atoll(e)), 128), 4096) : 4096ull;
    }();
    chunk = std::min<uint64_t>(std::max<uint64_t>(chunk, 128), chunk_max) & ~127ull;
    const uint64_t waves = (n + chunk - 1) / chunk;
    const dim3 grid(static_cast<uint32_t>((waves + 3) / 4)), block(256);
    if (d_counters) hipLaunchKernelGGL((sk_kmers_kernel<true>), grid, block, 0, s, idx->sk, idx->sk_aux, d_kmers, n, flags, static_cast<uint32_t>(chunk), o8, d_counters);
    else hipLaunchKernelGGL((sk_kmers_kernel<false>), grid, block, 0, s, idx->sk, idx->sk_aux, d_kmers, n, flags, static_cast<uint32_t>(chunk), o8, d_counters);
    FMSI_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_skdict(fmsi_hip_index* idx, const uint64_t* d_seq, uint64_t seq_words, uint64_t n, int k, uint32_t flags, uint8_t* o8, unsigned long long* d_counters,
                  hipStream_t s, const uint64_t* u, const uint64_t* h_runs, uint64_t n_runs) {
    (void)k;
    const uint64_t* d_runs = nullptr;
    uint64_t uu[3] = {0, 0, 0};
    if (u) {
        uu[0] = u[0]; uu[1] = u[1]; uu[2] = u[2];
    } else {   // the run table: n_runs starts, n_runs + 1 offsets (a slot of the handle's page-locked ring)
        int rc = ensure_cap(&idx->d_tasks, &idx->d_tasks_cap, (2 * n_runs + 1) * 8);
        if (rc) return rc;
        FMSI_HIP_TRY(hipMemcpyAsync(idx->d_tasks, h_runs, (2 * n_runs + 1) * 8, hipMemcpyHostToDevice, s));
        d_runs = static_cast<const uint64_t*>(idx->d_tasks);
    }
    const uint64_t waves = (n + kSkChunk - 1) / kSkChunk;
    const dim3 grid(static_cast<uint32_t>((waves + 1) / 2)), block(128);
    auto go = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, grid, block, 0, s, idx->sk, idx->sk_aux, d_seq, n, flags, o8, d_counters, uu[0], uu[1], uu[2], nullptr, nullptr, nullptr, nullptr, 0ull, d_runs,
                           d_runs ? d_runs + n_runs : nullptr, n_runs, seq_words ? seq_words - 1 : 0);
    };
    const bool strict = (flags & FMSI_HIP_STRICT) != 0u;
    if (d_counters) {
        if (strict) go(sk_runs_kernel<true, false, true>);
        else go(sk_runs_kernel<true, false, false>);
    } else {
        if (strict) go(sk_runs_kernel<false, false, true>);
        else go(sk_runs_kernel<false, false, false>);
    }
    FMSI_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace fmsi
