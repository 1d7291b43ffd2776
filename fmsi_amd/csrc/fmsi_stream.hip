// kLCP streaming queries on the device: counterpart of query_kmers_streaming<>() (reference
// src/fms_index.h:181-254) with extend_range_with_klcp() (:106-115).
//
// The reference walks the windows of a chunk right-to-left on the first strand (fresh k-step search when the
// previous interval is empty, otherwise kLCP extension + ONE LF step), then left-to-right on the other strand
// for the windows still undecided.  Here a *task* is a run of <= 64 consecutive windows; one lane runs a task
// exactly that way (forward strand right-to-left, then reverse-complement strand left-to-right, skipping decided
// windows unless STRICT), keeping the first pass' results in two 64-bit registers.  Intervals are the same as
// fresh searches would give (the kLCP extension yields exactly the (k-1)-mer interval), so results are
// independent of how a sequence is cut into tasks; only the cost changes.
//
// Memory rounds per lane and iteration, as in the refill kernel: one round of independent reads -- the
// interleaved block(s) of the interval (whose mask plane answers the current window and whose planes rank the
// step to the next window) together with the kLCP word(s) around i-1 and j-1.  A positive window therefore
// costs ONE block read + ONE kLCP word instead of k block reads.  Tasks are staged by a preparation kernel as
// 64-byte records (128 bases of sequence + output index), prefetched one iteration ahead and handed out with
// ballot + mbcnt prefix ranks.
//
// Two kernels live here.  query_stream_kernel (rounds 1-2, the text above: 64-byte blocks + separate kLCP words) is kept for A/B runs
// (FMSI_HIP_KERNEL_REFILL / FMSI_HIP_STREAM_V1) and for a handle without the room for the S-block layout.  query_stream2_kernel (round 3:
// 128-byte S-blocks that carry the kLCP plane, lines fetched cooperatively through LDS; round 4: ABSENCE CERTIFICATES -- a fresh search that
// follows a failed window first asks the prefix table for a (t + 4)-mer that the next k - t - 4 windows share, and a string the text does
// not hold settles them all with one request -- the strand probe on the task's first window, `lookup -S` (ORD) and 33 <= k <= 64 (WIDE)
// as instantiations) is what fmsi_hip_query_stream runs, and what the `fmsi` binary runs for every read when the index has its kLCP file.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/fmsi_hip.h"
#include "common.h"
#include "device_fm.h"
#include "index_priv.h"

namespace fmsi {
namespace {

constexpr uint32_t kTaskWindows = 64;

struct alignas(64) TaskRec {   // device-built, one cache line
    uint64_t s[4];             // bases base .. base+127, base+m in bits [2(m%32), +1] of s[m/32]
    uint64_t out;              // result index of the task's first window
    uint32_t n;                // windows in the task (1..64)
    uint32_t pad;
    uint64_t pad2[2];
};
static_assert(sizeof(TaskRec) == 64, "TaskRec layout");

// Runs -> tasks of <= 64 windows.  Per run the host supplies only its first base, first task and first result index
// (24 bytes per run; a 10-kb read is 156 tasks); task t belongs to the run r with task_off[r] <= t < task_off[r + 1].
// (u_wpr != 0: UNIFORM runs -- run r is the u_wpr windows from base u_start0 + r * u_stride, reads of one length back to back -- and there is
// no table to fetch, to stage on the host or to search)
__global__ __launch_bounds__(256) void stream_prep_kernel(const uint64_t* __restrict__ seq, const uint64_t* __restrict__ run_start,
                                                          const uint64_t* __restrict__ task_off, const uint64_t* __restrict__ out_off,
                                                          uint64_t n_runs, uint64_t n_tasks, uint64_t seq_words, TaskRec* __restrict__ rec,
                                                          uint64_t u_start0, uint64_t u_stride, uint64_t u_wpr) {
    const uint64_t t = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= n_tasks) return;
    uint64_t j, run_windows, first_base, first_out;
    if (u_wpr) {
        const uint64_t tpr = (u_wpr + kTaskWindows - 1) / kTaskWindows, r = t / tpr;
        j = t - r * tpr;
        run_windows = u_wpr;
        first_base = u_start0 + r * u_stride;
        first_out = r * u_wpr;
    } else {
        uint64_t lo = 0, hi = n_runs;   // invariant: task_off[lo] <= t < task_off[hi]
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (task_off[mid] <= t) lo = mid;
            else hi = mid;
        }
        j = t - task_off[lo];
        run_windows = out_off[lo + 1] - out_off[lo];
        first_base = run_start[lo];
        first_out = out_off[lo];
    }
    const uint64_t base = first_base + j * kTaskWindows;
    const uint64_t w = base >> 5;
    const uint32_t sh = static_cast<uint32_t>(base & 31u) * 2u;
    TaskRec r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // never reads past the caller's buffer (seq_words = seq_bases/32 + 2)
        uint64_t v = (w + q < seq_words ? seq[w + q] : 0ull) >> sh;
        if (sh && w + q + 1 < seq_words) v |= seq[w + q + 1] << (64u - sh);
        r.s[q] = v;
    }
    r.out = first_out + j * kTaskWindows;
    r.n = static_cast<uint32_t>(min(static_cast<uint64_t>(kTaskWindows), run_windows - j * kTaskWindows));
    r.pad = 0;
    r.pad2[0] = r.pad2[1] = 0;
    rec[t] = r;
}

// slots[task] (one aligned 64-byte line of result bytes per task) -> the caller's contiguous result array: four threads per task, 16 bytes
// each (an aligned load, a store at whatever alignment the task's first window has; the tail of a run's last task byte by byte)
__global__ __launch_bounds__(256) void stream_compact_kernel(const uint8_t* __restrict__ slots, const TaskRec* __restrict__ rec,
                                                             uint64_t n_tasks, uint8_t* __restrict__ out) {
    const uint64_t g = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint64_t t = g >> 2;
    const uint32_t at = static_cast<uint32_t>(g & 3u) * 16u;
    if (t >= n_tasks) return;
    const uint32_t n = rec[t].n;
    if (at >= n) return;
    const uint4 v = *reinterpret_cast<const uint4*>(slots + (t << 6) + at);
    uint8_t* dst = out + rec[t].out + at;
    if (at + 16u <= n) {
        __builtin_memcpy(dst, &v, 16);
    } else {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        for (uint32_t b = 0; b < n - at; ++b) dst[b] = static_cast<uint8_t>(w[b >> 2] >> (8u * (b & 3u)));
    }
}

// The same for UNIFORM runs (reads of one length): a run's tasks are neighbours, so its wpr result bytes are ONE contiguous piece of the
// slots -- slots[r * tpr * 64 ...) -- that goes to out[r * wpr ...): a copy per run, 16 bytes per thread (aligned loads, stores at
// whatever alignment the run's first window has), instead of a thread per byte and a look at the task's record.
__global__ __launch_bounds__(256) void stream_compact_uniform_kernel(const uint8_t* __restrict__ slots, uint32_t wpr, uint32_t tpr, uint64_t n_pieces,
                                                                     uint8_t* __restrict__ out) {
    const uint64_t g = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= n_pieces) return;
    const uint32_t ppr = tpr * 4u;   // 16-byte pieces per run
    const uint64_t r = g / ppr;
    const uint32_t at = static_cast<uint32_t>(g - r * ppr) * 16u;   // first byte of the piece in its run
    if (at >= wpr) return;
    const uint4 v = *reinterpret_cast<const uint4*>(slots + (r * tpr << 6) + at);
    uint8_t* dst = out + r * wpr + at;
    if (at + 16u <= wpr) {
        __builtin_memcpy(dst, &v, 16);
    } else {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        for (uint32_t b = 0; b < wpr - at; ++b) dst[b] = static_cast<uint8_t>(w[b >> 2] >> (8u * (b & 3u)));
    }
}
// ... and with FMSI_HIP_OUT_BITS (one bit per window, no STRICT): a thread per 64-bit word of the output gathers its 64 result bytes -- four
// 16-byte loads at the run's alignment; byte by byte where the word spans two runs -- and packs their low bits: no pass of bytes in between
__global__ __launch_bounds__(256) void stream_compact_uniform_bits_kernel(const uint8_t* __restrict__ slots, uint32_t wpr, uint32_t tpr, uint64_t n_windows,
                                                                          unsigned long long* __restrict__ out) {
    const uint64_t g = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint64_t q0 = g << 6;
    if (q0 >= n_windows) return;
    uint64_t r = q0 / wpr;
    uint32_t w = static_cast<uint32_t>(q0 - r * wpr);
    unsigned long long word = 0;
    if (w + 64u <= wpr && q0 + 64u <= n_windows) {
        const uint8_t* src = slots + (r * tpr << 6) + w;
#pragma unroll
        for (uint32_t d = 0; d < 4; ++d) {
            uint4 v;
            __builtin_memcpy(&v, src + 16u * d, 16);
            const uint64_t lo = (static_cast<uint64_t>(v.y) << 32) | v.x, hi = (static_cast<uint64_t>(v.w) << 32) | v.z;
            const uint64_t bl = ((lo & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56, bh = ((hi & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56;
            word |= (bl | (bh << 8)) << (16u * d);
        }
    } else {
        const uint32_t cnt = static_cast<uint32_t>(min(static_cast<uint64_t>(64u), n_windows - q0));
        for (uint32_t b = 0; b < cnt; ++b) {
            word |= static_cast<unsigned long long>(slots[(r * tpr << 6) + w] & 1u) << b;
            if (++w == wpr) {
                w = 0;
                ++r;
            }
        }
    }
    out[g] = word;
}

// ---- the S-block kernel's slots: a task's FOUR RESULT PLANES (32 bytes: bit w of plane p = bit p of window w's result byte) ----
// Round 5: the kernel no longer spreads a finished task's planes to 64 result bytes itself (some 280 vector instructions per task -- and in a
// wave whose tasks end in different rounds the whole wave ran them in each of those rounds); the passes below, which read the slots anyway,
// take the bits they need.
__device__ __forceinline__ uint32_t spread4(uint32_t nib) { return ((nib & 0xFu) * 0x00204081u) & 0x01010101u; }
// expands the 16 bits [16 d, 16 d + 16) of four bit planes to 16 result bytes (plane p contributes bit p of each byte)
__device__ __forceinline__ uint4 result_bytes16(uint64_t p0, uint64_t p1, uint64_t p2, uint64_t p3, uint32_t d) {
    const uint32_t a = static_cast<uint32_t>(p0 >> (16u * d)) & 0xFFFFu, b = static_cast<uint32_t>(p1 >> (16u * d)) & 0xFFFFu;
    const uint32_t c = static_cast<uint32_t>(p2 >> (16u * d)) & 0xFFFFu, e = static_cast<uint32_t>(p3 >> (16u * d)) & 0xFFFFu;
    uint4 r;
    r.x = spread4(a) | (spread4(b) << 1) | (spread4(c) << 2) | (spread4(e) << 3);
    r.y = spread4(a >> 4) | (spread4(b >> 4) << 1) | (spread4(c >> 4) << 2) | (spread4(e >> 4) << 3);
    r.z = spread4(a >> 8) | (spread4(b >> 8) << 1) | (spread4(c >> 8) << 2) | (spread4(e >> 8) << 3);
    r.w = spread4(a >> 12) | (spread4(b >> 12) << 1) | (spread4(c >> 12) << 2) | (spread4(e >> 12) << 3);
    return r;
}
__device__ __forceinline__ void store_piece(uint8_t* dst, const uint4& v, uint32_t cnt) {   // cnt of the 16 bytes, at any alignment
    if (cnt >= 16u) {
        __builtin_memcpy(dst, &v, 16);
    } else {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        for (uint32_t b = 0; b < cnt; ++b) dst[b] = static_cast<uint8_t>(w[b >> 2] >> (8u * (b & 3u)));
    }
}
// planes -> the caller's result bytes: four threads per task, 16 windows each
__global__ __launch_bounds__(256) void stream_planes_bytes_kernel(const uint64_t* __restrict__ planes, const TaskRec* __restrict__ rec, uint64_t n_tasks,
                                                                  uint8_t* __restrict__ out) {
    const uint64_t g = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint64_t t = g >> 2;
    const uint32_t d = static_cast<uint32_t>(g & 3u);
    if (t >= n_tasks) return;
    const uint32_t n = rec[t].n;
    if (16u * d >= n) return;
    const uint64_t* p = planes + 4 * t;
    store_piece(out + rec[t].out + 16u * d, result_bytes16(p[0], p[1], p[2], p[3], d), n - 16u * d);
}
// ... of UNIFORM runs (run r = tasks [r * tpr, (r + 1) * tpr), windows [r * wpr, (r + 1) * wpr)): no look at the records
__global__ __launch_bounds__(256) void stream_planes_bytes_uniform_kernel(const uint64_t* __restrict__ planes, uint32_t wpr, uint32_t tpr, uint64_t n_pieces,
                                                                          uint8_t* __restrict__ out) {
    const uint64_t g = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= n_pieces) return;
    const uint32_t ppr = tpr * 4u;   // 16-window pieces per run
    const uint64_t r = g / ppr;
    const uint32_t piece = static_cast<uint32_t>(g - r * ppr), at = piece * 16u;
    if (at >= wpr) return;
    const uint64_t* p = planes + 4 * (r * tpr + (piece >> 2));
    store_piece(out + r * wpr + at, result_bytes16(p[0], p[1], p[2], p[3], piece & 3u), wpr - at);
}
// ... and FMSI_HIP_OUT_BITS without STRICT (one bit per window: plane 0): a thread per 64-bit word of the output, which is one or two tasks'
// plane words shifted together, or, where runs end inside it, a few such pieces
__global__ __launch_bounds__(256) void stream_planes_bits_uniform_kernel(const uint64_t* __restrict__ planes, uint32_t wpr, uint32_t tpr, uint64_t n_windows,
                                                                         unsigned long long* __restrict__ out) {
    const uint64_t g = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint64_t q0 = g << 6;
    if (q0 >= n_windows) return;
    uint64_t r = q0 / wpr;
    uint32_t w = static_cast<uint32_t>(q0 - r * wpr);
    const uint32_t want = static_cast<uint32_t>(min(static_cast<uint64_t>(64u), n_windows - q0));
    unsigned long long word = 0;
    for (uint32_t got = 0; got < want;) {
        const uint32_t o = w & 63u;
        const uint32_t take = min(min(want - got, 64u - o), wpr - w);   // to the end of the word, of the task's plane, of the run
        const uint64_t bits = (planes[4 * (r * tpr + (w >> 6))] >> o) & (take == 64u ? ~0ull : ((1ull << take) - 1ull));
        word |= bits << got;
        got += take;
        w += take;
        if (w == wpr) {
            w = 0;
            ++r;
        }
    }
    out[g] = word;
}

enum : uint32_t { SS_IDLE = 0, SS_LOOKUP = 1, SS_LF = 2, SS_HAVE = 3, SS_EXTSCAN = 4, SS_EXTLF = 5, SS_CERT = 6, SS_DONE = 7, SS_PASS1 = 8, SS_FRESH_CERT = 9, SS_FRESH_PLAIN = 10, SS_DIED = 11 };

struct StreamCounters {
    uint32_t lf_steps = 0, blocks = 0, probes = 0, searches = 0, lookups = 0, klcp_words = 0;
};

// 2-bit base m (0..127) of a task's sequence words
__device__ __forceinline__ uint32_t task_base(uint64_t s0, uint64_t s1, uint64_t s2, uint64_t s3, uint32_t m) {
    const uint32_t q = m >> 5;
    const uint64_t lo = (q & 1u) ? s1 : s0, hi = (q & 1u) ? s3 : s2;
    const uint64_t w = (q & 2u) ? hi : lo;
    return static_cast<uint32_t>(w >> (2u * (m & 31u))) & 3u;
}
// k bases starting at base m (m + k <= 128), base m in bits 0..1
__device__ __forceinline__ uint64_t task_kmer(uint64_t s0, uint64_t s1, uint64_t s2, uint64_t s3, uint32_t m, int k) {
    const uint32_t q = m >> 5, sh = (m & 31u) * 2u;
    const uint64_t w0 = q == 0 ? s0 : (q == 1 ? s1 : (q == 2 ? s2 : s3));
    const uint64_t w1 = q == 0 ? s1 : (q == 1 ? s2 : (q == 2 ? s3 : 0ull));
    uint64_t f = w0 >> sh;
    if (sh) f |= w1 << (64u - sh);
    return k == 32 ? f : (f & ((1ull << (2 * k)) - 1ull));
}

// WIDE (33 <= k <= 64, the reference's __uint128_t k-mers): no k-mer register; the bases of the current window are taken
// from the task's 128 bases when needed -- base t of the forward k-mer of window w is S[w + t], of its reverse
// complement 3 - S[w + k-1-t].
template <bool COUNT, bool PTABLE, bool WIDE>
__global__ __launch_bounds__(256) void query_stream_kernel(DevIndex ix, const uint64_t* __restrict__ klcp, const TaskRec* __restrict__ tasks,
                                                           uint64_t n_tasks, int k, uint32_t flags, uint32_t chunk,
                                                           uint4* __restrict__ slots, long long* __restrict__ out_orders,
                                                           unsigned long long* __restrict__ counters) {
    // A task's <= 64 result bytes are collected in the lane's LDS row and leave as ONE aligned 64-byte line
    // (slots[task]); stream_compact_kernel moves them to their final positions.  One store request per window would
    // otherwise be a third of a streamed window's memory requests.
    __shared__ uint4 s_res[256][4];
    uint8_t* my_res = reinterpret_cast<uint8_t*>(s_res[threadIdx.x]);
    const uint32_t mode_all = flags & FMSI_HIP_MODE_ALL;
    const uint32_t strict = flags & FMSI_HIP_STRICT;
    const uint32_t orders = flags & FMSI_HIP_ORDERS;
    const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    uint64_t next = wave * chunk;
    const uint64_t end = min(n_tasks, next + static_cast<uint64_t>(chunk));
    const int T = PTABLE ? static_cast<int>(ix.ptable_len) : 0;
    const uint64_t kmask = (WIDE || k == 32) ? ~0ull : ((1ull << (2 * k)) - 1ull);

    // current task
    uint64_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, obase = 0;
    uint32_t n = 0;
    // prefetched task
    uint64_t ns0 = 0, ns1 = 0, ns2 = 0, ns3 = 0, nobase = 0;
    uint32_t nn = 0;
    bool has_next = false;
    // walk state
    uint64_t kmer = 0;            // k-mer of the current window on the current strand (base t in bits [2t, 2t+1])
    uint64_t rf_lo = 0, rf_hi = 0;   // pass-0 result + 1 of window w: bit w of rf_lo / rf_hi (0 = empty interval)
    uint64_t todo = 0;            // pass 1: windows still to visit
    uint32_t i = 0, j = 0, st = SS_IDLE, pass = 0, widx = 0;
    // Strand order of the task.  Pass 0 walks ALL windows on one strand, pass 1 the undecided ones on the other.
    // Default: forward strand first (right-to-left), like the reference without its predictor.  Without STRICT the
    // answer does not depend on the order, so when the task's last window is absent on the forward strand the task's FIRST
    // window is searched as reverse complement (the probe: one extra search per task): if it is there the read is taken to
    // be reverse-complement oriented and the RC strand goes first (left-to-right, from the probe's interval on) -- otherwise
    // every window of such a read would pay a failed forward search before the RC pass streams through it.  This is the job
    // the reference gives its strand predictor.  (Rounds 1-3 probed the LAST window and searched the first one again.)
    bool cur_rc = false;          // strand of the current pass
    bool probing = false;         // the RC probe of the first window is running
    uint32_t special = 0;         // set by window_done for advance(): 1 start the probe
    uint64_t task_id = 0, ntask_id = 0;
    uint32_t scan_i = 0, scan_j = 0;   // EXTSCAN: sides still extending
    int t = 0;
    StreamCounters cn;

    // ---- the current window's k-mer on the current strand ----
    auto kmer_base = [&](int tt) -> uint32_t {        // base tt (0 = first)
        if (WIDE) return !cur_rc ? task_base(s0, s1, s2, s3, widx + static_cast<uint32_t>(tt))
                                   : 3u - task_base(s0, s1, s2, s3, widx + static_cast<uint32_t>(k - 1 - tt));
        return static_cast<uint32_t>(kmer >> (2 * tt)) & 3u;
    };
    auto kmer_table_index = [&](int T_) -> uint32_t {   // its last T_ bases
        if (WIDE) return static_cast<uint32_t>(!cur_rc ? task_kmer(s0, s1, s2, s3, widx + static_cast<uint32_t>(k - T_), T_)
                                                         : revcomp_2bit(task_kmer(s0, s1, s2, s3, widx, T_), T_));
        return static_cast<uint32_t>(kmer >> (2 * (k - T_))) & static_cast<uint32_t>((1ull << (2 * T_)) - 1ull);
    };
    auto load_kmer = [&]() {                          // (re)load it for window widx
        if (WIDE) return;
        const uint64_t f = task_kmer(s0, s1, s2, s3, widx, k);
        kmer = !cur_rc ? f : revcomp_2bit(f, k);
    };
    auto fresh = [&]() {   // start a fresh k-step search for `kmer`
        if (COUNT) cn.searches++;
        if (PTABLE) {
            st = SS_LOOKUP;
        } else {
            i = 0;
            j = ix.n;
            t = k - 1;
            st = SS_LF;
        }
    };
    auto final_byte = [&](int f, int r) -> uint8_t {
        if (strict) return static_cast<uint8_t>((f + 1) | ((r + 1) << 2));
        if (mode_all) return static_cast<uint8_t>((f != -1 ? f : r) == 1);
        return static_cast<uint8_t>(f == 1 || r == 1);
    };
    auto end_task = [&]() {
        if (!orders) {
            uint4* dst = slots + (task_id << 2);
            dst[0] = s_res[threadIdx.x][0];
            dst[1] = s_res[threadIdx.x][1];
            dst[2] = s_res[threadIdx.x][2];
            dst[3] = s_res[threadIdx.x][3];
        }
        st = SS_IDLE;
    };
    // begin pass 1 (the other strand: reverse complement = windows ascending, forward = descending) or finish the task
    auto begin_pass1 = [&]() {
        pass = 1;
        cur_rc = !cur_rc;
        const uint64_t all = n == 64 ? ~0ull : ((1ull << n) - 1ull);
        if (strict) todo = all;
        else if (orders || !mode_all) todo = all & ~(rf_hi & ~rf_lo);   // OR / orders: only a forward result of 1 decides (rf = 2)
        else todo = all & ~(rf_lo | rf_hi);                            // ALL: any non-empty forward interval decides
        if (!todo) {
            end_task();
            return;
        }
        if (cur_rc) {
            widx = static_cast<uint32_t>(__ffsll(static_cast<unsigned long long>(todo))) - 1u;
            todo &= todo - 1;
        } else {
            widx = 63u - static_cast<uint32_t>(__clzll(static_cast<long long>(todo)));
            todo &= ~(1ull << widx);
        }
        load_kmer();
        fresh();
    };
    // the current window's result on the current strand is known
    auto window_done = [&](int res, long long ord) {
        if (!strict && pass == 0 && n > 1u) {
            if (probing) {
                probing = false;
                if (res == -1) {          // absent there, too: forward strand first after all, from the last window on (its forward result: -1)
                    cur_rc = false;
                    widx = n - 1u;
                    load_kmer();
                    ord = -1;
                }
                // (else: the probe was the FIRST window of a pass 0 on the reverse-complement strand -- its result is recorded below, the pass goes on from it)
            } else if (!cur_rc && res == -1 && widx == n - 1u) {
                probing = true;
                special = 1;
                return;
            }
        }
        if (pass == 0) {
            const uint64_t bit = 1ull << widx;
            const uint32_t v = static_cast<uint32_t>(res + 1);
            if (v & 1u) rf_lo |= bit;
            if (v & 2u) rf_hi |= bit;
            if (orders) {
                if (strict) out_orders[2 * (obase + widx)] = ord;
                else out_orders[obase + widx] = ord;
            } else if (!strict) {
                // decided windows are final now; undecided ones are rewritten by pass 1
                my_res[widx] = final_byte(res, -1);
            }
        } else {
            const int f = static_cast<int>(((rf_lo >> widx) & 1ull) | (((rf_hi >> widx) & 1ull) << 1)) - 1;
            if (orders) {
                if (strict) out_orders[2 * (obase + widx) + 1] = ord;
                else out_orders[obase + widx] = ord;   // visited only when the forward strand had no order
            } else {
                my_res[widx] = final_byte(f, res);
            }
        }
    };
    // move to the next window of the pass; `have` = the current window's interval [i, j) is non-empty.
    // Returns true when the walk continues by streaming (extension + one LF step from [i, j)).
    auto shift_fwd = [&](uint32_t w) {      // forward k-mer of window w from that of window w + 1: prepend S[w]
        if (!WIDE) kmer = ((kmer << 2) | task_base(s0, s1, s2, s3, w)) & kmask;
    };
    auto shift_rc = [&](uint32_t w) {       // RC k-mer of window w + 1 from that of window w: prepend comp(S[w + k])
        if (!WIDE) kmer = ((kmer << 2) | (3u - task_base(s0, s1, s2, s3, w + static_cast<uint32_t>(k)))) & kmask;
    };
    auto advance = [&](bool have) -> bool {
        if (special) {   // the probe: the task's FIRST window on the reverse-complement strand (where a pass on that strand starts)
            special = 0;
            cur_rc = true;
            widx = 0;
            load_kmer();
            fresh();
            return false;
        }
        if (pass == 0) {
            if (!cur_rc) {
                if (widx == 0) {
                    begin_pass1();
                    return false;
                }
                --widx;
                shift_fwd(widx);
            } else {
                if (widx + 1u == n) {
                    begin_pass1();
                    return false;
                }
                shift_rc(widx);
                ++widx;
            }
            if (have) return true;
            fresh();
            return false;
        }
        if (!todo) {
            end_task();
            return false;
        }
        uint32_t nw;
        bool adjacent;
        if (cur_rc) {
            nw = static_cast<uint32_t>(__ffsll(static_cast<unsigned long long>(todo))) - 1u;
            todo &= todo - 1;
            adjacent = nw == widx + 1u;
            if (adjacent) shift_rc(widx);
        } else {
            nw = 63u - static_cast<uint32_t>(__clzll(static_cast<long long>(todo)));
            todo &= ~(1ull << nw);
            adjacent = nw + 1u == widx;
            if (adjacent) shift_fwd(nw);
        }
        widx = nw;
        if (adjacent) {
            if (have) return true;
        } else {   // skipped windows: the reference resets the interval (fms_index.h:212-216)
            load_kmer();
        }
        fresh();
        return false;
    };

    for (;;) {
        if (st == SS_IDLE && has_next) {
            s0 = ns0; s1 = ns1; s2 = ns2; s3 = ns3; obase = nobase; n = nn;
            task_id = ntask_id;
            has_next = false;
            pass = 0;
            cur_rc = false;
            probing = false;
            special = 0;
            rf_lo = rf_hi = 0;
            widx = n - 1u;
            load_kmer();
            fresh();
        }
        {
            const bool want = !has_next;
            const uint64_t m = __ballot(want);
            if (m) {
                const uint64_t cand = next + lanes_below(m);
                if (want && cand < end) {
                    const uint4* p = reinterpret_cast<const uint4*>(tasks + cand);
                    const uint4 r0 = p[0], r1 = p[1], r2 = p[2];
                    ns0 = (static_cast<uint64_t>(r0.y) << 32) | r0.x;
                    ns1 = (static_cast<uint64_t>(r0.w) << 32) | r0.z;
                    ns2 = (static_cast<uint64_t>(r1.y) << 32) | r1.x;
                    ns3 = (static_cast<uint64_t>(r1.w) << 32) | r1.z;
                    nobase = (static_cast<uint64_t>(r2.y) << 32) | r2.x;
                    nn = r2.z;
                    ntask_id = cand;
                    has_next = true;
                }
                next += static_cast<uint64_t>(__popcll(m));
            }
        }
        if (!__any(st != SS_IDLE)) {
            if (!__any(has_next)) break;
            continue;
        }
        {   // every lane of the wave takes part (idle lanes with all predicates false): the table lines are fetched by lane
            // pairs, so a lane whose partner is idle still needs the partner's loads and shuffles
            // ---- one round of reads ----
            const uint4* pi = ix.blocks;
            const uint4* pj = ix.blocks;
            uint32_t bi = 0, bj = 0, tidx = 0;
            const bool need_blocks = st != SS_IDLE && st != SS_EXTSCAN;
            const bool lk = PTABLE && st == SS_LOOKUP;
            if (lk) {
                tidx = kmer_table_index(T);
            } else if (need_blocks) {
                bi = i >> kBlockRowsLog2;
                bj = (j - 1u) >> kBlockRowsLog2;
                pi = ix.blocks + (static_cast<size_t>(bi) << 2);
                pj = ix.blocks + (static_cast<size_t>(bj) << 2);
            }
            const bool two = need_blocks && !lk && pj != pi;
            // Table lines are fetched by LANE PAIRS: both lanes of a pair load their 32-byte half of the line of each lane
            // of the pair that is looking up (two adjacent lanes on one line are one request; 64 lanes on 64 different
            // lines of a 34 GB table run at a third of the rate -- measured in fmsi_pair.hip's lookup kernel).
            uint4 te0 = make_uint4(0, 0, 0, 0), te1 = te0, to0 = te0, to1 = te0;
            uint32_t t_even = 0, t_odd = 0;
            if (PTABLE) {
                const uint32_t ln = threadIdx.x & 63u, sub = ln & 1u;
                const uint64_t lkm = __ballot(lk);
                if (lkm) {
                    t_even = static_cast<uint32_t>(__shfl(static_cast<int>(tidx), static_cast<int>(ln & ~1u), 64));
                    t_odd = static_cast<uint32_t>(__shfl(static_cast<int>(tidx), static_cast<int>(ln | 1u), 64));
                    if ((lkm >> (ln & ~1u)) & 1ull) {
                        const uint4* p4 = reinterpret_cast<const uint4*>(ix.ptable + (t_even & ~7u)) + sub * 2u;
                        te0 = p4[0];
                        te1 = p4[1];
                    }
                    if ((lkm >> (ln | 1u)) & 1ull) {
                        const uint4* p4 = reinterpret_cast<const uint4*>(ix.ptable + (t_odd & ~7u)) + sub * 2u;
                        to0 = p4[0];
                        to1 = p4[1];
                    }
                }
            }
            uint4 b0 = make_uint4(0, 0, 0, 0), b1 = b0, b2 = b0, b3 = b0;
            uint4 a0 = b0, a1 = b0, a2 = b0, a3 = b0;
            uint64_t kwi = 0, kwj = 0;
            // kLCP words: around i-1 / j-1 for a window that will be extended, or the next words of a running scan
            const bool more = pass == 0 ? (cur_rc ? widx + 1u < n : widx > 0u)
                                        : (todo != 0 && (cur_rc ? (static_cast<uint32_t>(__ffsll(static_cast<unsigned long long>(todo))) - 1u) == widx + 1u
                                                                : 64u - static_cast<uint32_t>(__clzll(static_cast<long long>(todo))) == widx));
            const bool want_klcp = st == SS_HAVE && more;
            uint32_t wi_idx = 0, wj_idx = 0;
            if (want_klcp) {
                wi_idx = (i - 1u) >> 6;
                wj_idx = (j - 1u) >> 6;
            } else if (st == SS_EXTSCAN) {
                wi_idx = (i - 1u) >> 6;   // i, j already moved to the frontier of the scan
                wj_idx = (j - 1u) >> 6;
            }
            if (two) {
                b0 = pj[0];
                b1 = pj[1];
                b2 = pj[2];
                b3 = pj[3];
            }
            if (need_blocks && !lk) {
                a0 = pi[0];
                a1 = pi[1];
                a2 = pi[2];
                a3 = pi[3];
            }
            if (want_klcp || (st == SS_EXTSCAN && scan_i)) kwi = klcp[wi_idx];
            if (want_klcp || (st == SS_EXTSCAN && scan_j)) kwj = klcp[wj_idx];
            if (need_blocks && !two) {
                b0 = a0;
                b1 = a1;
                b2 = a2;
                b3 = a3;
            }
            if (COUNT) {
                cn.blocks += (need_blocks ? 1u : 0u) + (two ? 1u : 0u);
                cn.klcp_words += (want_klcp ? 1u + (wi_idx != wj_idx) : 0u) + (st == SS_EXTSCAN ? scan_i + scan_j : 0u);
            }

            // ---- arithmetic ----
            bool do_lf = false;       // perform the LF step of the walk with the loaded blocks
            uint32_t c = 0;
            // the looked-up entries travel from the half that holds them to their owners: entry e of a line is slot e & 3 of
            // the 32 bytes held by lane e >> 2 of the pair
            uint32_t ent_x = 0, ent_y = 0;
            if (PTABLE && __any(lk)) {
                const uint32_t ln = threadIdx.x & 63u, sub = ln & 1u, pb = ln & ~1u;
                const uint32_t ee = t_even & 7u, eo = t_odd & 7u;
                const uint4 ve = (ee & 2u) ? te1 : te0, vo = (eo & 2u) ? to1 : to0;
                const uint32_t vex = (ee & 1u) ? ve.z : ve.x, vey = (ee & 1u) ? ve.w : ve.y;
                const uint32_t vox = (eo & 1u) ? vo.z : vo.x, voy = (eo & 1u) ? vo.w : vo.y;
                const int src = static_cast<int>(pb | ((sub ? eo : ee) >> 2));
                const uint32_t gex = static_cast<uint32_t>(__shfl(static_cast<int>(vex), src, 64)), gey = static_cast<uint32_t>(__shfl(static_cast<int>(vey), src, 64));
                const uint32_t gox = static_cast<uint32_t>(__shfl(static_cast<int>(vox), src, 64)), goy = static_cast<uint32_t>(__shfl(static_cast<int>(voy), src, 64));
                ent_x = sub ? gox : gex;
                ent_y = sub ? goy : gey;
            }
            if (st == SS_LOOKUP) {
                if (COUNT) cn.lookups++;
                t = k - 1 - T;
                uint32_t q4 = 0;
                if (t >= 3) q4 = kmer_base(t - 3) | (kmer_base(t - 2) << 2) | (kmer_base(t - 1) << 4) | (kmer_base(t) << 6);
                decode_table_entry(ent_x, ent_y, ix.ptable_packed, t >= 3, q4, i, j);
                if (i == j) {
                    window_done(-1, -1);
                    advance(false);
                } else {
                    st = t < 0 ? SS_HAVE : SS_LF;
                }
            } else if (st == SS_LF) {
                do_lf = true;
                c = kmer_base(t);
            } else if (st == SS_EXTLF) {
                do_lf = true;
                c = kmer_base(0);   // the k-mer already is the next window's: its first base is the new one
            } else if (st == SS_HAVE) {
                // result of the current window from the mask plane (infer_presence, fms_index.h:126-144)
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
                    if (bj == bi) ones = popc_prefix128(a3, pj_off) - popc_prefix128(a3, pi_off);
                    else ones = (ix.mrank[bj] + popc_prefix128(b3, pj_off)) - (ix.mrank[bi] + popc_prefix128(a3, pi_off));
                    res = ones > 0u;
                    if (orders && res) ord = static_cast<long long>(ix.mrank[bi] + popc_prefix128(a3, pi_off));
                }
                window_done(res, ord);
                if (advance(true)) {
                    // extend_range_with_klcp (fms_index.h:106-115): j grows while klcp[j-1], i shrinks while klcp[i-1]
                    const uint32_t pj = (j - 1u) & 63u, pi_ = (i - 1u) & 63u;
                    const uint64_t up = kwj >> pj;                       // bit 0 = klcp[j-1]
                    const uint32_t run_j = static_cast<uint32_t>(__ffsll(static_cast<unsigned long long>(~up))) - 1u;   // consecutive ones; ~up != 0 unless all 64 set
                    const uint32_t lim_j = 64u - pj;
                    const uint64_t dn = kwi << (63u - pi_);              // bit 63 = klcp[i-1]
                    const uint32_t run_i = static_cast<uint32_t>(__clzll(static_cast<long long>(~dn)));
                    const uint32_t lim_i = pi_ + 1u;
                    const uint32_t rj = (~up == 0ull || run_j > lim_j) ? lim_j : run_j;
                    const uint32_t ri = (~dn == 0ull || run_i > lim_i) ? lim_i : run_i;
                    scan_j = rj == lim_j ? 1u : 0u;   // ran into the word boundary: keep scanning in the next word
                    scan_i = (ri == lim_i && i - ri > 0u) ? 1u : 0u;
                    if (rj == 0u && ri == 0u) {
                        do_lf = true;                 // common case: same interval, blocks already loaded
                        c = kmer_base(0);
                    } else {
                        j += rj;
                        i -= ri;
                        st = (scan_i | scan_j) ? SS_EXTSCAN : SS_EXTLF;
                    }
                }
            } else if (st == SS_EXTSCAN) {
                if (scan_j) {
                    const uint32_t pj = (j - 1u) & 63u;   // 0 by construction
                    const uint64_t up = kwj >> pj;
                    const uint32_t run = static_cast<uint32_t>(__ffsll(static_cast<unsigned long long>(~up))) - 1u;
                    const uint32_t lim = 64u - pj;
                    const uint32_t r = (~up == 0ull || run > lim) ? lim : run;
                    j += r;
                    scan_j = r == lim ? 1u : 0u;
                }
                if (scan_i) {
                    const uint32_t pi_ = (i - 1u) & 63u;  // 63 by construction
                    const uint64_t dn = kwi << (63u - pi_);
                    const uint32_t run = static_cast<uint32_t>(__clzll(static_cast<long long>(~dn)));
                    const uint32_t lim = pi_ + 1u;
                    const uint32_t r = (~dn == 0ull || run > lim) ? lim : run;
                    i -= r;
                    scan_i = (r == lim && i > 0u) ? 1u : 0u;
                }
                if (!(scan_i | scan_j)) st = SS_EXTLF;
            }
            if (do_lf) {
                uint32_t ri = rank_in_block(a0, a1, a2, c, i & (kBlockRows - 1u));
                uint32_t rj = rank_in_block(b0, b1, b2, c, j - (bj << kBlockRowsLog2));
                if (c == 0u) {
                    ri -= (i > ix.dollar);
                    rj -= (j > ix.dollar);
                }
                const uint32_t base = sel4(c, ix.C0, ix.C1, ix.C2, ix.C3);
                i = base + ri;
                j = base + rj;
                if (COUNT) cn.lf_steps++;
                if (st == SS_LF) {
                    --t;
                    if (i == j) {
                        window_done(-1, -1);
                        advance(false);
                    } else if (t < 0) {
                        st = SS_HAVE;
                    }
                } else {   // the single step of a streamed window (from SS_HAVE or SS_EXTLF)
                    if (i == j) {
                        window_done(-1, -1);
                        advance(false);
                    } else {
                        st = SS_HAVE;
                    }
                }
            }
        }
    }
    if (COUNT) {
        uint32_t v[7] = {cn.lf_steps, cn.blocks, cn.probes, cn.searches, cn.lookups, cn.klcp_words, 0};
        const int slot[6] = {0, 1, 2, 3, 4, 5};
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            uint32_t x = v[q];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if ((threadIdx.x & 63) == 0 && x) atomicAdd(&counters[slot[q]], static_cast<unsigned long long>(x));
        }
    }
}

// ======================================================================================================================
// Round 3: the streaming kernel over a layout of its own ("S-blocks"); round 4: certificates, orders, k up to 64 (further down).
//
// What bound the kernel above (SQ counters + tools/gather_bench): a streamed window cost its lane FOUR 16-byte loads
// of its own 64-byte block plus one or two 8-byte kLCP words -- the memory path charges per load instruction and distinct line, so
// that is 5-6 units per window where one would do -- and some 950 VALU instructions per wave round went into a six-state machine
// that kept its result bytes in LDS.  Here:
//   * one S-block = 128 bytes = 192 rows carrying EVERYTHING a streamed window needs -- cumulative symbol counts, counts after 64 and
//     128 rows (a rank is one 64-bit popcount, not four masked 32-bit ones), the hi / lo / mask planes AND the kLCP plane of its rows
//     (plus the kLCP bit of the row before the block and the cumulative mask rank) -- so a positive window is ONE 128-byte request;
//   * the lines are fetched cooperatively: eight lanes load 16 bytes each of one lane's line in ONE instruction (eight instructions
//     serve the wave's 64 lines), pass them through LDS (144-byte pitch: conflict-free b128 reads) and the owner reads the few words
//     it needs -- one unit per line, as in the dictionary's kernel;
//   * the rare second line of an interval that straddles two blocks is read by its lane directly (six small loads);
//   * prefix-table entries are 8-byte loads of the owning lane (one instruction, one unit per entry);
//   * a task's results live in registers (two bit planes per strand) and leave as one 64-byte line per task.
// Control flow -- tasks of <= 64 windows, pass 0 on one strand, pass 1 over the undecided windows of the other, the reverse-complement
// probe of the task's last window -- is the kernel's above, and so are the answers (tests/test_gpu_stream.py: stream2 == stream == runs).
constexpr uint32_t kStreamNoCert = 1u << 30, kStreamCertPlain = 1u << 31, kStreamUtil = 1u << 29;   // internal bits of the kernel's flags word (FMSI_HIP_STREAM_CERT=0 / 1: measurements)
constexpr uint32_t kSRows = 192;
constexpr uint32_t kSDwords = 32;
constexpr uint32_t kSPitch = 8;   // uint4 per staged line in LDS: 128 bytes (LineView: where its pieces are)

// S-blocks from the single-step blocks + the plain kLCP words; one thread per S-block
__global__ __launch_bounds__(256) void sblock_build_kernel(const uint32_t* __restrict__ blocks, const uint32_t* __restrict__ mrank, uint64_t n_blocks,
                                                           const uint32_t* __restrict__ klcp32, uint64_t klcp_dwords, uint64_t n_sblocks,
                                                           uint32_t* __restrict__ out) {
    const uint64_t b = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (b >= n_sblocks) return;
    uint32_t hi[6], lo[6], mk[6], kl[6];
#pragma unroll
    for (uint32_t d = 0; d < 6; ++d) {
        const uint64_t g = 6 * b + d;             // dword of the planes (32 rows each)
        const uint64_t sb = g >> 2;
        const uint32_t w = static_cast<uint32_t>(g & 3u);
        const bool in = sb < n_blocks;
        hi[d] = in ? blocks[sb * kBlockDwords + 4 + w] : 0u;
        lo[d] = in ? blocks[sb * kBlockDwords + 8 + w] : 0u;
        mk[d] = in ? blocks[sb * kBlockDwords + 12 + w] : 0u;
        kl[d] = g < klcp_dwords ? klcp32[g] : 0u;
    }
    // counts before row 192 b: through source block (192 b) >> 7 at offset 0 or 64
    const uint64_t row = b * kSRows;
    const uint64_t sb0 = row >> kBlockRowsLog2;
    const uint32_t off = static_cast<uint32_t>(row & (kBlockRows - 1u));
    uint32_t cnt[4] = {0, 0, 0, 0}, mr = 0;
    if (sb0 < n_blocks) {
        const uint32_t* src = blocks + sb0 * kBlockDwords;
        mr = mrank[sb0];
        for (uint32_t c = 0; c < 4; ++c) cnt[c] = src[c];
        if (off) {   // off == 64: the first two plane dwords of the source block come before this S-block
            for (uint32_t w = 0; w < 2; ++w) {
                const uint32_t h = src[4 + w], l = src[8 + w];
                cnt[0] += __popc(~h & ~l);
                cnt[1] += __popc(~h & l);
                cnt[2] += __popc(h & ~l);
                cnt[3] += __popc(h & l);
                mr += __popc(src[12 + w]);
            }
        }
    }
    uint32_t sub[2] = {0, 0}, msub = 0;
    {
        uint32_t a[4] = {0, 0, 0, 0}, m = 0;
        for (uint32_t d = 0; d < 4; ++d) {
            a[0] += __popc(~hi[d] & ~lo[d]);
            a[1] += __popc(~hi[d] & lo[d]);
            a[2] += __popc(hi[d] & ~lo[d]);
            a[3] += __popc(hi[d] & lo[d]);
            m += __popc(mk[d]);
            if (d == 1) {
                sub[0] = a[0] | (a[1] << 8) | (a[2] << 16) | (a[3] << 24);
                msub = m;
            }
        }
        sub[1] = a[0] | (a[1] << 8) | (a[2] << 16) | (a[3] << 24);
        msub |= m << 8;
    }
    uint32_t prev_klcp = 0;
    if (row > 0) {
        const uint64_t g = (row - 1) >> 5;
        prev_klcp = g < klcp_dwords ? (klcp32[g] >> ((row - 1) & 31u)) & 1u : 0u;
    }
    uint32_t* o = out + b * kSDwords;
    o[0] = cnt[0]; o[1] = cnt[1]; o[2] = cnt[2]; o[3] = cnt[3];
    o[4] = sub[0]; o[5] = sub[1]; o[6] = mr; o[7] = msub | (prev_klcp << 31);
#pragma unroll
    for (uint32_t d = 0; d < 6; ++d) {
        o[8 + d] = hi[d];
        o[14 + d] = lo[d];
        o[20 + d] = mk[d];
        o[26 + d] = kl[d];
    }
}

// what a lane takes out of one S-block for the boundaries it ranks: by pointer into LDS (line A) or from registers (line B)
struct SPiece {
    uint32_t cnt, sub;        // cumulative count of c before the block; count of c in the block's first 64 q rows
    uint64_t hw, lw;          // hi / lo words of rows 64 q .. 64 q + 63
    uint32_t mrank, msub;     // the same for the mask
    uint64_t mw;
};
__device__ __forceinline__ uint64_t below64(uint32_t r) { return r >= 64u ? ~0ull : ((1ull << r) - 1ull); }
__device__ __forceinline__ uint32_t s_quarter(uint32_t p) { return p >= 128u ? 2u : (p >> 6); }   // p in [0, 192]
__device__ __forceinline__ uint32_t spiece_rank(const SPiece& x, uint32_t c, uint32_t p, uint32_t q) {
    const uint64_t xh = (c & 2u) ? 0ull : ~0ull, xl = (c & 1u) ? 0ull : ~0ull;
    return x.cnt + x.sub + static_cast<uint32_t>(__popcll((x.hw ^ xh) & (x.lw ^ xl) & below64(p - 64u * q)));
}
__device__ __forceinline__ uint32_t spiece_mask_rank(const SPiece& x, uint32_t p, uint32_t q) {
    return x.mrank + x.msub + static_cast<uint32_t>(__popcll(x.mw & below64(p - 64u * q)));
}
// A lane's staged line in LDS.  The lines arrive by LDS-direct loads (global_load_lds_dwordx4: sixteen bytes per lane straight from memory into
// LDS at wave base + 16 x lane -- no registers, no ds_write), so where a piece lands is the loading lane's number: load instruction g
// (0..7) brings the lines of owners g, g + 8, ..., g + 56, lane (grp, p) of it piece p ^ g of owner g + 8 grp.  Line `owner` is then the
// 128 bytes at 1024 (owner & 7) + 128 (owner >> 3) with its piece q in slot q ^ (owner & 7): the owners' reads of one dword spread over
// sixteen bank quads, four lanes each -- what the 144-byte pitch of rounds 3-4 gave -- in 8 KB per wave instead of 9.
struct LineView {
    const char* wave_base;   // the wave's 8 KB
    uint32_t at;             // this lane's line + its swizzle key in the piece bits: a dword d is at (at ^ 4 d)
    __device__ __forceinline__ uint32_t dw(uint32_t d) const { return *reinterpret_cast<const uint32_t*>(wave_base + (at ^ (d << 2))); }
    __device__ __forceinline__ uint64_t u64(uint32_t d) const { return *reinterpret_cast<const uint64_t*>(wave_base + (at ^ (d << 2))); }   // (d even)
};
__device__ __forceinline__ SPiece spiece_lds(const LineView& L, uint32_t c, uint32_t q) {
    SPiece x;
    x.cnt = L.dw(c);
    x.sub = q ? (L.dw(3u + q) >> (8u * c)) & 0xFFu : 0u;
    x.hw = L.u64(8u + 2u * q);
    x.lw = L.u64(14u + 2u * q);
    x.mrank = L.dw(6);
    x.msub = q ? (L.dw(7) >> (8u * (q - 1u))) & 0xFFu : 0u;
    x.mw = L.u64(20u + 2u * q);
    return x;
}
// straight from memory (the second line of a straddling interval: few lanes, small loads)
__device__ __forceinline__ SPiece spiece_global(const uint32_t* __restrict__ B, uint32_t c, uint32_t q) {
    SPiece x;
    const uint4 h0 = *reinterpret_cast<const uint4*>(B), h1 = *reinterpret_cast<const uint4*>(B + 4);
    x.cnt = sel4(c, h0.x, h0.y, h0.z, h0.w);
    const uint32_t sw = q == 2u ? h1.y : h1.x;
    x.sub = q ? (sw >> (8u * c)) & 0xFFu : 0u;
    x.hw = *reinterpret_cast<const uint64_t*>(B + 8u + 2u * q);
    x.lw = *reinterpret_cast<const uint64_t*>(B + 14u + 2u * q);
    x.mrank = h1.z;
    x.msub = q ? (h1.w >> (8u * (q - 1u))) & 0xFFu : 0u;
    x.mw = *reinterpret_cast<const uint64_t*>(B + 20u + 2u * q);
    return x;
}


// UNI: reads of one length -- run r is the u.wpr windows from base u.start0 + r * u.stride -- are cut into tasks HERE (a division and five
// 8-byte loads when a lane draws a task) instead of by a preparation pass that writes a 64-byte record per task for this kernel to read.
struct UniformRuns {
    const uint64_t* seq;
    uint64_t seq_words, start0, stride;
    uint32_t wpr, tpr;
};
// WIDE (33 <= k <= 64): no k-mer register -- the bases of the current window come from the task's 128 bases, as in the first kernel.
// ORD (`lookup -S`, kmer_order_if_present of fms_index.h:151-157 in the loops of :181-254): a window's strand result is the mask rank of its
// interval's first row when a row of the interval is masked.  The kernel stores the orders it finds the moment it finds them (8 bytes each,
// straight to their place: no slots, no compaction) and keeps, as for the presence modes, which windows had one in the result planes; when
// a task is over the WAVE writes the -1 of every strand result without an order -- lane l the task's window l, the finished lanes' planes
// handed round by readlane: a read that is not in the text leaves as whole lines of -1.  (Round 4 filled the caller's array with -1 before
// the kernel: a pass of its own over 16 bytes per window, 0.5 of the leg's 4.1 ms.)
template <bool COUNT, bool PTABLE, bool UNI, bool WIDE, bool ORD>
__global__ __launch_bounds__(256, 4) void query_stream2_kernel(DevIndex ix, const uint4* __restrict__ sblocks, const uint64_t* __restrict__ klcp,
                                                            const TaskRec* __restrict__ tasks, uint64_t n_tasks, int k, uint32_t flags, uint32_t chunk,
                                                            uint4* __restrict__ slots, long long* __restrict__ out_orders,
                                                            unsigned long long* __restrict__ counters, UniformRuns u) {
    extern __shared__ uint4 s_line[];   // [waves of the workgroup][64 * kSPitch]
    const uint32_t ln = threadIdx.x & 63u;
    uint4* my_wave_lines = s_line + (threadIdx.x >> 6) * (64u * kSPitch);
    const LineView L{reinterpret_cast<const char*>(my_wave_lines), 1024u * (ln & 7u) + 128u * (ln >> 3) + 16u * (ln & 7u)};   // this lane's staged line
    const uint32_t mode_all = ORD ? 0u : (flags & FMSI_HIP_MODE_ALL);   // (orders: a strand decides a window when it HAS an order -- OR's schedule)
    const uint32_t strict = flags & FMSI_HIP_STRICT;
    const bool pair16 = ORD && strict && (reinterpret_cast<uintptr_t>(out_orders) & 15u) == 0u;   // (STRICT orders: a window's two are one aligned 16 bytes)
    const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    uint64_t next = wave * chunk;
    const uint64_t end = min(n_tasks, next + static_cast<uint64_t>(chunk));
    const int T = PTABLE ? static_cast<int>(ix.ptable_len) : 0;
    const uint64_t kmask = (WIDE || k == 32) ? ~0ull : ((1ull << (2 * k)) - 1ull);
    // ABSENCE CERTIFICATES (round 4).  A fresh search that follows a failed window starts with a table lookup of its own: the entry of bases
    // 4 .. T + 3 of the window's k-mer (on the strand being walked) with the signature test on bases 0 .. 3 -- a (T + 4)-mer at the END the pass
    // is walking TOWARDS (the forward strand goes through the windows right to left and this is its k-mer's first T + 4 bases; the
    // reverse-complement strand goes left to right and its k-mer's first bases are the window's last).  If that string does not occur in the
    // text, no k-mer containing it does: the window AND the cert_d = k - T - 4 windows the pass would visit next are absent on this strand --
    // one 8-byte request instead of cert_d + 1 searches (each one lookup, or, next to a substitution, up to k - T steps that end at the
    // same base).  The ordinary lookup's entry -- the k-mer's LAST T bases -- says the same about the windows the pass has just left: that is
    // why it cannot be used for this.  Exact: get_range_with_pattern (src/fms_index.h:117-124) returns an empty range for a pattern with an
    // absent substring.  A certificate that fails (the string occurs, or its entry has more than three rows) costs its round.
    // WHERE in the k-mer the certificate's string lies (offset o, 0 .. cert_d; it then settles cert_d + 1 - o windows): when a STREAMED step
    // dies -- a read that follows the text until a substituted base -- only strings that contain that base are absent, and the windows that
    // contain it are a run of k (aff_end: its last window in the pass' order).  While more than cert_d windows of the run are left the
    // string at offset 0 holds the base; for the run's last windows it moves towards the run's far end by the difference and settles
    // exactly those.  Everywhere else (reads that are not from the text, the strand a read does not follow) any string will do: offset 0.
    const bool cert_ok = PTABLE && k >= T + 4 && !(flags & kStreamNoCert);
    const bool cert_plain = (flags & kStreamCertPlain) != 0u;   // (A/B runs: always offset 0)
    const uint32_t cert_d = static_cast<uint32_t>(k - T - 4);
    constexpr int kAffFar = 1 << 20;
    int aff_end = 0;

    uint64_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, obase = 0;
    uint32_t n = 0;
    uint64_t kmer = 0;
    uint64_t rf_lo = 0, rf_hi = 0;   // pass-0 result + 1 of window w in bit w of the two planes (0 = empty interval)
    uint64_t rr_lo = 0, rr_hi = 0;   // pass-1 result + 1 (0: not visited or empty)
    uint64_t todo = 0;
    bool first_dead = false;          // the probe found window 0 absent on the reverse-complement strand: pass 1 need not search it again
    uint32_t i = 0, j = 0, st = SS_IDLE, pass = 0, widx = 0;
    bool cur_rc = false, probing = false;
    uint32_t special = 0;
    uint64_t task_id = 0;
    uint32_t scan_i = 0, scan_j = 0;
    int t = 0;
    StreamCounters cn;
    uint32_t util_rounds = 0, util_busy = 0;

    auto kmer_base = [&](int tt) -> uint32_t {   // base tt of the current window's k-mer on the current strand
        if (WIDE) return !cur_rc ? task_base(s0, s1, s2, s3, widx + static_cast<uint32_t>(tt))
                                   : 3u - task_base(s0, s1, s2, s3, widx + static_cast<uint32_t>(k - 1 - tt));
        return static_cast<uint32_t>(kmer >> (2 * tt)) & 3u;
    };
    auto kmer_table_index = [&]() -> uint32_t {   // its last T bases
        if (WIDE) return static_cast<uint32_t>(!cur_rc ? task_kmer(s0, s1, s2, s3, widx + static_cast<uint32_t>(k - T), T)
                                                         : revcomp_2bit(task_kmer(s0, s1, s2, s3, widx, T), T));
        return static_cast<uint32_t>(kmer >> (2 * (k - T))) & static_cast<uint32_t>((1ull << (2 * T)) - 1ull);
    };
    auto load_kmer = [&]() {
        if (WIDE) return;
        const uint64_t f = task_kmer(s0, s1, s2, s3, widx, k);
        kmer = !cur_rc ? f : revcomp_2bit(f, k);
    };
    auto cert_offset = [&]() -> uint32_t {
        if (cert_plain) return 0u;
        // windows left in the run of windows that hold the base a streamed step died at, the current one included (<= 0: behind it, or none known)
        const int r = !cur_rc ? static_cast<int>(widx) - aff_end + 1 : aff_end - static_cast<int>(widx) + 1;
        return (r <= 0 || r > static_cast<int>(cert_d)) ? 0u : cert_d + 1u - static_cast<uint32_t>(r);
    };
    auto cert_string = [&](uint32_t o) -> uint64_t {   // bases o .. o + T + 3 of the current window's k-mer on the current strand (the first in bits 0..1)
        if (WIDE) {
            const uint64_t z = task_kmer(s0, s1, s2, s3, !cur_rc ? widx + o : widx + cert_d - o, T + 4);
            return !cur_rc ? z : revcomp_2bit(z, T + 4);
        }
        return (kmer >> (2u * o)) & ((1ull << (2 * (T + 4))) - 1ull);
    };
    // after_fail: the window before this one (in the pass' order) had an empty interval -- try a certificate first
    auto fresh = [&](bool after_fail = false) {
        if (COUNT) cn.searches++;
        if (PTABLE) {
            st = (after_fail && cert_ok) ? SS_CERT : SS_LOOKUP;
        } else {
            i = 0;
            j = ix.n;
            t = k - 1;
            st = SS_LF;
        }
    };
    // A finished task's results stay in the lane's planes (SS_DONE) until the lane draws its next task or the wave ends: ONE code site writes
    // them, and in a wave of one task per lane it runs once -- the state machine's branches each ended tasks in place before, and in a wave
    // whose lanes finish in different rounds (reads with substitutions) the hundred instructions of the byte spread ran round after round.
    auto end_task = [&]() { st = SS_DONE; };
    auto flush_task = [&]() {
        if (ORD) return;
        uint64_t p0, p1, p2, p3;
        if (strict) {   // byte = (f + 1) | (r + 1) << 2
            p0 = rf_lo; p1 = rf_hi; p2 = rr_lo; p3 = rr_hi;
        } else {
            const uint64_t f1 = rf_hi & ~rf_lo, r1 = rr_hi & ~rr_lo;
            p0 = mode_all ? (((rf_lo | rf_hi) & f1) | (~(rf_lo | rf_hi) & r1)) : (f1 | r1);   // ALL: (f != -1 ? f : r) == 1; OR: f == 1 || r == 1
            p1 = p2 = p3 = 0;
        }
        uint4* dst = slots + (task_id << 1);   // (the planes as they are: stream_planes_* make bytes or bits of them)
        dst[0] = make_uint4(static_cast<uint32_t>(p0), static_cast<uint32_t>(p0 >> 32), static_cast<uint32_t>(p1), static_cast<uint32_t>(p1 >> 32));
        dst[1] = make_uint4(static_cast<uint32_t>(p2), static_cast<uint32_t>(p2 >> 32), static_cast<uint32_t>(p3), static_cast<uint32_t>(p3 >> 32));
    };
    // ORD: the -1s of the tasks that are over, by the whole wave (wave-uniform call); their lanes are free then
    auto lane64 = [](uint64_t v, int lane) -> uint64_t {
        const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(v)), lane));
        const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(v >> 32)), lane));
        return (static_cast<uint64_t>(hi) << 32) | lo;
    };
    auto flush_orders = [&]() {
        uint64_t dm = __ballot(st == SS_DONE);
        const uint64_t mine_f = rf_hi, mine_r = rr_hi;   // windows with an order (strand result + 1 == 2; ORD keeps no low planes: window_done)
        while (dm) {
            const int owner = __ffsll(static_cast<unsigned long long>(dm)) - 1;
            dm &= dm - 1;
            const uint64_t f1 = lane64(mine_f, owner), r1 = lane64(mine_r, owner), ob = lane64(obase, owner);
            const uint32_t nn = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(n), owner));
            if (ln < nn) {
                const bool fh = (f1 >> ln) & 1ull, rh = (r1 >> ln) & 1ull;
                if (strict) {
                    // (a window's pair is 16 bytes; the stores are what this mode costs -- the memory path counts requests -- so a pair without
                    // an order is ONE store, and a pair whose forward order pass 0 stored has its -1 already: window_done)
                    long long* p = out_orders + 2ull * (ob + ln);
                    if (pair16 && !fh && !rh) {
                        typedef long long ll2 __attribute__((ext_vector_type(2)));
                        __builtin_nontemporal_store(ll2{-1ll, -1ll}, reinterpret_cast<ll2*>(p));
                    } else {
                        if (!fh) __builtin_nontemporal_store(-1ll, p);
                        if (!rh && !(pair16 && fh)) __builtin_nontemporal_store(-1ll, p + 1);
                    }
                } else if (!(fh | rh)) {
                    __builtin_nontemporal_store(-1ll, out_orders + ob + ln);
                }
            }
        }
        if (st == SS_DONE) st = SS_IDLE;
    };
    // The set-up work between windows has ONE code site each, at the top of the round loop: the state machine's branches only name it --
    // SS_PASS1 (pass 0 is over), SS_FRESH_CERT / SS_FRESH_PLAIN (cut the k-mer of window widx from the task's bases and start a search, with
    // or without a certificate first).  In a wave whose lanes are in different states a round runs every branch some lane is in; what the
    // branches share is better run once.
    auto begin_pass1 = [&]() { st = SS_PASS1; };
    auto begin_pass1_now = [&]() {
        pass = 1;
        cur_rc = !cur_rc;
        aff_end = cur_rc ? -kAffFar : kAffFar;
        const uint64_t all = n == 64 ? ~0ull : ((1ull << n) - 1ull);
        if (strict) todo = all;
        else if (!mode_all) todo = all & ~(rf_hi & ~rf_lo);   // OR: only a pass-0 result of 1 decides
        else todo = all & ~(rf_lo | rf_hi);                   // ALL: any non-empty pass-0 interval decides
        if (first_dead) todo &= ~1ull;                        // (its pass-1 result is "empty interval": the planes' zero says so)
        if (!todo) {
            end_task();
            return;
        }
        if (cur_rc) {
            widx = static_cast<uint32_t>(__ffsll(static_cast<unsigned long long>(todo))) - 1u;
            todo &= todo - 1;
        } else {
            widx = 63u - static_cast<uint32_t>(__clzll(static_cast<long long>(todo)));
            todo &= ~(1ull << widx);
        }
        st = SS_FRESH_CERT;   // (pass 1 visits the windows pass 0 left undecided: mostly windows that are in the text on neither strand)
    };
    auto window_done = [&](int res, uint32_t ord = 0u) {
        if (!strict && pass == 0 && n > 1u) {
            if (probing) {
                probing = false;
                if (res == -1) {   // absent there, too: forward strand first after all, from the last window on
                    first_dead = true;
                    cur_rc = false;
                    widx = n - 1u;
                    load_kmer();
                }
                // (else: the probe was the first window of a pass 0 on the reverse-complement strand: recorded below, the pass streams on from its interval)
            } else if (!cur_rc && res == -1 && widx == n - 1u) {
                probing = true;
                special = 1;
                return;
            }
        }
        const uint64_t bit = 1ull << widx;
        const uint32_t v = static_cast<uint32_t>(res + 1);
        // (ORD: "present without an order" and "absent" print the same -1 and decide nothing, so the low planes stay empty -- and out of the
        // registers: with them the ORD instantiation spilled its pass-0 planes to scratch, a memory request per window on a kernel bound by requests)
        if (pass == 0) {
            if (!ORD && (v & 1u)) rf_lo |= bit;
            if (v & 2u) rf_hi |= bit;
        } else {
            if (!ORD && (v & 1u)) rr_lo |= bit;
            if (v & 2u) rr_hi |= bit;
        }
        if (ORD && res == 1) {   // (every other strand result is a -1: flush_orders)
            if (strict && pair16 && pass == 0) {   // the pair at once, the other strand's -1 on trust: an order of pass 1 comes later and over it
                typedef long long ll2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<ll2*>(out_orders + 2ull * (obase + widx)) = ll2{static_cast<long long>(ord), -1ll};
            } else if (strict) out_orders[2ull * (obase + widx) + pass] = static_cast<long long>(ord);
            else out_orders[obase + widx] = static_cast<long long>(ord);   // (pass 1 visits a window only when pass 0 had no order for it)
        }
    };
    auto shift_fwd = [&](uint32_t w) {
        if (!WIDE) kmer = ((kmer << 2) | task_base(s0, s1, s2, s3, w)) & kmask;
    };
    auto shift_rc = [&](uint32_t w) {
        if (!WIDE) kmer = ((kmer << 2) | (3u - task_base(s0, s1, s2, s3, w + static_cast<uint32_t>(k)))) & kmask;
    };
    auto advance = [&](bool have) -> bool {
        if (special) {   // the probe: the task's FIRST window on the reverse-complement strand (where a pass on that strand starts)
            special = 0;
            cur_rc = true;
            aff_end = -kAffFar;
            widx = 0;
            st = SS_FRESH_PLAIN;
            return false;
        }
        if (pass == 0) {
            if (!cur_rc) {
                if (widx == 0) {
                    begin_pass1();
                    return false;
                }
                --widx;
                shift_fwd(widx);
            } else {
                if (widx + 1u == n) {
                    begin_pass1();
                    return false;
                }
                shift_rc(widx);
                ++widx;
            }
            if (have) return true;
            fresh(true);
            return false;
        }
        if (!todo) {
            end_task();
            return false;
        }
        uint32_t nw;
        bool adjacent;
        if (cur_rc) {
            nw = static_cast<uint32_t>(__ffsll(static_cast<unsigned long long>(todo))) - 1u;
            todo &= todo - 1;
            adjacent = nw == widx + 1u;
            if (adjacent) shift_rc(widx);
        } else {
            nw = 63u - static_cast<uint32_t>(__clzll(static_cast<long long>(todo)));
            todo &= ~(1ull << nw);
            adjacent = nw + 1u == widx;
            if (adjacent) shift_fwd(nw);
        }
        widx = nw;
        if (adjacent) {
            if (have) return true;
            fresh(true);
        } else {
            st = have ? SS_FRESH_PLAIN : SS_FRESH_CERT;
        }
        return false;
    };
    // a certificate said so: the current window and the cert_d windows the pass would visit after it have empty intervals on this strand
    // (their result planes hold "empty" already) -- on to the first window behind them
    auto skip_dead = [&](uint32_t o) {
        const uint32_t cnt = cert_d - o;   // windows settled beside the current one
        if (pass == 0) {
            if (!cur_rc) {
                if (widx <= cnt) {
                    begin_pass1();
                    return;
                }
                widx -= cnt + 1u;
            } else {
                if (widx + cnt + 1u >= n) {
                    begin_pass1();
                    return;
                }
                widx += cnt + 1u;
            }
        } else {
            if (cur_rc) {   // windows (widx, widx + cnt]
                if (cnt) todo &= ~((((2ull << (cnt - 1u)) - 1ull) << widx) << 1);
            } else {        // windows [widx - cnt, widx)
                todo &= ~((1ull << widx) - 1ull) | (widx > cnt ? (1ull << (widx - cnt)) - 1ull : 0ull);
            }
            if (!todo) {
                end_task();
                return;
            }
            if (cur_rc) {
                widx = static_cast<uint32_t>(__ffsll(static_cast<unsigned long long>(todo))) - 1u;
                todo &= todo - 1;
            } else {
                widx = 63u - static_cast<uint32_t>(__clzll(static_cast<long long>(todo)));
                todo &= ~(1ull << widx);
            }
        }
        st = SS_FRESH_CERT;
    };

    for (;;) {
        // a search that ended with an empty interval in the last round (a table entry without rows, an LF step, a streamed step): the window is
        // absent on this strand, on to the next -- ONE code site for the three (a wave whose lanes are in different states runs every site some
        // lane is in, and this one is some ninety vector instructions and a hundred scalar ones)
        if (st == SS_DIED) {
            window_done(-1);
            advance(false);
        }
        if (st == SS_PASS1) begin_pass1_now();
        if (ORD) flush_orders();
        {   // idle lanes draw the wave's next tasks (ballot + prefix rank, no atomics); a task's record is three 16-byte loads per 64 windows
            const bool want = st == SS_IDLE || st == SS_DONE;
            const uint64_t m = next < end ? __ballot(want) : 0ull;   // (nothing left to draw: the finished lanes keep their results for the end)
            if (m) {
                const uint64_t cand = next + lanes_below(m);
                if (want && cand < end) {
                    if (st == SS_DONE) flush_task();
                    if (UNI) {   // (n_tasks < 2^32: the host checks)
                        const uint32_t r = static_cast<uint32_t>(cand) / u.tpr, j = static_cast<uint32_t>(cand) - r * u.tpr;
                        const uint64_t base = u.start0 + static_cast<uint64_t>(r) * u.stride + 64ull * j;
                        const uint64_t w0 = base >> 5;
                        const uint32_t sh = static_cast<uint32_t>(base & 31u) * 2u;
                        uint64_t v[5];
#pragma unroll
                        for (uint32_t q = 0; q < 5; ++q) v[q] = w0 + q < u.seq_words ? u.seq[w0 + q] : 0ull;   // (never past the caller's buffer)
                        s0 = v[0] >> sh; s1 = v[1] >> sh; s2 = v[2] >> sh; s3 = v[3] >> sh;
                        if (sh) {
                            s0 |= v[1] << (64u - sh); s1 |= v[2] << (64u - sh); s2 |= v[3] << (64u - sh); s3 |= v[4] << (64u - sh);
                        }
                        n = min(64u, u.wpr - 64u * j);
                        if (ORD) obase = static_cast<uint64_t>(r) * u.wpr + 64ull * j;
                    } else {
                        const uint4* p = reinterpret_cast<const uint4*>(tasks + cand);
                        const uint4 r0 = p[0], r1 = p[1], r2 = p[2];
                        s0 = (static_cast<uint64_t>(r0.y) << 32) | r0.x;
                        s1 = (static_cast<uint64_t>(r0.w) << 32) | r0.z;
                        s2 = (static_cast<uint64_t>(r1.y) << 32) | r1.x;
                        s3 = (static_cast<uint64_t>(r1.w) << 32) | r1.z;
                        n = r2.z;
                        if (ORD) obase = (static_cast<uint64_t>(r2.y) << 32) | r2.x;
                    }
                    task_id = cand;
                    aff_end = kAffFar;    // (no run known: pass 0 starts on the forward strand, windows descending)
                    pass = 0;
                    cur_rc = false;
                    probing = false;
                    special = 0;
                    rf_lo = rf_hi = rr_lo = rr_hi = 0;
                    first_dead = false;
                    widx = n - 1u;
                    st = SS_FRESH_PLAIN;
                }
                next += static_cast<uint64_t>(__popcll(m));
            }
        }
        if (st == SS_FRESH_CERT || st == SS_FRESH_PLAIN) {
            const bool with_cert = st == SS_FRESH_CERT;
            load_kmer();
            fresh(with_cert);
        }
        if (!__any(st != SS_IDLE && st != SS_DONE)) break;
        if (COUNT && (flags & kStreamUtil)) {   // (FMSI_HIP_STREAM_UTIL: lane-rounds of the wave and the busy ones among them, in counters 6 and 7)
            util_rounds++;
            util_busy += (st != SS_IDLE && st != SS_DONE) ? 1u : 0u;
        }
        // ---- this round's reads ----
        const bool lk = PTABLE && (st == SS_LOOKUP || st == SS_CERT);
        const bool need_line = st == SS_LF || st == SS_HAVE || st == SS_EXTLF;
        uint32_t bi = 0, bj = 0;
        if (need_line) {
            bi = i / kSRows;
            bj = (j - 1u) / kSRows;
        }
        const bool two = need_line && bj != bi;
        uint2 ent = make_uint2(0, 0);
        uint32_t cert_q4 = 0;
        if (lk) {
            uint32_t tidx;
            if (st == SS_CERT) {
                const uint64_t z = cert_string(cert_offset());
                cert_q4 = static_cast<uint32_t>(z) & 0xFFu;   // bases 0 .. 3 in the signatures' packing (base 3, the next one a search would consume, on top)
                tidx = static_cast<uint32_t>(z >> 8);
            } else {
                tidx = kmer_table_index();
            }
            ent = ix.ptable[tidx];
        }
        uint64_t kwi = 0, kwj = 0;
        if (st == SS_EXTSCAN) {   // a running extension scan (rare): the plain kLCP words at the frontier
            if (scan_i) kwi = klcp[(i - 1u) >> 6];
            if (scan_j) kwj = klcp[(j - 1u) >> 6];
            if (COUNT) cn.klcp_words += scan_i + scan_j;
        }
        const uint64_t line_mask = __ballot(need_line);
        if (line_mask) {   // eight lanes per line, eight lines per instruction, straight into LDS (LineView)
            const uint32_t grp = ln >> 3;
#pragma unroll
            for (uint32_t g = 0; g < 8; ++g) {
                const uint32_t owner = g + 8u * grp;
                const uint32_t bo = static_cast<uint32_t>(__shfl(static_cast<int>(bi), static_cast<int>(owner), 64));
                if ((line_mask >> owner) & 1ull)
                    __builtin_amdgcn_global_load_lds(sblocks + ((static_cast<size_t>(bo) << 3) + ((ln & 7u) ^ g)), s_line + ((threadIdx.x >> 6) * (64u * kSPitch) + 64u * g), 16, 0, 0);
            }
        }
        if (COUNT) cn.blocks += (need_line ? 1u : 0u) + (two ? 1u : 0u) + (lk ? 1u : 0u);   // ([1] counts table lines too, [4] says how many)
        __builtin_amdgcn_wave_barrier();

        // ---- arithmetic ----
        bool do_lf = false;
        uint32_t c = 0;
        if (st == SS_CERT) {
            if (COUNT) cn.lookups++;
            uint32_t ci, cj;
            decode_table_entry(ent.x, ent.y, ix.ptable_packed, true, cert_q4, ci, cj);
            if (ci == cj) skip_dead(cert_offset());
            else st = SS_LOOKUP;
        } else if (st == SS_LOOKUP) {
            if (COUNT) cn.lookups++;
            t = k - 1 - T;
            uint32_t q4 = 0;
            if (t >= 3) q4 = kmer_base(t - 3) | (kmer_base(t - 2) << 2) | (kmer_base(t - 1) << 4) | (kmer_base(t) << 6);
            decode_table_entry(ent.x, ent.y, ix.ptable_packed, t >= 3, q4, i, j);
            if (i == j) st = SS_DIED;
            else st = t < 0 ? SS_HAVE : SS_LF;
        } else if (st == SS_LF) {
            do_lf = true;
            c = kmer_base(t);
        } else if (st == SS_EXTLF) {
            do_lf = true;
            c = kmer_base(0);
        } else if (st == SS_HAVE) {
            // the current window's result off the mask plane (infer_presence, fms_index.h:126-144), the kLCP bits of rows i - 1 and
            // j - 1 off the same line(s)
            if (COUNT) cn.probes++;
            const uint32_t pi_off = i - bi * kSRows, pj_off = j - bj * kSRows;   // 0..191, 1..192
            const uint32_t* B = reinterpret_cast<const uint32_t*>(sblocks) + static_cast<size_t>(bj) * kSDwords;
            int res;
            uint32_t ord = 0;
            if (mode_all) {
                res = static_cast<int>((L.dw(20u + (pi_off >> 5)) >> (pi_off & 31u)) & 1u);
            } else if (!ORD && j - i == 1u) {
                res = static_cast<int>((L.dw(20u + (pi_off >> 5)) >> (pi_off & 31u)) & 1u);
            } else {
                const uint32_t qi = s_quarter(pi_off), qj = s_quarter(pj_off);
                const uint32_t lo_r = spiece_mask_rank(spiece_lds(L, 0, qi), pi_off, qi);
                const uint32_t hi_r = two ? spiece_mask_rank(spiece_global(B, 0, qj), pj_off, qj) : spiece_mask_rank(spiece_lds(L, 0, qj), pj_off, qj);
                res = hi_r > lo_r;
                ord = lo_r;   // kmer_order (fms_index.h:147-149): the mask rank of the interval's first row
            }
            // klcp[i - 1]: the bit before the block when i is its first row; klcp[j - 1]: row j - 1 is row pj_off - 1 of block bj
            const uint32_t ki = i == 0u ? 0u : (pi_off == 0u ? (L.dw(7) >> 31) : (L.dw(26u + ((pi_off - 1u) >> 5)) >> ((pi_off - 1u) & 31u)) & 1u);
            const uint32_t kjw = two ? B[26u + ((pj_off - 1u) >> 5)] : L.dw(26u + ((pj_off - 1u) >> 5));
            const uint32_t kj = (kjw >> ((pj_off - 1u) & 31u)) & 1u;
            window_done(res, ord);
            if (advance(true)) {
                if (!(ki | kj)) {
                    do_lf = true;   // common case: the (k-1)-mer's interval is the k-mer's; the line is here
                    c = kmer_base(0);
                } else {
                    // extend_range_with_klcp (fms_index.h:106-115) over the plain kLCP words, as a scan of its own
                    scan_i = ki;
                    scan_j = kj;
                    st = SS_EXTSCAN;
                }
            }
        } else if (st == SS_EXTSCAN) {
            if (scan_j) {   // j grows while klcp[j - 1]
                const uint32_t pj = (j - 1u) & 63u;
                const uint64_t up = kwj >> pj;
                const uint32_t run = static_cast<uint32_t>(__ffsll(static_cast<unsigned long long>(~up))) - 1u;
                const uint32_t lim = 64u - pj;
                const uint32_t r = (~up == 0ull || run > lim) ? lim : run;
                j += r;
                scan_j = (r == lim && j < ix.n) ? 1u : 0u;
            }
            if (scan_i) {   // i shrinks while klcp[i - 1]
                const uint32_t pi_ = (i - 1u) & 63u;
                const uint64_t dn = kwi << (63u - pi_);
                const uint32_t run = static_cast<uint32_t>(__clzll(static_cast<long long>(~dn)));
                const uint32_t lim = pi_ + 1u;
                const uint32_t r = (~dn == 0ull || run > lim) ? lim : run;
                i -= r;
                scan_i = (r == lim && i > 0u) ? 1u : 0u;
            }
            if (!(scan_i | scan_j)) st = SS_EXTLF;
        }
        if (do_lf) {
            const uint32_t pi_off = i - bi * kSRows, pj_off = j - bj * kSRows;
            const uint32_t qi = s_quarter(pi_off), qj = s_quarter(pj_off);
            uint32_t ri = spiece_rank(spiece_lds(L, c, qi), c, pi_off, qi);
            uint32_t rj;
            if (two) {
                const uint32_t* B = reinterpret_cast<const uint32_t*>(sblocks) + static_cast<size_t>(bj) * kSDwords;
                rj = spiece_rank(spiece_global(B, c, qj), c, pj_off, qj);
            } else {
                rj = spiece_rank(spiece_lds(L, c, qj), c, pj_off, qj);
            }
            if (c == 0u) {
                ri -= (i > ix.dollar);
                rj -= (j > ix.dollar);
            }
            const uint32_t base = sel4(c, ix.C0, ix.C1, ix.C2, ix.C3);
            i = base + ri;
            j = base + rj;
            if (COUNT) cn.lf_steps++;
            if (st == SS_LF) {
                if (i == j) st = SS_DIED;
                else if (--t < 0) st = SS_HAVE;
            } else {
                if (i == j) {
                    // a streamed step with an empty interval: the window's new base -- its first on the forward strand, its last on the other -- is
                    // one the text does not have there; the windows that hold it end k - 1 windows on in the pass' order
                    aff_end = !cur_rc ? static_cast<int>(widx) - (k - 1) : static_cast<int>(widx) + (k - 1);
                    st = SS_DIED;
                } else {
                    st = SS_HAVE;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();   // (the next round's staging must not overtake this round's reads of the lines)
    }
    if (ORD) flush_orders();
    if (st == SS_DONE) flush_task();
    if (COUNT && (flags & kStreamUtil)) {
        uint32_t x = util_busy, y = util_rounds;
        for (int off = 32; off > 0; off >>= 1) {
            x += __shfl_down(x, off, 64);
            y += __shfl_down(y, off, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&counters[6], static_cast<unsigned long long>(y));
            atomicAdd(&counters[7], static_cast<unsigned long long>(x));
        }
    }
    if (COUNT) {
        uint32_t v[6] = {cn.lf_steps, cn.blocks, cn.probes, cn.searches, cn.lookups, cn.klcp_words};
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            uint32_t x = v[q];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if ((threadIdx.x & 63) == 0 && x) atomicAdd(&counters[q], static_cast<unsigned long long>(x));
        }
    }
}

template <bool COUNT, bool PTABLE, bool WIDE, bool ORD>
void launch_stream2(fmsi_hip_index* idx, const TaskRec* recs, uint64_t n_tasks, int k, uint32_t flags, uint4* slots, long long* o64,
                    unsigned long long* d_counters, hipStream_t s, const UniformRuns* uni) {
    uint64_t chunk = n_tasks / (256ull * 32 * 8);
    chunk = std::min<uint64_t>(std::max<uint64_t>(chunk, 64), 1024);
    static const char* chunk_env = std::getenv("FMSI_HIP_STREAM_CHUNK");   // (measurements: tasks per wave)
    if (chunk_env) chunk = std::max(1, std::atoi(chunk_env));
    const uint64_t waves = (n_tasks + chunk - 1) / chunk;
    // ONE wave per workgroup (round 5; rounds 3-4: four).  A wave lasts as long as the slowest of its 64 tasks and a workgroup as long as the
    // slowest of its waves: with four, the device kept the three that were done -- their LDS, their wave slots -- until the fourth was
    // (kLCP kernel at t = 15, 2^27 windows: 57 -> 61 G windows/s on the clean mix, 40 -> 48 G with substitutions; FMSI_HIP_STREAM_WPB for A/B runs)
    static const char* wpb_env = std::getenv("FMSI_HIP_STREAM_WPB");
    const uint32_t wpb = wpb_env ? static_cast<uint32_t>(std::min(4, std::max(1, std::atoi(wpb_env)))) : 1u;
    const uint32_t blocks = static_cast<uint32_t>((waves + wpb - 1) / wpb);
    const uint32_t lds = wpb * 64u * kSPitch * static_cast<uint32_t>(sizeof(uint4));
    static const char* cert_env = std::getenv("FMSI_HIP_STREAM_CERT");   // (measurements: 0 = no certificates, 1 = always at offset 0)
    flags &= ~(kStreamNoCert | kStreamCertPlain | kStreamUtil);
    static const char* util_env = std::getenv("FMSI_HIP_STREAM_UTIL");   // (measurements: counters 6 / 7 = lane-rounds / busy lane-rounds)
    if (util_env) flags |= kStreamUtil;
    if (cert_env && cert_env[0] == '0') flags |= kStreamNoCert;
    // A certificate can only come from an entry that is empty or carries signatures (<= 3 rows): with n / 4^t rows per key on average that is
    // nearly every entry at 0.65 (t = 16 at human scale), 73 % at 2.6 (t = 15), a quarter at 5 -- and one in a thousand at 10 (the CLI's
    // 14-deep table), where a certificate is a wasted request per failed window.  Break-even (a success spares a dozen requests) is near 7.5.
    if (PTABLE && !cert_env && static_cast<double>(idx->n) > 7.0 * static_cast<double>(1ull << (2u * idx->dev.ptable_len))) flags |= kStreamNoCert;
    if (cert_env && cert_env[0] == '1') flags |= kStreamCertPlain;
    if (uni)
        hipLaunchKernelGGL((query_stream2_kernel<COUNT, PTABLE, true, WIDE, ORD>), dim3(blocks), dim3(64u * wpb), lds, s, idx->dev,
                           static_cast<const uint4*>(idx->d_sblocks), static_cast<const uint64_t*>(idx->d_klcp), recs, n_tasks, k, flags,
                           static_cast<uint32_t>(chunk), slots, o64, d_counters, *uni);
    else
        hipLaunchKernelGGL((query_stream2_kernel<COUNT, PTABLE, false, WIDE, ORD>), dim3(blocks), dim3(64u * wpb), lds, s, idx->dev,
                           static_cast<const uint4*>(idx->d_sblocks), static_cast<const uint64_t*>(idx->d_klcp), recs, n_tasks, k, flags,
                           static_cast<uint32_t>(chunk), slots, o64, d_counters, UniformRuns{});
}
template <bool WIDE, bool ORD>
void launch_stream2_variant(fmsi_hip_index* idx, bool count, bool table, const TaskRec* recs, uint64_t n_tasks, int k, uint32_t flags, uint4* slots,
                            long long* o64, unsigned long long* d_counters, hipStream_t s, const UniformRuns* uni) {
    if (table) {
        if (count) launch_stream2<true, true, WIDE, ORD>(idx, recs, n_tasks, k, flags, slots, o64, d_counters, s, uni);
        else launch_stream2<false, true, WIDE, ORD>(idx, recs, n_tasks, k, flags, slots, o64, d_counters, s, uni);
    } else {
        if (count) launch_stream2<true, false, WIDE, ORD>(idx, recs, n_tasks, k, flags, slots, o64, d_counters, s, uni);
        else launch_stream2<false, false, WIDE, ORD>(idx, recs, n_tasks, k, flags, slots, o64, d_counters, s, uni);
    }
}

template <bool COUNT, bool PTABLE, bool WIDE>
void launch_stream(fmsi_hip_index* idx, const TaskRec* recs, uint64_t n_tasks, int k, uint32_t flags, uint4* o8, long long* o64,
                   unsigned long long* d_counters, hipStream_t s) {
    uint64_t chunk = n_tasks / (256ull * 32 * 8);
    chunk = std::min<uint64_t>(std::max<uint64_t>(chunk, 64), 1024);
    const uint64_t waves = (n_tasks + chunk - 1) / chunk;
    const uint32_t blocks = static_cast<uint32_t>((waves + 3) / 4);
    hipLaunchKernelGGL((query_stream_kernel<COUNT, PTABLE, WIDE>), dim3(blocks), dim3(256), 0, s, idx->dev,
                       static_cast<const uint64_t*>(idx->d_klcp), recs, n_tasks, k, flags, static_cast<uint32_t>(chunk), o8, o64,
                       d_counters);
}

}  // namespace
}  // namespace fmsi

namespace fmsi {
void warm_stream_module() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&sblock_build_kernel));
    (void)hipGetLastError();
}
}  // namespace fmsi

using namespace fmsi;

// The streaming layout of a handle with a kLCP array (idempotent; fmsi_hip_query_stream builds it on first use): 128 bytes per 192 rows.
extern "C" int fmsi_hip_index_enable_stream_layout(fmsi_hip_index* idx) {
    if (!idx) return FMSI_HIP_EINVAL;
    FMSI_HIP_NARROW_ONLY(idx, "kLCP streaming");
    if (!idx->d_klcp) {
        set_last_error("the streaming layout needs an index created with its kLCP array");
        return FMSI_HIP_EINVAL;
    }
    if (idx->d_sblocks) return 0;
    ::fmsi::ScratchScope scratch_scope_(idx->device);
    FMSI_HIP_ON_DEVICE(idx->device);
    const uint64_t n_s = (idx->n + kSRows - 1) / kSRows + 1;
    void* p = nullptr;
    if (dev_malloc(&p, n_s * kSDwords * 4) != hipSuccess) {
        (void)hipGetLastError();
        set_last_error("hipMalloc failed for the streaming layout");
        return FMSI_HIP_ENOMEM;
    }
    hipLaunchKernelGGL(sblock_build_kernel, dim3(static_cast<uint32_t>((n_s + 255) / 256)), dim3(256), 0, 0, static_cast<const uint32_t*>(idx->d_blocks),
                       static_cast<const uint32_t*>(idx->d_mrank), idx->n_blocks, static_cast<const uint32_t*>(idx->d_klcp), ((idx->n + 63) / 64) * 2, n_s,
                       static_cast<uint32_t*>(p));
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)dev_free(p);
        set_last_error("building the streaming layout failed");
        return FMSI_HIP_EHIP;
    }
    idx->d_sblocks = p;
    idx->n_sblocks = n_s;
    idx->device_bytes += n_s * kSDwords * 4;
    return 0;
}

extern "C" int fmsi_hip_query_stream(fmsi_hip_index* idx, const uint64_t* packed_seq, uint64_t seq_bases, const uint64_t* run_start,
                                     const uint32_t* run_windows, uint64_t n_runs, uint64_t n_windows, int k, uint32_t flags,
                                     void* out, uint64_t* counters, void* stream) {
    if (n_windows == 0 || n_runs == 0) return 0;
    if (!idx || !packed_seq || !run_start || !run_windows || !out) {
        set_last_error("null argument");
        return FMSI_HIP_EINVAL;
    }
    // 64-bit rows: no kLCP kernel -- and no need for one to ANSWER: the extension is the reference's short cut to the interval a fresh
    // search finds (src/fms_index.h:106-115), so the plain search over the same runs gives these results, in this order
    if (idx->wide) return fmsi_hip_query_runs(idx, packed_seq, seq_bases, run_start, run_windows, n_runs, n_windows, k, flags, out, counters, stream);
    if (!idx->d_klcp) {
        set_last_error("streaming queries need an index created with its kLCP array");
        return FMSI_HIP_EINVAL;
    }
    if (k != idx->k) {       // the kLCP bits say "these two rows share k - 1 bases" for the index's k and no other
        set_last_error("streaming queries need k = " + std::to_string(idx->k) + ", the k the kLCP array was built for");
        return FMSI_HIP_EINVAL;
    }
    if (k < 1 || k > 64) {   // a kLCP array exists for k <= 64 only (reference src/main.cpp:223-226)
        set_last_error("streaming queries need 1 <= k <= 64");
        return FMSI_HIP_ERANGE;
    }
    FMSI_HIP_ON_DEVICE(idx->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the S-block kernel (its layout is built on the handle's first such call; FMSI_HIP_STREAM_V1 / FMSI_HIP_KERNEL_REFILL: the first kernel,
    // for A/B runs) -- since round 4 also for ORDERS (`lookup -S`) and 33 <= k <= 64
    static const bool force_v1 = std::getenv("FMSI_HIP_STREAM_V1") != nullptr;
    bool v2 = !force_v1 && (flags & FMSI_HIP_KERNEL_MASK) != FMSI_HIP_KERNEL_REFILL;
    if (v2 && !idx->d_sblocks && idx->stream_layout_failed) v2 = false;
    if (v2 && !idx->d_sblocks) {
        // built here on the handle's first such call (a kernel on the null stream and a device synchronisation: this ONE call is not
        // asynchronous on `stream`; fmsi_hip_index_enable_stream_layout at load time avoids that).  A handle without the ROOM for the
        // layout (2/3 byte per row) keeps streaming through the first kernel -- the same answers, slower; that is remembered (no second
        // attempt with every batch) and said once.  Any other failure is the call's.
        const int rc_layout = fmsi_hip_index_enable_stream_layout(idx);
        if (rc_layout == FMSI_HIP_ENOMEM) {
            (void)hipGetLastError();
            idx->stream_layout_failed = true;
            fprintf(stderr, "WARNING: not enough device memory for the streaming layout (%llu MiB); streaming through the first kernel (same answers, slower)\n",
                    static_cast<unsigned long long>((((idx->n + kSRows - 1) / kSRows + 1) * kSDwords * 4) >> 20));
            v2 = false;
        } else if (rc_layout != 0) {
            return rc_layout;
        }
    }
    // run_start / run_windows are HOST arrays in every mode (one entry per run); per run the device gets its first base,
    // first task and first result index, and cuts the tasks itself (stream_prep_kernel)
    // (in a page-locked slot of the handle's ring: the call stays asynchronous on `stream`)
    // UNIFORM runs (equal lengths at one stride: reads of one length laid out back to back) need no table at all: three numbers
    const uint64_t wpr = run_windows[0], stride = n_runs > 1 ? run_start[1] - run_start[0] : wpr;
    bool uniform = n_runs == 1 || (run_start[1] > run_start[0] && stride >= wpr);
    for (uint64_t r = 0; r < n_runs && uniform; ++r)
        if (run_windows[r] != wpr || (r > 0 && run_start[r] - run_start[r - 1] != stride)) uniform = false;
    void* meta_host = nullptr;
    unsigned meta_slot = 0;
    int rc;
    uint64_t* meta = nullptr;   // [0, R): run_start   [R, 2R]: task_off   [2R + 1, 3R + 1]: out_off
    const size_t meta_words = uniform ? 0 : 3 * n_runs + 2;
    if (!uniform) {
        if ((rc = meta_stage(idx, (3 * n_runs + 2) * 8, &meta_host, &meta_slot))) return rc;
        meta = static_cast<uint64_t*>(meta_host);
    }
    uint64_t* h_task_off = meta ? meta + n_runs : nullptr;
    uint64_t* h_out_off = meta ? meta + 2 * n_runs + 1 : nullptr;
    uint64_t n_tasks = 0, out_pos = 0;
    for (uint64_t r = 0; r < n_runs; ++r) {
        const uint64_t w = run_windows[r];
        if (w == 0 || run_start[r] + w + static_cast<uint64_t>(k) - 1 > seq_bases) {
            set_last_error("an empty run, or a run of windows that extends past the end of the sequence");
            return FMSI_HIP_EINVAL;
        }
        if (meta) {
            meta[r] = run_start[r];
            h_task_off[r] = n_tasks;
            h_out_off[r] = out_pos;
        }
        n_tasks += (w + kTaskWindows - 1) / kTaskWindows;
        out_pos += w;
    }
    if (meta) {
        h_task_off[n_runs] = n_tasks;
        h_out_off[n_runs] = out_pos;
    }
    if (out_pos != n_windows) {
        set_last_error("n_windows does not equal the sum of run_windows");
        return FMSI_HIP_EINVAL;
    }
    const bool dev_ptrs = flags & FMSI_HIP_DEVICE_PTRS;
    const size_t out_bytes = out_buffer_bytes(n_windows, flags);
    const size_t seq_bytes = (seq_bases / 32 + 2) * 8;
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
    // run metadata, then the task records and the result slots, in one staging buffer
    const size_t meta_bytes = (meta_words * 8 + 63) / 64 * 64;
    if ((rc = ensure_cap(&idx->d_tasks, &idx->d_tasks_cap, meta_bytes + 2 * n_tasks * sizeof(TaskRec)))) return rc;
    auto* d_meta = static_cast<uint64_t*>(idx->d_tasks);
    auto* d_rec = reinterpret_cast<TaskRec*>(static_cast<char*>(idx->d_tasks) + meta_bytes);
    auto* d_slots = reinterpret_cast<uint4*>(d_rec + n_tasks);   // per task: one 64-byte line of result bytes (first kernel), or its four result planes (32 bytes)
    if (meta) {
        FMSI_HIP_TRY(hipMemcpyAsync(d_meta, meta, meta_words * 8, hipMemcpyHostToDevice, s));
        if ((rc = meta_commit(idx, meta_slot, s))) return rc;
    }
    // FMSI_HIP_OUT_BITS: the compaction writes the bytes to the handle's scratch buffer and one more pass packs them
    PackAfter pack;
    const uint32_t caller_flags = flags;
    if ((rc = pack.begin(idx, flags, n_windows, d_out))) return rc;
    // the profiling events bracket everything the call runs on the device: prep, the query kernel and the compaction
    size_t slot;
    if ((rc = prof_begin(idx, s, &slot))) return rc;
    // (reads of one length through the S-block kernel: it cuts its tasks itself)
    const bool cut_in_kernel = v2 && uniform && n_tasks < (1ull << 32) && wpr < (1ull << 31);
    UniformRuns uni{};
    if (cut_in_kernel) {
        uni.seq = d_seq;
        uni.seq_words = seq_bases / 32 + 2;
        uni.start0 = run_start[0];
        uni.stride = stride;
        uni.wpr = static_cast<uint32_t>(wpr);
        uni.tpr = static_cast<uint32_t>((wpr + kTaskWindows - 1) / kTaskWindows);
    } else {
        hipLaunchKernelGGL(stream_prep_kernel, dim3(static_cast<uint32_t>((n_tasks + 255) / 256)), dim3(256), 0, s, d_seq, d_meta, d_meta + n_runs,
                           d_meta + 2 * n_runs + 1, n_runs, n_tasks, seq_bases / 32 + 2, d_rec, uniform ? run_start[0] : 0ull, uniform ? stride : 0ull, uniform ? wpr : 0ull);
        FMSI_HIP_TRY(hipGetLastError());
    }
    const bool use_table = idx->d_ptable && !(flags & FMSI_HIP_NO_PREFIX_TABLE) && k >= idx->prefix_len;
    long long* o64 = static_cast<long long*>(d_out);
    // (a table loaded with FMSI_HIP_LOAD_ASYNC_TABLE gets its signatures once enough windows have been streamed for the pass to pay for itself,
    // note_streamed -- then beside this call and the next ones, behind every allocation of the call, so that none of them waits for the pass)
    if (use_table) ::fmsi::note_streamed(idx, n_windows);
    if (v2) {
        const UniformRuns* up = cut_in_kernel ? &uni : nullptr;
        const bool ord = flags & FMSI_HIP_ORDERS;   // (every strand result without an order is -1: the kernel writes those, too -- flush_orders)
        static const bool prefill = std::getenv("FMSI_HIP_STREAM_ORD_PREFILL") != nullptr;   // (A/B runs: round 4's pass over the array before the kernel as well)
        if (ord && prefill) FMSI_HIP_TRY(hipMemsetAsync(d_out, 0xFF, out_bytes, s));
        if (k > 32) {
            if (ord) launch_stream2_variant<true, true>(idx, d_cnt != nullptr, use_table, d_rec, n_tasks, k, flags, d_slots, o64, d_cnt, s, up);
            else launch_stream2_variant<true, false>(idx, d_cnt != nullptr, use_table, d_rec, n_tasks, k, flags, d_slots, o64, d_cnt, s, up);
        } else {
            if (ord) launch_stream2_variant<false, true>(idx, d_cnt != nullptr, use_table, d_rec, n_tasks, k, flags, d_slots, o64, d_cnt, s, up);
            else launch_stream2_variant<false, false>(idx, d_cnt != nullptr, use_table, d_rec, n_tasks, k, flags, d_slots, o64, d_cnt, s, up);
        }
    } else if (k > 32) {
        if (use_table) {
            if (d_cnt) launch_stream<true, true, true>(idx, d_rec, n_tasks, k, flags, d_slots, o64, d_cnt, s);
            else launch_stream<false, true, true>(idx, d_rec, n_tasks, k, flags, d_slots, o64, d_cnt, s);
        } else {
            if (d_cnt) launch_stream<true, false, true>(idx, d_rec, n_tasks, k, flags, d_slots, o64, d_cnt, s);
            else launch_stream<false, false, true>(idx, d_rec, n_tasks, k, flags, d_slots, o64, d_cnt, s);
        }
    } else if (use_table) {
        if (d_cnt) launch_stream<true, true, false>(idx, d_rec, n_tasks, k, flags, d_slots, o64, d_cnt, s);
        else launch_stream<false, true, false>(idx, d_rec, n_tasks, k, flags, d_slots, o64, d_cnt, s);
    } else {
        if (d_cnt) launch_stream<true, false, false>(idx, d_rec, n_tasks, k, flags, d_slots, o64, d_cnt, s);
        else launch_stream<false, false, false>(idx, d_rec, n_tasks, k, flags, d_slots, o64, d_cnt, s);
    }
    FMSI_HIP_TRY(hipGetLastError());
    const uint32_t tpr = uniform ? static_cast<uint32_t>((wpr + kTaskWindows - 1) / kTaskWindows) : 0u;
    const bool run_wise = uniform && !(flags & FMSI_HIP_ORDERS) && wpr < (1ull << 31);   // (reads of one length: a copy per run)
    const bool bits_here = run_wise && pack.on && !(caller_flags & FMSI_HIP_STRICT);       // ... straight to the packed words
    const uint64_t* d_planes = reinterpret_cast<const uint64_t*>(d_slots);   // (the S-block kernel's slots)
    if (v2 && bits_here) {
        const uint64_t words = (n_windows + 63) / 64;
        hipLaunchKernelGGL(stream_planes_bits_uniform_kernel, dim3(static_cast<uint32_t>((words + 255) / 256)), dim3(256), 0, s, d_planes,
                           static_cast<uint32_t>(wpr), tpr, n_windows, static_cast<unsigned long long*>(d_out));
        FMSI_HIP_TRY(hipGetLastError());
    } else if (v2 && run_wise) {
        const uint64_t pieces = n_tasks * 4;
        hipLaunchKernelGGL(stream_planes_bytes_uniform_kernel, dim3(static_cast<uint32_t>((pieces + 255) / 256)), dim3(256), 0, s, d_planes,
                           static_cast<uint32_t>(wpr), tpr, pieces, pack.bytes);
        FMSI_HIP_TRY(hipGetLastError());
    } else if (v2 && !(flags & FMSI_HIP_ORDERS)) {
        const uint64_t threads = n_tasks * 4;
        hipLaunchKernelGGL(stream_planes_bytes_kernel, dim3(static_cast<uint32_t>((threads + 255) / 256)), dim3(256), 0, s, d_planes, d_rec, n_tasks, pack.bytes);
        FMSI_HIP_TRY(hipGetLastError());
    } else if (bits_here) {
        const uint64_t words = (n_windows + 63) / 64;
        hipLaunchKernelGGL(stream_compact_uniform_bits_kernel, dim3(static_cast<uint32_t>((words + 255) / 256)), dim3(256), 0, s,
                           reinterpret_cast<const uint8_t*>(d_slots), static_cast<uint32_t>(wpr), tpr, n_windows, static_cast<unsigned long long*>(d_out));
        FMSI_HIP_TRY(hipGetLastError());
    } else if (run_wise) {
        const uint64_t pieces = n_tasks * 4;
        hipLaunchKernelGGL(stream_compact_uniform_kernel, dim3(static_cast<uint32_t>((pieces + 255) / 256)), dim3(256), 0, s,
                           reinterpret_cast<const uint8_t*>(d_slots), static_cast<uint32_t>(wpr), tpr, pieces, pack.bytes);
        FMSI_HIP_TRY(hipGetLastError());
    } else if (!(flags & FMSI_HIP_ORDERS)) {
        const uint64_t threads = n_tasks * 4;
        hipLaunchKernelGGL(stream_compact_kernel, dim3(static_cast<uint32_t>((threads + 255) / 256)), dim3(256), 0, s,
                           reinterpret_cast<const uint8_t*>(d_slots), d_rec, n_tasks, pack.bytes);
        FMSI_HIP_TRY(hipGetLastError());
    }
    if (!bits_here && (rc = pack.end(n_windows, caller_flags, d_out, s))) return rc;
    if ((rc = prof_end(idx, s, slot))) return rc;
    if (!dev_ptrs) {
        FMSI_HIP_TRY(hipMemcpyAsync(out, idx->d_out, out_bytes, hipMemcpyDeviceToHost, s));
        if (counters) FMSI_HIP_TRY(hipMemcpyAsync(counters, idx->d_counters, FMSI_HIP_N_COUNTERS * 8, hipMemcpyDeviceToHost, s));
    }
    return 0;
}
