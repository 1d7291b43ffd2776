// Host-side builder of the interleaved device layout (common.h) from the reference's three
// wavelet-tree node bit-vectors (src/fms_index.h:427-450 builds them; rank_support_v5 is rebuilt at
// load and never stored, :506-512, so the in-memory layout is free to choose) and the decoded mask.
#include "layout.h"

#include <algorithm>
#include <thread>

#include "common.h"

namespace fmsi {
namespace {

uint64_t deposit_sw(uint64_t src, uint64_t mask) {
    uint64_t r = 0;
    while (mask) {
        uint64_t b = mask & (~mask + 1);
        if (src & 1) r |= b;
        src >>= 1;
        mask &= mask - 1;
    }
    return r;
}
__attribute__((target("bmi2"))) uint64_t deposit_hw(uint64_t src, uint64_t mask) { return __builtin_ia32_pdep_di(src, mask); }

inline uint64_t take_bits(const uint64_t* words, uint64_t pos, unsigned len) {
    if (len == 0) return 0;
    const uint64_t* w = words + (pos >> 6);
    unsigned off = pos & 63;
    uint64_t v = w[0] >> off;
    if (off + len > 64) v |= w[1] << (64 - off);
    return len >= 64 ? v : (v & ((1ull << len) - 1));
}

}  // namespace

// Turns the three wavelet-tree node bit-vectors + decoded mask into interleaved blocks (common.h).
int build_layout(const fmsi_index_view& v, HostLayout& L) {
    const uint64_t n = v.mask_bits;
    if (v.ac_gt_bits != n || v.ac_bits + v.gt_bits != n) {
        set_last_error("index view: ac_gt/ac/gt/mask sizes are inconsistent");
        return FMSI_HIP_EINVAL;
    }
    // the scalars of .misc and the optional kLCP vector must belong to these rows (the reference reads them blindly, fms_index.h:513-524):
    // LF steps add C[c] to a rank and -S scans klcp[] by row, so anything outside [0, n] would index past the arrays on the device
    if (v.dollar_position >= n || v.counts[0] != 1 || v.counts[0] > v.counts[1] || v.counts[1] > v.counts[2] || v.counts[2] > v.counts[3] ||
        v.counts[3] > n || v.counts[2] != v.ac_bits) {
        set_last_error("index view: dollar_position / counts[] do not belong to these bit-vectors");
        return FMSI_HIP_EINVAL;
    }
    if (v.klcp && v.klcp_bits && v.klcp_bits != n) {
        set_last_error("index view: the kLCP vector has another number of rows");
        return FMSI_HIP_EINVAL;
    }
    const uint64_t n_words = (n + 63) >> 6;
    const uint64_t n_blocks = ((n + kBlockRows - 1) >> kBlockRowsLog2) + 1;
    L.n_blocks = n_blocks;
    L.blocks.resize(n_blocks * kBlockDwords);   // (not cleared: pass 2 writes every dword)
    L.mrank.assign(n_blocks + 1, 0);
    const bool hw_pdep = __builtin_cpu_supports("bmi2");

    // pass 1: gt offsets per chunk of words
    const uint64_t chunk_words = 1u << 14;   // even, so chunks hold whole blocks
    const uint64_t n_chunks = (n_words + chunk_words - 1) / chunk_words;
    std::vector<uint64_t> gt_off(n_chunks + 1, 0);
    for (uint64_t c = 0; c < n_chunks; ++c) {
        uint64_t s = 0;
        const uint64_t w1 = std::min(n_words, (c + 1) * chunk_words);
        for (uint64_t w = c * chunk_words; w < w1; ++w) s += __builtin_popcountll(v.ac_gt[w]);
        gt_off[c + 1] = gt_off[c] + s;
    }
    if (gt_off[n_chunks] != v.gt_bits) {
        set_last_error("index view: popcount(ac_gt) != |gt|");
        return FMSI_HIP_EINVAL;
    }
    // pass 2 (parallel): planes + per-block (non-cumulative) counts in dw 0..3, mask popcount in mrank[b+1]
    auto fill = [&](uint64_t c0, uint64_t c1) {
        for (uint64_t c = c0; c < c1; ++c) {
            uint64_t gt_pos = gt_off[c];
            const uint64_t w1 = std::min(n_words, (c + 1) * chunk_words);
            for (uint64_t w = c * chunk_words; w < w1; ++w) {
                const uint64_t valid = (w == n_words - 1 && (n & 63)) ? ((1ull << (n & 63)) - 1) : ~0ull;
                const uint64_t g = v.ac_gt[w] & valid;
                const uint64_t ng = ~g & valid;
                const unsigned cg = __builtin_popcountll(g), ca = __builtin_popcountll(ng);
                const uint64_t ac_pos = (w << 6) - gt_pos;
                const uint64_t gbits = take_bits(v.gt, gt_pos, cg);
                const uint64_t abits = take_bits(v.ac, ac_pos, ca);
                const uint64_t lo = hw_pdep ? (deposit_hw(gbits, g) | deposit_hw(abits, ng))
                                            : (deposit_sw(gbits, g) | deposit_sw(abits, ng));
                const uint64_t m = v.mask[w] & valid;
                gt_pos += cg;
                uint32_t* blk = &L.blocks[(w >> 1) * kBlockDwords];
                const unsigned half = (w & 1) * 2;
                if (half == 0) std::fill(blk, blk + kBlockDwords, 0u);   // (a block's two words belong to one chunk)
                blk[4 + half] = static_cast<uint32_t>(g);
                blk[5 + half] = static_cast<uint32_t>(g >> 32);
                blk[8 + half] = static_cast<uint32_t>(lo);
                blk[9 + half] = static_cast<uint32_t>(lo >> 32);
                blk[12 + half] = static_cast<uint32_t>(m);
                blk[13 + half] = static_cast<uint32_t>(m >> 32);
                blk[0] += __builtin_popcountll(ng & ~lo);   // A (incl. the '$' slot)
                blk[1] += __builtin_popcountll(ng & lo);    // C
                blk[2] += __builtin_popcountll(g & ~lo);    // G
                blk[3] += __builtin_popcountll(g & lo);     // T
                L.mrank[(w >> 1) + 1] += __builtin_popcountll(m);
            }
        }
    };
    // blocks past the last word (the padding block, and the second half's block of an odd word count is covered above)
    const uint64_t blocks_filled = (n_words + 1) >> 1;
    std::fill(L.blocks.begin() + static_cast<std::ptrdiff_t>(blocks_filled * kBlockDwords), L.blocks.end(), 0u);
    // pass 3: exclusive prefix sums of the per-block counts (dw 0..3) and inclusive ones of the mask popcounts -- per range of
    // blocks in parallel: totals per range, offsets from a scan over the ranges, then each range scans itself
    unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (n_chunks < 4) hw = 1;
    const uint64_t per = (n_chunks + hw - 1) / hw;
    const uint64_t blocks_per_chunk = chunk_words / 2;
    struct Range {
        uint64_t c0, c1, b0, b1;
        uint32_t tot[5];
    };
    std::vector<Range> ranges;
    for (unsigned t = 0; t < hw; ++t) {
        const uint64_t a = std::min(n_chunks, per * t), b = std::min(n_chunks, per * (t + 1));
        if (a >= b) continue;
        Range r{a, b, a * blocks_per_chunk, b == n_chunks ? n_blocks : b * blocks_per_chunk, {0, 0, 0, 0, 0}};
        ranges.push_back(r);
    }
    auto run_all = [&](auto&& fn) {
        if (ranges.size() == 1) {
            fn(ranges[0]);
            return;
        }
        std::vector<std::thread> pool;
        for (Range& r : ranges) pool.emplace_back([&fn, &r] { fn(r); });
        for (auto& th : pool) th.join();
    };
    run_all([&](Range& r) {
        fill(r.c0, r.c1);
        for (uint64_t b = r.b0; b < r.b1; ++b) {
            const uint32_t* blk = &L.blocks[b * kBlockDwords];
            for (int s = 0; s < 4; ++s) r.tot[s] += blk[s];
            r.tot[4] += L.mrank[b + 1];
        }
    });
    uint32_t off[5] = {0, 0, 0, 0, 0};
    for (Range& r : ranges)
        for (int s = 0; s < 5; ++s) {
            const uint32_t t = r.tot[s];
            r.tot[s] = off[s];   // now: what lies before the range
            off[s] += t;
        }
    run_all([&](Range& r) {
        uint32_t run[4] = {r.tot[0], r.tot[1], r.tot[2], r.tot[3]};
        uint32_t mr = r.tot[4];
        for (uint64_t b = r.b0; b < r.b1; ++b) {
            uint32_t* blk = &L.blocks[b * kBlockDwords];
            for (int s = 0; s < 4; ++s) {
                const uint32_t here = blk[s];
                blk[s] = run[s];
                run[s] += here;
            }
            mr += L.mrank[b + 1];
            L.mrank[b + 1] = mr;
        }
    });
    return 0;
}


}  // namespace fmsi

extern "C" {

uint64_t fmsi_host_layout_blocks(const fmsi_index_view* view) {
    if (!view) return 0;
    return ((view->mask_bits + fmsi::kBlockRows - 1) >> fmsi::kBlockRowsLog2) + 1;
}

int fmsi_host_layout_build(const fmsi_index_view* view, uint32_t* blocks, uint32_t* mrank) {
    if (!view || !blocks || !mrank) return FMSI_HIP_EINVAL;
    fmsi::HostLayout L;
    int rc = fmsi::build_layout(*view, L);
    if (rc) return rc;
    std::copy(L.blocks.begin(), L.blocks.end(), blocks);
    std::copy(L.mrank.begin(), L.mrank.end(), mrank);
    return 0;
}

}  // extern "C"
