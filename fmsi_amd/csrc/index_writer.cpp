// Writer of the reference's on-disk index (<prefix>.fmsi.{ac_gt,ac,gt,mask,klcp,misc}): counterpart of
// dump_index() (reference src/fms_index.h:484-500) and of the sdsl serialisers it calls -- int_vector<1>
// (sdsl/int_vector.hpp:593-609, 1563-1582) and rrr_vector<63> incl. its encoder (sdsl/rrr_vector.hpp:145-238,
// 350-362; rrr_helper.hpp:304-320).  Product code, independent of oracle/; must be byte-identical to the reference
// (tests/test_host.py::test_rrr_encoder_bytes, tests/test_gpu_index_cli.py).
#include "index_writer.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace fmsi {
namespace {

constexpr unsigned kBs = 63, kK = 32;

struct Binom {
    uint64_t c[65][65];
    uint8_t space[64];
    Binom() {
        std::memset(c, 0, sizeof c);
        for (int n = 0; n <= 64; ++n) c[n][0] = 1;
        for (int n = 1; n <= 64; ++n)
            for (int k = 1; k <= 64; ++k) c[n][k] = c[n - 1][k - 1] + c[n - 1][k];
        for (int k = 0; k <= 63; ++k) space[k] = c[63][k] == 1 ? 0 : static_cast<uint8_t>(64 - __builtin_clzll(c[63][k]));
    }
};
const Binom& binom() {
    static const Binom b;
    return b;
}

inline unsigned hi_bit(uint64_t x) { return x ? 63u - static_cast<unsigned>(__builtin_clzll(x)) : 0u; }   // bits::hi, hi(0) = 0
inline uint64_t low_mask(unsigned len) { return len >= 64 ? ~0ull : ((1ull << len) - 1); }

inline uint64_t get_bits(const uint64_t* w, uint64_t pos, unsigned len) {
    if (!len) return 0;
    const uint64_t* p = w + (pos >> 6);
    const unsigned off = pos & 63;
    uint64_t v = p[0] >> off;
    if (off + len > 64) v |= p[1] << (64 - off);
    return v & low_mask(len);
}
inline void put_bits(std::vector<uint64_t>& w, uint64_t pos, uint64_t x, unsigned len) {
    if (!len) return;
    x &= low_mask(len);
    const unsigned off = pos & 63;
    w[pos >> 6] |= x << off;
    if (off + len > 64) w[(pos >> 6) + 1] |= x >> (64 - off);
}

// combinatorial number of a 63-bit block, positions enumerated from the least significant bit (rrr_helper.hpp:304-320)
inline uint64_t bin_to_nr(uint64_t bin, const Binom& B) {
    if (bin == 0 || bin == low_mask(kBs)) return 0;
    uint64_t nr = 0;
    unsigned k = static_cast<unsigned>(__builtin_popcountll(bin)), nn = kBs;
    while (bin) {
        if (bin & 1) {
            nr += B.c[nn - 1][k];
            --k;
        }
        bin >>= 1;
        --nn;
    }
    return nr;
}

struct Out {
    FILE* f = nullptr;
    bool ok = true;
    explicit Out(const std::string& path) : f(std::fopen(path.c_str(), "wb")) { ok = f != nullptr; }
    ~Out() {
        if (f) std::fclose(f);
    }
    void raw(const void* p, size_t n) {
        if (ok && n && std::fwrite(p, 1, n, f) != n) ok = false;
    }
    void u64(uint64_t v) { raw(&v, 8); }
    void u8(uint8_t v) { raw(&v, 1); }
    // int_vector<1>: bit count + ceil(bits/64) words
    void bit_vector(const uint64_t* words, uint64_t bits) {
        u64(bits);
        raw(words, static_cast<size_t>((bits + 63) >> 6) * 8);
    }
    // int_vector<0>: bit count + width + words
    void int_vector(const std::vector<uint64_t>& words, uint64_t bits, uint8_t width) {
        u64(bits);
        u8(width);
        raw(words.data(), static_cast<size_t>((bits + 63) >> 6) * 8);
    }
};

}  // namespace

// rrr_vector<63> image of `bits` (n bits), written to `o` -- the constructor of sdsl/rrr_vector.hpp:145-238 followed
// by serialize() (:350-362)
// Threads: the encoder's loops go superblock by superblock, and everything a superblock writes -- 32 class fields of 6 bits (3 words), one
// sample each of the offset pointers and of the ranks, one inversion bit, its blocks' offsets -- sits in words of its own once the
// superblocks are taken 64 at a time (64 samples of w bits are w words), except the offsets, whose bit positions run through: those go in
// with atomic ORs.  What a range of superblocks needs from the ones before it (offset bits and ones so far) comes from a first pass that
// also yields the classes; the widths do not depend on the inversion (space[x] == space[63 - x]).
inline void put_bits_shared(uint64_t* w, uint64_t pos, uint64_t x, unsigned len) {
    if (!len) return;
    x &= low_mask(len);
    const unsigned off = pos & 63;
    __atomic_fetch_or(w + (pos >> 6), x << off, __ATOMIC_RELAXED);
    if (off + len > 64) __atomic_fetch_or(w + (pos >> 6) + 1, x >> (64 - off), __ATOMIC_RELAXED);
}

template <class F>
void in_ranges(uint64_t items, uint64_t grain, F&& fn) {   // fn(lo, hi) over [0, items) cut at multiples of `grain`, on up to 16 threads
    const uint64_t chunks = (items + grain - 1) / grain;
    const unsigned threads = static_cast<unsigned>(std::max<uint64_t>(1, std::min<uint64_t>({16, std::thread::hardware_concurrency(), chunks / 4 + 1})));
    if (threads <= 1) {
        fn(uint64_t(0), items);
        return;
    }
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t) {
        const uint64_t lo = std::min(items, chunks * t / threads * grain), hi = t + 1 == threads ? items : std::min(items, chunks * (t + 1) / threads * grain);
        if (hi > lo) pool.emplace_back([&fn, lo, hi] { fn(lo, hi); });
    }
    for (std::thread& th : pool) th.join();
}

static void write_rrr63(Out& o, const uint64_t* bits, uint64_t n) {
    const Binom& B = binom();
    const uint64_t n_bt = (n + kBs) / kBs;                      // incl. the dummy block when n % 63 == 0
    const uint64_t n_super = (n_bt + kK - 1) / kK;
    std::vector<uint8_t> bt(n_bt, 0);
    // pass 1: classes; per superblock its ones and its offset bits
    std::vector<uint32_t> sb_ones(n_super + 1, 0), sb_bits(n_super + 1, 0);
    in_ranges(n_super, 64, [&](uint64_t s0, uint64_t s1) {
        for (uint64_t s = s0; s < s1; ++s) {
            uint32_t ones = 0, width = 0;
            for (uint64_t b = s * kK; b < std::min<uint64_t>((s + 1) * kK, n_bt) && b * kBs < n; ++b) {
                const uint64_t pos = b * kBs;
                const unsigned len = static_cast<unsigned>(std::min<uint64_t>(kBs, n - pos));
                const unsigned x = static_cast<unsigned>(__builtin_popcountll(get_bits(bits, pos, len)));
                bt[b] = static_cast<uint8_t>(x);
                ones += x;
                width += B.space[x];                            // space is symmetric in x <-> 63 - x
            }
            sb_ones[s] = ones;
            sb_bits[s] = width;
        }
    });
    std::vector<uint64_t> ones_before(n_super + 1, 0), bits_before(n_super + 1, 0);
    for (uint64_t s = 0; s < n_super; ++s) {
        ones_before[s + 1] = ones_before[s] + sb_ones[s];
        bits_before[s + 1] = bits_before[s] + sb_bits[s];
    }
    const uint64_t sum_rank = ones_before[n_super], btnr_bits = bits_before[n_super];
    const uint64_t btnr_size = std::max<uint64_t>(btnr_bits, 64);
    std::vector<uint64_t> btnr((btnr_size + 63) / 64 + 1, 0);
    const uint8_t btnrp_w = static_cast<uint8_t>(hi_bit(btnr_bits) + 1), rank_w = static_cast<uint8_t>(hi_bit(sum_rank) + 1);
    const uint64_t n_rank = n_super + ((n % (kK * kBs)) > 0 ? 1 : 0);
    std::vector<uint64_t> btnrp((n_super * btnrp_w + 63) / 64 + 1, 0), rank((n_rank * rank_w + 63) / 64 + 1, 0);
    std::vector<uint64_t> invert((n_super + 63) / 64 + 1, 0), bt_packed((n_bt * 6 + 63) / 64 + 1, 0);
    // pass 2: samples, inversion, class fields, offsets
    in_ranges(n_super, 64, [&](uint64_t s0, uint64_t s1) {
        for (uint64_t s = s0; s < s1; ++s) {
            uint64_t pos_bits = bits_before[s];
            const uint64_t b0 = s * kK, b1 = std::min<uint64_t>(b0 + kK, n_bt);
            if (b0 * kBs < n) {   // a superblock that consists of the dummy block alone gets no samples (rrr_vector.hpp:180-233)
                put_bits(btnrp, s * btnrp_w, pos_bits, btnrp_w);
                put_bits(rank, s * rank_w, ones_before[s], rank_w);
            }
            bool inv = false;
            if (b0 + kK <= n_bt) {                                  // only complete superblocks may be inverted (:186-199)
                unsigned heavy = 0;
                for (uint64_t b = b0; b < b1; ++b) heavy += bt[b] > kBs / 2;
                inv = heavy > kK / 2;
            }
            if (inv) invert[s >> 6] |= 1ull << (s & 63);
            for (uint64_t b = b0; b < b1; ++b) {
                const unsigned x = bt[b];                           // true popcount
                const unsigned stored = inv ? kBs - x : x;
                put_bits(bt_packed, b * 6, stored, 6);
                const uint64_t first = b * kBs;
                if (first >= n) continue;                           // dummy block: class 0, no offset
                const unsigned width = B.space[stored];
                if (width) {
                    const unsigned len = static_cast<unsigned>(std::min<uint64_t>(kBs, n - first));
                    put_bits_shared(btnr.data(), pos_bits, bin_to_nr(get_bits(bits, first, len), B), width);
                }
                pos_bits += width;
            }
        }
    });
    const uint64_t ones = sum_rank;
    put_bits(rank, (n_rank - 1) * rank_w, ones, rank_w);        // last element = total number of ones (:236)
    o.u64(n);
    o.int_vector(bt_packed, n_bt * 6, 6);
    o.bit_vector(btnr.data(), btnr_size);
    o.int_vector(btnrp, n_super * btnrp_w, btnrp_w);
    o.int_vector(rank, n_rank * rank_w, rank_w);
    o.bit_vector(invert.data(), n_super);
}

int write_index_files(const std::string& prefix, const fmsi_index_view& v, std::string& err) {
    const std::string base = prefix + ".fmsi";
    // the plain bit-vectors go out side by side while the mask is encoded (each file has its own thread; the encoder has its own)
    std::string errs[4];
    bool oks[4] = {true, true, true, true};
    auto bv_into = [&](int slot, const char* ext, const uint64_t* words, uint64_t bits) {
        try {
            Out o(base + "." + ext);
            o.bit_vector(words, bits);
            oks[slot] = o.ok;
        } catch (...) {   // (out of memory in a writer thread: an error of this call, not std::terminate)
            oks[slot] = false;
        }
        if (!oks[slot]) errs[slot] = "cannot write " + base + "." + ext;
    };
    std::vector<std::thread> writers;
    struct JoinAll {   // whatever leaves this function -- the encoder may throw std::bad_alloc -- no thread is destroyed while joinable
        std::vector<std::thread>& t;
        ~JoinAll() {
            for (std::thread& x : t)
                if (x.joinable()) x.join();
        }
    } join_all{writers};
    bool mask_ok = false;
    try {
        writers.reserve(4);
        writers.emplace_back(bv_into, 0, "ac_gt", v.ac_gt, v.ac_gt_bits);
        writers.emplace_back(bv_into, 1, "ac", v.ac, v.ac_bits);
        writers.emplace_back(bv_into, 2, "gt", v.gt, v.gt_bits);
        if (v.klcp && v.klcp_bits) writers.emplace_back(bv_into, 3, "klcp", v.klcp, v.klcp_bits);   // fms_index.h:490-492
        Out o(base + ".mask");
        write_rrr63(o, v.mask, v.mask_bits);
        mask_ok = o.ok;
    } catch (const std::exception& e) {
        for (std::thread& t : writers)
            if (t.joinable()) t.join();
        err = std::string("cannot write the index files: ") + e.what();
        return FMSI_HIP_ENOMEM;
    }
    for (std::thread& t : writers) t.join();
    for (int i = 0; i < 4; ++i)
        if (!oks[i]) {
            err = errs[i];
            return FMSI_HIP_EIO;
        }
    if (!mask_ok) {
        err = "cannot write " + base + ".mask";
        return FMSI_HIP_EIO;
    }
    FILE* f = std::fopen((base + ".misc").c_str(), "w");
    if (!f) {
        err = "cannot write " + base + ".misc";
        return FMSI_HIP_EIO;
    }
    std::fprintf(f, "%llu\n", static_cast<unsigned long long>(v.dollar_position));
    for (int i = 0; i < 4; ++i) std::fprintf(f, "%llu\n", static_cast<unsigned long long>(v.counts[i]));
    std::fprintf(f, "%d\n", v.k);
    std::fclose(f);
    return 0;
}

namespace {
uint64_t extract_sw(uint64_t src, uint64_t mask) {   // software pext
    uint64_t r = 0;
    unsigned k = 0;
    while (mask) {
        const uint64_t b = mask & (~mask + 1);
        if (src & b) r |= 1ull << k;
        ++k;
        mask &= mask - 1;
    }
    return r;
}
__attribute__((target("bmi2"))) uint64_t extract_hw(uint64_t src, uint64_t mask) { return __builtin_ia32_pext_di(src, mask); }
}  // namespace

void vectors_from_blocks(const uint32_t* blocks, uint64_t n, HostVectors& out) {
    const uint64_t words = (n + 63) >> 6;
    out.n = n;
    out.ac_gt.assign(words + 2, 0);
    out.mask.assign(words + 2, 0);
    std::vector<uint64_t> lo(words + 2, 0);
    // ranges of 4096 words on threads: the planes of a word come out of its block; how many G/T (A/C) bits lie before a range is a
    // prefix sum over the ranges, and the compacted bits go in with atomic ORs (neighbouring ranges share a word of `ac` / `gt`)
    constexpr uint64_t kGrain = 4096;
    const uint64_t chunks = (words + kGrain - 1) / kGrain;
    std::vector<uint64_t> chunk_gt(chunks + 1, 0);
    in_ranges(words, kGrain, [&](uint64_t w0, uint64_t w1) {
        for (uint64_t c = w0 / kGrain; c * kGrain < w1; ++c) {
            uint64_t gt = 0;
            for (uint64_t w = c * kGrain; w < std::min(w1, (c + 1) * kGrain); ++w) {
                const uint32_t* blk = blocks + (w >> 1) * kBlockDwords;
                const unsigned half = static_cast<unsigned>(w & 1) * 2;
                const uint64_t valid = (w == words - 1 && (n & 63)) ? low_mask(static_cast<unsigned>(n & 63)) : ~0ull;
                out.ac_gt[w] = ((static_cast<uint64_t>(blk[5 + half]) << 32) | blk[4 + half]) & valid;
                lo[w] = ((static_cast<uint64_t>(blk[9 + half]) << 32) | blk[8 + half]) & valid;
                out.mask[w] = ((static_cast<uint64_t>(blk[13 + half]) << 32) | blk[12 + half]) & valid;
                gt += static_cast<uint64_t>(__builtin_popcountll(out.ac_gt[w]));
            }
            chunk_gt[c + 1] = gt;
        }
    });
    for (uint64_t c = 0; c < chunks; ++c) chunk_gt[c + 1] += chunk_gt[c];
    const uint64_t gt_bits = chunk_gt[chunks];
    out.gt_bits = gt_bits;
    out.ac_bits = n - gt_bits;
    out.ac.assign((out.ac_bits + 63) / 64 + 2, 0);
    out.gt.assign((out.gt_bits + 63) / 64 + 2, 0);
    const bool hw = __builtin_cpu_supports("bmi2");
    in_ranges(words, kGrain, [&](uint64_t w0, uint64_t w1) {
        uint64_t gp = chunk_gt[w0 / kGrain];
        uint64_t ap = std::min(w0 * 64, n) - gp;   // valid rows before the range that are not G/T
        for (uint64_t w = w0; w < w1; ++w) {
            const uint64_t valid = (w == words - 1 && (n & 63)) ? low_mask(static_cast<unsigned>(n & 63)) : ~0ull;
            const uint64_t g = out.ac_gt[w], ng = ~g & valid;
            const unsigned cg = static_cast<unsigned>(__builtin_popcountll(g)), ca = static_cast<unsigned>(__builtin_popcountll(ng));
            put_bits_shared(out.gt.data(), gp, hw ? extract_hw(lo[w], g) : extract_sw(lo[w], g), cg);
            put_bits_shared(out.ac.data(), ap, hw ? extract_hw(lo[w], ng) : extract_sw(lo[w], ng), ca);
            gp += cg;
            ap += ca;
        }
    });
}

}  // namespace fmsi

extern "C" int fmsi_host_index_write(const char* prefix, const fmsi_index_view* view) {
    if (!prefix || !view || !view->ac_gt || !view->ac || !view->gt || !view->mask) {
        fmsi::set_last_error("null argument");
        return FMSI_HIP_EINVAL;
    }
    std::string err;
    int rc = fmsi::write_index_files(prefix, *view, err);
    if (rc) fmsi::set_last_error(err);
    return rc;
}
