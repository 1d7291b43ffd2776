// Host-side parser of the reference's index files.  See host_index.h.
#include "host_index.h"
#include "common.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>

namespace fmsi {
namespace {

constexpr unsigned kRrrBlock = 63;       // RRR_BLOCK_SIZE, reference src/fms_index.h:51
constexpr unsigned kRrrSuper = 32;       // t_k default, rrr_vector.hpp:72

struct Binomials {
    uint64_t c[65][65];
    uint8_t space[64];   // bits used by the offset of a block of class k (rrr_helper.hpp:247-257)
    Binomials() {
        std::memset(c, 0, sizeof c);
        for (int n = 0; n <= 64; ++n) c[n][0] = 1;
        for (int n = 1; n <= 64; ++n)
            for (int k = 1; k <= 64; ++k) c[n][k] = c[n - 1][k - 1] + c[n - 1][k];
        for (int k = 0; k <= 63; ++k) {
            uint64_t v = c[63][k];
            space[k] = v == 1 ? 0 : static_cast<uint8_t>(64 - __builtin_clzll(v));
        }
    }
};
const Binomials& binomials() {
    static const Binomials b;
    return b;
}

inline uint64_t low_mask(unsigned len) { return len >= 64 ? ~0ull : ((1ull << len) - 1); }

// LSB-first bit field read; `words` must be readable one word past the field.
inline uint64_t read_bits(const uint64_t* words, uint64_t pos, unsigned len) {
    if (len == 0) return 0;
    const uint64_t* w = words + (pos >> 6);
    unsigned off = pos & 63;
    uint64_t v = w[0] >> off;
    if (off + len > 64) v |= w[1] << (64 - off);
    return v & low_mask(len);
}

bool read_file(const std::string& path, std::vector<uint8_t>& out) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize(sz > 0 ? static_cast<size_t>(sz) : 0);
    size_t got = out.empty() ? 0 : std::fread(out.data(), 1, out.size(), f);
    std::fclose(f);
    return got == out.size();
}

// Cursor over a serialized sdsl object.
struct Cursor {
    const uint8_t* p;
    size_t left;
    bool take(void* dst, size_t n) {
        if (left < n) return false;
        std::memcpy(dst, p, n);
        p += n;
        left -= n;
        return true;
    }
    // int_vector<1>: u64 bit count + ceil(bits/64) words
    bool bit_vector(BitVector& bv) {
        uint64_t bits;
        if (!take(&bits, 8)) return false;
        size_t nw = static_cast<size_t>((bits + 63) >> 6);
        if (left < nw * 8) return false;
        bv.bits = bits;
        bv.words.assign(nw + 2, 0);
        if (nw) std::memcpy(bv.words.data(), p, nw * 8);
        p += nw * 8;
        left -= nw * 8;
        if (bits & 63) bv.words[nw - 1] &= low_mask(bits & 63);
        return true;
    }
    // int_vector<0>: u64 bit count + u8 width + words
    bool int_vector(std::vector<uint64_t>& words, uint64_t& bits, uint8_t& width) {
        if (!take(&bits, 8) || !take(&width, 1)) return false;
        size_t nw = static_cast<size_t>((bits + 63) >> 6);
        if (left < nw * 8) return false;
        words.assign(nw + 2, 0);
        if (nw) std::memcpy(words.data(), p, nw * 8);
        p += nw * 8;
        left -= nw * 8;
        return true;
    }
};

// One 63-bit block from (class, offset): inverse of rrr_helper<63>::bin_to_nr (rrr_helper.hpp:304-320) --
// combinatorial number system, positions enumerated from the least significant bit.
inline uint64_t decode_block(unsigned cls, uint64_t nr, const Binomials& B) {
    if (cls == 0) return 0;
    if (cls == kRrrBlock) return low_mask(kRrrBlock);
    uint64_t bits = 0;
    unsigned remaining = kRrrBlock;
    for (unsigned pos = 0; pos < kRrrBlock && cls > 0; ++pos, --remaining) {
        uint64_t skip = B.c[remaining - 1][cls];
        if (nr >= skip) {
            bits |= 1ull << pos;
            nr -= skip;
            --cls;
        }
    }
    return bits;
}

inline void or_bits(uint64_t* words, uint64_t pos, uint64_t v) {
    unsigned off = pos & 63;
    words[pos >> 6] |= v << off;
    // (the spill word is touched only when bits really spill: the last block of a decoding thread's range ends exactly
    // at a word boundary, and the next word belongs to the neighbouring thread)
    if (off && (v >> (64 - off))) words[(pos >> 6) + 1] |= v >> (64 - off);
}

}  // namespace

void rrr_tables(uint64_t* c, uint8_t* space) {
    const Binomials& B = binomials();
    for (int n = 0; n < 64; ++n)
        for (int k = 0; k < 64; ++k) c[n * 64 + k] = B.c[n][k];
    std::memcpy(space, B.space, 64);
}

int decode_rrr63(const uint8_t* data, size_t size, BitVector& out, std::string& err) {
    const Binomials& B = binomials();
    Cursor cur{data, size};
    uint64_t n;
    std::vector<uint64_t> bt, btnrp, rank;
    uint64_t bt_bits, btnrp_bits, rank_bits;
    uint8_t bt_w, btnrp_w, rank_w;
    BitVector btnr, invert;
    if (!cur.take(&n, 8) || !cur.int_vector(bt, bt_bits, bt_w) || !cur.bit_vector(btnr) ||
        !cur.int_vector(btnrp, btnrp_bits, btnrp_w) || !cur.int_vector(rank, rank_bits, rank_w) ||
        !cur.bit_vector(invert)) {
        err = "truncated rrr_vector<63> image";
        return FMSI_HIP_EIO;
    }
    const uint64_t n_blocks = (n + kRrrBlock) / kRrrBlock;   // incl. dummy block when n % 63 == 0 (rrr_vector.hpp:148)
    const uint64_t n_super = (n_blocks + kRrrSuper - 1) / kRrrSuper;
    if (bt_w != 6 || bt_bits != n_blocks * 6 || btnrp_w == 0 || btnrp_bits != n_super * btnrp_w ||
        invert.bits != n_super || rank_w == 0) {
        err = "rrr_vector<63> image has inconsistent geometry";
        return FMSI_HIP_EIO;
    }
    out.bits = n;
    out.words.assign(static_cast<size_t>((n + 63) >> 6) + 3, 0);

    std::atomic<bool> corrupt{false};   // an offset pointer or class width that runs past the offsets' bit-vector
    auto decode_range = [&](uint64_t s0, uint64_t s1) {
        for (uint64_t s = s0; s < s1; ++s) {
            uint64_t pos = read_bits(btnrp.data(), s * btnrp_w, btnrp_w);
            if (pos > btnr.bits) {
                corrupt = true;
                return;
            }
            const bool inv = invert.get(s);
            const uint64_t b_end = std::min<uint64_t>((s + 1) * kRrrSuper, n_blocks);
            for (uint64_t b = s * kRrrSuper; b < b_end; ++b) {
                unsigned stored = static_cast<unsigned>(read_bits(bt.data(), b * 6, 6));
                unsigned cls = inv ? kRrrBlock - stored : stored;   // rrr_vector.hpp:262-263
                unsigned width = B.space[stored];
                if (width > btnr.bits - pos) {
                    corrupt = true;
                    return;
                }
                uint64_t nr = read_bits(btnr.words.data(), pos, width);
                pos += width;
                uint64_t first = b * kRrrBlock;
                if (first >= n) break;   // dummy block
                uint64_t v = decode_block(cls, nr, B);
                if (first + kRrrBlock > n) v &= low_mask(static_cast<unsigned>(n - first));
                if (v) or_bits(out.words.data(), first, v);
            }
        }
    };
    // two superblocks = 4032 bits = 63 words: ranges of an even number of superblocks are word-aligned
    unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    uint64_t pairs = (n_super + 1) / 2;
    if (pairs < 64 || hw == 1) {
        decode_range(0, n_super);
    } else {
        std::vector<std::thread> pool;
        uint64_t per = (pairs + hw - 1) / hw;
        for (unsigned t = 0; t < hw; ++t) {
            uint64_t a = std::min<uint64_t>(n_super, 2 * per * t), b = std::min<uint64_t>(n_super, 2 * per * (t + 1));
            if (a < b) pool.emplace_back(decode_range, a, b);
        }
        for (auto& th : pool) th.join();
    }
    if (corrupt) {
        err = "rrr_vector<63> image: block offsets run past the end of the offset vector";
        return FMSI_HIP_EIO;
    }
    return 0;
}

int peek_host_index(const std::string& prefix, uint64_t& n, int& k, bool& klcp_file) {
    const std::string base = prefix + ".fmsi";
    FILE* f = std::fopen((base + ".mask").c_str(), "rb");
    if (!f) return FMSI_HIP_EIO;
    uint64_t rows = 0;
    const bool ok = std::fread(&rows, 8, 1, f) == 1;
    std::fclose(f);
    if (!ok) return FMSI_HIP_EIO;
    f = std::fopen((base + ".misc").c_str(), "r");
    if (!f) return FMSI_HIP_EIO;
    unsigned long long d, c[4];
    int kk = 0;
    const int got = std::fscanf(f, "%llu %llu %llu %llu %llu %d", &d, &c[0], &c[1], &c[2], &c[3], &kk);
    std::fclose(f);
    if (got != 6) return FMSI_HIP_EIO;
    for (const char* part : {".ac_gt", ".ac", ".gt"}) {
        f = std::fopen((base + part).c_str(), "rb");
        if (!f) return FMSI_HIP_EIO;
        std::fclose(f);
    }
    f = std::fopen((base + ".klcp").c_str(), "rb");
    klcp_file = f != nullptr;
    if (f) std::fclose(f);
    n = rows;
    k = kk;
    return 0;
}

int load_host_index(const std::string& prefix, bool use_klcp, HostIndex& out, std::string& err) {
    const std::string base = prefix + ".fmsi";
    // the parts are independent files: read (and decode) them side by side, report the first failure in file order
    int rcs[5] = {0, 0, 0, 0, 0};
    std::string errs[5];
    auto load_bv_t = [&](int slot, const char* ext, BitVector& bv, bool required) {
        std::vector<uint8_t> raw;
        if (!read_file(base + "." + ext, raw)) {
            if (!required) {
                rcs[slot] = 1;
                return;
            }
            errs[slot] = "cannot read " + base + "." + ext;
            rcs[slot] = FMSI_HIP_EIO;
            return;
        }
        Cursor cur{raw.data(), raw.size()};
        if (!cur.bit_vector(bv)) {
            errs[slot] = "malformed bit vector " + base + "." + ext;
            rcs[slot] = FMSI_HIP_EIO;
        }
    };
    {
        std::vector<std::thread> pool;
        pool.emplace_back([&] { load_bv_t(0, "ac_gt", out.ac_gt, true); });
        pool.emplace_back([&] { load_bv_t(1, "ac", out.ac, true); });
        pool.emplace_back([&] { load_bv_t(2, "gt", out.gt, true); });
        if (use_klcp)   // fms_index.h:513-515: only when the file exists AND -S was given
            pool.emplace_back([&] { load_bv_t(4, "klcp", out.klcp, false); });
        {
            std::vector<uint8_t> raw;
            if (!read_file(base + ".mask", raw)) {
                errs[3] = "cannot read " + base + ".mask";
                rcs[3] = FMSI_HIP_EIO;
            } else {
                rcs[3] = std::min(0, decode_rrr63(raw.data(), raw.size(), out.mask, errs[3]));
            }
        }
        for (auto& th : pool) th.join();
    }
    for (int slot = 0; slot < 5; ++slot)
        if (rcs[slot] < 0) {
            err = errs[slot];
            return rcs[slot];
        }
    {
        FILE* f = std::fopen((base + ".misc").c_str(), "r");
        if (!f) {
            err = "cannot read " + base + ".misc";
            return FMSI_HIP_EIO;
        }
        unsigned long long d, c[4];
        int k;
        int got = std::fscanf(f, "%llu %llu %llu %llu %llu %d", &d, &c[0], &c[1], &c[2], &c[3], &k);
        std::fclose(f);
        if (got != 6) {
            err = "malformed " + base + ".misc";
            return FMSI_HIP_EIO;
        }
        out.dollar_position = d;
        for (int i = 0; i < 4; ++i) out.counts[i] = c[i];
        out.k = k;
    }
    // structural sanity: the three wavelet-tree nodes must be consistent with each other and the mask
    if (out.ac_gt.bits != out.mask.bits || out.ac.bits + out.gt.bits != out.ac_gt.bits) {
        err = "index parts disagree on the number of rows";
        return FMSI_HIP_EIO;
    }
    return 0;
}

}  // namespace fmsi

namespace fmsi {
namespace {
thread_local std::string t_last_error;
}
void set_last_error(const std::string& msg) { t_last_error = msg; }
const char* last_error() { return t_last_error.c_str(); }
}  // namespace fmsi

extern "C" {

int fmsi_host_index_load(const char* prefix, int use_klcp, fmsi_host_index** out) {
    if (!prefix || !out) return FMSI_HIP_EINVAL;
    auto* h = new fmsi_host_index();
    std::string err;
    int rc = fmsi::load_host_index(prefix, use_klcp != 0, h->ix, err);
    if (rc < 0) {
        fmsi::set_last_error(err);
        delete h;
        return rc;
    }
    *out = h;
    return 0;
}

void fmsi_host_index_free(fmsi_host_index* h) { delete h; }

int fmsi_host_index_peek(const char* prefix, uint64_t* n, int32_t* k, int32_t* has_klcp_file) {
    if (!prefix || !n || !k || !has_klcp_file) return FMSI_HIP_EINVAL;
    int kk = 0;
    bool kl = false;
    const int rc = fmsi::peek_host_index(prefix, *n, kk, kl);
    if (rc) {
        fmsi::set_last_error(std::string("cannot read the index files of ") + prefix);
        return rc;
    }
    *k = kk;
    *has_klcp_file = kl ? 1 : 0;
    return 0;
}

int fmsi_host_index_view(const fmsi_host_index* h, fmsi_index_view* v) {
    if (!h || !v) return FMSI_HIP_EINVAL;
    const fmsi::HostIndex& ix = h->ix;
    std::memset(v, 0, sizeof *v);
    v->ac_gt = ix.ac_gt.words.data(); v->ac_gt_bits = ix.ac_gt.bits;
    v->ac = ix.ac.words.data(); v->ac_bits = ix.ac.bits;
    v->gt = ix.gt.words.data(); v->gt_bits = ix.gt.bits;
    v->mask = ix.mask.words.data(); v->mask_bits = ix.mask.bits;
    v->klcp = ix.klcp.bits ? ix.klcp.words.data() : nullptr; v->klcp_bits = ix.klcp.bits;
    for (int i = 0; i < 4; ++i) v->counts[i] = ix.counts[i];
    v->dollar_position = ix.dollar_position;
    v->k = ix.k;
    v->prefix_len = 0;
    return 0;
}

uint64_t fmsi_host_pack_2bit(const char* seq, uint64_t len, uint64_t* words) {
    // codes of reference src/kmers.h:3-20: A/a=0 C/c=1 G/g=2 T/t=3
    static const int8_t code[256] = {
#define X4 -1, -1, -1, -1
#define X16 X4, X4, X4, X4
        X16, X16, X16, X16,
        -1, 0, -1, 1, -1, -1, -1, 2, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 3, -1, -1, -1, X4, X4,
        -1, 0, -1, 1, -1, -1, -1, 2, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 3, -1, -1, -1, X4, X4,
        X16, X16, X16, X16, X16, X16, X16, X16
#undef X16
#undef X4
    };
    uint64_t nw = len / 32 + 2;
    std::memset(words, 0, nw * 8);
    for (uint64_t i = 0; i < len; ++i) {
        int c = code[static_cast<uint8_t>(seq[i])];
        if (c < 0) return i;
        words[i >> 5] |= static_cast<uint64_t>(c) << (2 * (i & 31));
    }
    return len;
}

}  // extern "C"
