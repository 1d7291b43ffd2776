// ParallelGunzip: see pgunzip.h.  Deflate as RFC 1951 defines it, the checks of a block header as zlib makes them (inftrees.c's rules for
// over-subscribed and incomplete codes, inflate.c's for the counts), so that what this decoder takes for a block is what zlib would.
#include "pgunzip.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <immintrin.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

namespace fmsi {
namespace {

constexpr uint32_t kWin = 32768;
// A chunk's text may be this long and no longer: text that deflates fifty-fold and more (a megabase of one letter) would otherwise make a round
// of 64 chunks hold tens of gigabytes; such a round fails and zlib -- which inflates that kind of stream at a gigabyte a second -- goes on.
constexpr size_t kMaxChunkText = size_t(96) << 20;
constexpr uint16_t kMarker = 0x8000;   // symbol = kMarker | index into the window the chunk was not given (kWin - 1: the byte right in front)

constexpr uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
constexpr uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
constexpr uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
constexpr uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
constexpr uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

inline uint64_t load64(const unsigned char* p) {
    uint64_t w;
    std::memcpy(&w, p, 8);
    return w;
}

// 64 bits of the stream from bit position `bit` on (zeros behind the end)
inline uint64_t peek64(const unsigned char* d, size_t size, uint64_t bit) {
    const size_t b = static_cast<size_t>(bit >> 3);
    const unsigned s = static_cast<unsigned>(bit & 7);
    if (b + 9 <= size) {
        uint64_t v = load64(d + b) >> s;
        if (s) v |= static_cast<uint64_t>(d[b + 8]) << (64 - s);
        return v;
    }
    uint64_t v = 0;
    for (unsigned i = 0; i < 9 && b + i < size; ++i) {
        const uint64_t x = d[b + i];
        if (i * 8 >= s) {
            if (i * 8 - s < 64) v |= x << (i * 8 - s);
        } else {
            v |= x >> (s - i * 8);
        }
    }
    return v;
}

struct BitReader {
    const unsigned char* d = nullptr;
    size_t size = 0, pos = 0;   // pos: the next byte to take
    uint64_t buf = 0;
    unsigned cnt = 0;           // valid bits in buf
    bool overrun = false;
    void init(const unsigned char* data, size_t n, uint64_t bit) {
        d = data;
        size = n;
        pos = static_cast<size_t>(bit >> 3);
        buf = 0;
        cnt = 0;
        overrun = pos > size;
        if (overrun) pos = size;
        refill();
        const unsigned s = static_cast<unsigned>(bit & 7);
        if (s) {
            if (cnt >= s) {
                buf >>= s;
                cnt -= s;
            } else {
                overrun = true;
            }
        }
    }
    inline void refill() {
        if (pos + 8 <= size) {   // (the bits above cnt that this leaves in buf are the stream's own next bits: OR-ed again by the next refill, unchanged)
            buf |= load64(d + pos) << cnt;
            const unsigned add = (63u - cnt) >> 3;
            pos += add;
            cnt += add * 8u;
        } else {
            while (cnt <= 56 && pos < size) {
                buf |= static_cast<uint64_t>(d[pos++]) << cnt;
                cnt += 8;
            }
        }
    }
    inline void drop(unsigned n) {
        if (n > cnt) {
            overrun = true;
            n = cnt;
        }
        buf >>= n;
        cnt -= n;
    }
    inline uint32_t take(unsigned n) {   // n <= 32
        const uint32_t v = static_cast<uint32_t>(buf & ((1ull << n) - 1ull));
        drop(n);
        return v;
    }
    uint64_t bit_pos() const { return static_cast<uint64_t>(pos) * 8u - cnt; }
    void align_byte() { drop(cnt & 7u); }
};

// A canonical Huffman code as a two-level table over the LSB-first bits.  Entry: bits 0-3 the code's length (what to drop), bit 4 "go to the
// sub-table at `value`, indexed by that many further bits", bit 5 no code, bit 6 literal, bit 7 end of block, bits 8-11 extra bits that follow,
// bits 16-31 value: the literal, the length's / distance's base, or the symbol itself (the code-length code).
struct Huff {
    static constexpr uint32_t kSub = 0x10, kBad = 0x20, kLit = 0x40, kEob = 0x80;
    enum Kind { CODES, LENS, DISTS };
    unsigned pb = 0;
    std::vector<uint32_t> t;
    std::vector<uint8_t> sb;
    // zlib's verdict on the lengths alone (inftrees.c): over-subscribed -> no; incomplete -> only a single one-bit code of LENS / DISTS; no code at
    // all -> a DISTS table that every use of fails (a CODES / LENS set without codes cannot make a valid block either: refused here)
    static bool lengths_ok(const uint8_t* lens, unsigned n, Kind kind) {
        unsigned count[16] = {0};
        for (unsigned i = 0; i < n; ++i) count[lens[i]]++;
        unsigned max = 15;
        while (max >= 1 && count[max] == 0) --max;
        if (max == 0) return kind == DISTS;
        int left = 1;
        for (unsigned l = 1; l <= 15; ++l) {
            left <<= 1;
            left -= static_cast<int>(count[l]);
            if (left < 0) return false;
        }
        return !(left > 0 && (kind == CODES || max != 1));
    }
    static uint32_t entry(unsigned sym, unsigned len, Kind kind) {
        if (kind == CODES) return (sym << 16) | len;
        if (kind == LENS) {
            if (sym < 256) return (sym << 16) | kLit | len;
            if (sym == 256) return kEob | len;
            if (sym >= 286) return kBad | len;   // (codes that the fixed code has and no stream may use)
            return (static_cast<uint32_t>(kLenBase[sym - 257]) << 16) | (static_cast<uint32_t>(kLenExtra[sym - 257]) << 8) | len;
        }
        if (sym >= 30) return kBad | len;
        return (static_cast<uint32_t>(kDistBase[sym]) << 16) | (static_cast<uint32_t>(kDistExtra[sym]) << 8) | len;
    }
    bool build(const uint8_t* lens, unsigned n, Kind kind, unsigned primary_bits) {
        if (!lengths_ok(lens, n, kind)) return false;
        pb = primary_bits;
        const uint32_t psize = 1u << pb, pmask = psize - 1u;
        unsigned count[16] = {0};
        for (unsigned i = 0; i < n; ++i) count[lens[i]]++;
        count[0] = 0;
        unsigned next[16];
        unsigned code = 0;
        for (unsigned l = 1; l <= 15; ++l) {
            code = (code + count[l - 1]) << 1;
            next[l] = code;
        }
        uint16_t rc[320];
        sb.assign(psize, 0);
        bool any_sub = false;
        for (unsigned s = 0; s < n; ++s) {
            const unsigned l = lens[s];
            if (!l) continue;
            unsigned c = next[l]++, r = 0;
            for (unsigned i = 0; i < l; ++i) {
                r = (r << 1) | (c & 1u);
                c >>= 1;
            }
            rc[s] = static_cast<uint16_t>(r);
            if (l > pb) {
                sb[r & pmask] = std::max<uint8_t>(sb[r & pmask], static_cast<uint8_t>(l - pb));
                any_sub = true;
            }
        }
        size_t total = psize;
        if (any_sub)
            for (uint32_t p = 0; p < psize; ++p) total += sb[p] ? (size_t(1) << sb[p]) : 0;
        t.assign(total, kBad | 1u);
        if (any_sub) {
            size_t cur = psize;
            for (uint32_t p = 0; p < psize; ++p)
                if (sb[p]) {
                    t[p] = static_cast<uint32_t>(cur << 16) | kSub | sb[p];   // (at most 2^pb + 2^15 entries: the offset fits)
                    cur += size_t(1) << sb[p];
                }
        }
        for (unsigned s = 0; s < n; ++s) {
            const unsigned l = lens[s];
            if (!l) continue;
            const uint32_t r = rc[s], e = entry(s, l, kind);
            if (l <= pb) {
                for (uint32_t i = r; i < psize; i += 1u << l) t[i] = e;
            } else {
                const uint32_t sub = t[r & pmask], off = sub >> 16, sbits = sub & 15u;
                for (uint32_t i = r >> pb; i < (1u << sbits); i += 1u << (l - pb)) t[off + i] = e;
            }
        }
        return true;
    }
    inline uint32_t lookup(uint64_t bits) const {
        uint32_t e = t[bits & ((1u << pb) - 1u)];
        if (e & kSub) e = t[(e >> 16) + ((bits >> pb) & ((1u << (e & 15u)) - 1u))];
        return e;
    }
};

// the code lengths of a dynamic block whose first three bits are taken: lens[0, nlen) literal / length, lens[nlen, nlen + ndist) distance
bool read_dynamic_lengths(BitReader& br, Huff& cl, uint8_t* lens, unsigned* nlen, unsigned* ndist) {
    br.refill();
    *nlen = br.take(5) + 257;
    *ndist = br.take(5) + 1;
    const unsigned ncode = br.take(4) + 4;
    if (*nlen > 286 || *ndist > 30) return false;
    uint8_t cll[19] = {0};
    for (unsigned i = 0; i < ncode; ++i) {
        if (i % 10 == 0) br.refill();
        cll[kClOrder[i]] = static_cast<uint8_t>(br.take(3));
    }
    if (br.overrun || !cl.build(cll, 19, Huff::CODES, 7)) return false;
    const unsigned total = *nlen + *ndist;
    unsigned have = 0;
    while (have < total) {
        br.refill();
        const uint32_t e = cl.lookup(br.buf);
        if (e & Huff::kBad) return false;
        br.drop(e & 15u);
        const unsigned s = e >> 16;
        if (s < 16) {
            lens[have++] = static_cast<uint8_t>(s);
        } else {
            unsigned len = 0, rep;
            if (s == 16) {
                if (have == 0) return false;
                len = lens[have - 1];
                rep = 3 + br.take(2);
            } else if (s == 17) {
                rep = 3 + br.take(3);
            } else {
                rep = 11 + br.take(7);
            }
            if (have + rep > total) return false;
            while (rep--) lens[have++] = static_cast<uint8_t>(len);
        }
        if (br.overrun) return false;
    }
    return lens[256] != 0;
}

struct Tables {
    Huff cl, lit, dist;
    Huff fixed_lit, fixed_dist;
    bool fixed_built = false;
    void build_fixed() {
        if (fixed_built) return;
        uint8_t l[288];
        for (unsigned i = 0; i < 144; ++i) l[i] = 8;
        for (unsigned i = 144; i < 256; ++i) l[i] = 9;
        for (unsigned i = 256; i < 280; ++i) l[i] = 7;
        for (unsigned i = 280; i < 288; ++i) l[i] = 8;
        fixed_lit.build(l, 288, Huff::LENS, 10);
        uint8_t dl[32];
        for (unsigned i = 0; i < 32; ++i) dl[i] = 5;
        fixed_dist.build(dl, 32, Huff::DISTS, 5);
        fixed_built = true;
    }
};

// output with an unknown window: 16-bit symbols
struct MarkerOut {
    PgzBuf b;
    size_t q = 0;
    uint16_t* data() const { return reinterpret_cast<uint16_t*>(b.data()); }
    size_t size() const { return b.cap() / 2; }
    // Is there a marker among the last kWin symbols?  Looked for from the end, block after block: while the window is unknown one is found within
    // a few symbols, and the one time none is the 64 KiB are read once.  (The copies used to OR what they moved and branch on it: a branch on
    // the data in the loop's hottest path.)
    bool window_known() const {
        if (q < kWin) return false;
        const uint16_t* p = data();
        size_t i = q;
        while (i > q - kWin && (i & 7u)) {
            if (p[--i] & kMarker) return false;
        }
        while (i >= q - kWin + 8) {
            i -= 8;
            if (_mm_movemask_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(p + i))) & 0xAAAA) return false;
        }
        while (i > q - kWin) {
            if (p[--i] & kMarker) return false;
        }
        return true;
    }
    void room(size_t n) {
        if (q + n > size()) b.grow(std::max(b.cap() * 2, (q + n) * 2 + (size_t(1) << 17)), q * 2);
    }
};

enum BlockEnd { kBlockDone, kBlockFinal, kBlockError };

// one deflate block from its first bit on; `limit`: stop with kBlockDone as soon as the output has reached that many symbols (trial decodes)
BlockEnd decode_block_markers(BitReader& br, Tables& tb, MarkerOut& out, size_t limit = ~size_t(0)) {
    br.refill();
    const bool final = br.take(1) != 0;
    const unsigned type = br.take(2);
    if (br.overrun) return kBlockError;
    if (type == 0) {
        br.align_byte();
        br.refill();
        const uint32_t len = br.take(16), nlen = br.take(16);
        if (br.overrun || (len ^ 0xFFFFu) != nlen) return kBlockError;
        // (the bytes in buf are whole bytes now: hand them back and copy from the stream)
        size_t at = br.pos - br.cnt / 8;
        if (at + len > br.size) return kBlockError;
        out.room(len);
        for (uint32_t i = 0; i < len; ++i) out.data()[out.q++] = br.d[at + i];
        br.init(br.d, br.size, static_cast<uint64_t>(at + len) * 8u);
        return final ? kBlockFinal : kBlockDone;
    }
    const Huff* lit;
    const Huff* dist;
    if (type == 1) {
        tb.build_fixed();
        lit = &tb.fixed_lit;
        dist = &tb.fixed_dist;
    } else if (type == 2) {
        uint8_t lens[320];
        unsigned nlen, ndist;
        if (!read_dynamic_lengths(br, tb.cl, lens, &nlen, &ndist)) return kBlockError;
        if (!tb.lit.build(lens, nlen, Huff::LENS, 10) || !tb.dist.build(lens + nlen, ndist, Huff::DISTS, 8)) return kBlockError;
        lit = &tb.lit;
        dist = &tb.dist;
    } else {
        return kBlockError;
    }
    const unsigned lpb = lit->pb, dpb = dist->pb;
    const uint32_t* lt = lit->t.data();
    const uint32_t* dt = dist->t.data();
    const uint64_t lmask = (1u << lpb) - 1u, dmask = (1u << dpb) - 1u;
    for (;;) {
        // ---- the fast loop: while 16 bytes of input and 320 symbols of room are there for certain (no bit of the reader's can run out)
        out.room(4096);
        if (br.pos + 16 <= br.size) {
            const unsigned char* d = br.d;
            size_t pos = br.pos;
            const size_t pos_end = br.size - 16;
            uint64_t buf = br.buf;
            unsigned cnt = br.cnt;
            uint16_t* const o0 = out.data();
            uint16_t* o = o0 + out.q;
            uint16_t* const o_stop = o0 + std::min(out.size() - 320, limit);
            int state = 0;   // 1: end of block, 2: error
#define PGZ_REFILL()                                   \
    do {                                               \
        buf |= load64(d + pos) << cnt;                 \
        const unsigned add_ = (63u - cnt) >> 3;        \
        pos += add_;                                   \
        cnt += add_ * 8u;                              \
    } while (0)
#define PGZ_LOOKUP(e_, tab_, mask_, pb_)                                                              \
    do {                                                                                              \
        e_ = tab_[buf & mask_];                                                                       \
        if (e_ & Huff::kSub) e_ = tab_[(e_ >> 16) + ((buf >> pb_) & ((1u << (e_ & 15u)) - 1u))];      \
    } while (0)
#define PGZ_DROP(n_)          \
    do {                      \
        const unsigned n2_ = (n_); \
        buf >>= n2_;          \
        cnt -= n2_;           \
    } while (0)
            while (pos <= pos_end && o < o_stop) {
                PGZ_REFILL();
                uint32_t e;
                PGZ_LOOKUP(e, lt, lmask, lpb);
                if (e & Huff::kLit) {   // up to three literals on one refill (45 bits at most)
                    PGZ_DROP(e & 15u);
                    *o++ = static_cast<uint16_t>(e >> 16);
                    PGZ_LOOKUP(e, lt, lmask, lpb);
                    if (e & Huff::kLit) {
                        PGZ_DROP(e & 15u);
                        *o++ = static_cast<uint16_t>(e >> 16);
                        PGZ_LOOKUP(e, lt, lmask, lpb);
                        if (e & Huff::kLit) {
                            PGZ_DROP(e & 15u);
                            *o++ = static_cast<uint16_t>(e >> 16);
                            continue;
                        }
                    }
                    PGZ_REFILL();
                }
                if (e & (Huff::kBad | Huff::kEob)) {
                    if (e & Huff::kBad) {
                        state = 2;
                    } else {
                        PGZ_DROP(e & 15u);
                        state = 1;
                    }
                    break;
                }
                PGZ_DROP(e & 15u);
                const unsigned lx = (e >> 8) & 15u;
                const uint32_t len = (e >> 16) + static_cast<uint32_t>(buf & ((1u << lx) - 1u));
                PGZ_DROP(lx);
                uint32_t de;
                PGZ_LOOKUP(de, dt, dmask, dpb);
                if (de & Huff::kBad) {
                    state = 2;
                    break;
                }
                PGZ_DROP(de & 15u);
                const unsigned dx = (de >> 8) & 15u;
                const uint32_t dd = (de >> 16) + static_cast<uint32_t>(buf & ((1u << dx) - 1u));
                PGZ_DROP(dx);
                const size_t q = static_cast<size_t>(o - o0);
                uint16_t* const oe = o + len;
                if (dd <= q) {
                    const uint16_t* src = o - dd;
                    if (dd >= 8) {
                        // sixteen symbols whatever the length (eight at a time: a source at least eight back is written before it is read; room for
                        // what goes beyond the match is there), more only for the rare long match: no branch on the length in the common case
                        _mm_storeu_si128(reinterpret_cast<__m128i*>(o), _mm_loadu_si128(reinterpret_cast<const __m128i*>(src)));
                        _mm_storeu_si128(reinterpret_cast<__m128i*>(o + 8), _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + 8)));
                        if (len > 16) {
                            src += 16;
                            o += 16;
                            do {
                                _mm_storeu_si128(reinterpret_cast<__m128i*>(o), _mm_loadu_si128(reinterpret_cast<const __m128i*>(src)));
                                src += 8;
                                o += 8;
                            } while (o < oe);
                        }
                    } else {
                        do {
                            *o++ = *src++;
                        } while (o < oe);
                    }
                    o = oe;
                } else {   // into the window this chunk was not given
                    for (uint32_t i = 0; i < len; ++i) {
                        const size_t at = q + i;
                        o0[at] = dd > at ? static_cast<uint16_t>(kMarker | (kWin + at - dd)) : o0[at - dd];
                    }
                    o = oe;
                }
            }
#undef PGZ_REFILL
#undef PGZ_LOOKUP
#undef PGZ_DROP
            br.pos = pos;
            br.buf = buf;
            br.cnt = cnt;
            out.q = static_cast<size_t>(o - o0);
            if (state == 2) return kBlockError;
            if (state == 1) break;
            if (out.q >= limit) return kBlockDone;
            if (pos <= pos_end) continue;   // (room: the next turn makes it)
        }
        // ---- the careful loop: the stream's last bytes, symbol by symbol
        br.refill();
        uint32_t e = lit->lookup(br.buf);
        if (e & Huff::kBad) return kBlockError;
        br.drop(e & 15u);
        if (e & Huff::kLit) {
            out.room(1);
            out.data()[out.q++] = static_cast<uint16_t>(e >> 16);
            if (br.overrun) return kBlockError;
            if (out.q >= limit) return kBlockDone;
            continue;
        }
        if (e & Huff::kEob) break;
        const uint32_t len = (e >> 16) + br.take((e >> 8) & 15u);
        const uint32_t de = dist->lookup(br.buf);
        if (de & Huff::kBad) return kBlockError;
        br.drop(de & 15u);
        const uint32_t dd = (de >> 16) + br.take((de >> 8) & 15u);
        if (br.overrun) return kBlockError;
        out.room(len);
        uint16_t* o = out.data();
        const size_t q = out.q;
        for (uint32_t i = 0; i < len; ++i) {
            const size_t at = q + i;
            o[at] = dd > at ? static_cast<uint16_t>(kMarker | (kWin + at - dd)) : o[at - dd];   // dd <= 32768: the index is in [0, kWin)
        }
        out.q = q + len;
        if (out.q >= limit) return kBlockDone;
    }
    if (br.overrun) return kBlockError;
    return final ? kBlockFinal : kBlockDone;
}

// RFC 1952 member header at byte `at`; the deflate stream's first byte
bool gzip_header(const unsigned char* d, size_t size, size_t at, size_t* deflate_at) {
    if (at + 10 > size || d[at] != 0x1f || d[at + 1] != 0x8b || d[at + 2] != 8) return false;
    const unsigned flg = d[at + 3];
    if (flg & 0xE2u) return false;   // reserved bits; a header CRC (zlib checks it: left to zlib)
    size_t p = at + 10;
    if (flg & 4u) {
        if (p + 2 > size) return false;
        const size_t xlen = d[p] | (static_cast<size_t>(d[p + 1]) << 8);
        p += 2 + xlen;
    }
    for (unsigned f : {8u, 16u})
        if (flg & f) {
            while (p < size && d[p]) ++p;
            ++p;
        }
    if (p >= size) return false;
    *deflate_at = p;
    return true;
}

// CRC-32 (the gzip polynomial, reflected) by carry-less multiplication: 64 bytes folded per step (Gopal et al., "Fast CRC Computation for Generic
// Polynomials Using PCLMULQDQ Instruction", Intel 2009; the constants are x^k mod P for the fold distances, bit-reflected).  zlib 1.2.11's
// table-driven crc32() does 0.6 GB/s per thread -- next to an inflate of 0.3 GB/s it was a third of a chunk's time.  Held against zlib's
// crc32() on random buffers of every length and alignment the first time it is used (crc_self_test): a processor without the instruction, or a
// mismatch, leaves zlib's.
__attribute__((target("pclmul,sse4.1"))) uint32_t crc32_clmul(uint32_t crc, const unsigned char* buf, size_t len) {   // len >= 64, a multiple of 16; crc: the register (inverted)
    alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
    alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
    alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
    alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x00));
    x2 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x10));
    x3 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x20));
    x4 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128(static_cast<int>(crc)));
    x0 = _mm_load_si128(reinterpret_cast<const __m128i*>(k1k2));
    buf += 64;
    len -= 64;
    while (len >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
        x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
        x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x00));
        y6 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x10));
        y7 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x20));
        y8 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
        x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
        x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64;
        len -= 64;
    }
    x0 = _mm_load_si128(reinterpret_cast<const __m128i*>(k3k4));   // four lanes -> one
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {
        x2 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf));
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16;
        len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);   // 128 -> 64 bits
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64(reinterpret_cast<const __m128i*>(k5k0));
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128(reinterpret_cast<const __m128i*>(poly));   // Barrett reduction to 32 bits
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return static_cast<uint32_t>(_mm_extract_epi32(x1, 1));
}

uint32_t crc32_zlib(uint32_t crc, const unsigned char* p, size_t n) {
    while (n) {
        const size_t nb = std::min<size_t>(n, size_t(1) << 30);
        crc = static_cast<uint32_t>(crc32(crc, p, static_cast<uInt>(nb)));
        p += nb;
        n -= nb;
    }
    return crc;
}

uint32_t crc32_fast_unchecked(uint32_t crc, const unsigned char* p, size_t n) {
    if (n >= 64) {
        const size_t body = n & ~size_t(15);
        crc = ~crc32_clmul(~crc, p, body);
        p += body;
        n -= body;
    }
    return n ? crc32_zlib(crc, p, n) : crc;
}

bool crc_self_test() {
    if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1") || std::getenv("FMSI_HIP_NO_CLMUL_CRC")) return false;
    std::vector<unsigned char> b(4096 + 300);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (auto& c : b) {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        c = static_cast<unsigned char>(x >> 32);
    }
    for (size_t off = 0; off < 17; ++off)
        for (size_t n : {size_t(0), size_t(1), size_t(15), size_t(16), size_t(63), size_t(64), size_t(65), size_t(79), size_t(80), size_t(127), size_t(128), size_t(129), size_t(255), size_t(1000), size_t(4096), size_t(4111)}) {
            const uint32_t seed = static_cast<uint32_t>(0x12345678u * (off + 1) + n);
            if (crc32_fast_unchecked(seed, b.data() + off, n) != crc32_zlib(seed, b.data() + off, n)) return false;
        }
    return true;
}

// CRC-32 of p[0, n) continued from crc (zlib's convention: crc32(0, NULL, 0) = 0 starts one)
uint32_t crc32_of(uint32_t crc, const unsigned char* p, size_t n) {
    static const bool fast = crc_self_test();
    return fast ? crc32_fast_unchecked(crc, p, n) : crc32_zlib(crc, p, n);
}

inline uint32_t le32(const unsigned char* p) { return p[0] | (static_cast<uint32_t>(p[1]) << 8) | (static_cast<uint32_t>(p[2]) << 16) | (static_cast<uint32_t>(p[3]) << 24); }

struct MemberEnd {
    size_t out_off;      // the member's last byte is the chunk's output byte out_off - 1
    uint32_t crc, isize;
};

struct ByteBuf {
    PgzBuf b;
    size_t len = 0;
    unsigned char* data() const { return b.data(); }
    size_t cap() const { return b.cap(); }
    void reserve(size_t n) {
        if (n > b.cap()) b.grow(std::max(n, b.cap() + b.cap() / 2), len);
    }
};

struct StopRule {
    const std::vector<uint64_t>* later = nullptr;   // the starts the chunks behind this one found, ascending
    size_t idx = 0;
    uint64_t round_end_bit = 0;
    bool ends_at(uint64_t p) {
        while (idx < later->size() && (*later)[idx] < p) ++idx;
        if (idx < later->size() && (*later)[idx] == p) return true;
        return p >= round_end_bit;
    }
};

struct Chunk {
    uint64_t start_bit = 0, end_bit = 0;
    bool has_start = false, error = false, stream_end = false;
    MarkerOut mk;                 // the first part of its output, window unknown
    ByteBuf out;                  // [0, mk.q): the markers' bytes once resolved; from mk.q on: what zlib decoded
    std::vector<MemberEnd> member_ends;
    std::vector<std::pair<uint32_t, size_t>> crc_pieces;   // CRC-32 and length of the output between member ends
    std::vector<unsigned char> window_before;
    size_t total() const { return out.len; }
};

// behind a member's last block, the reader at the first byte of the trailer: the trailer, then the next member's header or the end of the file
// returns false on anything else.  *next_bit: the next member's first block (only when !*stream_end)
bool cross_member(const unsigned char* d, size_t size, size_t trailer_at, size_t out_off, Chunk& c, uint64_t* next_bit) {
    if (trailer_at + 8 > size) return false;
    c.member_ends.push_back({out_off, le32(d + trailer_at), le32(d + trailer_at + 4)});
    const size_t at = trailer_at + 8;
    if (at == size) {
        c.stream_end = true;
        return true;
    }
    size_t deflate_at;
    if (!gzip_header(d, size, at, &deflate_at)) return false;
    *next_bit = static_cast<uint64_t>(deflate_at) * 8u;
    return true;
}

// zlib from a block boundary with a known window, until the stop rule says so or the stream ends
void zlib_phase(const unsigned char* d, size_t size, uint64_t bit, const unsigned char* dict, size_t dict_len, StopRule& stop, Chunk& c, size_t expect) {
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) {
        c.error = true;
        return;
    }
    auto start_at = [&](uint64_t b, const unsigned char* dc, size_t dl) -> bool {
        size_t byte = static_cast<size_t>(b >> 3);
        const unsigned s = static_cast<unsigned>(b & 7);
        if (byte >= size) return false;
        if (s) {
            if (inflatePrime(&zs, static_cast<int>(8 - s), d[byte] >> s) != Z_OK) return false;
            ++byte;
        }
        if (dl && inflateSetDictionary(&zs, dc, static_cast<uInt>(dl)) != Z_OK) return false;
        zs.next_in = const_cast<Bytef*>(d + byte);
        return true;
    };
    if (!start_at(bit, dict, dict_len)) {
        c.error = true;
        inflateEnd(&zs);
        return;
    }
    c.out.reserve(c.out.len + expect);
    for (;;) {
        const size_t in_left = size - static_cast<size_t>(zs.next_in - d);
        zs.avail_in = static_cast<uInt>(std::min<size_t>(in_left, size_t(1) << 30));
        if (c.out.cap() - c.out.len < (size_t(1) << 16)) c.out.reserve(c.out.cap() + (c.out.cap() >> 1) + (size_t(1) << 20));
        const size_t room = std::min<size_t>(c.out.cap() - c.out.len, size_t(16) << 20);
        zs.next_out = c.out.data() + c.out.len;
        zs.avail_out = static_cast<uInt>(room);
        const int rc = inflate(&zs, Z_BLOCK);
        c.out.len += room - zs.avail_out;
        if (c.out.len > kMaxChunkText) {
            c.error = true;
            break;
        }
        if (rc == Z_STREAM_END) {
            uint64_t nb = 0;
            if (!cross_member(d, size, static_cast<size_t>(zs.next_in - d), c.out.len, c, &nb)) {
                c.error = true;
                break;
            }
            if (c.stream_end) {
                c.end_bit = static_cast<uint64_t>(size) * 8u;
                break;
            }
            if (stop.ends_at(nb)) {
                c.end_bit = nb;
                break;
            }
            if (inflateReset(&zs) != Z_OK || !start_at(nb, nullptr, 0)) {
                c.error = true;
                break;
            }
            continue;
        }
        if (rc != Z_OK && rc != Z_BUF_ERROR) {
            c.error = true;
            break;
        }
        if ((zs.data_type & 128) && !(zs.data_type & 64)) {   // a block boundary inside the stream (not: the last block's header just read)
            const uint64_t p = static_cast<uint64_t>(zs.next_in - d) * 8u - static_cast<unsigned>(zs.data_type & 63);
            if (stop.ends_at(p)) {
                c.end_bit = p;
                break;
            }
            continue;
        }
        if (zs.avail_out == 0) continue;            // more room, same place
        if (zs.data_type & 128) continue;           // (the last block's header: go on into it)
        if (in_left > (size_t(1) << 30)) continue;  // the next piece of a very long input
        c.error = true;                             // the input ended inside the stream
        break;
    }
    inflateEnd(&zs);
}

// a chunk that starts with its window unknown
void decode_unknown(const unsigned char* d, size_t size, StopRule& stop, Chunk& c, Tables& tb, size_t expect) {
    BitReader br;
    br.init(d, size, c.start_bit);
    MarkerOut& mk = c.mk;
    for (;;) {
        const BlockEnd be = decode_block_markers(br, tb, mk, kMaxChunkText + 1);
        if (be == kBlockError || mk.q > kMaxChunkText) {
            c.error = true;
            return;
        }
        uint64_t p = br.bit_pos();
        bool fresh_member = false;
        if (be == kBlockFinal) {
            br.align_byte();
            const size_t trailer_at = static_cast<size_t>(br.bit_pos() >> 3);
            if (!cross_member(d, size, trailer_at, mk.q, c, &p)) {
                c.error = true;
                return;
            }
            if (c.stream_end) {
                c.end_bit = static_cast<uint64_t>(size) * 8u;
                c.out.reserve(mk.q);
                c.out.len = mk.q;
                return;
            }
            fresh_member = true;
        }
        if (stop.ends_at(p)) {
            c.end_bit = p;
            c.out.reserve(mk.q);
            c.out.len = mk.q;
            return;
        }
        if (fresh_member || mk.window_known()) {   // the window is known from here on: a new member has none, 32 KiB without a marker are one
            std::vector<unsigned char> dict;
            if (!fresh_member) {
                dict.resize(kWin);
                for (uint32_t i = 0; i < kWin; ++i) dict[i] = static_cast<unsigned char>(mk.data()[mk.q - kWin + i]);
            }
            c.out.reserve(mk.q + expect);
            c.out.len = mk.q;
            zlib_phase(d, size, p, dict.data(), dict.size(), stop, c, expect);
            return;
        }
    }
}

// is there a dynamic block's header at this bit, and does the block behind it decode?
struct Finder {
    Tables tb;
    MarkerOut trial;
    bool ok13[1 << 13];
    Finder() {
        for (unsigned v = 0; v < (1u << 13); ++v) ok13[v] = (v & 7u) == 4u && ((v >> 3) & 31u) <= 29u && ((v >> 8) & 31u) <= 29u;   // not final, dynamic, counts in range
    }
    bool candidate(const unsigned char* d, size_t size, uint64_t bit, uint64_t v) {
        // the code-length code must be complete
        const unsigned ncode = static_cast<unsigned>((v >> 13) & 15u) + 4u;
        const uint64_t w = peek64(d, size, bit + 17);
        unsigned cnt[8] = {0};
        for (unsigned i = 0; i < ncode; ++i) cnt[(w >> (3 * i)) & 7u]++;
        int left = 1;
        for (unsigned l = 1; l <= 7; ++l) {
            left = (left << 1) - static_cast<int>(cnt[l]);
            if (left < 0) return false;
        }
        if (left != 0) return false;
        BitReader br;
        br.init(d, size, bit + 3);
        uint8_t lens[320];
        unsigned nlen, ndist;
        if (!read_dynamic_lengths(br, tb.cl, lens, &nlen, &ndist)) return false;
        if (!Huff::lengths_ok(lens, nlen, Huff::LENS) || !Huff::lengths_ok(lens + nlen, ndist, Huff::DISTS)) return false;
        // the block itself (up to 2^17 symbols of it)
        br.init(d, size, bit);
        trial.q = 0;
        return decode_block_markers(br, tb, trial, size_t(1) << 17) != kBlockError;
    }
    // the first block start in [from_bit, to_bit), or ~0
    uint64_t find(const unsigned char* d, size_t size, uint64_t from_bit, uint64_t to_bit) {
        for (uint64_t byte = from_bit >> 3; byte * 8 < to_bit && byte < size; ++byte) {
            const uint64_t lo = peek64(d, size, byte * 8);
            const uint64_t hi = byte + 8 < size ? d[byte + 8] : 0;
            for (unsigned s = 0; s < 8; ++s) {
                const uint64_t bit = byte * 8 + s;
                if (bit < from_bit) continue;
                if (bit >= to_bit) break;
                const uint64_t v = s ? (lo >> s) | (hi << (64 - s)) : lo;
                if (!ok13[v & 0x1FFFu]) continue;
                if (candidate(d, size, bit, v)) return bit;
            }
        }
        return ~uint64_t(0);
    }
};

}  // namespace

PgzBuf::~PgzBuf() { std::free(p_); }

PgzBuf& PgzBuf::operator=(PgzBuf&& o) noexcept {
    if (this != &o) {
        std::free(p_);
        p_ = o.p_;
        cap_ = o.cap_;
        o.p_ = nullptr;
        o.cap_ = 0;
    }
    return *this;
}

void PgzBuf::grow(size_t bytes, size_t keep) {
    if (bytes <= cap_) return;
    constexpr size_t kHuge = size_t(2) << 20;
    void* np = nullptr;
    if (bytes >= 2 * kHuge) {
        bytes = (bytes + kHuge - 1) / kHuge * kHuge;
        if (posix_memalign(&np, kHuge, bytes) != 0) np = nullptr;
        else (void)madvise(np, bytes, MADV_HUGEPAGE);
    } else {
        np = std::malloc(bytes);
    }
    if (!np) throw std::bad_alloc();
    if (keep) std::memcpy(np, p_, std::min(keep, cap_));
    std::free(p_);
    p_ = np;
    cap_ = bytes;
}

ParallelGunzip::~ParallelGunzip() {
    if (mapped_ && data_) munmap(const_cast<unsigned char*>(data_), size_);
}

bool ParallelGunzip::open_memory(const unsigned char* data, size_t size) {
    size_t deflate_at;
    if (!gzip_header(data, size, 0, &deflate_at)) return false;
    data_ = data;
    size_ = size;
    next_bit_ = static_cast<uint64_t>(deflate_at) * 8u;
    crc_ = static_cast<uint32_t>(crc32(0L, Z_NULL, 0));
    member_len_ = 0;
    window_.clear();
    return true;
}

bool ParallelGunzip::open(const std::string& path, size_t min_bytes) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    bool ok = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && static_cast<size_t>(st.st_size) >= std::max<size_t>(min_bytes, 20);
    if (ok) {
        void* p = mmap(nullptr, static_cast<size_t>(st.st_size), PROT_READ, MAP_PRIVATE, fd, 0);
        ok = p != MAP_FAILED;
        if (ok) {
            (void)madvise(p, static_cast<size_t>(st.st_size), MADV_SEQUENTIAL);
            if (open_memory(static_cast<const unsigned char*>(p), static_cast<size_t>(st.st_size))) {
                mapped_ = true;
            } else {
                munmap(p, static_cast<size_t>(st.st_size));
                ok = false;
            }
        }
    }
    ::close(fd);
    return ok;
}

bool ParallelGunzip::round(unsigned team, const Runner& run) {
    const size_t b0 = static_cast<size_t>(next_bit_ >> 3);
    if (b0 >= size_) return false;
    const size_t left = size_ - b0;
    // chunks of 1 MiB of compressed bytes in the first round, one per thread (the stages behind wait for the first bytes), then of 2 MiB, FOUR per
    // thread, taken by the threads as they fall free: text like DNA never gives a chunk a known window, so a small chunk costs no more than a
    // large one (3 ms of looking for its start), and chunks that differ in length or decoder -- the round's first is zlib's -- no longer leave
    // threads waiting for the longest (one chunk per thread: 2.6 GB/s with 16 threads, 4.0 with 32 on the same cores)
    const size_t cb = chunk_bytes_ ? chunk_bytes_ : (stats.rounds ? size_t(2) << 20 : size_t(1) << 20);
    const unsigned per_thread = stats.rounds ? 4u : 1u;
    const unsigned n = static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>(static_cast<size_t>(std::max(1u, team)) * per_thread, (left + cb - 1) / cb)));
    const unsigned workers = std::min(n, std::max(1u, team));
    // fn(i) for every i in [0, count), the workers taking the next index as they finish one
    auto each = [&](unsigned count, const std::function<void(unsigned)>& fn) {
        std::atomic<unsigned> next{0};
        run(std::min(workers, std::max(1u, count)), [&](unsigned) {
            for (unsigned i; (i = next.fetch_add(1, std::memory_order_relaxed)) < count;) fn(i);
        });
    };
    const size_t round_end_byte = std::min(size_, b0 + static_cast<size_t>(n) * cb);
    const uint64_t round_end_bit = round_end_byte == size_ ? ~uint64_t(0) : static_cast<uint64_t>(round_end_byte) * 8u;   // (the last round ends with the stream)
    static const bool dbg = std::getenv("FMSI_HIP_PGZ_DEBUG") != nullptr;
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    auto since = [&](clk::time_point a) { return std::chrono::duration<double>(clk::now() - a).count(); };
    std::vector<double> tfind(n, 0), tdec(n, 0);
    std::vector<Chunk> ch(n);
    ch[0].start_bit = next_bit_;
    ch[0].has_start = true;
    for (unsigned i = 0; i < n; ++i) {
        if (i && !sym_pool_.empty()) {   // (chunk 0 knows its window: no symbols)
            ch[i].mk.b = std::move(sym_pool_.back());
            sym_pool_.pop_back();
        }
        if (!byte_pool_.empty()) {
            ch[i].out.b = std::move(byte_pool_.back());
            byte_pool_.pop_back();
        }
    }
    auto give_back = [&] {   // (whatever ends the round: the buffers that were not handed out stay with the reader)
        for (Chunk& c : ch) {
            if (c.mk.b.cap()) sym_pool_.push_back(std::move(c.mk.b));
            if (c.out.b.cap()) byte_pool_.push_back(std::move(c.out.b));
        }
    };
    struct GiveBack {
        std::function<void()> f;
        ~GiveBack() { f(); }
    } give_back_at_exit{give_back};
    // ---- where the chunks begin
    each(n, [&](unsigned i) {
        if (i == 0) return;
        const auto ta = clk::now();
        Finder f;
        const uint64_t from = static_cast<uint64_t>(b0 + static_cast<size_t>(i) * cb) * 8u;
        const uint64_t to = std::min(static_cast<uint64_t>(b0 + static_cast<size_t>(i + 1) * cb), static_cast<uint64_t>(round_end_byte)) * 8u;
        const uint64_t at = f.find(data_, size_, std::max(from, next_bit_ + 1), to);
        if (at != ~uint64_t(0)) {
            ch[i].start_bit = at;
            ch[i].has_start = true;
        }
        tfind[i] = since(ta);
    });
    const double t_find = since(t0);
    std::vector<std::vector<uint64_t>> later(n);
    for (unsigned i = 0; i < n; ++i)
        for (unsigned j = i + 1; j < n; ++j)
            if (ch[j].has_start) later[i].push_back(ch[j].start_bit);
    // ---- the chunks, side by side
    each(n, [&](unsigned i) {
        Chunk& c = ch[i];
        if (!c.has_start) return;
        StopRule stop{&later[i], 0, round_end_bit};
        const auto ta = clk::now();
        const size_t expect = cb * 5 + (size_t(1) << 20);
        if (i == 0) {
            zlib_phase(data_, size_, c.start_bit, window_.data(), window_.size(), stop, c, expect);
        } else {
            Tables tb;
            decode_unknown(data_, size_, stop, c, tb, expect);
        }
        tdec[i] = since(ta);
    });
    const double t_dec = since(t0);
    // ---- the chain of equal positions from the round's first bit, and every chunk's window
    std::vector<unsigned> chain;
    {
        unsigned cur = 0;
        for (;;) {
            if (ch[cur].error) return false;
            chain.push_back(cur);
            if (ch[cur].stream_end) break;
            unsigned nx = n;
            for (unsigned j = cur + 1; j < n; ++j)
                if (ch[j].has_start && ch[j].start_bit == ch[cur].end_bit) {
                    nx = j;
                    break;
                }
            if (nx == n) break;
            cur = nx;
        }
    }
    std::vector<unsigned char> win = window_;
    auto resolve = [](const Chunk& c, size_t from, size_t to, unsigned char* dst) -> bool {   // the markers [from, to) of c -> dst[0, to - from)
        const uint16_t* s = c.mk.data();
        const size_t wl = c.window_before.size();
        const unsigned char* w = c.window_before.data();
        uint32_t bad = 0;
        size_t i = from;
        // eight symbols at a time where none of them is a marker (SSE2: the saturating pack is exact for values below 256)
        while (i + 8 <= to) {
            const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + i));
            if (_mm_movemask_epi8(v) & 0xAAAA) {   // a high byte with its top bit set: a marker among the eight
                for (size_t e = i + 8; i < e; ++i) {
                    const uint16_t x = s[i];
                    if (x & kMarker) {
                        const uint32_t idx = x & 0x7FFFu;
                        if (idx + wl < kWin) {
                            bad = 1;
                            dst[i - from] = 0;
                        } else {
                            dst[i - from] = w[idx + wl - kWin];
                        }
                    } else {
                        dst[i - from] = static_cast<unsigned char>(x);
                    }
                }
                continue;
            }
            _mm_storel_epi64(reinterpret_cast<__m128i*>(dst + (i - from)), _mm_packus_epi16(v, v));
            i += 8;
        }
        for (; i < to; ++i) {
            const uint16_t v = s[i];
            if (v & kMarker) {
                const uint32_t idx = v & 0x7FFFu;
                if (idx + wl < kWin) {
                    bad = 1;
                    dst[i - from] = 0;
                } else {
                    dst[i - from] = w[idx + wl - kWin];
                }
            } else {
                dst[i - from] = static_cast<unsigned char>(v);
            }
        }
        return bad == 0;
    };
    for (unsigned ci : chain) {
        Chunk& c = ch[ci];
        c.window_before = win;
        const size_t m = c.mk.q, total = c.total();
        size_t member_from = 0;   // the window does not reach back over a member's end
        bool crossed = false;
        if (!c.member_ends.empty()) {
            member_from = c.member_ends.back().out_off;
            crossed = true;
        }
        const size_t from = std::max(member_from, total > kWin ? total - kWin : 0);
        std::vector<unsigned char> nw;
        if (!crossed && total < kWin) {   // the old window's tail stays in front
            const size_t keep = std::min(win.size(), static_cast<size_t>(kWin) - total);
            nw.assign(win.end() - static_cast<std::ptrdiff_t>(keep), win.end());
        }
        const size_t at = nw.size();
        nw.resize(at + (total - from));
        if (from < m) {
            if (!resolve(c, from, m, nw.data() + at)) return false;
            std::memcpy(nw.data() + at + (m - from), c.out.data() + m, total - m);
        } else if (total > from) {
            std::memcpy(nw.data() + at, c.out.data() + from, total - from);
        }
        win.swap(nw);
    }
    // ---- the markers' bytes and the CRCs of the pieces, side by side
    std::vector<char> ok(chain.size(), 1);
    each(static_cast<unsigned>(chain.size()), [&](unsigned k) {
        Chunk& c = ch[chain[k]];
        if (c.mk.q && !resolve(c, 0, c.mk.q, c.out.data())) ok[k] = 0;
        size_t from = 0;
        auto piece = [&](size_t to) {
            c.crc_pieces.emplace_back(crc32_of(0u, c.out.data() + from, to - from), to - from);
            from = to;
        };
        for (const MemberEnd& me : c.member_ends) piece(me.out_off);
        piece(c.total());
    });
    if (std::find(ok.begin(), ok.end(), 0) != ok.end()) return false;
    if (dbg) {
        std::fprintf(stderr, "pgz round %llu: %u chunks of %zu B: find %.3f s, decode %.3f s, resolve+crc %.3f s |", (unsigned long long)stats.rounds, n, cb, t_find, t_dec - t_find, since(t0) - t_dec);
        for (unsigned i = 0; i < n; ++i) std::fprintf(stderr, " [%.3f %.3f mk %zu tot %zu]", tfind[i], tdec[i], ch[i].mk.q, ch[i].total());
        std::fprintf(stderr, "\n");
    }
    uint32_t crc = crc_;
    uint64_t mlen = member_len_;
    uint64_t members = 0;
    for (unsigned ci : chain) {
        const Chunk& c = ch[ci];
        for (size_t pi = 0; pi < c.crc_pieces.size(); ++pi) {
            crc = static_cast<uint32_t>(crc32_combine(crc, c.crc_pieces[pi].first, static_cast<z_off_t>(c.crc_pieces[pi].second)));
            mlen += c.crc_pieces[pi].second;
            if (pi < c.member_ends.size()) {
                if (crc != c.member_ends[pi].crc || static_cast<uint32_t>(mlen) != c.member_ends[pi].isize) return false;
                crc = static_cast<uint32_t>(crc32(0L, Z_NULL, 0));
                mlen = 0;
                ++members;
            }
        }
    }
    // ---- the round stands
    const Chunk& last = ch[chain.back()];
    crc_ = crc;
    member_len_ = mlen;
    window_.swap(win);
    next_bit_ = last.end_bit;
    done_ = last.stream_end;
    stats.rounds++;
    stats.members += members;
    stats.chunks += n;
    for (unsigned i = 0; i < n; ++i) {
        if (!ch[i].has_start) stats.chunks_without_start++;
        else if (std::find(chain.begin(), chain.end(), i) == chain.end()) stats.chunks_passed_over++;
    }
    for (unsigned ci : chain) {
        Chunk& c = ch[ci];
        stats.marker_symbols += c.mk.q;
        stats.plain_bytes += c.total() - c.mk.q;
        if (!c.total()) continue;
        Piece p;   // (the bytes stay where zlib wrote them)
        p.len = c.total();
        p.buf = std::move(c.out.b);
        ready_.push_back(std::move(p));
    }
    return true;
}

size_t ParallelGunzip::read(char* dst, size_t n, unsigned team, const Runner& run) {
    struct Seg {
        const unsigned char* src;
        char* dst;
        size_t len;
    };
    std::vector<Seg> segs;
    std::vector<Piece> spent;   // (handed out in full: alive until the copies are done)
    size_t got = 0;
    while (got < n && !failed_) {
        if (ready_.empty()) {
            if (done_) break;
            if (!round(team, run)) {
                failed_ = true;
                break;
            }
            continue;
        }
        Piece& p = ready_.front();
        const size_t take = std::min(n - got, p.len - p.from);
        segs.push_back({p.buf.data() + p.from, dst + got, take});
        p.from += take;
        got += take;
        if (p.from == p.len) {
            spent.push_back(std::move(p));
            ready_.pop_front();
        }
    }
    const unsigned copiers = got >= (size_t(8) << 20) ? std::max(1u, team) : 1u;
    run(copiers, [&](unsigned t) {   // share t: the bytes [got t / copiers, got (t + 1) / copiers) of what is handed out
        const size_t lo = got / copiers * t, hi = t + 1 == copiers ? got : got / copiers * (t + 1);
        size_t at = 0;
        for (const Seg& s : segs) {
            const size_t a = std::max(lo, at), b = std::min(hi, at + s.len);
            if (a < b) std::memcpy(s.dst + (a - at), s.src + (a - at), b - a);
            at += s.len;
            if (at >= hi) break;
        }
    });
    for (Piece& p : spent) byte_pool_.push_back(std::move(p.buf));
    if (done_ && ready_.empty()) {   // nothing more to decode: the buffers go
        byte_pool_.clear();
        sym_pool_.clear();
    }
    delivered_ += got;
    return got;
}

}  // namespace fmsi

// ---- the C-ABI's test / measurement entry (include/fmsi_hip.h)
#include <thread>

extern "C" int fmsi_host_gunzip(const char* path, int threads, uint64_t chunk_bytes, unsigned char* out, uint64_t cap, uint64_t* out_len,
                                uint64_t* stats7) {
    fmsi::ParallelGunzip g;
    if (!path || !out_len || !g.open(path, 0)) return -1;
    if (chunk_bytes) g.set_chunk_bytes(static_cast<size_t>(chunk_bytes));
    const unsigned team = static_cast<unsigned>(std::max(1, threads));
    const fmsi::ParallelGunzip::Runner run = [](unsigned n, const std::function<void(unsigned)>& fn) {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < n; ++t) th.emplace_back([&fn, t] { fn(t); });
        fn(0);
        for (auto& x : th) x.join();
    };
    uint64_t got = 0;
    std::vector<char> sink;
    for (;;) {   // (out == NULL or the room used up: the bytes are counted and dropped)
        size_t r;
        if (out && got < cap) {
            r = g.read(reinterpret_cast<char*>(out) + got, static_cast<size_t>(cap - got), team, run);
        } else {
            sink.resize(size_t(64) << 20);
            r = g.read(sink.data(), sink.size(), team, run);
        }
        got += r;
        if (r == 0) break;
    }
    *out_len = got;
    if (stats7) {
        stats7[0] = g.stats.rounds;
        stats7[1] = g.stats.chunks;
        stats7[2] = g.stats.chunks_passed_over;
        stats7[3] = g.stats.chunks_without_start;
        stats7[4] = g.stats.marker_symbols;
        stats7[5] = g.stats.plain_bytes;
        stats7[6] = g.stats.members;
    }
    return g.failed() ? 1 : 0;
}
