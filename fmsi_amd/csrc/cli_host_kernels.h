// The word-at-a-time loops of the CLI's host side (query_cli.cpp): the 2-bit packer and the lower-case bits (PEXT or plain), the scan for
// invalid characters (SSE2), the expansion of result bits into '0' / '1' (PDEP or plain), and the byte sums of a -S chunk's strand
// results (SSE2).  Each is held against its one-character-at-a-time definition by tests/native/cli_host_kernels.cpp.
#pragma once

#include <emmintrin.h>   // SSE2: part of x86-64

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace fmsi {
namespace {

struct CodeTable {   // nucleotideToInt, src/kmers.h:3-20 (4 = invalid)
    uint8_t t[256];
    CodeTable() {
        std::memset(t, 4, sizeof t);
        t['A'] = t['a'] = 0;
        t['C'] = t['c'] = 1;
        t['G'] = t['g'] = 2;
        t['T'] = t['t'] = 3;
    }
};
const CodeTable kCode;

// 2-bit packing of `len` VALID bases into zeroed words: base i in bits [2(i%32), +1] of dst[i/32].
// For valid characters the code is ((c >> 1) & 3) ^ ((c >> 2) & 1): A/a 0, C/c 1, G/g 2, T/t 3.
inline uint64_t codes_of8(const char* s) {   // 8 characters -> their codes in the low 2 bits of 8 bytes
    uint64_t v;
    std::memcpy(&v, s, 8);
    return ((v >> 1) & 0x0303030303030303ull) ^ ((v >> 2) & 0x0101010101010101ull);
}
inline void pack_tail(const char* s, uint64_t from, uint64_t len, uint64_t* dst) {
    if (from >= len) return;
    uint64_t w = 0;
    for (uint64_t j = from; j < len; ++j) w |= static_cast<uint64_t>(kCode.t[static_cast<unsigned char>(s[j])] & 3u) << (2 * (j - from));
    dst[from >> 5] = w;
}
__attribute__((target("bmi2"))) void pack_run_bmi2(const char* s, uint64_t len, uint64_t* dst) {
    uint64_t i = 0;
    for (; i + 32 <= len; i += 32) {
        uint64_t w = 0;
        for (int q = 0; q < 4; ++q) w |= __builtin_ia32_pext_di(codes_of8(s + i + 8 * q), 0x0303030303030303ull) << (16 * q);
        dst[i >> 5] = w;
    }
    pack_tail(s, i, len, dst);
}
void pack_run_plain(const char* s, uint64_t len, uint64_t* dst) {   // the same without PEXT: three shift-or rounds per 8 bases
    uint64_t i = 0;
    for (; i + 32 <= len; i += 32) {
        uint64_t w = 0;
        for (int q = 0; q < 4; ++q) {
            uint64_t x = codes_of8(s + i + 8 * q);
            x = (x | (x >> 6)) & 0x000F000F000F000Full;
            x = (x | (x >> 12)) & 0x000000FF000000FFull;
            x = (x | (x >> 24)) & 0xFFFFull;
            w |= x << (16 * q);
        }
        dst[i >> 5] = w;
    }
    pack_tail(s, i, len, dst);
}
// PEXT / PDEP or the plain forms.  Read ONCE, by the command's main thread before it starts any other (host_kernels_init): getenv() beside a
// setenv() in another thread -- the HIP runtime sets variables while it starts up -- is a race glibc does not forgive.
inline bool use_bmi2() {
    static const bool bmi2 = __builtin_cpu_supports("bmi2") && !std::getenv("FMSI_HIP_NO_BMI2");
    return bmi2;
}
inline void host_kernels_init() { (void)use_bmi2(); }
inline void pack_run(const char* s, uint64_t len, uint64_t* dst) {
    if (use_bmi2()) pack_run_bmi2(s, len, dst);
    else pack_run_plain(s, len, dst);
}

// Lower-case bits of `len` valid bases: bit i % 32 of dst32[i / 32] = character i is lower case (bit 5 of an ASCII letter) -- what
// fmsi_hip_query_runs_fn wants next to the packed bases (a self-complementary k-mer in asymmetric case counts twice, include/fmsi_hip.h)
inline void case_run(const char* s, uint64_t len, uint32_t* dst32) {
    uint64_t i = 0;
    for (; i + 8 <= len; i += 8) {
        uint64_t v;
        std::memcpy(&v, s + i, 8);
        const uint32_t bits = static_cast<uint32_t>((((v >> 5) & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);   // byte j's bit -> bit j
        if ((i & 31) == 0) dst32[i >> 5] = bits;
        else dst32[i >> 5] |= bits << (i & 31);
    }
    if (i < len && (i & 31) == 0) dst32[i >> 5] = 0;
    for (; i < len; ++i) dst32[i >> 5] |= static_cast<uint32_t>((static_cast<unsigned char>(s[i]) >> 5) & 1u) << (i & 31);
}

// next_invalid_character_or_end (src/parser.h:57-64): length of the longest prefix of s[0, n) made of ACGTacgt, 16 characters a step
inline int64_t valid_prefix(const char* s, int64_t n) {
    int64_t i = 0;
    const __m128i fold = _mm_set1_epi8(static_cast<char>(0xDF));   // a-z -> A-Z (and nothing else onto A, C, G, T)
    const __m128i a = _mm_set1_epi8('A'), c = _mm_set1_epi8('C'), g = _mm_set1_epi8('G'), t = _mm_set1_epi8('T');
    for (; i + 16 <= n; i += 16) {
        const __m128i v = _mm_and_si128(_mm_loadu_si128(reinterpret_cast<const __m128i*>(s + i)), fold);
        const __m128i ok = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(v, a), _mm_cmpeq_epi8(v, c)), _mm_or_si128(_mm_cmpeq_epi8(v, g), _mm_cmpeq_epi8(v, t)));
        const unsigned bad = ~static_cast<unsigned>(_mm_movemask_epi8(ok)) & 0xFFFFu;
        if (bad) return i + __builtin_ctz(bad);
    }
    while (i < n && kCode.t[static_cast<unsigned char>(s[i])] != 4) ++i;
    return i;
}

// One k-mer, k <= 32 characters at s -> its 2-bit word (base t in bits [2t, 2t + 1]: what fmsi_hip_query_kmers takes); false when one of the
// k characters is not ACGTacgt.  Reads 32 bytes at s whatever k is: the caller keeps them inside its buffer.
inline bool kmer_all_valid32(const char* s, int k) {
    const __m128i fold = _mm_set1_epi8(static_cast<char>(0xDF));
    const __m128i a = _mm_set1_epi8('A'), c = _mm_set1_epi8('C'), g = _mm_set1_epi8('G'), t = _mm_set1_epi8('T');
    const __m128i v0 = _mm_and_si128(_mm_loadu_si128(reinterpret_cast<const __m128i*>(s)), fold);
    const __m128i v1 = _mm_and_si128(_mm_loadu_si128(reinterpret_cast<const __m128i*>(s + 16)), fold);
    const __m128i ok0 = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(v0, a), _mm_cmpeq_epi8(v0, c)), _mm_or_si128(_mm_cmpeq_epi8(v0, g), _mm_cmpeq_epi8(v0, t)));
    const __m128i ok1 = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(v1, a), _mm_cmpeq_epi8(v1, c)), _mm_or_si128(_mm_cmpeq_epi8(v1, g), _mm_cmpeq_epi8(v1, t)));
    const uint32_t ok = static_cast<uint32_t>(_mm_movemask_epi8(ok0)) | (static_cast<uint32_t>(_mm_movemask_epi8(ok1)) << 16);
    const uint32_t want = k >= 32 ? 0xFFFFFFFFu : (1u << k) - 1u;
    return (ok & want) == want;
}
__attribute__((target("bmi2"))) inline bool kmer_word_bmi2(const char* s, int k, uint64_t* out) {
    if (!kmer_all_valid32(s, k)) return false;
    uint64_t w = 0;
    for (int q = 0; q < 4; ++q) w |= __builtin_ia32_pext_di(codes_of8(s + 8 * q), 0x0303030303030303ull) << (16 * q);
    *out = k >= 32 ? w : w & ((1ull << (2 * k)) - 1ull);
    return true;
}
inline bool kmer_word_plain(const char* s, int k, uint64_t* out) {
    if (!kmer_all_valid32(s, k)) return false;
    uint64_t w = 0;
    for (int q = 0; q < 4; ++q) {
        uint64_t x = codes_of8(s + 8 * q);
        x = (x | (x >> 6)) & 0x000F000F000F000Full;
        x = (x | (x >> 12)) & 0x000000FF000000FFull;
        x = (x | (x >> 24)) & 0xFFFFull;
        w |= x << (16 * q);
    }
    *out = k >= 32 ? w : w & ((1ull << (2 * k)) - 1ull);
    return true;
}
// the same one character at a time (the last bytes of a buffer, where 32 bytes cannot be read)
inline bool kmer_word_tail(const char* s, int k, uint64_t* out) {
    uint64_t w = 0;
    for (int j = 0; j < k; ++j) {
        const uint8_t code = kCode.t[static_cast<unsigned char>(s[j])];
        if (code == 4) return false;
        w |= static_cast<uint64_t>(code) << (2 * j);
    }
    *out = w;
    return true;
}

inline int fwd_of(uint8_t b) { return static_cast<int>(b & 3) - 1; }
inline int rc_of(uint8_t b) { return static_cast<int>((b >> 2) & 3) - 1; }

// first index >= from whose byte differs from `v` (or n)
inline uint64_t run_end(const uint8_t* p, uint64_t from, uint64_t n, uint8_t v) {
    const uint64_t pat = 0x0101010101010101ull * v;
    uint64_t e = from;
    while (e + 8 <= n) {
        uint64_t x;
        std::memcpy(&x, p + e, 8);
        x ^= pat;
        if (x) return e + static_cast<uint64_t>(__builtin_ctzll(x) >> 3);
        e += 8;
    }
    while (e < n && p[e] == v) ++e;
    return e;
}

// the same over the bytes' LOW NIBBLES (a replayed byte keeps its strand nibble and carries the printed bit on top: replay_presence_tight)
inline uint64_t run_end_nibble(const uint8_t* p, uint64_t from, uint64_t n, uint8_t nib) {
    uint64_t e = from;
    const __m128i want = _mm_set1_epi8(static_cast<char>(nib)), low = _mm_set1_epi8(0x0F);
    for (; e + 16 <= n; e += 16) {
        const unsigned same = static_cast<unsigned>(_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_and_si128(_mm_loadu_si128(reinterpret_cast<const __m128i*>(p + e)), low), want)));
        if (same != 0xFFFFu) return e + static_cast<uint64_t>(__builtin_ctz(~same & 0xFFFFu));
    }
    while (e < n && (p[e] & 15u) == nib) ++e;
    return e;
}

// One -S chunk of strict result bytes, 16 windows a step.  With F = f + 1 = b & 3 and R = r + 1 = (b >> 2) & 3 (0, 1, 2 for -1, 0, 1):
// the first strand searched settles a window when it says 1, or 0 under -O (replay_streaming_chunk above); the window prints 1
// when the settling strand says 1, or -- unsettled -- when the other one does.  sums[swap] = {sum of the first strand's results,
// sum of the second strand's over the unsettled windows}; returns whether some window prints differently for the two strand orders.
struct ChunkMasks {
    __m128i dec0, dec1, one0, one1, F, R4;
};
inline ChunkMasks chunk_masks(__m128i v, bool maximized_ones) {
    ChunkMasks m;
    m.F = _mm_and_si128(v, _mm_set1_epi8(3));
    m.R4 = _mm_and_si128(v, _mm_set1_epi8(12));   // 4 (r + 1)
    const __m128i f_one = _mm_cmpeq_epi8(m.F, _mm_set1_epi8(2)), r_one = _mm_cmpeq_epi8(m.R4, _mm_set1_epi8(8));
    m.dec0 = f_one;
    m.dec1 = r_one;
    if (maximized_ones) {
        m.dec0 = _mm_or_si128(m.dec0, _mm_cmpeq_epi8(m.F, _mm_set1_epi8(1)));
        m.dec1 = _mm_or_si128(m.dec1, _mm_cmpeq_epi8(m.R4, _mm_set1_epi8(4)));
    }
    m.one0 = _mm_or_si128(f_one, _mm_andnot_si128(m.dec0, r_one));
    m.one1 = _mm_or_si128(r_one, _mm_andnot_si128(m.dec1, f_one));
    return m;
}
inline bool chunk_sums(const uint8_t* b, uint32_t n, bool maximized_ones, int sums[2][2]) {
    int fs[2] = {0, 0}, ss[2] = {0, 0};
    bool matters = false;
    uint32_t i = 0;
    if (n >= 16) {
        const __m128i zero = _mm_setzero_si128(), one = _mm_set1_epi8(1);
        __m128i sF = zero, sR4 = zero, sR4u = zero, sFu = zero, nu0 = zero, nu1 = zero, diff = zero;   // byte sums (two 64-bit lanes each)
        for (; i + 16 <= n; i += 16) {
            const ChunkMasks m = chunk_masks(_mm_loadu_si128(reinterpret_cast<const __m128i*>(b + i)), maximized_ones);
            sF = _mm_add_epi64(sF, _mm_sad_epu8(m.F, zero));
            sR4 = _mm_add_epi64(sR4, _mm_sad_epu8(m.R4, zero));
            sR4u = _mm_add_epi64(sR4u, _mm_sad_epu8(_mm_andnot_si128(m.dec0, m.R4), zero));   // second strand = r where f does not settle
            sFu = _mm_add_epi64(sFu, _mm_sad_epu8(_mm_andnot_si128(m.dec1, m.F), zero));      // second strand = f where r does not settle
            nu0 = _mm_add_epi64(nu0, _mm_sad_epu8(_mm_andnot_si128(m.dec0, one), zero));
            nu1 = _mm_add_epi64(nu1, _mm_sad_epu8(_mm_andnot_si128(m.dec1, one), zero));
            diff = _mm_or_si128(diff, _mm_xor_si128(m.one0, m.one1));
        }
        auto total = [](__m128i x) { return static_cast<int>(_mm_cvtsi128_si32(x) + _mm_cvtsi128_si32(_mm_srli_si128(x, 8))); };
        const int done = static_cast<int>(i);
        fs[0] = total(sF) - done;              // sum of f = sum of F - windows
        fs[1] = total(sR4) / 4 - done;
        ss[0] = total(sR4u) / 4 - total(nu0);
        ss[1] = total(sFu) - total(nu1);
        matters = _mm_movemask_epi8(diff) != 0;
    }
    for (; i < n; ++i) {
        const int f = fwd_of(b[i]), r = rc_of(b[i]);
        const bool dec0 = f == 1 || (f == 0 && maximized_ones), dec1 = r == 1 || (r == 0 && maximized_ones);
        fs[0] += f;
        ss[0] += dec0 ? 0 : r;
        fs[1] += r;
        ss[1] += dec1 ? 0 : f;
        const int v0 = dec0 ? f : std::max(f, r), v1 = dec1 ? r : std::max(r, f);
        matters |= (v0 == 1) != (v1 == 1);
    }
    sums[0][0] = fs[0];
    sums[0][1] = ss[0];
    sums[1][0] = fs[1];
    sums[1][1] = ss[1];
    return matters;
}
// ... and its output bytes when the strand order does not matter: 1 where the window prints '1'
inline void chunk_output(uint8_t* b, uint32_t n, bool maximized_ones) {
    uint32_t i = 0;
    for (; i + 16 <= n; i += 16) {
        const ChunkMasks m = chunk_masks(_mm_loadu_si128(reinterpret_cast<const __m128i*>(b + i)), maximized_ones);
        _mm_storeu_si128(reinterpret_cast<__m128i*>(b + i), _mm_and_si128(m.one0, _mm_set1_epi8(1)));
    }
    for (; i < n; ++i) {
        const int f = fwd_of(b[i]), r = rc_of(b[i]);
        b[i] = static_cast<uint8_t>((f == 1 || (f == 0 && maximized_ones) ? f : std::max(f, r)) == 1);
    }
}

// `count` result bits from bit `first` of `words` (FMSI_HIP_OUT_BITS layout) -> '0' / '1' characters, 8 a step
inline uint64_t bits_from(const uint64_t* words, uint64_t at) {   // 64 bits starting at bit `at` (the buffer has a spare word)
    const unsigned sh = static_cast<unsigned>(at & 63);
    const uint64_t lo = words[at >> 6] >> sh;
    return sh ? lo | (words[(at >> 6) + 1] << (64 - sh)) : lo;
}
__attribute__((target("bmi2"))) void expand_bits_bmi2(const uint64_t* words, uint64_t first, uint64_t count, char* dst) {
    uint64_t i = 0;
    for (; i + 64 <= count; i += 64) {
        const uint64_t x = bits_from(words, first + i);
        for (int q = 0; q < 8; ++q) {
            const uint64_t c = __builtin_ia32_pdep_di(x >> (8 * q), 0x0101010101010101ull) | 0x3030303030303030ull;
            std::memcpy(dst + i + 8 * q, &c, 8);
        }
    }
    for (; i < count; ++i) dst[i] = static_cast<char>('0' + ((words[(first + i) >> 6] >> ((first + i) & 63)) & 1));
}
void expand_bits_plain(const uint64_t* words, uint64_t first, uint64_t count, char* dst) {
    uint64_t i = 0;
    for (; i + 64 <= count; i += 64) {
        const uint64_t x = bits_from(words, first + i);
        for (int q = 0; q < 8; ++q) {
            // byte j <- bit j of the q-th byte: spread by a multiply, keep one bit per byte, normalise it to 0 / 1
            uint64_t c = (((x >> (8 * q)) & 0xFF) * 0x0101010101010101ull) & 0x8040201008040201ull;
            c = ((c + 0x7F7F7F7F7F7F7F7Full) >> 7) & 0x0101010101010101ull;
            c |= 0x3030303030303030ull;
            std::memcpy(dst + i + 8 * q, &c, 8);
        }
    }
    for (; i < count; ++i) dst[i] = static_cast<char>('0' + ((words[(first + i) >> 6] >> ((first + i) & 63)) & 1));
}
inline void expand_bits(const uint64_t* words, uint64_t first, uint64_t count, char* dst) {
    if (use_bmi2()) expand_bits_bmi2(words, first, count, dst);
    else expand_bits_plain(words, first, count, dst);
}

}  // namespace
}  // namespace fmsi
