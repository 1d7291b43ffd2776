// Device-side FM primitives over the interleaved block layout (common.h).  Included by the query
// kernels and by the prefix-table builder.  Citations are relative to the reference root.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "index_priv.h"

namespace fmsi {

// Keeps a value opaque to the optimiser.  Without it hipcc folds a chain of selects over the four
// components of a uint4 into a dynamically indexed vector, which it then spills to scratch
// (cdna_hip_programming.md rule 20).
__device__ __forceinline__ uint32_t opaque(uint32_t x) {
    asm volatile("" : "+v"(x));
    return x;
}

__device__ __forceinline__ uint32_t sel4(uint32_t c, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3) {
    const uint32_t lo = (c & 1u) ? a1 : a0;
    const uint32_t hi = (c & 1u) ? a3 : a2;
    return (c & 2u) ? hi : lo;
}

// ones in bit positions [0, s) of a 32-bit word, s may be <= 0 or >= 32
__device__ __forceinline__ uint32_t prefix32(int s) {
    return s >= 32 ? 0xFFFFFFFFu : (s <= 0 ? 0u : ((1u << s) - 1u));
}

// occurrences of symbol c among the first p rows (p in [0,128]) of a block, plus the block's base count:
// rank() of src/fms_index.h:68-86 (two dependent rank_support_v5 probes there) on one 64-byte block.
__device__ __forceinline__ uint32_t rank_in_block(const uint4 cnt, const uint4 h, const uint4 l, uint32_t c, uint32_t p) {
    const uint32_t xh = (c & 2u) ? 0u : 0xFFFFFFFFu;
    const uint32_t xl = (c & 1u) ? 0u : 0xFFFFFFFFu;
    uint32_t r = sel4(c, opaque(cnt.x), opaque(cnt.y), opaque(cnt.z), opaque(cnt.w));
    const int s = static_cast<int>(p);
    r += __popc((h.x ^ xh) & (l.x ^ xl) & prefix32(s));
    r += __popc((h.y ^ xh) & (l.y ^ xl) & prefix32(s - 32));
    r += __popc((h.z ^ xh) & (l.z ^ xl) & prefix32(s - 64));
    r += __popc((h.w ^ xh) & (l.w ^ xl) & prefix32(s - 96));
    return r;
}

__device__ __forceinline__ uint32_t popc_prefix128(const uint4 m, uint32_t p) {
    const int s = static_cast<int>(p);
    return __popc(m.x & prefix32(s)) + __popc(m.y & prefix32(s - 32)) + __popc(m.z & prefix32(s - 64)) +
           __popc(m.w & prefix32(s - 96));
}

// One LF step on a non-empty [i, j): update_range, src/fms_index.h:98-103.
// The upper boundary is ranked through the block of row j-1 (offset 1..128), so a single-row interval
// touches ONE block; both fetches are issued unconditionally (same cache line when bj == bi).
__device__ __forceinline__ void lf_step(const DevIndex& ix, uint32_t c, uint32_t& i, uint32_t& j, uint32_t& blocks_read) {
    const uint32_t bi = i >> kBlockRowsLog2;
    const uint32_t bj = (j - 1u) >> kBlockRowsLog2;
    const uint4* pi = ix.blocks + (static_cast<size_t>(bi) << 2);
    const uint4* pj = ix.blocks + (static_cast<size_t>(bj) << 2);
    const uint4 ci = pi[0], hi = pi[1], li = pi[2];
    const uint4 cj = pj[0], hj = pj[1], lj = pj[2];
    blocks_read += 1u + (bj != bi);
    uint32_t ri = rank_in_block(ci, hi, li, c, i & (kBlockRows - 1u));
    uint32_t rj = rank_in_block(cj, hj, lj, c, j - (bj << kBlockRowsLog2));
    if (c == 0u) {   // the '$' slot is stored as an 'A' and subtracted here (src/fms_index.h:81)
        ri -= (i > ix.dollar);
        rj -= (j > ix.dollar);
    }
    const uint32_t base = sel4(c, ix.C0, ix.C1, ix.C2, ix.C3);
    i = base + ri;
    j = base + rj;
}

// rank1 on the SA-transformed mask through block b with in-block offset p (0..128)
__device__ __forceinline__ uint32_t mask_rank(const DevIndex& ix, uint32_t b, uint32_t p) {
    const uint4 m = ix.blocks[(static_cast<size_t>(b) << 2) + 3];
    return ix.mrank[b] + popc_prefix128(m, p);
}

__device__ __forceinline__ uint32_t mask_bit(const DevIndex& ix, uint32_t row) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(ix.blocks) + (static_cast<size_t>(row >> kBlockRowsLog2) << 4) + 12u;
    return (w[(row & (kBlockRows - 1u)) >> 5] >> (row & 31u)) & 1u;
}

// Prefix-table entry -> interval [i, j).  Packed format (ptable_packed != 0): x = i; y = tag << 30 | ...
//   tag 0: y = j - i (< 2^30).  An interval of 1..5 rows may stand in this form, too -- without signatures, the interval itself: that is
//          how the entries are while the signatures are still being written beside the first queries (build_prefix_table, fmsi_hip.hip;
//          an entry is one aligned 8-byte store, a reader sees the one form or the other)
//   tag 1: 1..3 rows; y = 1 << 30 | rows << 24 | sig2 << 16 | sig1 << 8 | sig0, where sig_m = the four text symbols
//          preceding the suffix of row i + m (c1 << 6 | c2 << 4 | c3 << 2 | c4, c1 = BWT[row] = the next base the
//          backward search consumes).  If the query still has >= 4 bases to go (have4) and its next four (q4, same
//          packing) equal none of the signatures, the interval would be empty within four more update_range() calls --
//          get_range_with_pattern() would return an empty range either way (src/fms_index.h:117-124), so it is returned
//          empty right here and the block reads of those steps never happen.  A row whose preceding text is shorter
//          than the signature carries an arbitrary one (a false match only costs the ordinary steps).
//   tags 2, 3 (round 5): 4 resp. 5 rows with signatures of THREE symbols, 6 bits each (c1 << 4 | c2 << 2 | c3), row i + m in bits
//          [6 m, 6 m + 5] -- the same test on the query's next three bases (q4 >> 2).  At the depth the `fmsi` binary takes at human
//          scale (t = 15: 2.6 rows per key) a quarter of the keys have four rows or more; with these two tags 97 % of the keys instead
//          of 74 % can say "this string does not occur" (the streaming kernel's absence certificates, fmsi_stream.hip).
__device__ __forceinline__ void decode_table_entry(uint32_t x, uint32_t y, uint32_t packed, bool have4, uint32_t q4, uint32_t& i, uint32_t& j) {
    i = x;
    if (!packed) {
        j = y;
        return;
    }
    const uint32_t tag = y >> 30;
    if (tag == 0u) {
        j = x + y;
        return;
    }
    if (tag == 1u) {
        const uint32_t rows = (y >> 24) & 3u;
        const bool hit = (y & 0xFFu) == q4 || (rows > 1u && ((y >> 8) & 0xFFu) == q4) || (rows > 2u && ((y >> 16) & 0xFFu) == q4);
        j = (have4 && !hit) ? x : x + rows;
        return;
    }
    const uint32_t rows = 2u + tag, q3 = q4 >> 2;
    const bool hit = (y & 63u) == q3 || ((y >> 6) & 63u) == q3 || ((y >> 12) & 63u) == q3 || ((y >> 18) & 63u) == q3 || (tag == 3u && ((y >> 24) & 63u) == q3);
    j = (have4 && !hit) ? x : x + rows;
}

// reverse complement of a 2-bit packed k-mer (base t in bits [2t,2t+1]); ReverseComplementString, src/kmers.h:53-59
__device__ __forceinline__ uint64_t revcomp_2bit(uint64_t f, int k) {
    uint64_t x = ~f;                                                               // complement: 3 - c
    x = __brevll(x);                                                               // reverse all bits ...
    x = ((x & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((x & 0x5555555555555555ull) << 1);   // ... restoring the order inside each pair
    x >>= (64 - 2 * k);
    return k == 32 ? x : (x & ((1ull << (2 * k)) - 1ull));
}

__device__ __forceinline__ uint64_t window_kmer(const uint64_t* __restrict__ seq, uint64_t start, int k) {
    const uint64_t w = start >> 5;
    const uint32_t sh = static_cast<uint32_t>(start & 31u) * 2u;
    uint64_t f = seq[w] >> sh;
    if (sh) f |= seq[w + 1] << (64u - sh);
    return k == 32 ? f : (f & ((1ull << (2 * k)) - 1ull));
}

// lanes below this one in a 64-bit ballot mask
__device__ __forceinline__ uint32_t lanes_below(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}


// FMSI_HIP_OUT_BITS: a chunk's result bytes (LDS, `len` of them, the chunk starting at window `start`, a multiple of 128) leave
// bit-packed -- window w is bit w & 63 of u64 word w >> 6 (its byte's bit 0), STRICT: nibble w & 15 of word w >> 4 (the byte's low
// four bits: (fwd + 1) | (rc + 1) << 2).  One 8-byte store per 64 (16) windows instead of 64 (16) bytes.
__device__ __forceinline__ void flush_packed_results(const uint8_t* res, uint32_t len, bool strict, uint8_t* out, uint64_t start, uint32_t lane) {
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(out);
    if (!strict) {
        const uint32_t words = (len + 63u) >> 6;
        for (uint32_t w = lane; w < words; w += 64u) {
            const uint64_t* src = reinterpret_cast<const uint64_t*>(res + 64u * w);
            unsigned long long v = 0;
#pragma unroll
            for (uint32_t b = 0; b < 8u; ++b) v |= (((src[b] & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56) << (8u * b);
            const uint32_t valid = len - 64u * w;
            if (valid < 64u) v &= (1ull << valid) - 1ull;
            __builtin_nontemporal_store(v, dst + (start >> 6) + w);
        }
    } else {
        const uint32_t words = (len + 15u) >> 4;
        for (uint32_t w = lane; w < words; w += 64u) {
            const uint4 x = *reinterpret_cast<const uint4*>(res + 16u * w);
            auto nib = [](uint32_t d) -> unsigned long long {
                return (d & 0xFu) | ((d >> 4) & 0xF0u) | ((d >> 8) & 0xF00u) | ((d >> 12) & 0xF000u);
            };
            unsigned long long v = nib(x.x) | (nib(x.y) << 16) | (nib(x.z) << 32) | (nib(x.w) << 48);
            const uint32_t valid = len - 16u * w;
            if (valid < 16u) v &= (1ull << (4u * valid)) - 1ull;
            __builtin_nontemporal_store(v, dst + (start >> 4) + w);
        }
    }
}

}  // namespace fmsi
