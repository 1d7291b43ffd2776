// device_fm_wide.h -- device helpers for handles with 64-bit rows (fmsi_wide.hip's search, fmsi_export.hip's walk): the same 64-byte blocks,
// their u32 cumulative counts wrapped, absolute u64 counts per superblock (index_priv.h: DevIndexW).
#pragma once
#include <hip/hip_runtime.h>

#include "device_fm.h"
#include "index_priv.h"

namespace fmsi {

__device__ __forceinline__ uint64_t abs_count(const DevIndexW& ix, uint64_t block, uint32_t slot, uint32_t wrapped) {
    const uint64_t s = ix.super[(block >> ix.sb_log2) * 8u + slot];
    return s + static_cast<uint32_t>(wrapped - static_cast<uint32_t>(s));
}

// one LF step on a non-empty [i, j): update_range, src/fms_index.h:98-103 (the upper boundary through the block of row j - 1)
__device__ __forceinline__ void lf_step_w(const DevIndexW& ix, uint32_t c, uint64_t& i, uint64_t& j, uint32_t& blocks_read) {
    const uint64_t bi = i >> kBlockRowsLog2, bj = (j - 1u) >> kBlockRowsLog2;
    const uint4* pi = ix.blocks + (bi << 2);
    const uint4* pj = ix.blocks + (bj << 2);
    const uint4 ci = pi[0], hi = pi[1], li = pi[2];
    const uint4 cj = pj[0], hj = pj[1], lj = pj[2];
    blocks_read += 1u + (bj != bi);
    const uint4 z = make_uint4(0, 0, 0, 0);
    const uint32_t in_i = rank_in_block(z, hi, li, c, static_cast<uint32_t>(i & (kBlockRows - 1u)));
    const uint32_t in_j = rank_in_block(z, hj, lj, c, static_cast<uint32_t>(j - (bj << kBlockRowsLog2)));
    uint64_t ri = abs_count(ix, bi, c, sel4(c, opaque(ci.x), opaque(ci.y), opaque(ci.z), opaque(ci.w))) + in_i;
    uint64_t rj = abs_count(ix, bj, c, sel4(c, opaque(cj.x), opaque(cj.y), opaque(cj.z), opaque(cj.w))) + in_j;
    if (c == 0u) {   // the '$' slot is stored as an 'A' and subtracted here (src/fms_index.h:81)
        ri -= (i > ix.dollar);
        rj -= (j > ix.dollar);
    }
    const uint64_t base = c == 0u ? ix.C0 : (c == 1u ? ix.C1 : (c == 2u ? ix.C2 : ix.C3));
    i = base + ri;
    j = base + rj;
}

__device__ __forceinline__ uint64_t mask_rank_w(const DevIndexW& ix, uint64_t b, uint32_t p) {
    const uint4 m = ix.blocks[(b << 2) + 3];
    return abs_count(ix, b, 4u, ix.mrank[b]) + popc_prefix128(m, p);
}


// ---- one set of names for the kernels that are templates over the handle's view (32-bit rows: DevIndex, 64-bit rows: DevIndexW) ----
template <class IX> struct RowOf;
template <> struct RowOf<DevIndex> { typedef uint32_t type; };
template <> struct RowOf<DevIndexW> { typedef uint64_t type; };

__device__ __forceinline__ void view_lf_step(const DevIndex& ix, uint32_t c, uint32_t& i, uint32_t& j, uint32_t& blocks) { lf_step(ix, c, i, j, blocks); }
__device__ __forceinline__ void view_lf_step(const DevIndexW& ix, uint32_t c, uint64_t& i, uint64_t& j, uint32_t& blocks) { lf_step_w(ix, c, i, j, blocks); }
// the mask bit of a row; the ON bits among the rows below `row` (row may be n)
__device__ __forceinline__ uint32_t view_mask_bit(const DevIndex& ix, uint32_t row) { return mask_bit(ix, row); }
__device__ __forceinline__ uint32_t view_mask_bit(const DevIndexW& ix, uint64_t row) {
    const uint32_t* m = reinterpret_cast<const uint32_t*>(ix.blocks) + ((row >> kBlockRowsLog2) << 4) + 12u;
    return (m[(static_cast<uint32_t>(row) & (kBlockRows - 1u)) >> 5] >> (row & 31u)) & 1u;
}
// ON bits among the rows [i, j), i < j (two mask ranks: the upper one through the block of row j - 1)
__device__ __forceinline__ uint64_t view_ones_in(const DevIndex& ix, uint32_t i, uint32_t j, uint32_t& blocks, uint64_t* rank_i = nullptr) {
    const uint32_t bi = i >> kBlockRowsLog2, bj = (j - 1u) >> kBlockRowsLog2;
    blocks += 1u + (bj != bi);
    const uint32_t ri = mask_rank(ix, bi, i & (kBlockRows - 1u));
    if (rank_i) *rank_i = ri;
    return mask_rank(ix, bj, j - (bj << kBlockRowsLog2)) - ri;
}
__device__ __forceinline__ uint64_t view_ones_in(const DevIndexW& ix, uint64_t i, uint64_t j, uint32_t& blocks, uint64_t* rank_i = nullptr) {
    const uint64_t bi = i >> kBlockRowsLog2, bj = (j - 1u) >> kBlockRowsLog2;
    blocks += 1u + (bj != bi);
    const uint64_t ri = mask_rank_w(ix, bi, static_cast<uint32_t>(i & (kBlockRows - 1u)));
    if (rank_i) *rank_i = ri;
    return mask_rank_w(ix, bj, static_cast<uint32_t>(j - (bj << kBlockRowsLog2))) - ri;
}
// the interval of a pattern's last T bases from the handle's prefix table (tidx: those bases, the first of them in the low bits)
__device__ __forceinline__ bool view_has_table(const DevIndex& ix, int k) { return ix.ptable && ix.ptable_len > 0 && k >= static_cast<int>(ix.ptable_len); }
__device__ __forceinline__ bool view_has_table(const DevIndexW& ix, int k) { return ix.ptable && ix.ptable_len > 0 && k >= static_cast<int>(ix.ptable_len); }
__device__ __forceinline__ void view_table_entry(const DevIndex& ix, uint64_t tidx, uint32_t& i, uint32_t& j) {
    const uint2 e = ix.ptable[static_cast<uint32_t>(tidx)];
    decode_table_entry(e.x, e.y, ix.ptable_packed, false, 0u, i, j);   // (no signature shortcut: the interval itself)
}
__device__ __forceinline__ void view_table_entry(const DevIndexW& ix, uint64_t tidx, uint64_t& i, uint64_t& j) {
    const ulonglong2 e = ix.ptable[tidx];
    i = e.x;
    j = e.y;
}

}  // namespace fmsi
