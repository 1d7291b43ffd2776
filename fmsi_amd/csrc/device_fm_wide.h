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

}  // namespace fmsi
