// The pair layout (fmsi_pair.hip builds it: 128-byte blocks of 128 rows for two-base LF steps) as the kernels of more than one
// translation unit see it.  Block format: the head of fmsi_pair.hip.
#pragma once

#include <cstdint>

#include "common.h"
#include "index_priv.h"

namespace fmsi {

constexpr uint32_t kPairBlockDwords = 32;
constexpr uint32_t kSuperLog2 = 23;                       // rows per superblock
constexpr uint32_t kSuperBlocksLog2 = kSuperLog2 - kBlockRowsLog2;

struct PairIndex {
    const uint32_t* pb;       // n_blocks x 32 dwords
    const uint32_t* sbt;      // n_super x 16 absolute pair counts
    const uint32_t* c2;       // 16 x C2[p]
    const uint32_t* mrank;    // n_blocks + 1 cumulative mask popcounts (shared with the single-step layout)
    const uint4* sblocks;     // the single-step layout (its mask plane serves the explicit probes)
    const uint2* ptable;
    const uint4* vtable;      // text verification (fmsi_hip_index_enable_verify) or nullptr: the lookup kernel's own copy of the
                              // table, 16 bytes per entry: (i, tagged word + mask bits 26-28, candidate 0, candidate 1)
    const uint32_t* stable;   // ... candidate 2 per entry (intervals of three rows)
    const uint64_t* text2;    // ... and the 2-bit packed superstring
    int verify_k;             // the k the candidate tables were made for (the index's): a query with another k does not use them
    uint32_t vwide;           // 1: vtable holds 32-byte entries (double-stranded entries of up to seven occurrences; lookup_lane_kernel<.., WIDE>)
    uint32_t verify_bases;    // 1: the words ARE the k - t bases in front of each row (k - t <= 15; bit 31 = no such k-mer);
                              // 0: the words are text positions and the bases are read from text2
    uint32_t ptable_len, ptable_packed;
    uint32_t n, r0, r1, p1;   // rows; row of suffix 0 ('$' slot); row of suffix 1; pair index stored for r1
    uint32_t C0, C1, C2_, C3;
};

// the handle's pair layout as a kernel argument (fmsi_pair.hip)
PairIndex make_pair_index(const fmsi_hip_index* idx);

}  // namespace fmsi
