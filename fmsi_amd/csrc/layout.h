#pragma once

#include <cstdint>
#include <vector>

#include "../../include/fmsi_hip.h"

namespace fmsi {

struct HostLayout {
    std::vector<uint32_t> blocks;   // n_blocks * 16 dwords, see common.h
    std::vector<uint32_t> mrank;    // n_blocks + 1 cumulative mask popcounts
    uint64_t n_blocks = 0;
};

// Returns 0 or a negative FMSI_HIP_E* code (message via set_last_error).
int build_layout(const fmsi_index_view& view, HostLayout& out);

}  // namespace fmsi
