// Shared declarations of libfmsi_hip.so's translation units.
#pragma once

#include <cstdint>
#include <string>

namespace fmsi {

// thread-local last error behind fmsi_hip_last_error()
void set_last_error(const std::string& msg);
const char* last_error();

// ---- device layout (see DESIGN.md "Data layout in HBM") -------------------------------------------
// One interleaved block = 64 bytes = 16 dwords covering 128 consecutive BWT rows:
//   dw 0..3   cumulative symbol counts before the block: #A(incl. the '$' slot) #C #G #T   (u32)
//   dw 4..7   hi plane: bit r = ac_gt[row]            (row = 128*b + r)
//   dw 8..11  lo plane: bit r = ac[...] or gt[...]    (the second wavelet-tree level, scattered back to row order)
//   dw 12..15 mask plane: bit r = SA-transformed mask[row]
// so one rank(c, row) -- two dependent rank_support_v5 probes in the reference
// (src/fms_index.h:68-86) -- is ONE aligned 64-byte read plus popcounts, and the final mask access
// (rrr_vector<63>::operator[]) is a bit of the same structure.
constexpr uint32_t kBlockRowsLog2 = 7;
constexpr uint32_t kBlockRows = 1u << kBlockRowsLog2;
constexpr uint32_t kBlockDwords = 16;
constexpr uint32_t kBlockBytes = 64;

}  // namespace fmsi
