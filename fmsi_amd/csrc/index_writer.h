// Writer of the reference's on-disk index format; see index_writer.cpp.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/fmsi_hip.h"

namespace fmsi {

// Writes <prefix>.fmsi.{ac_gt,ac,gt,mask,misc} (+ .klcp when the view has one); the mask is RRR-63 encoded.
// Returns 0 or a negative FMSI_HIP_E* code with the message in `err`.
int write_index_files(const std::string& prefix, const fmsi_index_view& view, std::string& err);

// Reference-format bit-vectors from the single-step device layout (blocks of 16 dwords per 128 rows, common.h):
// ac_gt = hi plane, mask = mask plane, ac / gt = the lo bits of the rows with hi = 0 / 1 in row order
// (the split of reference src/fms_index.h:427-450).
struct HostVectors {
    std::vector<uint64_t> ac_gt, ac, gt, mask;
    uint64_t n = 0, ac_bits = 0, gt_bits = 0;
};
void vectors_from_blocks(const uint32_t* blocks, uint64_t n, HostVectors& out);

}  // namespace fmsi
