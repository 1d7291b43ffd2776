// Host-side parser of the reference's on-disk index (<prefix>.fmsi.{ac_gt,ac,gt,mask,klcp,misc}).
// Counterpart of load_index() (reference src/fms_index.h:502-526); product code, independent of oracle/.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/fmsi_hip.h"

namespace fmsi {

// sdsl::int_vector<1> payload (reference src/sdsl-lite/include/sdsl/int_vector.hpp:1441-1447, 1563-1595):
// bit i = (words[i >> 6] >> (i & 63)) & 1.  One zero word of padding is always appended.
struct BitVector {
    uint64_t bits = 0;
    std::vector<uint64_t> words;
    bool get(uint64_t i) const { return (words[i >> 6] >> (i & 63)) & 1; }
};

struct HostIndex {
    BitVector ac_gt, ac, gt, mask, klcp;   // mask = SA-transformed mask decoded from the rrr_vector<63>
    uint64_t counts[4] = {0, 0, 0, 0};
    uint64_t dollar_position = 0;
    int k = 0;
    uint64_t n() const { return mask.bits; }
};

// Returns 0 or a negative FMSI_HIP_E* code; on failure `err` holds the message.
int load_host_index(const std::string& prefix, bool use_klcp, HostIndex& out, std::string& err);

// rows, k and whether a .klcp file exists, from the files' headers alone (the CLI's checks before the device load)
int peek_host_index(const std::string& prefix, uint64_t& n, int& k, bool& klcp_file);

// rrr_vector<63> file image -> plain bits (reference src/sdsl-lite/include/sdsl/rrr_vector.hpp:145-238, 350-373;
// rrr_helper.hpp:304-320).  Exposed for tests.
int decode_rrr63(const uint8_t* data, size_t size, BitVector& out, std::string& err);

// The RRR-63 decoder's tables for the on-device decoder (build_index.hip): c[n * 64 + k] = C(n, k) for n, k < 64 and the offset
// widths per class.
void rrr_tables(uint64_t* c /* 64 x 64 */, uint8_t* space /* 64 */);

}  // namespace fmsi

struct fmsi_host_index {
    fmsi::HostIndex ix;
};
