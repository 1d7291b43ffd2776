// Host-side code under AddressSanitizer + UBSan (CPU only; SURVEY.md 5: "host side: ASan/UBSan job for the loader").
// Built and run by tests/test_host.py::test_host_code_under_sanitizers from fmsi_amd/csrc/{host_index,index_writer,layout,
// fastx}.cpp -- no HIP.  argv: <index prefix> <use_klcp 0|1> <out prefix> <fasta file>
//   1. load the reference-format index files (parser + RRR-63 decoder), write them back (RRR-63 encoder): the caller
//      compares the bytes;  2. read the FASTA/FASTQ file with the ordinary reader and, cut at every '>' line, with one
//      memory reader per part: same records.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/fmsi_hip.h"
#include "../../fmsi_amd/csrc/fastx.h"
#include "../../fmsi_amd/csrc/index_writer.h"
#include "../../fmsi_amd/csrc/layout.h"

namespace fmsi { const char* last_error(); }
// (fmsi_hip_last_error lives in the HIP translation unit of the library; the host-only build supplies it)
extern "C" const char* fmsi_hip_last_error(void) { return fmsi::last_error(); }

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    fmsi_host_index* h = nullptr;
    if (fmsi_host_index_load(argv[1], atoi(argv[2]), &h) != 0) {
        std::cerr << "load failed: " << fmsi_hip_last_error() << "\n";
        return 1;
    }
    fmsi_index_view v;
    fmsi_host_index_view(h, &v);
    if (fmsi_host_index_write(argv[3], &v) != 0) {
        std::cerr << "write failed: " << fmsi_hip_last_error() << "\n";
        return 1;
    }
    // the device layout of the same vectors (layout.cpp: what the host path uploads) and back again (vectors_from_blocks: what the
    // writer makes of a downloaded handle, on threads): the same bits
    {
        fmsi::HostLayout L;
        if (fmsi::build_layout(v, L) != 0) {
            std::cerr << "layout failed: " << fmsi_hip_last_error() << "\n";
            return 1;
        }
        fmsi::HostVectors hv;
        fmsi::vectors_from_blocks(L.blocks.data(), v.ac_gt_bits, hv);
        auto same = [](const uint64_t* a, const std::vector<uint64_t>& b, uint64_t bits) {
            for (uint64_t w = 0; w < (bits + 63) / 64; ++w) {
                const uint64_t m = (w == bits / 64 && (bits & 63)) ? ((1ull << (bits & 63)) - 1) : ~0ull;
                if ((a[w] & m) != (b[w] & m)) return false;
            }
            return true;
        };
        if (hv.n != v.ac_gt_bits || hv.ac_bits != v.ac_bits || hv.gt_bits != v.gt_bits || !same(v.ac_gt, hv.ac_gt, v.ac_gt_bits) || !same(v.ac, hv.ac, v.ac_bits) ||
            !same(v.gt, hv.gt, v.gt_bits) || !same(v.mask, hv.mask, v.mask_bits)) {
            std::cerr << "layout -> vectors round trip differs\n";
            return 1;
        }
    }
    fmsi_host_index_free(h);
    // a missing index is an error, not a crash
    if (fmsi_host_index_load("/nonexistent/prefix", 0, &h) == 0) return 1;

    std::vector<std::pair<std::string, std::string>> a, b;
    {
        fmsi::FastxReader r(argv[4]);
        while (r.next()) a.emplace_back(r.name, r.seq);
    }
    std::ifstream f(argv[4], std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string data = ss.str();
    bool fastq = false;
    {
        size_t pos = 0;
        while (pos < data.size()) {
            size_t end = data.find("\n>", pos + 1);
            end = end == std::string::npos ? data.size() : end + 1;
            fmsi::FastxReader r(reinterpret_cast<const unsigned char*>(data.data()) + pos, end - pos);
            std::string text;
            size_t no, nl, so, sl;
            for (;;) {
                const size_t before = text.size();
                if (!r.next_append(text, &no, &nl, &so, &sl)) {
                    text.resize(before);
                    break;
                }
                b.emplace_back(text.substr(no, nl), text.substr(so, sl));
            }
            fastq |= r.saw_fastq();
            pos = end;
        }
    }
    if (!fastq && a != b) {
        std::cerr << "readers disagree: " << a.size() << " vs " << b.size() << " records\n";
        return 1;
    }
    std::cout << a.size() << " records, fastq=" << fastq << "\n";
    return 0;
}
