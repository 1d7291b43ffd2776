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
