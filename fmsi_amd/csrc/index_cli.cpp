// `fmsi index` on the GPU: counterpart of ms_index (reference src/main.cpp:172-236).  The masked superstring is
// read exactly as the reference reads it (first FASTA record, k inferred from the trailing lower-case run,
// src/parser.h:31-55), the suffix sort / BWT / SA-transformed mask / kLCP are computed on the device
// (build_index.hip) and the six files are written in the reference's format (index_writer.cpp): byte-identical
// to `fmsi index` of the reference (tests/test_gpu_index_cli.py).
#include <fcntl.h>
#include <getopt.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <cstdlib>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>

#include "../../include/fmsi_hip.h"
#include "common.h"
#include "fastx.h"
#include "pgunzip.h"

#include <algorithm>
#include <thread>
#include <vector>

namespace {

int usage_index() {
    std::cerr << std::endl;
    std::cerr << "Usage:   fmsi index [options] <masked-superstring-input>" << std::endl << std::endl;
    std::cerr << "Options:" << std::endl;
    std::cerr << "    -k INT  - size of k-mers [recommended, default: number of mask trailing zeros - 1]" << std::endl;
    std::cerr << "    -x      - do not compute the kLCP array used for faster streaming queries." << std::endl;
    std::cerr << "    --device INT - HIP device to use [default: 0]" << std::endl << std::endl;
    std::cerr << "Note: `fmsi index` accepts only masked superstrings - these can be computed e.g. by KmerCamel from any FASTA file."
              << std::endl << std::endl;
    return 1;
}

inline bool is_upper(char c) { return c >= 'A' && c <= 'Z'; }   // src/kmers.h:71-73

}  // namespace

namespace {
int run_index(int argc, char** argv) {
    bool usage = false, no_streaming = false;
    int k = 0, device = 0, c;
    std::string fn;
    if (argc > 1 && std::string(argv[argc - 1]) != "-h") {   // last argument = input, as in the reference (:177-180)
        fn = argv[argc - 1];
        argc--;
    }
    static const option long_opts[] = {{"device", required_argument, nullptr, 1000}, {nullptr, 0, nullptr, 0}};
    optind = 1;
    while ((c = getopt_long(argc, argv, "hk:x", long_opts, nullptr)) >= 0) {
        switch (c) {
        case 'h': usage = true; break;
        case 'k': k = atoi(optarg); break;
        case 'x': no_streaming = true; break;
        case 1000: device = atoi(optarg); break;
        default: return usage_index();
        }
    }
    if (usage) {
        usage_index();
        return 0;
    } else if (fn.empty()) {
        std::cerr << "ERROR: Path to the masked superstring is a required argument." << std::endl;
        return usage_index();
    }
    std::cerr << "Starting " << fn << std::endl;
    const bool verbose = std::getenv("FMSI_HIP_VERBOSE") != nullptr;   // phase times on stderr
    auto t_phase = std::chrono::steady_clock::now();
    auto phase = [&](const char* what) {
        const auto now = std::chrono::steady_clock::now();
        if (verbose) fprintf(stderr, "[fmsi index] %-28s %8.3f s\n", what, std::chrono::duration<double>(now - t_phase).count());
        t_phase = now;
    };
    // The first record's sequence is the masked superstring (read_masked_superstring, src/parser.h:41-55).  A plain file is mapped and
    // parsed in place -- a sequence on one line, the usual shape, is then a view of the file and never copied on the host --; a gzip
    // stream (or a file that cannot be mapped) goes through the ordinary reader.
    std::string ms_owned;
    const char* ms_data = nullptr;
    size_t ms_len = 0;
    bool more = false;
    struct Mapping {
        const unsigned char* data = nullptr;
        size_t size = 0;
        ~Mapping() {
            if (data) munmap(const_cast<unsigned char*>(data), size);
        }
    } map;
    fmsi::FastxViewParser::Arena arena;
    fmsi::PgzBuf inflated;   // a gzip file's text, inflated by several threads (pgunzip.h)
    size_t inflated_len = 0;
    {
        const int fd = open(fn.c_str(), O_RDONLY);
        struct stat st;
        if (fd >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size >= 2) {
            void* p = mmap(nullptr, static_cast<size_t>(st.st_size), PROT_READ, MAP_PRIVATE, fd, 0);
            if (p != MAP_FAILED) {
                map.data = static_cast<const unsigned char*>(p);
                map.size = static_cast<size_t>(st.st_size);
                if (map.data[0] == 0x1f && map.data[1] == 0x8b) {   // gzip
                    // An ordinary gzip file of a megabyte and more is inflated by several threads (a human-scale superstring is 0.8 GB of
                    // gzip: 8 s of zlib's one thread in front of a build of 2.3 s) and parsed like a mapped plain file; a stream that
                    // reader does not take (pgunzip.h), or FMSI_HIP_NO_PARALLEL_GUNZIP, leaves the file to the ordinary reader below.
                    fmsi::ParallelGunzip pgz;
                    if (!std::getenv("FMSI_HIP_NO_PARALLEL_GUNZIP") && pgz.open(fn, size_t(1) << 20)) {
                        const unsigned team = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
                        const fmsi::ParallelGunzip::Runner run = [](unsigned n, const std::function<void(unsigned)>& f) {
                            std::vector<std::thread> th;
                            for (unsigned t = 1; t < n; ++t) th.emplace_back([&f, t] { f(t); });
                            f(0);
                            for (auto& x : th) x.join();
                        };
                        // (the trailer's length field is the text's length modulo 2^32: a first guess at the room)
                        const size_t isize = map.data[map.size - 4] | (static_cast<size_t>(map.data[map.size - 3]) << 8) | (static_cast<size_t>(map.data[map.size - 2]) << 16) |
                                             (static_cast<size_t>(map.data[map.size - 1]) << 24);
                        inflated.grow(std::max(isize, map.size * 3) + (size_t(1) << 20), 0);
                        for (;;) {
                            if (inflated.cap() - inflated_len < (size_t(64) << 20)) inflated.grow(inflated.cap() + inflated.cap() / 2, inflated_len);
                            const size_t got = pgz.read(reinterpret_cast<char*>(inflated.data()) + inflated_len, inflated.cap() - inflated_len, team, run);
                            inflated_len += got;
                            if (got == 0) break;
                        }
                        if (pgz.failed()) inflated_len = 0;   // (whatever it is: zlib reads the file from its first byte, as it always did)
                    }
                    munmap(p, map.size);
                    map.data = nullptr;
                }
            }
        }
        if (fd >= 0) close(fd);
    }
    if (map.data || inflated_len) {
        fmsi::FastxViewParser parser(map.data ? map.data : inflated.data(), map.data ? map.size : inflated_len);
        const char *name, *seq;
        size_t name_len;
        if (!parser.next(arena, &name, &name_len, &seq, &ms_len))
            throw std::invalid_argument("Error reading the fasta file. The fasta file should contain a single entry - the masked superstring.");
        ms_data = seq;
        const char* seq2;
        size_t len2;
        more = parser.next(arena, &name, &name_len, &seq2, &len2);
    } else {
        fmsi::FastxReader reader(fn);   // throws std::invalid_argument like OpenFile(), src/parser.h:22
        if (!reader.next()) throw std::invalid_argument("Error reading the fasta file. The fasta file should contain a single entry - the masked superstring.");
        ms_owned.swap(reader.seq);
        ms_data = ms_owned.data();
        ms_len = ms_owned.size();
        more = reader.next();
    }
    if (more) std::cerr << "Warning: The fasta file contains more than one entry. Only the first entry will be used." << std::endl;
    struct MsView {   // (what the code below knows as `ms`)
        const char* d;
        size_t n;
        bool empty() const { return n == 0; }
        size_t size() const { return n; }
        const char* data() const { return d; }
        char operator[](size_t i) const { return d[i]; }
    } ms{ms_data, ms_len};
    if (ms.empty()) {
        std::cerr << "ERROR: The file '" << fn
                  << "' is in incorrect format. It is supposed to be a fasta file with a single entry, the masked superstring" << std::endl;
        return usage_index();
    }
    std::cerr << "Read masked superstring of length " << ms.size() << std::endl;
    phase("superstring read");
    int inferred_k = 1;                                       // infer_k, src/parser.h:31-37
    while (static_cast<size_t>(inferred_k) <= ms.size() && !is_upper(ms[ms.size() - inferred_k])) inferred_k++;
    if (k == 0) {
        k = inferred_k;
        std::cerr << "Inferred k from the masked case convention: " << k << std::endl;
    }
    if (k != inferred_k) {
        std::cerr << "WARNING: The provided k (" << k << ") does not match the k inferred from the mask convention (" << inferred_k
                  << "). The provided k is used but we recommend double checking that it is correct." << std::endl;
    }
    if (k > 64 && !no_streaming) {
        std::cerr << "WARNING: Construction of kLCP array for streaming support is only available for k <= 64. The index will be "
                     "constructed without streaming support, which results in slower positive streaming queries." << std::endl;
        no_streaming = true;
    }
    fmsi_hip_index* dev = nullptr;
    if (fmsi_hip_index_build(ms.data(), ms.size(), k, !no_streaming, device, 0, &dev) != 0) {
        std::cerr << "ERROR: cannot construct the index on the device: " << fmsi_hip_last_error() << std::endl;
        return 1;
    }
    std::cerr << "Constructed index" << std::endl;
    phase("index built on the device");
    const int rc = fmsi_hip_index_write_files(dev, fn.c_str());
    fmsi_hip_index_destroy(dev);
    if (rc != 0) {
        std::cerr << "ERROR: cannot write the index: " << fmsi_hip_last_error() << std::endl;
        return 1;
    }
    std::cerr << "Written index" << std::endl;
    phase("files written");
    return 0;
}
}  // namespace

// No exception crosses the C boundary: the reference lets std::invalid_argument from its parser terminate the
// process (src/parser.h:22-29); here the message is printed and the exit code is 1.
extern "C" int fmsi_cli_index(int argc, char** argv) {
    // the command is a chain of builder calls in a process of its own: they share the library's block cache (include/fmsi_hip.h)
    struct KeepScratch {
        KeepScratch() { fmsi_hip_keep_scratch(1); }
        ~KeepScratch() { fmsi_hip_keep_scratch(0); }
    } keep_scratch_;
    try {
        return run_index(argc, argv);
    } catch (const std::exception& e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
        return 1;
    }
}

// ---- `fmsi export` (ms_export, reference src/main.cpp:508-541) ------------------------------------------------------
#include "host_index.h"

namespace {
int usage_export() {
    std::cerr << std::endl << "Usage:   fmsi export <index-prefix>" << std::endl << std::endl;
    return 1;
}

int run_export(int argc, char** argv) {
    bool usage = false;
    int c;
    std::string fn;
    if (argc > 1 && std::string(argv[argc - 1]) != "-h") {
        fn = argv[argc - 1];
        argc--;
    }
    int device = 0;
    static const option long_opts[] = {{"device", required_argument, nullptr, 1000}, {nullptr, 0, nullptr, 0}};
    optind = 1;
    while ((c = getopt_long(argc, argv, "h", long_opts, nullptr)) >= 0) {
        switch (c) {
        case 'h': usage = true; break;
        case 1000: device = atoi(optarg); break;
        default: return usage_export();
        }
    }
    if (usage) {
        usage_export();
        return 0;
    }
    // the index goes straight into HBM (fmsi_hip_index_load: RRR decode and block layout on the device), the walk runs there
    // (fmsi_export.hip), and the text comes back into a buffer nobody clears first
    uint64_t rows = 0;
    int32_t index_k = 0, klcp_file = 0;
    if (fmsi_host_index_peek(fn.c_str(), &rows, &index_k, &klcp_file) != 0 || rows == 0) {
        std::cerr << "ERROR: index not correctly loaded. Ensure that you correctly call `fmsi index` before." << std::endl;
        return usage_export();
    }
    fmsi_hip_index* dev = nullptr;
    const int rc_load = fmsi_hip_index_load(fn.c_str(), 0, device, 0, &dev);
    if (rc_load == FMSI_HIP_EIO) {
        std::cerr << "ERROR: index not correctly loaded. Ensure that you correctly call `fmsi index` before." << std::endl;
        return usage_export();
    }
    if (rc_load != 0) {
        std::cerr << "ERROR: cannot create the device index: " << fmsi_hip_last_error() << std::endl;
        return 1;
    }
    const uint64_t text_len = rows - 1;
    std::unique_ptr<char[]> ms(new char[text_len + 1]);
    uint64_t len = 0;
    const int rc = fmsi_hip_index_export(dev, ms.get(), text_len, &len);
    fmsi_hip_index_destroy(dev);
    if (rc != 0) {
        std::cerr << "ERROR: export failed: " << fmsi_hip_last_error() << std::endl;
        return 1;
    }
    ms[len] = '\n';
    static const char header[] = ">exported masked superstring\n";
    std::cout.flush();
    if (std::fwrite(header, 1, sizeof header - 1, stdout) != sizeof header - 1 || std::fwrite(ms.get(), 1, len + 1, stdout) != len + 1 || std::fflush(stdout) != 0) {
        std::cerr << "ERROR: write to stdout failed" << std::endl;
        return 1;
    }
    return 0;
}
}  // namespace

extern "C" int fmsi_cli_export(int argc, char** argv) {
    // the command is a chain of builder calls in a process of its own: they share the library's block cache (include/fmsi_hip.h)
    struct KeepScratch {
        KeepScratch() { fmsi_hip_keep_scratch(1); }
        ~KeepScratch() { fmsi_hip_keep_scratch(0); }
    } keep_scratch_;
    try {
        return run_export(argc, argv);
    } catch (const std::exception& e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
        return 1;
    }
}
