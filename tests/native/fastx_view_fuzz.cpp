// FastxViewParser (views into the bytes, fmsi_amd/csrc/fastx.cpp) against FastxReader over the same bytes: record by record the
// same names, sequences, end of stream, saw_fastq() and clean_end() -- on well-formed FASTA / FASTQ files and on mutated ones
// (stray header characters, CRs, empty lines, truncations).  Built with ASan + UBSan by tests/test_host.py.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

#include "../../fmsi_amd/csrc/fastx.h"

using fmsi::FastxReader;
using fmsi::FastxViewParser;

static std::string random_file(std::mt19937_64& rng) {
    auto pick = [&](size_t n) { return static_cast<size_t>(rng() % n); };
    std::string s;
    const bool fastq = pick(3) == 0, crlf = pick(4) == 0;
    const std::string nl = crlf ? "\r\n" : "\n";
    const size_t recs = pick(6);
    for (size_t r = 0; r < recs; ++r) {
        s += fastq ? '@' : '>';
        s += "r" + std::to_string(r);
        if (pick(3) == 0) s += (pick(2) ? " " : "\t") + std::string("comment > @ +");
        s += nl;
        const size_t lines = pick(4);
        size_t len = 0;
        for (size_t l = 0; l < lines; ++l) {
            const size_t n = pick(12);
            for (size_t i = 0; i < n; ++i) s += "ACGTNacgt"[pick(9)];
            len += n;
            s += nl;
            if (pick(8) == 0) s += nl;
        }
        if (fastq) {
            s += "+" + (pick(2) ? std::string("r") : std::string()) + nl;
            size_t left = len;
            while (left) {
                const size_t n = 1 + pick(left);
                for (size_t i = 0; i < n; ++i) s += "I@>+#5"[pick(6)];
                left -= n;
                s += nl;
            }
            if (!len && pick(2)) s += nl;
        }
    }
    // mutations
    const size_t muts = pick(4);
    for (size_t m = 0; m < muts && !s.empty(); ++m) {
        const size_t at = pick(s.size());
        switch (pick(5)) {
        case 0: s[at] = ">@+\n\r \tA"[pick(8)]; break;
        case 1: s.erase(at, 1 + pick(3)); break;
        case 2: s.insert(at, 1, ">@+\n\r \tA"[pick(8)]); break;
        case 3: s.resize(at); break;
        default: break;
        }
    }
    return s;
}

int main(int argc, char** argv) {
    const size_t rounds = argc > 1 ? strtoull(argv[1], nullptr, 10) : 200000;
    std::mt19937_64 rng(20260101);
    for (size_t it = 0; it < rounds; ++it) {
        const std::string f = random_file(rng);
        const auto* d = reinterpret_cast<const unsigned char*>(f.data());
        FastxReader a(d, f.size());
        FastxViewParser b(d, f.size());
        FastxViewParser::Arena arena;
        for (size_t rec = 0;; ++rec) {
            const bool ra = a.next();
            const char *name, *seq;
            size_t name_len, seq_len;
            const bool rb = b.next(arena, &name, &name_len, &seq, &seq_len);
            bool same = ra == rb;
            if (same && ra) same = a.name == std::string(name, name_len) && a.seq == std::string(seq, seq_len);
            if (same && !ra) same = a.saw_fastq() == b.saw_fastq() && a.clean_end() == b.clean_end();
            if (same && ra) same = a.saw_fastq() == b.saw_fastq();
            if (!same) {
                fprintf(stderr, "MISMATCH at input %zu record %zu: reader %d view %d\ninput (%zu bytes): ", it, rec, ra, rb, f.size());
                for (unsigned char c : f) fprintf(stderr, c >= 32 && c < 127 ? "%c" : "\\x%02x", c);
                fprintf(stderr, "\n");
                if (ra && rb) fprintf(stderr, "reader name '%s' seq '%s'\nview   name '%.*s' seq '%.*s'\n", a.name.c_str(), a.seq.c_str(),
                                      static_cast<int>(name_len), name, static_cast<int>(seq_len), seq);
                return 1;
            }
            if (!ra) break;
        }
    }
    printf("fastx view parser == reader on %zu inputs\n", rounds);
    return 0;
}
