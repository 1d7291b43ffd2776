// Per-stage cost of the CLI's host side (reader, pack, format, write) on ONE thread, without a device: which stage the cores go to.
// Includes query_cli.cpp to reach its file-local stages.  Build: see tools/host_stage_bench.sh.  Development tool, not shipped.
#include "../fmsi_amd/csrc/query_cli.cpp"

#include <random>

using namespace fmsi;
using hclock = std::chrono::steady_clock;
static double since(hclock::time_point t0) { return std::chrono::duration<double>(hclock::now() - t0).count(); }

int main(int argc, char** argv) {
    const size_t reads = argc > 1 ? strtoull(argv[1], nullptr, 10) : 20000, len = 10000;
    const char* out_path = argc > 2 ? argv[2] : "/dev/null";
    std::string fa;
    fa.reserve(reads * (len + 16));
    std::mt19937_64 rng(1);
    for (size_t r = 0; r < reads; ++r) {
        fa += ">r" + std::to_string(r) + "\n";
        for (size_t i = 0; i < len; i += 32) {
            uint64_t x = rng();
            for (size_t j = 0; j < 32 && i + j < len; ++j, x >>= 2) fa += "ACGT"[x & 3];
        }
        fa += "\n";
    }
    const double gb = static_cast<double>(reads * len) / 1e9;
    const auto* d = reinterpret_cast<const unsigned char*>(fa.data());
    for (int rep = 0; rep < 2; ++rep) {
        auto t0 = hclock::now();
        auto P = std::make_unique<Prepared>();
        P->raw.arenas.resize(1);
        parse_range(d, 0, fa.size(), P->raw.arenas[0], P->raw.recs, P->raw.seq_bytes, false);
        const double t_parse = since(t0);
        const double t_copy = 0;
        t0 = hclock::now();
        prepare_batch(*P, 31, 1);
        const double t_prep = since(t0);
        t0 = hclock::now();
        P->packed_bits = !getenv("HSB_BYTES");
        if (P->packed_bits) {
            P->bits.resize((P->windows + 63) / 64 + 1);
            std::memset(P->bits.data(), 0x5a, P->bits.size() * 8);
        } else {
            P->bytes.resize(P->windows);
            std::memset(P->bytes.data(), 1, P->windows);
        }
        const double t_fill = since(t0);
        t0 = hclock::now();
        std::vector<char> buf(P->rec_out[P->raw.recs.size()]);
        format_bits(*P, 1, buf.data());
        const double t_fmt = since(t0);
        t0 = hclock::now();
        FILE* f = std::fopen(out_path, "w");
        std::fwrite(buf.data(), 1, buf.size(), f);
        std::fclose(f);
        const double t_write = since(t0);
        printf("rep %d  %.3f G bases: parse %.3f s (%.2f ns/base)  copy %.3f  prepare %.3f (%.2f)  result-fill %.3f  format %.3f (%.2f)  write %.3f (%.2f)\n",
               rep, gb, t_parse, t_parse / gb, t_copy, t_prep, t_prep / gb, t_fill, t_fmt, t_fmt / gb, t_write, t_write / gb);
    }
    return 0;
}
