// The CLI's word-at-a-time host loops (fmsi_amd/csrc/query_cli.cpp) against their one-character-at-a-time definitions, on random
// inputs of every length and alignment: the validity scan (SSE2), the 2-bit packer (PEXT and plain), the expansion of result bits
// into '0' / '1' (PDEP and plain) and the -S chunk sums / output bytes (SSE2).  Includes query_cli.cpp to reach its file-local
// functions; links libfmsi_hip.so for the symbols that file refers to (none is called).  Built with ASan + UBSan by tests/test_host.py.
#include "../../fmsi_amd/csrc/query_cli.cpp"

#include <random>

using namespace fmsi;

static int fail(const char* what, size_t a, size_t b) {
    fprintf(stderr, "MISMATCH in %s (case %zu, %zu)\n", what, a, b);
    return 1;
}

int main() {
    std::mt19937_64 rng(7);
    // validity scan
    for (size_t it = 0; it < 20000; ++it) {
        const size_t n = rng() % 200, off = rng() % 16;
        std::string s(off + n, 'A');
        for (size_t i = 0; i < n; ++i) s[off + i] = "ACGTacgt"[rng() % 8];
        const size_t bad = rng() % 3 ? rng() % (n + 1) : n;
        if (bad < n) s[off + bad] = "NnX@[`{\0\xC1\x01 "[rng() % 11];
        if (valid_prefix(s.data() + off, static_cast<int64_t>(n)) != static_cast<int64_t>(bad)) return fail("valid_prefix", it, bad);
    }
    // packer
    for (size_t it = 0; it < 5000; ++it) {
        const size_t n = 1 + rng() % 300, off = rng() % 8;
        std::string s(off + n, 'A');
        for (size_t i = 0; i < n; ++i) s[off + i] = "ACGTacgt"[rng() % 8];
        std::vector<uint64_t> want((n + 31) / 32, 0), a(want.size(), 0), b(want.size(), 0);
        for (size_t i = 0; i < n; ++i) want[i / 32] |= static_cast<uint64_t>(kCode.t[static_cast<unsigned char>(s[off + i])]) << (2 * (i % 32));
        pack_run_plain(s.data() + off, n, a.data());
        if (a != want) return fail("pack_run_plain", it, n);
        if (__builtin_cpu_supports("bmi2")) {
            pack_run_bmi2(s.data() + off, n, b.data());
            if (b != want) return fail("pack_run_bmi2", it, n);
        }
    }
    // result bits -> characters
    for (size_t it = 0; it < 5000; ++it) {
        const size_t total = 64 + rng() % 1000;
        std::vector<uint64_t> words(total / 64 + 2);
        for (auto& w : words) w = rng();
        const size_t first = rng() % total, count = rng() % (total - first + 1);
        std::string want(count, '0'), a(count, 'x'), b(count, 'x');
        for (size_t i = 0; i < count; ++i) want[i] = static_cast<char>('0' + ((words[(first + i) >> 6] >> ((first + i) & 63)) & 1));
        expand_bits_plain(words.data(), first, count, a.data());
        if (a != want) return fail("expand_bits_plain", it, count);
        if (__builtin_cpu_supports("bmi2")) {
            expand_bits_bmi2(words.data(), first, count, b.data());
            if (b != want) return fail("expand_bits_bmi2", it, count);
        }
    }
    // -S chunk sums and output bytes
    for (size_t it = 0; it < 20000; ++it) {
        const uint32_t n = 1 + static_cast<uint32_t>(rng() % 440);
        const bool maximized_ones = rng() & 1;
        std::vector<uint8_t> b(n);
        const int shape = static_cast<int>(rng() % 4);   // mostly one strand present, as in reads; or anything
        for (auto& x : b) {
            const int f = shape == 0 ? 1 : (shape == 1 ? -1 : static_cast<int>(rng() % 3) - 1), r = shape == 1 ? 1 : (shape == 0 && rng() % 50 ? -1 : static_cast<int>(rng() % 3) - 1);
            x = static_cast<uint8_t>((f + 1) | ((r + 1) << 2));
        }
        int want[2][2] = {{0, 0}, {0, 0}}, got[2][2];
        bool matters = false;
        std::vector<uint8_t> out(n);
        for (uint32_t i = 0; i < n; ++i) {
            const int f = fwd_of(b[i]), r = rc_of(b[i]);
            const bool dec0 = f == 1 || (f == 0 && maximized_ones), dec1 = r == 1 || (r == 0 && maximized_ones);
            want[0][0] += f;
            want[0][1] += dec0 ? 0 : r;
            want[1][0] += r;
            want[1][1] += dec1 ? 0 : f;
            const int v0 = dec0 ? f : std::max(f, r), v1 = dec1 ? r : std::max(r, f);
            matters |= (v0 == 1) != (v1 == 1);
            out[i] = static_cast<uint8_t>(v0 == 1);
        }
        if (chunk_sums(b.data(), n, maximized_ones, got) != matters || std::memcmp(want, got, sizeof want) != 0) return fail("chunk_sums", it, n);
        chunk_output(b.data(), n, maximized_ones);
        if (b != out) return fail("chunk_output", it, n);
    }
    printf("cli host kernels OK\n");
    return 0;
}
