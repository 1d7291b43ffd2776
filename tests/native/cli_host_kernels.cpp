// The CLI's word-at-a-time host loops (fmsi_amd/csrc/query_cli.cpp) against their one-character-at-a-time definitions, on random
// inputs of every length and alignment: the validity scan (SSE2), the 2-bit packer (PEXT and plain), the expansion of result bits
// into '0' / '1' (PDEP and plain) and the -S chunk sums / output bytes (SSE2).  Includes query_cli.cpp to reach its file-local
// functions; links libfmsi_hip.so for the symbols that file refers to (none is called).  Built with ASan + UBSan by tests/test_host.py.
#include "../../fmsi_amd/csrc/query_cli.cpp"

#include <charconv>
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
    // lower-case bits
    for (size_t it = 0; it < 5000; ++it) {
        const size_t n = 1 + rng() % 300, off = rng() % 8;
        std::string s(off + n, 'A');
        for (size_t i = 0; i < n; ++i) s[off + i] = "ACGTacgt"[rng() % 8];
        std::vector<uint32_t> want((n + 31) / 32, 0), got(want.size(), 0);
        for (size_t i = 0; i < n; ++i) want[i / 32] |= static_cast<uint32_t>(s[off + i] >= 'a') << (i % 32);
        case_run(s.data() + off, n, got.data());
        if (got != want) return fail("case_run", it, n);
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
    // lookup replay (runs of windows with the same signs skipped behind a step that left the predictor unchanged) against one
    // replay_single per window: absent strands, orders that fit an int, orders of 2^31 and more (absent to the reference's int)
    // and -- as indexes of 2^32 rows and more produce them, replayed value by value -- orders whose low 32 bits are anything
    for (size_t it = 0; it < 60; ++it) {
        const uint64_t nw = it < 50 ? 1 + rng() % 3000 : 200000 + rng() % 300000;
        Prepared P;
        P.windows = nw;
        P.orders.resize(2 * nw);
        const int shape = static_cast<int>(it % 5);
        const bool wide = it % 3 == 0;
        auto draw = [&](int strand) -> int64_t {
            const uint64_t x = rng();
            const int kind = shape == 4 ? static_cast<int>(x % (wide ? 6 : 3)) : (shape == strand ? 1 + static_cast<int>(x % 2) : (x % 40 ? 0 : 1 + static_cast<int>(x % (wide ? 5 : 2))));
            switch (kind) {
            case 0: return -1;
            case 1: return static_cast<int64_t>((x >> 8) % 2000000000u);
            case 2: return static_cast<int64_t>(0x80000000ull + (x >> 8) % 0x7FFFFF00ull);
            case 3: return static_cast<int64_t>((1ull << 32) + (x >> 8) % 1000u);
            case 4: return static_cast<int64_t>((3ull << 32) - 1 - (x >> 8) % 14u);   // (int) in [-14, -1]
            default: return static_cast<int64_t>((5ull << 32) - 20 - (x >> 8) % 100000u);
            }
        };
        uint64_t i = 0;
        while (i < nw) {   // runs of similar windows, as reads give them
            const uint64_t run = 1 + rng() % 200;
            const int64_t f = draw(0), r = draw(1);
            for (uint64_t j = 0; j < run && i < nw; ++j, ++i) {
                P.orders[2 * i] = f < 0 || rng() % 8 ? f : draw(0);
                P.orders[2 * i + 1] = r < 0 || rng() % 8 ? r : draw(1);
            }
        }
        StrandPredictor a, b;
        for (int warm = static_cast<int>(rng() % 5); warm > 0; --warm) {
            const int x = static_cast<int>(rng() % 3) - 1, y = static_cast<int>(rng() % 3) - 1;
            a.log_result(x, y);
            b.log_result(x, y);
        }
        std::vector<int64_t> want(nw);
        for (uint64_t w = 0; w < nw; ++w) want[w] = replay_single(a, Mode::Or, true, P.orders[2 * w], P.orders[2 * w + 1]);
        replay_batch(P, b, Mode::Or, true, false, 31, it % 2 ? 5u : 1u, wide);
        if (!same_state(a, b) || std::memcmp(want.data(), P.orders.data(), nw * 8) != 0) return fail("lookup replay", it, nw);
    }
    // the lookup formatter's integers (put_order: digit pairs from a table, written backwards into place) against std::to_chars: every power
    // of ten and its neighbours, the ends of int and int64, random values of every width, -1 and other negatives
    {
        std::vector<int64_t> vals = {0, -1, -2, -14, 9, 10, 11, 99, 100, 101, INT32_MAX, static_cast<int64_t>(INT32_MAX) + 1, UINT32_MAX, static_cast<int64_t>(UINT32_MAX) + 1, INT64_MAX, INT64_MIN + 1};
        for (int64_t t = 10; t > 0 && t < INT64_MAX / 10; t *= 10)
            for (int64_t d : {-1, 0, 1}) vals.push_back(t + d);
        for (int i = 0; i < 200000; ++i) {
            const uint64_t x = rng();
            vals.push_back(static_cast<int64_t>(x >> (x % 64)) * (i % 17 == 0 ? -1 : 1));
        }
        for (int64_t v : vals) {
            char a[32], b[32];
            char* pa = put_order(a, v);
            const auto rb = std::to_chars(b, b + sizeof b, v);
            if (pa - a != rb.ptr - b || std::memcmp(a, b, static_cast<size_t>(pa - a)) != 0) return fail("put_order", static_cast<size_t>(v & 0xFFFF), static_cast<size_t>(pa - a));
        }
    }
    printf("cli host kernels OK\n");
    // the predictor's recurrence as a table (k-mer batches) against replay_single(), window by window, from every kind of state
    for (size_t it = 0; it < 400; ++it) {
        const size_t n = 1 + rng() % 5000;
        const Mode mode = it % 2 ? Mode::All : Mode::Or;
        std::vector<uint8_t> in(n), a, bb;
        const int style = static_cast<int>(rng() % 4);
        for (auto& x : in) {
            int f, r;
            if (style == 0) { f = static_cast<int>(rng() % 3) - 1; r = static_cast<int>(rng() % 3) - 1; }
            else if (style == 1) { f = rng() % 2 ? 1 : -1; r = -1; if (rng() % 7 == 0) std::swap(f, r); }
            else if (style == 2) { f = -1; r = rng() % 3 ? -1 : static_cast<int>(rng() % 2); }
            else { f = static_cast<int>(rng() % 2); r = static_cast<int>(rng() % 2); }
            x = static_cast<uint8_t>((f + 1) | ((r + 1) << 2));
        }
        StrandPredictor p0;
        if (it % 3) {   // a state from the middle of a run (otherwise: the initial one, previous == -1)
            p0.score = static_cast<int>(rng() % 15) - 7;
            p0.result_scores[0] = static_cast<int>(rng() % 15) - 7;
            p0.result_scores[1] = static_cast<int>(rng() % 15) - 7;
            p0.previous = static_cast<int>(rng() % 2);
        }
        StrandPredictor pa = p0, pb = p0;
        a = in;
        for (size_t w = 0; w < n; ++w) a[w] = static_cast<uint8_t>(replay_single(pa, mode, false, fwd_of(in[w]), rc_of(in[w])) == 1);
        bb = in;
        if (it % 2) replay_presence_tight<true, true>(bb.data(), n, pb, mode);
        else replay_presence_tight<true, false>(bb.data(), n, pb, mode);
        for (size_t w = 0; w < n; ++w)
            if ((bb[w] & 15u) != in[w] || (bb[w] >> 7) != a[w]) return fail("replay_presence_tight", it, w);
        if (!same_state(pa, pb)) return fail("replay_presence_tight (state)", it, n);
    }
    // ... and cut into stretches walked side by side from warmed-up states (replay_presence_team): the sequential walk's results whatever the
    // inputs are -- random ones, where the guesses hold, and runs that never move the scores, where stretches are walked again
    for (size_t it = 0; it < 24; ++it) {
        const size_t n = 300000 + rng() % 400000;
        const Mode mode = it % 2 ? Mode::All : Mode::Or;
        std::vector<uint8_t> in(n);
        const int style = static_cast<int>(it % 4);
        for (size_t w = 0; w < n; ++w) {
            int f, r;
            if (style == 0) { f = static_cast<int>(rng() % 3) - 1; r = static_cast<int>(rng() % 3) - 1; }
            else if (style == 1) { if (rng() % 2) { f = r = -1; } else { f = 1; r = -1; if (rng() % 2) std::swap(f, r); } }
            else if (style == 2) { f = r = -1; if (w % 100000 < 50) f = 1; }                 // absent k-mers nearly throughout: the scores stand still
            else { f = r = 0; if ((w / 70000) % 2) { f = static_cast<int>(rng() % 3) - 1; } }   // stretches of (0, 0), stretches of noise
            in[w] = static_cast<uint8_t>((f + 1) | ((r + 1) << 2));
        }
        StrandPredictor p0;
        if (it % 3) {
            p0.score = static_cast<int>(rng() % 15) - 7;
            p0.result_scores[0] = static_cast<int>(rng() % 15) - 7;
            p0.result_scores[1] = static_cast<int>(rng() % 15) - 7;
            p0.previous = static_cast<int>(rng() % 2);
        }
        StrandPredictor pa = p0, pb = p0;
        std::vector<uint8_t> a = in, bb = in;
        replay_presence_tight<true, false>(a.data(), n, pa, mode);
        const uint64_t reruns = it % 3 ? replay_presence_team<true>(bb.data(), n, pb, mode, 2 + static_cast<unsigned>(it % 5))
                                       : replay_presence_team<false>(bb.data(), n, pb, mode, 2 + static_cast<unsigned>(it % 5));
        if (a != bb || !same_state(pa, pb)) return fail("replay_presence_team", it, static_cast<size_t>(reruns));
        if (style == 1 && reruns != 0) return fail("replay_presence_team: a guess failed on random k-mers", it, static_cast<size_t>(reruns));
    }
    // one k-mer of text -> its 2-bit word (PEXT and plain and the tail form), invalid characters refused
    for (size_t it = 0; it < 20000; ++it) {
        const int k = 1 + static_cast<int>(rng() % 32);
        std::string s(64, 'A');
        for (auto& c : s) c = "ACGTacgt"[rng() % 8];
        const int bad = rng() % 4 == 0 ? static_cast<int>(rng() % 40) : -1;
        if (bad >= 0) s[static_cast<size_t>(bad)] = "NnX@[`{\n\xC1\x01 >"[rng() % 12];
        uint64_t want = 0, a = 0, b = 0, c = 0;
        bool want_ok = true;
        for (int j = 0; j < k; ++j) {
            const uint8_t code = kCode.t[static_cast<unsigned char>(s[static_cast<size_t>(j)])];
            if (code == 4) want_ok = false;
            want |= static_cast<uint64_t>(code & 3) << (2 * j);
        }
        const bool oa = kmer_word_plain(s.data(), k, &a), oc = kmer_word_tail(s.data(), k, &c);
        if (oa != want_ok || oc != want_ok || (want_ok && (a != want || c != want))) return fail("kmer_word_plain/tail", it, static_cast<size_t>(k));
        if (__builtin_cpu_supports("bmi2")) {
            const bool ob = kmer_word_bmi2(s.data(), k, &b);
            if (ob != want_ok || (want_ok && b != want)) return fail("kmer_word_bmi2", it, static_cast<size_t>(k));
        }
    }
    // -S replay of a batch of short reads (replay_batch, streaming): the chunks listed and summed by the team, the predictor's state walked
    // list by list from warmed-up states -- the same bytes and the same final state as with one thread, whatever the strand results are:
    // without windows whose output depends on the strand order (the state walk alone), and with them (the plain walk)
    for (size_t it = 0; it < 12; ++it) {
        const int k = 13;
        const Mode mode = it % 2 ? Mode::All : Mode::Or;
        const bool with_order_windows = it % 4 >= 2;
        const size_t nrec = 60000 + rng() % 20000;
        auto build = [&](Prepared& P, std::vector<RecRef>& recs) {
            uint64_t w = 0;
            P.rec_ops.push_back(0);
            for (size_t r = 0; r < nrec; ++r) {
                const uint64_t L = 20 + (r * 2654435761ull) % 300;
                recs.push_back(RecRef{nullptr, nullptr, L, 0});
                P.rec_first.push_back(w);
                const uint64_t windows = L - k + 1;
                if (r % 7 == 3) {   // an invalid character in the middle: two runs and a fill
                    const uint64_t a = windows / 2;
                    if (a) P.ops.push_back({a, true});
                    P.ops.push_back({static_cast<uint64_t>(k), false});
                    w += a;
                } else {
                    P.ops.push_back({windows, true});
                    w += windows;
                }
                P.rec_ops.push_back(P.ops.size());
            }
            P.rec_first.push_back(w);
            P.windows = w;
        };
        Prepared A, B;
        build(A, A.raw.recs);
        build(B, B.raw.recs);
        A.bytes.resize(A.windows);
        std::mt19937_64 gen(it * 77 + 1);
        for (uint64_t w = 0; w < A.windows; ++w) {
            int f, r;
            const uint64_t x = gen();
            if ((w / 5000) % 3 == 0) { f = r = -1; }                                   // stretches that never move the scores
            else if (x % 2) { f = 1; r = -1; if (x % 8 < 3) std::swap(f, r); }
            else { f = r = static_cast<int>(x % 3) - 1; }
            if (with_order_windows && x % 1000 == 7) { f = 0; r = 1; }
            A.bytes[w] = static_cast<uint8_t>((f + 1) | ((r + 1) << 2));
        }
        B.bytes = A.bytes;
        StrandPredictor pa, pb;
        if (it % 3) { pa.score = 3; pa.result_scores[0] = -2; pa.result_scores[1] = 5; pa.previous = 1; pb = pa; }
        replay_batch(A, pa, mode, false, true, k, 1);
        replay_batch(B, pb, mode, false, true, k, 2 + static_cast<unsigned>(it % 7));
        if (A.bytes != B.bytes || !same_state(pa, pb)) return fail("replay_batch -S by the team", it, static_cast<size_t>(A.windows));
    }
    return 0;
}
