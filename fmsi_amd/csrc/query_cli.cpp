// `fmsi query` / `fmsi lookup` on the GPU: counterpart of ms_query (reference src/main.cpp:238-375).
//
// The host keeps what the reference's caller does -- option parsing, index load + sanity checks, the
// FASTA/FASTQ record loop, invalid-character splitting, chunk bookkeeping, the strand predictor and the
// stdout grammar -- and hands the k-mer windows to libfmsi_hip through the C-ABI (include/fmsi_hip.h).
// Output is byte-identical to the reference: per-window results of both strands come back from the
// device and the sequential strand predictor (src/fms_index.h:18-49) is replayed here.
#include <getopt.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/fmsi_hip.h"
#include "common.h"
#include "fastx.h"
#include "host_index.h"

namespace fmsi {
namespace {

// ---- strand predictor, src/fms_index.h:18-49 ---------------------------------------------------
struct StrandPredictor {
    int score = 0;
    int result_scores[2] = {0, 0};
    int previous = -1;
    static int clipped(int x) { return std::max(-7, std::min(7, x)); }
    void log_result(int fwd, int rev) {
        const int difference = fwd - rev;
        score = clipped(score + difference);
        if (previous != -1) result_scores[previous] = clipped(result_scores[previous] + difference);
        previous = fwd > rev;
    }
    bool predict_swap() const {
        if (previous != -1 && result_scores[previous] != 0) return result_scores[previous] < 0;
        return score < 0;
    }
};

enum class Mode { Or, All };

struct Options {
    std::string index_prefix, query_fn = "-";
    int k = 0;
    bool has_klcp = false;
    Mode mode = Mode::Or;
    bool output_orders = false;
    bool fast = false;        // --fast: skip the second strand when the first decides (exact for OR; for -O under its precondition)
    bool stats = false;       // --stats: timings on stderr
    int device = 0;
    uint64_t batch_windows = 1ull << 24;
};

int usage_query(bool lookup) {
    // text of the reference's usage_query()/usage_lookup() (src/main.cpp:102-130) plus the GPU options
    std::cerr << std::endl;
    std::cerr << "Usage:   fmsi " << (lookup ? "lookup" : "query") << " [options] <index-prefix>" << std::endl << std::endl;
    std::cerr << "Options (stable):" << std::endl;
    std::cerr << "  -q FILE - Path to FASTA/FASTQ with queries [default: stdin]" << std::endl;
    std::cerr << "  -k INT  - Size of k-mers [default: infer automatically from index]" << std::endl;
    std::cerr << "  -S      - Use kLCP array for streamed queries (increses memory consumption)" << std::endl;
    if (!lookup) {
        std::cerr << "  -O      - FMSI uses properties of max-one masked superstrings to speed up queries" << std::endl;
        std::cerr << "            Use only if a masked superstring with maximum number of ones is indexed." << std::endl;
        std::cerr << "  -f FUNCTION - Demasking function; the GPU path supports: or [default], all (= -O)" << std::endl;
    }
    std::cerr << "Options (MI355X path):" << std::endl;
    std::cerr << "  --device INT - HIP device to use [default: 0]" << std::endl;
    std::cerr << "  --fast       - search the reverse strand only when the forward strand does not decide;" << std::endl;
    std::cerr << "                 identical output for the default mode, and for -O on max-ones masks" << std::endl;
    std::cerr << "  --batch INT  - k-mer windows per device batch [default: 16777216]" << std::endl;
    std::cerr << "  --stats      - print timings to stderr" << std::endl;
    std::cerr << std::endl;
    return 1;
}

// What one record prints, in order: runs of windows (results come from the device) and fills of
// '0' / '-1' after invalid characters (src/main.cpp:359-370).
struct Op {
    bool is_run;
    uint64_t count;
};
struct RecordPlan {
    std::string name;
    int64_t max_chunk;       // k + max(10, min(400, 2*sqrt(L))), src/main.cpp:330-331
    std::vector<Op> ops;
    uint64_t first_result;   // offset of the record's first window in the batch results
};

struct Batch {
    std::vector<RecordPlan> records;
    std::vector<uint64_t> packed;        // 2-bit bases of all runs, concatenated
    uint64_t bases = 0;
    std::vector<uint64_t> window_start;  // per window: first base in `packed`
    std::vector<uint64_t> run_start;     // per run of consecutive windows: first base / number of windows (-S kernel)
    std::vector<uint32_t> run_windows;
    void clear() {
        records.clear();
        packed.assign(2, 0);
        bases = 0;
        window_start.clear();
        run_start.clear();
        run_windows.clear();
    }
};

inline int code_of(unsigned char ch) {   // nucleotideToInt, src/kmers.h:3-20
    switch (ch) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}

void append_bases(Batch& b, const char* s, uint64_t len) {
    const uint64_t need_words = (b.bases + len) / 32 + 2;
    if (b.packed.size() < need_words) b.packed.resize(std::max<uint64_t>(need_words, b.packed.size() * 2), 0);
    for (uint64_t i = 0; i < len; ++i) {
        const uint64_t p = b.bases + i;
        b.packed[p >> 5] |= static_cast<uint64_t>(code_of(static_cast<unsigned char>(s[i]))) << (2 * (p & 31));
    }
    b.bases += len;
}

// Walk of one record exactly as src/main.cpp:337-371 does it, recording instead of querying.
void plan_record(Batch& b, const std::string& name, const std::string& seq, int k) {
    RecordPlan rp;
    rp.name = name;
    int64_t sequence_length = static_cast<int64_t>(seq.size());
    rp.max_chunk = k + std::max<int64_t>(10, std::min<int64_t>(400, 2 * static_cast<int64_t>(std::sqrt(sequence_length))));
    rp.first_result = b.window_start.size();
    const char* sequence = seq.data();
    while (sequence_length > 0) {
        int64_t current_length = 0;   // next_invalid_character_or_end, src/parser.h:57-64
        while (current_length < sequence_length && code_of(static_cast<unsigned char>(sequence[current_length])) != 4) ++current_length;
        if (current_length >= k) {
            const uint64_t windows = static_cast<uint64_t>(current_length - k + 1);
            const uint64_t base0 = b.bases;
            append_bases(b, sequence, static_cast<uint64_t>(current_length));
            for (uint64_t w = 0; w < windows; ++w) b.window_start.push_back(base0 + w);
            for (uint64_t w = 0; w < windows; w += 0x40000000ull) {   // run lengths are 32-bit in the C-ABI
                b.run_start.push_back(base0 + w);
                b.run_windows.push_back(static_cast<uint32_t>(std::min<uint64_t>(windows - w, 0x40000000ull)));
            }
            rp.ops.push_back({true, windows});
            // the reference consumes chunk_length - k + 1 characters per chunk: k - 1 remain
            sequence += windows;
            sequence_length -= static_cast<int64_t>(windows);
            current_length -= static_cast<int64_t>(windows);
        }
        sequence_length -= current_length + 1;
        sequence += current_length + 1;
        if (sequence_length >= 0) rp.ops.push_back({false, static_cast<uint64_t>(std::min<int64_t>(k, current_length + 1))});
    }
    b.records.push_back(std::move(rp));
}

struct Emitter {
    std::string out;
    bool orders;
    bool first_in_record = true;
    void begin_record(const std::string& name) {
        out += name;
        out += '\t';
        first_in_record = true;
    }
    void bit(bool one) { out += one ? '1' : '0'; }
    void order(int64_t v) {
        if (!first_in_record) out += ',';
        first_in_record = false;
        out += std::to_string(v);
    }
    void end_record() { out += '\n'; }
    void flush() {
        if (!out.empty()) std::fwrite(out.data(), 1, out.size(), stdout);
        out.clear();
    }
};

// Per-window device results of one batch.
struct Results {
    std::vector<uint8_t> bytes;      // presence modes: fast -> 0/1 ; strict -> (fwd+1)|((rc+1)<<2)
    std::vector<int64_t> orders;     // lookup: fast -> order ; strict -> {fwd, rc}
    bool strict = true;
    int fwd(uint64_t w) const { return static_cast<int>(bytes[w] & 3) - 1; }
    int rc(uint64_t w) const { return static_cast<int>((bytes[w] >> 2) & 3) - 1; }
};

// query_kmers_single<mode> for one window given both strands' results (src/fms_index.h:263-331).
inline void replay_single(StrandPredictor& p, Mode mode, bool orders, int64_t f, int64_t r, int64_t* got_out) {
    const bool should_swap = p.predict_swap();
    const int64_t first = should_swap ? r : f, second = should_swap ? f : r;
    int fpr = 0, bpr = 0;
    int64_t got = first;
    fpr = static_cast<int>(got);
    if (orders) {
        if (fpr >= 0) fpr = 1;
        else { got = second; bpr = got >= 0 ? 1 : -1; }
    } else if (mode == Mode::Or) {
        if (got != 1) { got = second; bpr = static_cast<int>(got); }
    } else {
        if (got == -1) { got = second; bpr = static_cast<int>(got); }
    }
    if (should_swap) std::swap(fpr, bpr);
    p.log_result(fpr, bpr);
    *got_out = got;
}

// query_kmers_streaming<> for one chunk of `n` windows starting at result index w0 (src/fms_index.h:181-254).
void replay_streaming_chunk(StrandPredictor& p, Mode mode, bool orders, const Results& R, uint64_t w0, uint64_t n, Emitter& em) {
    const bool should_swap = p.predict_swap();
    const bool maximized_ones = mode == Mode::All;
    int fpr = 0, bpr = 0;
    std::vector<int64_t> result(n);
    for (uint64_t i = 0; i < n; ++i) {
        // result[] is indexed in the orientation of the (possibly swapped) first strand; window w of the
        // forward string is window n-1-w of the reverse-complement string
        const uint64_t w = should_swap ? (n - 1 - i) : i;
        const int64_t first = orders ? (should_swap ? R.orders[2 * (w0 + w) + 1] : R.orders[2 * (w0 + w)])
                                     : (should_swap ? R.rc(w0 + w) : R.fwd(w0 + w));
        result[i] = first;
        if (orders) fpr += first >= 0 ? 1 : -1;
        else fpr += static_cast<int>(first);
    }
    for (uint64_t i = 0; i < n; ++i) {
        if ((result[i] >= 0 && orders) || result[i] == 1 || (result[i] == 0 && maximized_ones)) continue;
        // second pass walks the other strand's string: its window i_back pairs with result[i]
        const uint64_t w = should_swap ? (n - 1 - i) : i;
        const int64_t second = orders ? (should_swap ? R.orders[2 * (w0 + w)] : R.orders[2 * (w0 + w) + 1])
                                      : (should_swap ? R.fwd(w0 + w) : R.rc(w0 + w));
        if (orders) bpr += second >= 0 ? 1 : -1;
        else bpr += static_cast<int>(second);
        result[i] = std::max(result[i], second);
    }
    if (should_swap) {
        std::reverse(result.begin(), result.end());
        std::swap(fpr, bpr);
    }
    p.log_result(fpr, bpr);
    for (uint64_t i = 0; i < n; ++i) {
        if (orders) em.order(result[i]);
        else em.bit(result[i] == 1);
    }
}

void check(int rc, const char* what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + fmsi_hip_last_error());
}

}  // namespace
}  // namespace fmsi

using namespace fmsi;

extern "C" int fmsi_cli_query(int argc, char** argv, int output_orders_i) {
    const bool output_orders = output_orders_i != 0;
    Options opt;
    opt.output_orders = output_orders;
    bool usage = false;
    std::string f_name = "or";
    // the reference takes the LAST argument as the index prefix before getopt (src/main.cpp:244-247)
    if (argc > 1 && std::string(argv[argc - 1]) != "-h") {
        opt.index_prefix = argv[argc - 1];
        argc--;
    }
    static const option long_opts[] = {{"device", required_argument, nullptr, 1000}, {"fast", no_argument, nullptr, 1001},
                                       {"batch", required_argument, nullptr, 1002},  {"stats", no_argument, nullptr, 1003},
                                       {"strict", no_argument, nullptr, 1004},       {nullptr, 0, nullptr, 0}};
    optind = 1;
    int c;
    while ((c = getopt_long(argc, argv, "f:hq:k:OS", long_opts, nullptr)) >= 0) {
        switch (c) {
        case 'f':
            f_name = optarg;
            if (f_name != "or" && f_name != "all") {
                std::cerr << "ERROR: Function '" << optarg << "' is not available on the GPU path (supported: or, all)." << std::endl;
                return usage_query(output_orders);
            }
            break;
        case 'h': usage = true; break;
        case 'q': opt.query_fn = optarg; break;
        case 'k': opt.k = atoi(optarg); break;
        case 'O':
            if (f_name != "or") std::cerr << "WARNING: Parameter -O is ignored when parameter -f is specified." << std::endl;
            else f_name = "all";
            break;
        case 'S': opt.has_klcp = true; break;
        case 1000: opt.device = atoi(optarg); break;
        case 1001: opt.fast = true; break;
        case 1002: opt.batch_windows = std::max<uint64_t>(1024, strtoull(optarg, nullptr, 10)); break;
        case 1003: opt.stats = true; break;
        case 1004: opt.fast = false; break;
        default: return usage_query(output_orders);
        }
    }
    if (usage) {
        usage_query(output_orders);
        return 0;
    } else if (opt.index_prefix.empty()) {
        std::cerr << "ERROR: Path to the fasta file is a required argument." << std::endl;
        return usage_query(output_orders);
    }
    if (output_orders && f_name == "all") {
        std::cerr << "WARNING: The current version of FMSI has speed benefits only if output as (minimum) perfect hash function is not used. Additionally, if you desire minimum perfect hash function, please minimize the number of ones in the mask." << std::endl;
    }
    opt.mode = f_name == "all" ? Mode::All : Mode::Or;

    using clock = std::chrono::steady_clock;
    auto t_start = clock::now();
    fmsi_host_index* host = nullptr;
    if (fmsi_host_index_load(opt.index_prefix.c_str(), opt.has_klcp, &host) != 0 || host->ix.n() == 0) {
        std::cerr << "ERROR: index not correctly loaded. Ensure that you correctly call `fmsi index` before." << std::endl;
        return usage_query(output_orders);
    }
    if (opt.has_klcp != (host->ix.klcp.bits > 0)) {
        std::cerr << "ERROR: kLCP array was not constructed for the given index. Either construct it again without the `-s` flag or use `query -s` which slows down streaming queries." << std::endl;
        fmsi_host_index_free(host);
        return usage_query(output_orders);
    }
    int k = opt.k;
    if (k != 0 && k != host->ix.k) {
        std::cerr << "ERROR: Mismatch. Provided k (" << k << ") does not match the k of the index (" << host->ix.k << ")." << std::endl;
        fmsi_host_index_free(host);
        return usage_query(output_orders);
    }
    if (k == 0) k = host->ix.k;

    fmsi_hip_index* dev = nullptr;
    fmsi_index_view view;
    fmsi_host_index_view(host, &view);
    // private accelerators (results unchanged): a prefix table of min(k, 12) steps -- smaller for small indexes, where
    // 4^t entries would dwarf the index -- and, for non-streaming queries, the two-base-step pair layout
    int t = std::min(k, 12);
    while (t > 0 && (1ull << (2 * t)) > 4 * host->ix.n()) --t;
    view.prefix_len = t;
    if (fmsi_hip_index_create(&view, opt.device, &dev) != 0 || (!opt.has_klcp && fmsi_hip_index_enable_pair_layout(dev) != 0)) {
        std::cerr << "ERROR: cannot create the device index: " << fmsi_hip_last_error() << std::endl;
        if (dev) fmsi_hip_index_destroy(dev);
        fmsi_host_index_free(host);
        return 1;
    }
    auto t_loaded = clock::now();

    // OR-mode presence output does not depend on the strand order, so the cheaper schedule is exact;
    // everything else replays the predictor from both strands unless --fast.
    const bool need_both = opt.has_klcp || output_orders || opt.mode == Mode::All;
    const bool strict = need_both && !opt.fast;
    uint32_t flags = (opt.mode == Mode::All ? FMSI_HIP_MODE_ALL : FMSI_HIP_MODE_OR);
    if (strict) flags |= FMSI_HIP_STRICT;
    if (output_orders) flags |= FMSI_HIP_ORDERS;

    int exit_code = 0;
    double gpu_s = 0;
    uint64_t total_windows = 0;
    try {
        FastxReader reader(opt.query_fn);   // throws std::invalid_argument like OpenFile(), src/parser.h:22
        StrandPredictor predictor;
        Batch batch;
        batch.clear();
        Results R;
        R.strict = strict;
        Emitter em;
        em.orders = output_orders;

        auto run_batch = [&]() {
            const uint64_t nw = batch.window_start.size();
            const size_t elem = output_orders ? (strict ? 2 : 1) : 1;
            if (output_orders) R.orders.resize(nw * elem);
            else R.bytes.resize(nw);
            auto t0 = clock::now();
            if (nw) {
                void* dst = output_orders ? static_cast<void*>(R.orders.data()) : static_cast<void*>(R.bytes.data());
                if (opt.has_klcp)   // -S: kLCP streaming kernel (query_kmers_streaming, src/fms_index.h:181-254)
                    check(fmsi_hip_query_stream(dev, batch.packed.data(), batch.bases, batch.run_start.data(), batch.run_windows.data(),
                                                batch.run_start.size(), nw, k, flags, dst, nullptr, nullptr),
                          "fmsi_hip_query_stream");
                else
                    check(fmsi_hip_query_windows(dev, batch.packed.data(), batch.bases, batch.window_start.data(), nw, k, flags, dst,
                                                 nullptr, nullptr),
                          "fmsi_hip_query_windows");
                check(fmsi_hip_sync(dev, nullptr), "fmsi_hip_sync");
            }
            gpu_s += std::chrono::duration<double>(clock::now() - t0).count();
            total_windows += nw;
            for (const RecordPlan& rp : batch.records) {
                em.begin_record(rp.name);
                uint64_t w = rp.first_result;
                for (const Op& op : rp.ops) {
                    if (!op.is_run) {
                        for (uint64_t i = 0; i < op.count; ++i) {
                            if (output_orders) em.order(-1);
                            else em.bit(false);
                        }
                        continue;
                    }
                    if (!strict) {
                        // the device already applied the strand schedule
                        for (uint64_t i = 0; i < op.count; ++i) {
                            if (output_orders) em.order(R.orders[w + i]);
                            else em.bit(R.bytes[w + i] != 0);
                        }
                    } else if (!opt.has_klcp) {
                        for (uint64_t i = 0; i < op.count; ++i) {
                            int64_t got;
                            const int64_t f = output_orders ? R.orders[2 * (w + i)] : R.fwd(w + i);
                            const int64_t r = output_orders ? R.orders[2 * (w + i) + 1] : R.rc(w + i);
                            replay_single(predictor, opt.mode, output_orders, f, r, &got);
                            if (output_orders) em.order(got);
                            else em.bit(got == 1);
                        }
                    } else {
                        // chunks of the run as src/main.cpp:340-354 cuts them
                        int64_t current_length = static_cast<int64_t>(op.count) + k - 1;
                        uint64_t cw = w;
                        while (current_length >= k) {
                            const int64_t chunk_length = std::min<int64_t>(current_length, rp.max_chunk);
                            const uint64_t n = static_cast<uint64_t>(chunk_length - k + 1);
                            replay_streaming_chunk(predictor, opt.mode, output_orders, R, cw, n, em);
                            cw += n;
                            current_length -= static_cast<int64_t>(n);
                        }
                    }
                    w += op.count;
                }
                em.end_record();
                if (em.out.size() > (1u << 22)) em.flush();
            }
            em.flush();
            batch.clear();
        };

        while (reader.next()) {
            plan_record(batch, reader.name, reader.seq, k);
            if (batch.window_start.size() >= opt.batch_windows) run_batch();
        }
        run_batch();
        std::fflush(stdout);
    } catch (const std::invalid_argument&) {
        fmsi_hip_index_destroy(dev);
        fmsi_host_index_free(host);
        throw;   // uncaught in the reference too: abort, exit status 134
    } catch (const std::exception& e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
        exit_code = 1;
    }
    if (opt.stats) {
        const double total = std::chrono::duration<double>(clock::now() - t_start).count();
        const double load = std::chrono::duration<double>(t_loaded - t_start).count();
        fprintf(stderr, "{\"windows\": %llu, \"total_seconds\": %.6f, \"load_seconds\": %.6f, \"device_seconds\": %.6f, \"strict\": %s}\n",
                static_cast<unsigned long long>(total_windows), total, load, gpu_s, strict ? "true" : "false");
    }
    fmsi_hip_index_destroy(dev);
    fmsi_host_index_free(host);
    return exit_code;
}
