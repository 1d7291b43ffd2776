// `fmsi query` / `fmsi lookup` on the GPU: counterpart of ms_query (reference src/main.cpp:238-375).
//
// The host keeps what the reference's caller does -- option parsing, index load + sanity checks, the
// FASTA/FASTQ record loop, invalid-character splitting, chunk bookkeeping, the strand predictor and the
// stdout grammar -- and hands the k-mer windows to libfmsi_hip through the C-ABI (include/fmsi_hip.h).
// Output is byte-identical to the reference: per-window results of both strands come back from the
// device and the sequential strand predictor (src/fms_index.h:18-49) is replayed here.
//
// Host pipeline, every stage on threads of its own, batches in bounded queues, output order = input order:
//   readers (cli_reader.h: mapped plain files and block-gzip files parsed in place, the ordinary reader for gzip / stdin)
//   -> two workers per device handle: planning + 2-bit packing on a team (cli_host_kernels.h), one device call per batch
//   -> this thread, batches in order: predictor replay (sequential by nature; skipped when nothing depends on it)
//   -> formatter (a team writes the batch's lines into one buffer) -> sink (write()).
#include <getopt.h>
#include <sys/stat.h>
#include <unistd.h>
#include <sys/mman.h>
#include <fcntl.h>
#include <linux/falloc.h>

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <charconv>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fmsi_hip.h"
#include "common.h"
#include "cli_host_kernels.h"
#include "cli_reader.h"
#include "fastx.h"
#include "trace.h"
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
    std::vector<int> devices;              // --devices a,b,...: one replicated handle per device, batches dealt out among them
    uint64_t batch_windows = 1ull << 26;   // bases of query sequence per batch (>= windows)
    unsigned threads = 0;                  // 0 = all CPUs this process may use, at most 16
    bool full_tables = false;              // --tables full: the k-mer dictionary (what bench.py measures); lookup: the 16-deep prefix table
    bool compact_tables = false;           // --tables compact: the super-k-mer dictionary alone (records filed by minimizer; a third of the dictionary's HBM)
    bool candidate_tables = false;         // --tables candidates: the 16-deep prefix table + candidate tables instead (A/B runs)
    bool tables_given = false;             // (--tables on the command line: no choice by the size of the query file)
    bool background_dict = false;          // --tables background, or a large query file: the dictionary is built beside the first batches
    int fn = 0;                            // -f and|xor|r-s: FMSI_HIP_FN_* (query_mode::general), 0 = the presence modes
    uint64_t fn_r = 0, fn_s = 0;
};

// mask_function(name), src/functions.h:23-57: 0 = a general function (o.fn, o.fn_r, o.fn_s set), 1 = "or", 2 = "all", -1 = not recognized.
// A bound that std::stoi rejects leaves the reference with an uncaught std::out_of_range: the same exception from here.
int parse_function(const std::string& name, Options& o) {
    if (name == "or") return 1;
    if (name == "and") {
        o.fn = FMSI_HIP_FN_AND;
        return 0;
    }
    if (name == "xor") {
        o.fn = FMSI_HIP_FN_XOR;
        return 0;
    }
    if (name == "all") return 2;
    for (size_t i = 1; i + 1 < name.size(); ++i) {
        if (name[i] != '-') continue;
        const std::string r_string = name.substr(0, i), s_string = name.substr(i + 1);
        const auto digits = [](const std::string& x) { return x.find_first_not_of("0123456789") == std::string::npos; };
        if (!digits(r_string) || !digits(s_string)) break;
        o.fn = FMSI_HIP_FN_RANGE;
        o.fn_r = static_cast<uint64_t>(std::stoi(r_string));
        o.fn_s = static_cast<uint64_t>(std::stoi(s_string));
        return 0;
    }
    return -1;
}

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
        std::cerr << "Parameters (experimental, using f-MS framework):" << std::endl;   // usage_functions(), src/main.cpp:85-100
        std::cerr << "  -f FUNCTION - Demasking function to determine k-mer presence; recognized functions:" << std::endl;
        std::cerr << "    or      - represented when at least 1 ON occurrence [default]" << std::endl;
        std::cerr << "    all     - all occurrence are either ON or OFF (equivalent to -O flag for queries)" << std::endl;
        std::cerr << "    and     - represented when no OFF occurrence" << std::endl;
        std::cerr << "    xor     - represented when an odd number of ON occurrences" << std::endl;
        std::cerr << "    INT-INT - represented when in the bounds" << std::endl;
    }
    std::cerr << "Options (MI355X path):" << std::endl;
    std::cerr << "  --device INT - HIP device to use [default: 0]" << std::endl;
    std::cerr << "  --devices LIST - several devices (e.g. 0,1,2,3): the index is replicated, batches are dealt out, output order is kept" << std::endl;
    std::cerr << "  --fast       - search the reverse strand only when the forward strand does not decide;" << std::endl;
    std::cerr << "                 identical output for the default mode, and for -O on max-ones masks" << std::endl;
    std::cerr << "  --batch INT  - bases of query sequence per device batch [default: 67108864]" << std::endl;
    std::cerr << "  --threads INT - host threads for packing and formatting [default: available CPUs, at most 16]" << std::endl;
    std::cerr << "  --stats      - print timings to stderr" << std::endl;
    std::cerr << "  --tables auto|background|full|compact|candidates - device-side accelerators (results unaffected).  auto [default]: a prefix" << std::endl;
    std::cerr << "                 table of up to 14 bases (15 where that makes the streaming kernel's shortcuts possible), which keeps the device" << std::endl;
    std::cerr << "                 ahead of this program's reader and loads fastest; an index with its kLCP file streams its windows with or" << std::endl;
    std::cerr << "                 without -S -- the answers are the same (query files of 8 GB and more get `background`); background: the same," << std::endl;
    std::cerr << "                 and the k-mer dictionary is built" << std::endl;
    std::cerr << "                 beside the first batches, which the LF walk answers -- every batch after it stands goes to the dictionary;" << std::endl;
    std::cerr << "                 full: the dictionary before the first batch -- one 128-byte read per k-mer, about 13 bytes of HBM per" << std::endl;
    std::cerr << "                 distinct k-mer, the fastest device path for isolated k-mers (also for -S); compact: the super-k-mer" << std::endl;
    std::cerr << "                 dictionary instead (23 <= k <= 32) -- records filed by minimizer, one read per ~6 windows of a sequence," << std::endl;
    std::cerr << "                 about 6 bytes of HBM per distinct k-mer, the fastest path for reads; candidates: the 16-base prefix table with" << std::endl;
    std::cerr << "                 candidate tables (up to 180 GB of HBM)" << std::endl;
    std::cerr << std::endl;
    return 1;
}

// ---- threading helpers ---------------------------------------------------------------------------
unsigned available_cpus() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::max(1, CPU_COUNT(&set));
    std::ifstream f("/sys/fs/cgroup/cpu.max");   // "<quota> <period>" or "max <period>"
    std::string quota;
    double period = 0;
    if (f >> quota >> period && quota != "max" && period > 0) {
        const double q = std::atof(quota.c_str()) / period;
        if (q >= 1) n = std::min<unsigned>(n, static_cast<unsigned>(q + 0.999));
    }
    return std::max(1u, n);
}

// ---- batches -----------------------------------------------------------------------------------
// What one record prints, in order: runs of windows (results come from the device) and fills of
// '0' / '-1' after invalid characters (src/main.cpp:359-370).
struct Op {
    uint64_t count;
    bool is_run;
};
// Page-locked memory for what crosses the link every batch (packed bases in, result bytes / orders out): blocks come from
// fmsi_hip_host_alloc and are kept in a small cache by size, since pinning costs far more than a batch's copy.  Small or
// failed requests fall back to the heap.  resize() does not clear: the device writes every element before it is read.
class PinnedPool {
public:
    static constexpr size_t kMin = 1u << 20;
    static size_t bucket(size_t bytes) {
        size_t b = kMin;
        while (b < bytes) b <<= 1;
        return b;
    }
    static void* get(size_t bytes) {
        if (bytes < kMin) return std::malloc(bytes ? bytes : 1);
        const size_t b = bucket(bytes);
        {
            std::lock_guard<std::mutex> lock(mutex());
            auto& v = cache()[b];
            if (!v.empty()) {
                void* p = v.back();
                v.pop_back();
                return p;
            }
        }
        void* p = fmsi_hip_host_alloc(b);
        if (p) {
            std::lock_guard<std::mutex> lock(mutex());
            pinned().push_back(p);
            return p;
        }
        return std::malloc(b);
    }
    static void put(void* p, size_t bytes) {
        if (bytes < kMin) {
            std::free(p);
            return;
        }
        std::lock_guard<std::mutex> lock(mutex());
        if (std::find(pinned().begin(), pinned().end(), p) == pinned().end()) {
            std::free(p);
            return;
        }
        cache()[bucket(bytes)].push_back(p);   // (kept until the process ends)
    }

private:
    static std::mutex& mutex() { static std::mutex m; return m; }
    static std::map<size_t, std::vector<void*>>& cache() { static auto* c = new std::map<size_t, std::vector<void*>>; return *c; }
    static std::vector<void*>& pinned() { static auto* v = new std::vector<void*>; return *v; }
};
template <class T>
struct PinnedAlloc {
    using value_type = T;
    PinnedAlloc() = default;
    template <class U> PinnedAlloc(const PinnedAlloc<U>&) {}
    T* allocate(size_t n) {
        void* p = PinnedPool::get(n * sizeof(T));
        if (!p) throw std::bad_alloc();
        return static_cast<T*>(p);
    }
    void deallocate(T* p, size_t n) { PinnedPool::put(p, n * sizeof(T)); }
    template <class U> void construct(U*) {}   // resize(): no clearing
    template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
    template <class U> bool operator==(const PinnedAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const PinnedAlloc<U>&) const { return false; }
};
template <class T> using PinVec = std::vector<T, PinnedAlloc<T>>;

struct Prepared {
    explicit Prepared(std::unique_ptr<RawBatch> r = std::make_unique<RawBatch>()) : raw_owner(std::move(r)), raw(*raw_owner) {}
    std::unique_ptr<RawBatch> raw_owner;
    RawBatch& raw;
    std::vector<Op> ops;                  // all records' ops, record r owns [rec_ops[r], rec_ops[r+1])
    std::vector<uint64_t> rec_ops;
    std::vector<uint64_t> rec_first;      // first result (window) index of record r; [recs] = total
    std::vector<uint64_t> rec_out;        // presence modes: byte offset of record r's line in the batch output; [recs] = total
    std::vector<uint64_t> run_start;      // device runs: first base (a multiple of 32: runs are word aligned in `packed`)
    std::vector<uint32_t> run_windows;
    PinVec<uint64_t> packed;
    PinVec<uint64_t> lower;               // -f with an even k: the lower-case bit of every packed base (case_run)
    uint64_t bases = 0, windows = 0;
    PinVec<uint8_t> bytes;                // device results, then (after the replay) the final 0/1 per window ...
    PinVec<uint64_t> bits;                // ... or, when nothing is replayed, one BIT per window straight from the device (FMSI_HIP_OUT_BITS)
    bool packed_bits = false;
    bool bit7 = false;                    // `bytes` after the replay: the printed bit in bit 7 on top of the strand nibble (replay_presence_team), else 0 / 1
    uint64_t replay_reruns = 0;
    PinVec<int64_t> orders;               // lookup: device results, then the final order per window
};

struct alignas(128) TeamPlan {   // what one thread of the team found in its records (a cache line pair of its own: the team's plans are neighbours in one array)
    std::vector<Op> ops;
    std::vector<uint32_t> rec_nops;
    std::vector<uint64_t> rec_windows, rec_symbols;
    std::vector<std::pair<const char*, uint64_t>> runs;   // (first base, length in bases >= k)
    uint64_t windows = 0, words = 0, dev_runs = 0;
};

constexpr uint64_t kMaxRunWindows = 0x40000000ull;   // run lengths are 32-bit in the C-ABI

// Walk of one record exactly as src/main.cpp:337-371 does it, recording instead of querying.
inline void plan_record(const char* sequence, int64_t sequence_length, int k, TeamPlan& tp) {
    uint32_t nops = 0;
    uint64_t windows_total = 0, symbols = 0;
    while (sequence_length > 0) {
        int64_t current_length = valid_prefix(sequence, sequence_length);
        if (current_length >= k) {
            const uint64_t windows = static_cast<uint64_t>(current_length - k + 1);
            tp.runs.emplace_back(sequence, static_cast<uint64_t>(current_length));
            tp.words += (static_cast<uint64_t>(current_length) + 31) / 32;
            tp.dev_runs += (windows + kMaxRunWindows - 1) / kMaxRunWindows;
            tp.ops.push_back({windows, true});
            ++nops;
            windows_total += windows;
            symbols += windows;
            // the reference consumes chunk_length - k + 1 characters per chunk: k - 1 remain
            sequence += windows;
            sequence_length -= static_cast<int64_t>(windows);
            current_length -= static_cast<int64_t>(windows);
        }
        sequence_length -= current_length + 1;
        sequence += current_length + 1;
        if (sequence_length >= 0) {
            const uint64_t fill = static_cast<uint64_t>(std::min<int64_t>(k, current_length + 1));
            tp.ops.push_back({fill, false});
            ++nops;
            symbols += fill;
        }
    }
    tp.rec_nops.push_back(nops);
    tp.rec_windows.push_back(windows_total);
    tp.rec_symbols.push_back(symbols);
    tp.windows += windows_total;
}

// records -> ops, device runs and the packed sequence, on `team` threads
void prepare_batch(Prepared& P, int k, unsigned team, bool with_case = false) {
    const size_t nrec = P.raw.recs.size();
    team = static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>(team, nrec / 64 + 1)));
    // contiguous record ranges of about equal sequence length
    std::vector<size_t> cut(team + 1, nrec);
    cut[0] = 0;
    {
        const uint64_t per = P.raw.seq_bytes / team + 1;
        uint64_t acc = 0;
        unsigned t = 1;
        for (size_t r = 0; r < nrec && t < team; ++r) {
            acc += P.raw.recs[r].seq_len;
            if (acc >= per * t) cut[t++] = r + 1;
        }
    }
    std::vector<TeamPlan> tp(team);
    parallel_for(team, [&](unsigned t) {
        for (size_t r = cut[t]; r < cut[t + 1]; ++r) {
            const RecRef& rr = P.raw.recs[r];
            plan_record(rr.seq, static_cast<int64_t>(rr.seq_len), k, tp[t]);
        }
    });
    // offsets of every thread's share
    std::vector<uint64_t> ops_at(team + 1, 0), win_at(team + 1, 0), word_at(team + 1, 0), run_at(team + 1, 0);
    for (unsigned t = 0; t < team; ++t) {
        ops_at[t + 1] = ops_at[t] + tp[t].ops.size();
        win_at[t + 1] = win_at[t] + tp[t].windows;
        word_at[t + 1] = word_at[t] + tp[t].words;
        run_at[t + 1] = run_at[t] + tp[t].dev_runs;
    }
    P.ops.resize(ops_at[team]);
    P.rec_ops.resize(nrec + 1);
    P.rec_first.resize(nrec + 1);
    P.rec_out.resize(nrec + 1);
    P.run_start.resize(run_at[team]);
    P.run_windows.resize(run_at[team]);
    P.packed.assign(word_at[team] + 2, 0);
    if (with_case) P.lower.assign(word_at[team] / 2 + 2, 0);
    P.windows = win_at[team];
    P.bases = word_at[team] * 32;
    // byte offsets of the records' output lines (presence modes): name + '\t' + symbols + '\n'
    std::vector<uint64_t> out_at(team + 1, 0);
    for (unsigned t = 0; t < team; ++t) {
        uint64_t b = 0;
        for (size_t r = cut[t]; r < cut[t + 1]; ++r) b += P.raw.recs[r].name_len + 2 + tp[t].rec_symbols[r - cut[t]];
        out_at[t + 1] = out_at[t] + b;
    }
    parallel_for(team, [&](unsigned t) {
        std::copy(tp[t].ops.begin(), tp[t].ops.end(), P.ops.begin() + static_cast<std::ptrdiff_t>(ops_at[t]));
        uint64_t o = ops_at[t], w = win_at[t], b = out_at[t];
        for (size_t r = cut[t]; r < cut[t + 1]; ++r) {
            P.rec_ops[r] = o;
            P.rec_first[r] = w;
            P.rec_out[r] = b;
            o += tp[t].rec_nops[r - cut[t]];
            w += tp[t].rec_windows[r - cut[t]];
            b += P.raw.recs[r].name_len + 2 + tp[t].rec_symbols[r - cut[t]];
        }
        uint64_t word = word_at[t], run = run_at[t];
        for (const auto& [first, len] : tp[t].runs) {
            pack_run(first, len, P.packed.data() + word);
            if (with_case) case_run(first, len, reinterpret_cast<uint32_t*>(P.lower.data()) + word);
            uint64_t windows = len - static_cast<uint64_t>(k) + 1, base = word * 32;
            while (windows) {
                const uint64_t take = std::min(windows, kMaxRunWindows);
                P.run_start[run] = base;
                P.run_windows[run] = static_cast<uint32_t>(take);
                ++run;
                base += take;
                windows -= take;
            }
            word += (len + 31) / 32;
        }
    });
    P.rec_ops[nrec] = ops_at[team];
    P.rec_first[nrec] = win_at[team];
    P.rec_out[nrec] = out_at[team];
}

// ---- predictor replay: device results of both strands -> what the reference prints ------------------

// query_kmers_single<mode> for one window given both strands' results (src/fms_index.h:263-331).
inline int64_t replay_single(StrandPredictor& p, Mode mode, bool orders, int64_t f, int64_t r) {
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
    return got;
}

// query_kmers_streaming<> for one chunk of `n` windows starting at result index w0 (src/fms_index.h:181-254); the
// final values replace the device results in place (element w0 + i <- window i of the chunk).
// The reference fills result[] in the orientation of the first strand (reversed when the strands are swapped), sums
// the first strand's results over all windows and the second strand's over the undecided ones, reverses result[] back
// and logs the two sums in forward / reverse order.  Per window that is: first = the first strand's result,
// decided = it settles the window, value = decided ? first : max(first, second) -- independent of the walk order,
// so one pass in window order gives the same sums and the same output.
void replay_streaming_chunk(StrandPredictor& p, Mode mode, bool orders, Prepared& P, uint64_t w0, uint64_t n) {
    const bool should_swap = p.predict_swap();
    const bool maximized_ones = mode == Mode::All;
    int first_sum = 0, second_sum = 0;
    if (orders) {
        // In-place is safe: element w0 + i <= 2 (w0 + i), and the pair of window i is read before element w0 + i is written
        // (for i = 0 the pair is read first; for i > 0 element w0 + i lies below the pair of window i).
        for (uint64_t i = 0; i < n; ++i) {
            const int64_t f = P.orders[2 * (w0 + i)], r = P.orders[2 * (w0 + i) + 1];
            const int64_t first = should_swap ? r : f, second = should_swap ? f : r;
            first_sum += first >= 0 ? 1 : -1;
            int64_t v = first;
            if (first < 0) {
                second_sum += second >= 0 ? 1 : -1;
                v = std::max(first, second);
            }
            P.orders[w0 + i] = v;
        }
    } else {
        uint8_t* b = P.bytes.data() + w0;
        for (uint64_t i = 0; i < n; ++i) {
            const int f = fwd_of(b[i]), r = rc_of(b[i]);
            const int first = should_swap ? r : f, second = should_swap ? f : r;
            const bool decided = first == 1 || (first == 0 && maximized_ones);
            first_sum += first;
            second_sum += decided ? 0 : second;
            b[i] = static_cast<uint8_t>((decided ? first : std::max(first, second)) == 1);
        }
    }
    if (should_swap) p.log_result(second_sum, first_sum);
    else p.log_result(first_sum, second_sum);
}

inline bool same_state(const StrandPredictor& a, const StrandPredictor& b) {
    return a.score == b.score && a.result_scores[0] == b.result_scores[0] && a.result_scores[1] == b.result_scores[1] && a.previous == b.previous;
}

// The same step for presence bytes as a table: what a window's pair of strand results (the STRICT byte's low nibble) does under either
// strand order -- the difference log_result() adds to the scores, the new `previous`, the character printed.  Filled from replay_single()
// itself.  The loop below is the predictor's recurrence and nothing else: two table reads that do not depend on the state, one select by
// predict_swap(), two clipped adds -- about 2 ns per window where the general loop (copy, compare, run search) takes 8-16: isolated
// k-mers have no runs of equal inputs to skip, and a file of 10^8 of them is 10^8 sequential steps.
struct PredictorSteps {
    int32_t e[2][2][16];   // [mode all][swap][nibble] = (diff & 0xFF) | previous << 8 | out << 16
    PredictorSteps() {
        for (int all = 0; all < 2; ++all)
            for (int swap = 0; swap < 2; ++swap)
                for (int in = 0; in < 16; ++in) {
                    StrandPredictor p;
                    p.score = swap ? -1 : 1;   // (previous == -1: predict_swap() is score < 0, and no result score is touched)
                    const int64_t got = replay_single(p, all ? Mode::All : Mode::Or, false, fwd_of(static_cast<uint8_t>(in)), rc_of(static_cast<uint8_t>(in)));
                    const int diff = p.score - (swap ? -1 : 1);
                    e[all][swap][in] = (diff & 0xFF) | (p.previous << 8) | (static_cast<int32_t>(got == 1) << 16);
                }
    }
};
const PredictorSteps kPredictorSteps;

// WRITE: b[w] keeps its strand nibble and gets the character's bit on top (bit 7), so that a stretch can be walked again from another state
// RUNS: windows of reads -- runs of equal strand results are taken out of the loop (below); isolated k-mers have none, and the test for them
// alone doubled the step's cost there (4.5 -> 10 ns on the build container's core): their batches take the plain loop
template <bool WRITE, bool RUNS>
void replay_presence_tight(uint8_t* b, uint64_t n, StrandPredictor& p, Mode mode) {
    uint64_t w = 0;
    for (; w < n && p.previous == -1; ++w) {   // (the very first window: no result score is updated)
        const uint8_t in = b[w] & 15u;
        const uint8_t out = static_cast<uint8_t>(replay_single(p, mode, false, fwd_of(in), rc_of(in)) == 1);
        if (WRITE) b[w] = static_cast<uint8_t>(in | (out << 7));
    }
    const int32_t(*T)[16] = kPredictorSteps.e[mode == Mode::All ? 1 : 0];
    int score = p.score, rs0 = p.result_scores[0], rs1 = p.result_scores[1], prev = p.previous;
    for (; w < n; ++w) {
        const unsigned in = b[w] & 15u;
        const int32_t e0 = T[0][in], e1 = T[1][in];
        const int rsp = prev ? rs1 : rs0;
        const int sel = rsp ? rsp : score;
        const int32_t e = sel < 0 ? e1 : e0;
        const int diff = static_cast<int8_t>(e & 0xFF);
        const int score_was = score;
        score = StrandPredictor::clipped(score + diff);
        const int nrs = StrandPredictor::clipped(rsp + diff);
        const bool unchanged = score == score_was && nrs == rsp && prev == ((e >> 8) & 1);   // (a zero difference, or scores that are saturated)
        rs1 = prev ? nrs : rs1;
        rs0 = prev ? rs0 : nrs;
        prev = (e >> 8) & 1;
        const uint8_t done = static_cast<uint8_t>(in | ((e >> 16) << 7));
        if (WRITE) b[w] = done;
        // ... then every following window with the same pair of strand results leaves it unchanged, too, and prints the same: reads that lie
        // on one strand and absent reads are such runs (the general loop's skip: 0.05 s instead of 2.9 s per 10^9 windows of 10-kb reads)
        // (only runs of eight and more are taken out of the loop: among isolated k-mers equal neighbours come in twos and threes, and a
        // mispredicted branch per window costs more than the steps it saves)
        // (... and the test is ONE branch that is almost never taken there: whether the step changed the state is a coin toss on such input)
        if (RUNS && w + 9 <= n) {
            uint64_t x;
            std::memcpy(&x, b + w + 1, 8);
            const uint64_t differ = ((x & 0x0F0F0F0F0F0F0F0Full) ^ (0x0101010101010101ull * in)) | static_cast<uint64_t>(!unchanged);
            if (__builtin_expect(differ == 0, 0)) {
                const uint64_t run = run_end_nibble(b, w + 9, n, static_cast<uint8_t>(in));
                if (WRITE) std::memset(b + w + 1, done, run - (w + 1));
                w = run - 1;
            }
        }
    }
    p.score = score;
    p.result_scores[0] = rs0;
    p.result_scores[1] = rs1;
    p.previous = prev;
}

// ... and the batch cut into one stretch per thread.  The recurrence is sequential, but its state is four small saturating numbers: walked
// from ANY state, the scores of two runs over the same inputs close in on each other whenever one of them is clipped, and after a few
// hundred windows of ordinary input they are the same run.  So every thread but the first warms a predictor up over the kWarm windows in
// front of its stretch (from the zero state) and walks its stretch from there; afterwards the stretches are taken in order: where the
// state a stretch was started from IS the state its predecessor ended in, its results stand -- it was the sequential walk; where not
// (inputs that never move the scores, such as a long run of absent k-mers), the stretch is walked again from the true state, one thread,
// as without any of this.  Exact either way; `reruns` says how often the guess failed.
template <bool RUNS>
uint64_t replay_presence_team(uint8_t* b, uint64_t n, StrandPredictor& p, Mode mode, unsigned team) {
    constexpr uint64_t kWarm = 4096;
    const unsigned parts = static_cast<unsigned>(std::min<uint64_t>(team, n / (16 * kWarm)));
    if (parts < 2) {
        replay_presence_tight<true, RUNS>(b, n, p, mode);
        return 0;
    }
    std::vector<StrandPredictor> from(parts), to(parts);
    auto lo_of = [&](unsigned t) { return n / parts * t; };
    auto hi_of = [&](unsigned t) { return t + 1 == parts ? n : n / parts * (t + 1); };
    from[0] = p;
    parallel_for(parts, [&](unsigned t) {   // (the warm-ups read what the stretches in front of them will write: first all of the one, then the other)
        if (!t) return;
        // (round 6, as replay_orders_team: the warm-up reaches back until it has seen windows on EITHER strand -- absent windows move no score,
        // and a read of 10^4 windows that is not in the text is longer than kWarm: 217 of 240 guesses failed on the benchmark's reads)
        constexpr uint64_t kMaxBack = uint64_t(1) << 20;
        const uint64_t lo = lo_of(t), floor = lo > kMaxBack ? lo - kMaxBack : 0;
        uint64_t from_w = lo, seen_f = 0, seen_r = 0;
        while (from_w > floor && (lo - from_w < kWarm || seen_f < 64 || seen_r < 64)) {
            --from_w;
            const bool f_in = fwd_of(b[from_w] & 15u) == 1;
            seen_f += f_in;
            seen_r += !f_in && rc_of(b[from_w] & 15u) == 1;
        }
        StrandPredictor q;
        q.previous = 0;
        replay_presence_tight<false, RUNS>(b + from_w, lo - from_w, q, mode);
        from[t] = q;
    });
    parallel_for(parts, [&](unsigned t) {
        StrandPredictor q = from[t];
        replay_presence_tight<true, RUNS>(b + lo_of(t), hi_of(t) - lo_of(t), q, mode);
        to[t] = q;
    });
    uint64_t reruns = 0;
    StrandPredictor cur = to[0];
    for (unsigned t = 1; t < parts; ++t) {
        if (same_state(cur, from[t])) {
            cur = to[t];
        } else {
            replay_presence_tight<true, RUNS>(b + lo_of(t), hi_of(t) - lo_of(t), cur, mode);
            ++reruns;
        }
    }
    p = cur;
    return reruns;
}

// strict results (both strands per window) -> final per-window values, in the order the reference visits them
// lookup, a stretch of n windows: the pairs in[2 w], in[2 w + 1] -> out[w] (out may be `in` itself: a window's place is never behind its pair's;
// or null: only the predictor is walked)
void replay_orders_tight(const int64_t* in, int64_t* out, uint64_t n, StrandPredictor& predictor, Mode mode, bool wide_orders) {
    auto cls = [](int64_t x, int64_t y) {
        return (static_cast<int>(x) >= 0 ? 1 : 0) | (x >= 0 ? 2 : 0) | (static_cast<int>(y) >= 0 ? 4 : 0) | (y >= 0 ? 8 : 0);
    };
    uint64_t w = 0;
    while (w < n) {
        const int64_t f = in[2 * w], r = in[2 * w + 1];
        const StrandPredictor before = predictor;
        const int64_t got = replay_single(predictor, mode, true, f, r);
        if (out) out[w] = got;
        ++w;
        if (wide_orders || !same_state(before, predictor)) continue;
        const int c0 = cls(f, r);
        const bool swap = predictor.predict_swap();
        while (w < n && cls(in[2 * w], in[2 * w + 1]) == c0) {
            const int64_t first = swap ? in[2 * w + 1] : in[2 * w], second = swap ? in[2 * w] : in[2 * w + 1];
            if (out) out[w] = static_cast<int>(first) >= 0 ? first : second;
            ++w;
        }
    }
}

// ... and the batch by the team: stretches walked side by side from warmed-up states, exact (a stretch whose guessed state is not the one its
// predecessor ended in is walked again from that one); the results stand beside the pairs until all pairs are read, then move to o[0, n)
uint64_t replay_orders_team(int64_t* o, uint64_t n, StrandPredictor& p, Mode mode, unsigned team, bool wide_orders) {
    constexpr uint64_t kWarm = 4096, kMaxBack = uint64_t(1) << 20;
    const unsigned parts = static_cast<unsigned>(std::min<uint64_t>(team, n / (16 * kWarm)));
    if (parts < 2) {
        replay_orders_tight(o, o, n, p, mode, wide_orders);
        return 0;
    }
    static thread_local std::vector<int64_t> chosen;   // (the one thread that replays: batches come in order)
    if (chosen.size() < n) chosen.resize(n);
    int64_t* const c = chosen.data();
    std::vector<StrandPredictor> from(parts), to(parts);
    auto lo_of = [&](unsigned t) { return n / parts * t; };
    auto hi_of = [&](unsigned t) { return t + 1 == parts ? n : n / parts * (t + 1); };
    from[0] = p;
    parallel_for(parts, [&](unsigned t) {
        StrandPredictor q = from[0];
        if (t) {
            // (the warm-up reaches back until it has seen windows that move EVERY score: absent windows move none, a read on the forward strand
            // moves the score and result_scores[1], one on the other strand the score and result_scores[0] -- and a read is 10^4 windows)
            const uint64_t lo = lo_of(t), floor = lo > kMaxBack ? lo - kMaxBack : 0;
            uint64_t from_w = lo, seen_f = 0, seen_r = 0;
            while (from_w > floor && (lo - from_w < kWarm || seen_f < 64 || seen_r < 64)) {
                --from_w;
                const bool f_in = static_cast<int>(o[2 * from_w]) >= 0;
                seen_f += f_in;
                seen_r += !f_in && o[2 * from_w + 1] >= 0;
            }
            q = StrandPredictor();
            q.previous = 0;
            replay_orders_tight(o + 2 * from_w, nullptr, lo - from_w, q, mode, wide_orders);
            from[t] = q;
        }
        replay_orders_tight(o + 2 * lo_of(t), c + lo_of(t), hi_of(t) - lo_of(t), q, mode, wide_orders);
        to[t] = q;
    });
    uint64_t reruns = 0;
    StrandPredictor cur = to[0];
    for (unsigned t = 1; t < parts; ++t) {
        if (same_state(cur, from[t])) {
            cur = to[t];
        } else {
            replay_orders_tight(o + 2 * lo_of(t), c + lo_of(t), hi_of(t) - lo_of(t), cur, mode, wide_orders);
            ++reruns;
        }
    }
    p = cur;
    parallel_for(parts, [&](unsigned t) { std::memcpy(o + lo_of(t), c + lo_of(t), (hi_of(t) - lo_of(t)) * sizeof(int64_t)); });
    return reruns;
}

void replay_batch(Prepared& P, StrandPredictor& predictor, Mode mode, bool orders, bool streaming, int k, unsigned team, bool wide_orders = false) {
    if (!streaming) {
        // One predictor step per window, in output order == result order; fills do not touch the predictor.  The
        // predictor is a small saturating state machine: when a step leaves it unchanged, every following window with
        // the same input (the same pair of strand results; for lookup the same pair of signs) leaves it unchanged too
        // and takes the same decision -- such runs (reads that lie on one strand, absent reads) are skipped over.
        const uint64_t nw = P.windows;
        if (!orders) {   // (the recurrence as a table, runs skipped, stretches walked side by side by the team: replay_presence_team)
            P.replay_reruns += P.raw.kmer_k ? replay_presence_team<false>(P.bytes.data(), nw, predictor, mode, team)
                                            : replay_presence_team<true>(P.bytes.data(), nw, predictor, mode, team);
            P.bit7 = true;
            return;
        }
        // lookup.  What a window prints is its first strand's order if that strand has the k-mer, else the second's
        // (src/fms_index.h:281-287: the first strand's order is looked at as an int -- an order of 2^31 and more counts as absent
        // there -- the second's sign as it is), and what the predictor is told are those signs.  A first strand that is negative
        // as an int goes into the predictor with its value (fms_index.h:283-291); any value below -14 saturates the scores alike,
        // and orders below 2^32 - 128 (every index with 32-bit rows) are such values, so there the four signs stand for the
        // window; an index of 2^32 rows and more takes every step with the values themselves.
        // (Measured and dropped, profiles/README.md r02n: signs by the team, predictor over the sign bytes, choice by the team --
        // three passes over 16 bytes per window cost more than this one: 2.0 against 1.0 s per 10^9 windows at 1.4 G rows, 3.9
        // against 2.9 s at 2.8 G rows, where every other present k-mer's order is above 2^31 and equal neighbours are rare.)
        // (Round 6: the walk is cut into one stretch per thread, each started from a predictor warmed up over the windows in front of it
        // and kept where that state IS the one its predecessor ended in -- replay_presence_team's scheme, here with the pairs read from
        // the batch and the chosen orders written beside them until every stretch has read its pairs: `fmsi lookup` on 997 M windows
        // spent 2.9 s of its 3.7 s in this one-thread loop.)
        P.replay_reruns += replay_orders_team(P.orders.data(), nw, predictor, mode, team, wide_orders);
        return;
    }
    // -S: the predictor steps once per CHUNK (src/fms_index.h:181-254), on the sums of the chunk's strand results.
    // Presence modes, in two phases: (A) the team takes the chunks in parallel -- what the predictor will be told for either
    // strand order (the two pairs of sums), and the chunk's output bytes, which do not depend on the order unless a window
    // has one strand 0 and the other 1 under -O (then the first strand searched decides: such chunks are left for B);
    // (B) one pass over the chunks with the predictor itself.  Same sums, same decisions, same bytes as the walk chunk by
    // chunk; `lookup -S` keeps that walk.
    struct Chunk {
        uint64_t w0;
        uint32_t n;
        int16_t sums[2][2];   // [swap][first, second]
        bool order_matters;
    };
    // the chunks of run `o` of record r as src/main.cpp:340-354 cuts them
    auto for_chunks_of = [&](size_t r, auto&& fn) {
        const uint64_t L = P.raw.recs[r].seq_len;
        const int64_t max_chunk = k + std::max<int64_t>(10, std::min<int64_t>(400, 2 * static_cast<int64_t>(std::sqrt(static_cast<int64_t>(L)))));   // src/main.cpp:330-331
        uint64_t w = P.rec_first[r];
        for (uint64_t o = P.rec_ops[r]; o < P.rec_ops[r + 1]; ++o) {
            if (!P.ops[o].is_run) continue;
            int64_t current_length = static_cast<int64_t>(P.ops[o].count) + k - 1;
            while (current_length >= k) {
                const int64_t chunk_length = std::min<int64_t>(current_length, max_chunk);
                const uint64_t n = static_cast<uint64_t>(chunk_length - k + 1);
                fn(w, n);
                w += n;
                current_length -= static_cast<int64_t>(n);
            }
        }
    };
    const size_t nrec = P.rec_ops.empty() ? 0 : P.rec_ops.size() - 1;
    if (orders) {
        for (size_t r = 0; r < nrec; ++r) for_chunks_of(r, [&](uint64_t w, uint64_t n) { replay_streaming_chunk(predictor, mode, orders, P, w, n); });
        return;
    }
    // (A) by the team: the records' chunks listed -- a read of 150 bases is five of them, a batch of such reads millions: one thread's list
    // was 0.2 s per 7 x 10^8 windows --, then their sums and output bytes
    const bool maximized_ones = mode == Mode::All;
    // (a share is records: enough of them for the lists not to be crumbs -- 4096 short reads -- or enough windows, 2^20: a batch of 10-kb reads is
    // 6 700 records and 6 x 10^7 windows, and by the records' count alone two threads did the team's work, 0.2 s per 10^9 windows)
    const unsigned listers = static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>({static_cast<size_t>(team), nrec, std::max<size_t>(nrec / 4096, P.windows >> 20) + 1})));
    std::vector<std::vector<Chunk>> listed(listers);
    std::vector<char> any_matters(listers, 0);
    parallel_for(listers, [&](unsigned t) {
        std::vector<Chunk> mine;   // (local until done: the lists' headers are neighbours)
        const size_t lo = nrec / listers * t, hi = t + 1 == listers ? nrec : nrec / listers * (t + 1);
        mine.reserve((hi - lo) * 2 + 16);
        bool matters = false;
        for (size_t r = lo; r < hi; ++r)
            for_chunks_of(r, [&](uint64_t w, uint64_t n) {
                Chunk ch{w, static_cast<uint32_t>(n), {{0, 0}, {0, 0}}, false};
                uint8_t* b = P.bytes.data() + w;
                int sums[2][2];
                ch.order_matters = chunk_sums(b, ch.n, maximized_ones, sums);
                for (int sw = 0; sw < 2; ++sw)
                    for (int j = 0; j < 2; ++j) ch.sums[sw][j] = static_cast<int16_t>(sums[sw][j]);
                if (!ch.order_matters) chunk_output(b, ch.n, maximized_ones);
                matters |= ch.order_matters;
                mine.push_back(ch);
            });
        any_matters[t] = matters ? 1 : 0;
        listed[t] = std::move(mine);
    });
    auto step = [](StrandPredictor& p, const Chunk& ch) {
        const bool swap = p.predict_swap();
        const int first_sum = ch.sums[swap][0], second_sum = ch.sums[swap][1];
        if (swap) p.log_result(second_sum, first_sum);
        else p.log_result(first_sum, second_sum);
    };
    // (B) the predictor over the chunks, in order.  A chunk whose output depends on the strand order (non-max-ones masks under -O) needs
    // the predictor's decision at that point: with one of those in the batch it is the plain walk.
    if (std::find(any_matters.begin(), any_matters.end(), 1) != any_matters.end()) {
        for (auto& list : listed)
            for (Chunk& ch : list) {
                if (ch.order_matters) replay_streaming_chunk(predictor, mode, false, P, ch.w0, ch.n);
                else step(predictor, ch);
            }
        return;
    }
    // Otherwise all that is left is the state the predictor ends in -- and that is replay_presence_team's situation: every list but the
    // first is walked from a state warmed up over the 4096 chunks in front of it; where that state IS the one the list before ended in, the
    // walk stands, where not it is done again from the true state.  (7 x 10^8 windows of 150-base reads under -S: 3 x 10^7 chunks, 0.3 s of
    // one thread.)
    constexpr size_t kWarm = 4096;
    std::vector<StrandPredictor> from(listers), to(listers);
    from[0] = predictor;
    parallel_for(listers, [&](unsigned t) {
        StrandPredictor q = from[0];
        if (t) {
            q = StrandPredictor();
            q.previous = 0;
            size_t need = kWarm;   // the last kWarm chunks in front of list t (they may lie in several lists)
            unsigned first_list = t;
            size_t first_at = 0;
            while (need && first_list > 0) {
                --first_list;
                const size_t have = listed[first_list].size();
                first_at = have > need ? have - need : 0;
                need -= have - first_at;
            }
            for (unsigned l = first_list; l < t; ++l)
                for (size_t c = l == first_list ? first_at : 0; c < listed[l].size(); ++c) step(q, listed[l][c]);
            from[t] = q;
        }
        for (const Chunk& ch : listed[t]) step(q, ch);
        to[t] = q;
    });
    StrandPredictor cur = to[0];
    for (unsigned t = 1; t < listers; ++t) {
        if (same_state(cur, from[t])) {
            cur = to[t];
        } else {
            for (const Chunk& ch : listed[t]) step(cur, ch);
        }
    }
    predictor = cur;
}

// ---- formatting ----------------------------------------------------------------------------------
// presence modes: every line has a known length, the team writes straight into one buffer
void format_bits(const Prepared& P, unsigned team, char* out) {   // out: P.rec_out[records] bytes
    const size_t nrec = P.raw.recs.size();
    team = static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>(team, nrec / 64 + 1)));
    parallel_for(team, [&](unsigned t) {
        // ranges of about equal output size
        const uint64_t lo = P.rec_out[nrec] / team * t, hi = t + 1 == team ? P.rec_out[nrec] : P.rec_out[nrec] / team * (t + 1);
        size_t r = static_cast<size_t>(std::lower_bound(P.rec_out.begin(), P.rec_out.begin() + static_cast<std::ptrdiff_t>(nrec), lo) - P.rec_out.begin());
        for (; r < nrec && P.rec_out[r] < hi; ++r) {
            char* dst = out + P.rec_out[r];
            const RecRef& rr = P.raw.recs[r];
            std::memcpy(dst, rr.name, rr.name_len);
            dst += rr.name_len;
            *dst++ = '\t';
            uint64_t w = P.rec_first[r];
            for (uint64_t o = P.rec_ops[r]; o < P.rec_ops[r + 1]; ++o) {
                const uint64_t c = P.ops[o].count;
                if (P.ops[o].is_run) {
                    if (P.packed_bits) {
                        expand_bits(P.bits.data(), w, c, dst);
                    } else {
                        const uint8_t* src = P.bytes.data() + w;
                        if (P.bit7) for (uint64_t i = 0; i < c; ++i) dst[i] = static_cast<char>('0' + (src[i] >> 7));
                        else for (uint64_t i = 0; i < c; ++i) dst[i] = static_cast<char>('0' + (src[i] != 0));
                    }
                    w += c;
                } else {
                    std::memset(dst, '0', c);
                }
                dst += c;
            }
            *dst = '\n';
        }
    });
}

// lookup: comma separated integers (src/main.cpp:341-342,361-365); every thread formats a range of records.
// Round 6: straight into a buffer with room for the worst case (no string that checks its capacity per number, no std::to_chars into a
// temporary): an order is at most ten digits below 2^32 rows -- its digit count from a comparison ladder, the digits two at a time from a table,
// written backwards into place.  997 M orders (6.3 GB of text) were 30-50 core-seconds of the team, the longest item of `fmsi lookup`.
struct OrderText {
    PgzBuf buf;       // (pgunzip.h's buffer: not zero-filled, huge pages)
    size_t len = 0;
};
inline char* put_order(char* p, int64_t v) {
    static const char kPairs[] =
        "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
    if (v < 0) {   // (-1, or whatever an index of 2^32 rows and more makes of a negative int)
        *p++ = '-';
        v = -v;
    }
    uint64_t u = static_cast<uint64_t>(v);
    unsigned nd = 1;
    if (u >= 10) {
        if (u < 100) nd = 2;
        else if (u < 1000) nd = 3;
        else if (u < 10000) nd = 4;
        else if (u < 100000) nd = 5;
        else if (u < 1000000) nd = 6;
        else if (u < 10000000) nd = 7;
        else if (u < 100000000) nd = 8;
        else if (u < 1000000000ull) nd = 9;
        else {
            nd = 10;
            for (uint64_t t = 10000000000ull; u >= t && nd < 20; t *= 10) ++nd;
        }
    }
    char* const end = p + nd;
    char* q = end;
    while (u >= 100) {
        const unsigned r = static_cast<unsigned>(u % 100);
        u /= 100;
        q -= 2;
        std::memcpy(q, kPairs + 2 * r, 2);
    }
    if (u >= 10) std::memcpy(q - 2, kPairs + 2 * u, 2);
    else q[-1] = static_cast<char>('0' + u);
    return end;
}
void format_orders(const Prepared& P, unsigned team, std::vector<OrderText>& parts, bool wide_orders) {
    const size_t nrec = P.raw.recs.size();
    team = static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>(team, nrec / 64 + 1)));
    parts.resize(team);   // (the buffers keep their room from batch to batch)
    const uint64_t total = P.rec_first[nrec] + nrec;   // balance by windows (+1 per record)
    (void)wide_orders;
    const size_t per_order = 21;   // the longest int64 and the comma (room that is not written is not touched: the buffer is not cleared)
    parallel_for(team, [&](unsigned t) {
        OrderText& ot = parts[t];
        const uint64_t lo = total / team * t, hi = t + 1 == team ? total + 1 : total / team * (t + 1);
        // room for the worst case of this share: every window an order of full length, every filled window "-1,", the names and the record ends
        size_t room = 64;
        size_t r0 = nrec, r1 = nrec;
        for (size_t r = 0; r < nrec; ++r) {
            const uint64_t key = P.rec_first[r] + r;
            if (key < lo) continue;
            if (key >= hi) {
                r1 = r;
                break;
            }
            if (r0 == nrec) r0 = r;
            uint64_t windows = 0;
            for (uint64_t o = P.rec_ops[r]; o < P.rec_ops[r + 1]; ++o) windows += P.ops[o].count;
            room += P.raw.recs[r].name_len + 2 + windows * per_order;
        }
        if (r0 == nrec) r1 = nrec;
        ot.buf.grow(room, 0);
        char* const base = reinterpret_cast<char*>(ot.buf.data());
        char* p = base;
        for (size_t r = r0; r < r1; ++r) {
            const RecRef& rr = P.raw.recs[r];
            std::memcpy(p, rr.name, rr.name_len);
            p += rr.name_len;
            *p++ = '\t';
            bool first = true;
            uint64_t w = P.rec_first[r];
            for (uint64_t o = P.rec_ops[r]; o < P.rec_ops[r + 1]; ++o) {
                const uint64_t count = P.ops[o].count;
                if (P.ops[o].is_run) {
                    const int64_t* v = P.orders.data() + w;
                    for (uint64_t i = 0; i < count; ++i) {
                        if (!first) *p++ = ',';
                        first = false;
                        p = put_order(p, v[i]);
                    }
                    w += count;
                } else {
                    for (uint64_t i = 0; i < count; ++i) {
                        if (!first) *p++ = ',';
                        first = false;
                        *p++ = '-';
                        *p++ = '1';
                    }
                }
            }
            *p++ = '\n';
        }
        ot.len = static_cast<size_t>(p - base);
    });
}

// a k-mer batch (cli_reader.h: records of one k-mer each, their lines written by the parser): the answers go into the lines, one character
// per record, slices of 2^16 records dealt out to the team
void patch_kmer_lines(Prepared& P, unsigned team) {
    struct Slice {
        KmerPart* part;
        size_t lo, hi;
        uint64_t first;   // result index of the part's record `lo`
    };
    std::vector<Slice> slices;
    uint64_t first = 0;
    for (auto& kp : P.raw.kparts) {
        const size_t n = kp->n;
        for (size_t lo = 0; lo < n; lo += size_t(1) << 16) slices.push_back({kp.get(), lo, std::min(n, lo + (size_t(1) << 16)), first + lo});
        first += n;
    }
    if (slices.empty()) return;
    team = static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>(team, slices.size())));
    std::atomic<size_t> next{0};
    parallel_for(team, [&](unsigned) {
        for (size_t t; (t = next.fetch_add(1, std::memory_order_relaxed)) < slices.size();) {
            const Slice& sl = slices[t];
            char* text = sl.part->text.get();
            const uint32_t* at = sl.part->at.get();
            if (P.packed_bits) {
                const uint64_t* bits = P.bits.data();
                for (size_t i = sl.lo; i < sl.hi; ++i) {
                    const uint64_t w = sl.first + (i - sl.lo);
                    text[at[i]] = static_cast<char>('0' + ((bits[w >> 6] >> (w & 63)) & 1u));
                }
            } else {
                const uint8_t* b = P.bytes.data() + sl.first;
                if (P.bit7) for (size_t i = sl.lo; i < sl.hi; ++i) text[at[i]] = static_cast<char>('0' + (b[i - sl.lo] >> 7));   // (replay_presence_team's bit)
                else for (size_t i = sl.lo; i < sl.hi; ++i) text[at[i]] = static_cast<char>('0' + (b[i - sl.lo] != 0));
            }
        }
    });
}

void check(int rc, const char* what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + fmsi_hip_last_error());
}

}  // namespace
}  // namespace fmsi

using namespace fmsi;

namespace {
void (*g_exit_hook)(int) = nullptr;
}
extern "C" void fmsi_cli_set_exit_hook(void (*hook)(int)) { g_exit_hook = hook; }

namespace {
// stdout is a regular file: the pages of its first gigabytes are allocated WHILE THE INDEX LOADS -- fallocate(KEEP_SIZE): beyond the end of
// the file, its size untouched -- by a thread that stops the moment the sink writes its first byte.  One thread's write() into a tmpfs file
// is the longest stage of this program (0.15-0.18 s per GB of output: 1 GB per 10^9 windows), and 0.065 s of that is the allocation of the
// file's pages, which costs the same when nobody is waiting for it: write() into pages that stand takes 0.085 s per GB (tools/micro/
// write_bench.cpp; beside a writer that is writing the same call makes both slower -- the inode's lock -- hence the stop).  What was allocated
// and not written is given back at the end.  FMSI_HIP_NO_OUT_PREALLOC: off.
// How many bytes will a presence query print for this file?  From its first megabyte: per record the name, a tab, a character per window, a
// newline -- 0.997 of the file for reads of 10 kb, a quarter of it for records of one 31-mer.  It has to be about right: pages allocated and
// not written are given back at the end, where somebody waits for it (a k-mer file taken for reads: 2 GB of them, + 0.1 s).
uint64_t estimate_output_bytes(const std::string& path, uint64_t file_bytes, int k, unsigned bytes_per_window = 1) {
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return 0;
    std::vector<char> buf(size_t(1) << 20);
    ssize_t got = pread(fd, buf.data(), buf.size(), 0);
    close(fd);
    if (got >= 2 && static_cast<unsigned char>(buf[0]) == 0x1f && static_cast<unsigned char>(buf[1]) == 0x8b) {
        // a gzip / block-gzip file: its first four megabytes of text through zlib (a few milliseconds), and the file's text taken to be as many
        // times its size as those were of the compressed bytes they came from
        gzFile g = gzopen(path.c_str(), "rb");
        if (!g) return 0;
        (void)gzbuffer(g, 1u << 16);
        buf.resize(size_t(4) << 20);
        const int n = gzread(g, buf.data(), static_cast<unsigned>(buf.size()));
        const z_off_t used = gzoffset(g);
        (void)gzclose(g);
        if (n <= 0 || used <= 0) return 0;
        got = n;
        if (static_cast<uint64_t>(used) < file_bytes) file_bytes = static_cast<uint64_t>(static_cast<double>(file_bytes) * static_cast<double>(n) / static_cast<double>(used));
        else file_bytes = static_cast<uint64_t>(n);   // (the whole file was those four megabytes, or less)
    }
    if (got < 2 || (buf[0] != '>' && buf[0] != '@')) return 0;
    const bool fastq = buf[0] == '@';
    const char* p = buf.data();
    const char* const end = p + got;
    uint64_t in = 0, out = 0;
    while (p < end) {
        const char* const rec = p;
        const char* nl = static_cast<const char*>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
        if (!nl) break;
        size_t name = 0;
        for (const char* q = p + 1; q < nl && !std::isspace(static_cast<unsigned char>(*q)); ++q) ++name;
        p = nl + 1;
        uint64_t bases = 0;
        bool whole = false;
        if (fastq) {   // sequence, '+', quality: one line each
            const char* a = static_cast<const char*>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
            const char* b = a ? static_cast<const char*>(std::memchr(a + 1, '\n', static_cast<size_t>(end - a - 1))) : nullptr;
            const char* c = b ? static_cast<const char*>(std::memchr(b + 1, '\n', static_cast<size_t>(end - b - 1))) : nullptr;
            if (c) {
                bases = static_cast<uint64_t>(a - p);
                p = c + 1;
                whole = true;
            }
        } else {
            while (p < end && *p != '>') {
                const char* a = static_cast<const char*>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
                if (!a) break;
                bases += static_cast<uint64_t>(a - p);
                p = a + 1;
            }
            whole = p < end || static_cast<uint64_t>(got) == file_bytes;
        }
        if (!whole) break;
        in += static_cast<uint64_t>(p - rec);
        out += name + 2 + (bases >= static_cast<uint64_t>(k) ? bases - static_cast<uint64_t>(k) + 1 : 0) * bytes_per_window;
    }
    if (in == 0) return 0;
    return static_cast<uint64_t>(static_cast<double>(file_bytes) * static_cast<double>(out) / static_cast<double>(in) * 1.01) + (uint64_t(1) << 20);
}

class OutPrealloc {
public:
    ~OutPrealloc() { finish(); }
    void start(std::function<uint64_t()> estimate) {   // (the estimate is made by the thread itself: a compressed file's costs a few milliseconds of zlib)
        if (std::getenv("FMSI_HIP_NO_OUT_PREALLOC")) return;
        struct stat st;
        if (fstat(STDOUT_FILENO, &st) != 0 || !S_ISREG(st.st_mode)) return;
        const int fl = fcntl(STDOUT_FILENO, F_GETFL);
        if (fl < 0 || (fl & O_APPEND)) return;   // (`>>`: somebody else may append to that file, too, and the end of it is not this program's to trim)
        const off_t at = lseek(STDOUT_FILENO, 0, SEEK_CUR);
        if (at < 0 || at < st.st_size) return;   // (writing into the middle of a file that is there: its pages are, too)
        base_ = at;
        end_.store(at, std::memory_order_relaxed);
        active_ = true;
        th_ = std::thread([this, estimate] {
            const uint64_t want = std::min<uint64_t>(estimate(), kMost);
            for (uint64_t off = 0; off < want && !stop_.load(std::memory_order_acquire); off += kPiece) {
                const off_t len = static_cast<off_t>(std::min<uint64_t>(kPiece, want - off));
                if (fallocate(STDOUT_FILENO, FALLOC_FL_KEEP_SIZE, base_ + static_cast<off_t>(off), len) != 0) break;   // (a file system without it, no room: nothing lost)
                end_.store(base_ + static_cast<off_t>(off) + len, std::memory_order_release);
            }
        });
    }
    void writing_starts() { stop_.store(true, std::memory_order_release); }
    void finish() {   // behind the last write (or on any way out): what stands beyond the end of the output goes back
        if (!active_) return;
        active_ = false;
        stop_.store(true, std::memory_order_release);
        if (th_.joinable()) th_.join();
        struct stat st;
        off_t pos = lseek(STDOUT_FILENO, 0, SEEK_CUR);
        if (fstat(STDOUT_FILENO, &st) == 0) pos = std::max(pos, st.st_size);
        const off_t end = end_.load(std::memory_order_acquire);
        if (pos >= 0 && end > pos) {
            // (tmpfs frees pages beyond the end by the hole; ext4 ignores a hole out there and frees them when the file is cut to the size it has)
            (void)fallocate(STDOUT_FILENO, FALLOC_FL_PUNCH_HOLE | FALLOC_FL_KEEP_SIZE, pos, end - pos);
            (void)!ftruncate(STDOUT_FILENO, pos);
        }
    }

private:
    static constexpr uint64_t kPiece = 32ull << 20, kMost = 3ull << 30;   // (what a load of 0.2 s has the time for)
    std::thread th_;
    std::atomic<bool> stop_{false};
    std::atomic<off_t> end_{0};
    off_t base_ = 0;
    bool active_ = false;
};
}  // namespace

extern "C" int fmsi_cli_query(int argc, char** argv, int output_orders_i) {
    // the command is a chain of builder calls in a process of its own: they share the library's block cache (include/fmsi_hip.h)
    struct KeepScratch {
        KeepScratch() { fmsi_hip_keep_scratch(1); }
        ~KeepScratch() { fmsi_hip_keep_scratch(0); }
    } keep_scratch_;
    host_kernels_init();          // (every environment switch of the host side is read here, before a second thread exists)
    (void)team_spawns_threads();
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
                                       {"strict", no_argument, nullptr, 1004},       {"threads", required_argument, nullptr, 1005},
                                       {"tables", required_argument, nullptr, 1006}, {"devices", required_argument, nullptr, 1007},
                                       {nullptr, 0, nullptr, 0}};
    optind = 1;
    int c;
    while ((c = getopt_long(argc, argv, "f:hq:k:OS", long_opts, nullptr)) >= 0) {
        switch (c) {
        case 'f': {   // src/main.cpp:255-263
            f_name = optarg;
            opt.fn = 0;
            if (parse_function(f_name, opt) < 0) {
                std::cerr << "ERROR: Function '" << optarg << "' not recognized." << std::endl;
                return usage_query(false);   // (the reference prints the query usage here, also for lookup)
            }
            break;
        }
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
        case 1005: opt.threads = static_cast<unsigned>(std::max(1, atoi(optarg))); break;
        case 1006:
            if (std::string(optarg) == "full") opt.full_tables = opt.tables_given = true;
            else if (std::string(optarg) == "compact") opt.full_tables = opt.compact_tables = opt.tables_given = true;
            else if (std::string(optarg) == "background") opt.background_dict = opt.tables_given = true;
            else if (std::string(optarg) == "candidates") opt.full_tables = opt.candidate_tables = opt.tables_given = true;
            else if (std::string(optarg) != "auto") return usage_query(output_orders);
            break;
        case 1007: {
            opt.devices.clear();
            std::string list = optarg;
            size_t pos = 0;
            while (pos <= list.size()) {
                const size_t comma = std::min(list.find(',', pos), list.size());
                const std::string item = list.substr(pos, comma - pos);
                if (item.empty() || item.find_first_not_of("0123456789") != std::string::npos) return usage_query(output_orders);
                opt.devices.push_back(atoi(item.c_str()));
                pos = comma + 1;
            }
            break;
        }
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
    } else if (f_name != "or" && output_orders) {   // src/main.cpp:297-300
        std::cerr << "ERROR: FMSI as Minimum Perfect Hash Function is not allowed with f-masked superstrings in the current version." << std::endl;
        return usage_query(output_orders);
    }
    opt.mode = f_name == "all" ? Mode::All : Mode::Or;
    // -f and|xor|r-s: query_mode::general (src/fms_index.h:316-327).  f(ones, total) over both strands per window: the strand
    // predictor is never told anything and -S changes nothing (:336), so no strand is replayed and the kLCP array stays on disk --
    // though -S still asks for its file, as load_index() does.
    const bool general = opt.fn != 0;
    const bool load_klcp = opt.has_klcp && !general;

    using clock = std::chrono::steady_clock;
    auto t_start = clock::now();
    const double epoch_start = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
    if (opt.devices.empty()) opt.devices.push_back(opt.device);
    const unsigned team = opt.threads ? opt.threads : std::min(16u, available_cpus());

    // ---- the reader starts first: it needs the query file only, so parsing the first batches overlaps the index load ----------
    // (only when the file is there: a missing query file must fail where the reference's OpenFile() does -- after the index
    // checks, uncaught, src/parser.h:22 -- so that case keeps the old order)
    double gpu_s = 0, prep_s = 0, replay_s = 0, format_s = 0, write_s = 0;
    uint64_t replay_reruns = 0;   // (k-mer batches: stretches of the predictor's replay that had to be walked again, replay_presence_team)
    BoundedQueue<std::unique_ptr<RawBatch>> raw_q(2);
    BoundedQueue<std::unique_ptr<Prepared>> out_q(2);
    std::exception_ptr writer_err;
    OutPrealloc out_prealloc;   // (declared before the reader: whatever ends this function gives the unwritten pages back)
    QueryReader query_reader(opt.query_fn, opt.batch_windows, team, raw_q);
    auto start_reader = [&] { query_reader.start(); };
    auto stop_reader = [&] {   // every way out below: the reader must not stay blocked on a full queue
        raw_q.abort();
        query_reader.join();
    };

    // the index: its scalars from the files' headers now (the checks of ms_query, src/main.cpp:302-320), its payload straight into
    // HBM below (fmsi_hip_index_load: RRR decode and block layout on the device)
    uint64_t index_rows = 0;
    int32_t index_k = 0, klcp_file = 0;
    const int rc_load = fmsi_host_index_peek(opt.index_prefix.c_str(), &index_rows, &index_k, &klcp_file);
    if (rc_load != 0 || index_rows == 0) {
        std::cerr << "ERROR: index not correctly loaded. Ensure that you correctly call `fmsi index` before." << std::endl;
        stop_reader();
        return usage_query(output_orders);
    }
    if (opt.has_klcp && !klcp_file) {
        std::cerr << "ERROR: kLCP array was not constructed for the given index. Either construct it again without the `-s` flag or use `query -s` which slows down streaming queries." << std::endl;
        stop_reader();
        return usage_query(output_orders);
    }
    int k = opt.k;
    if (k != 0 && k != index_k) {
        std::cerr << "ERROR: Mismatch. Provided k (" << k << ") does not match the k of the index (" << index_k << ")." << std::endl;
        stop_reader();
        return usage_query(output_orders);
    }
    if (k == 0) k = index_k;
    // (the header checks above are a few file reads; the reader starts here, before the index's payload is touched)
    // Records of exactly one k-mer -- the query files of BASELINE's isolated-k-mer configurations -- take a path of their own from the
    // parser on (cli_reader.h: KmerPart; fmsi_hip_query_kmers): presence modes, k <= 32.  FMSI_HIP_CLI_KMER_BATCHES=0: the general path (A/B runs).
    {
        static const char* kmer_env = std::getenv("FMSI_HIP_CLI_KMER_BATCHES");
        if (!output_orders && !general && k >= 1 && k <= 32 && !(kmer_env && kmer_env[0] == '0')) query_reader.allow_kmer_batches(k);
        struct stat st;
        if (opt.query_fn == "-" || stat(opt.query_fn.c_str(), &st) == 0) {
            start_reader();
            // (`lookup` prints an order and a comma per window: about six characters on a file of which half is in the text; what the estimate
            // is short of is simply not allocated early, what it is over is given back)
            if (opt.query_fn != "-" && S_ISREG(st.st_mode))
                out_prealloc.start([path = opt.query_fn, bytes = static_cast<uint64_t>(st.st_size), k, per = output_orders ? 6u : 1u] { return estimate_output_bytes(path, bytes, k, per); });
        }
    }

    // private accelerators (results unchanged).  How much of them: end to end this program moves 3-4 G windows/s at human scale
    // with a 14-deep prefix table -- the device stage (LF walk at about 5 G k-mers/s) is then the longest, just ahead of write()
    // -- and 5-7 G windows/s with the k-mer dictionary (45 G k-mers/s on resident batches: bench.py), which takes about half a
    // second more to build at that scale.  So the dictionary is built when asked for (--tables full) or when the
    // query file is so large (16 GB; uncompressed bytes ~ windows) that the difference pays for the build several times over.
    // Round 3: that dictionary is built BESIDE the queries -- a second handle in a thread of its own, the first batches answered by the LF
    // walk (or the kLCP kernel), every batch from the moment the dictionary stands by the dictionary: same bytes (the dictionary holds the
    // index's own answers), no half second of load.  It pays from some 8 GB of queries on (the build is half a second of device time the
    // batches queue behind, and the host's write() of one byte per window caps the program at about 5 G windows/s either way: DESIGN.md 8),
    // so that is where `--tables auto` starts it; `--tables background` always does, `--tables full` builds it before the first batch.
    if (!opt.tables_given && opt.query_fn != "-") {
        struct stat st;
        const bool gz = opt.query_fn.size() > 3 && opt.query_fn.compare(opt.query_fn.size() - 3, 3, ".gz") == 0;
        if (stat(opt.query_fn.c_str(), &st) == 0 && S_ISREG(st.st_mode) &&
            static_cast<uint64_t>(st.st_size) * (gz ? 4u : 1u) >= 8000000000ull)
            opt.background_dict = true;
    }
    // `full`: presence queries (also -S ones: their answers do not depend on the kLCP array) are answered by the k-mer
    // dictionary, which needs no prefix table (a small one serves the few searches of its construction); lookup keeps the
    // 16-deep table
    const bool dict_possible = !general && !opt.candidate_tables && !output_orders && k <= 32 && index_rows > static_cast<uint64_t>(k);
    const bool want_dict = dict_possible && opt.full_tables;
    const bool want_bg_dict = dict_possible && !opt.full_tables && opt.background_dict;
    int t = std::min(k, want_dict ? 12 : (opt.full_tables ? 16 : 14));
    if (const char* e = std::getenv("FMSI_HIP_CLI_TABLE_DEPTH")) t = std::min(k, std::max(0, std::atoi(e)));   // (measurements)
    while (t > 0 && (1ull << (2 * t)) > 4 * index_rows) --t;
    // Which kernel walks the reads.  -S asks for the kLCP array; WITHOUT -S the reference searches every window afresh -- but the interval a
    // streamed step reaches is the interval the fresh search finds (src/fms_index.h:106-115), so when the index HAS its .klcp file the windows
    // can be streamed all the same: identical per-strand answers (the predictor's replay below stays the mode's own), one block read per window
    // of a read that follows the text instead of k - t LF steps (human scale, 997 M windows: device stage 0.21 -> 0.03 s for the default mode,
    // 0.27 -> 0.06 s for -O).  `--tables auto` does so; FMSI_HIP_CLI_STREAM=0 keeps the searches (A/B runs).
    static const char* stream_env = std::getenv("FMSI_HIP_CLI_STREAM");
    const bool stream_without_S = !opt.has_klcp && klcp_file && !general && !opt.tables_given && k <= 64 && !(stream_env && stream_env[0] == '0');
    bool stream_kernel = load_klcp || stream_without_S;   // (without -S: only as long as the kLCP file loads -- the reference would never open it)
    // ... and the streaming kernel's absence certificates need table keys of few rows (fmsi_stream.hip: off above 7 rows per key): one level
    // more than the default where that decides it (human scale: 15 -- 2.6 rows per key, 8.6 GB -- instead of 14)
    const int t_before_raise = t;
    if (stream_kernel && !std::getenv("FMSI_HIP_CLI_TABLE_DEPTH") && !opt.tables_given)
        while (t >= 14 && t < 16 && t < k && static_cast<double>(index_rows) > 7.0 * static_cast<double>(1ull << (2 * t)) && (1ull << (2 * (t + 1))) <= 4 * index_rows) ++t;

    // ---- one replicated handle per device (the index is read-only, the batches are independent) ---------------
    struct DeviceCtx {
        int device = 0;
        fmsi_hip_index* dev = nullptr;
        bool stream = false;   // this handle has the kLCP array: its reads go through the streaming kernel
        bool have_dict = false;
        // the dictionary built beside the queries (--tables background / large inputs): a second handle, published when it stands
        fmsi_hip_index* dict_dev = nullptr;
        std::atomic<bool> dict_ready{false};
        std::thread dict_builder;
        uint64_t dict_batches = 0;
        std::string error;
        std::unique_ptr<std::mutex> busy = std::make_unique<std::mutex>();   // calls on one handle are not re-entrant: its two workers take turns
    };
    // (on the heap, and left there when the binary skips the teardown: a dictionary builder that is still running then must not find
    // its context gone)
    auto* ctx_heap = new std::vector<DeviceCtx>(opt.devices.size());
    std::vector<DeviceCtx>& ctx = *ctx_heap;
    struct CtxOwner {
        std::vector<DeviceCtx>* p;
        bool keep = false;
        ~CtxOwner() {
            if (keep) return;
            for (DeviceCtx& c : *p)
                if (c.dict_builder.joinable()) c.dict_builder.join();
            delete p;
        }
    } ctx_owner{ctx_heap};
    for (size_t d = 0; d < ctx.size(); ++d) ctx[d].device = opt.devices[d];
    bool rows64 = false;
    auto join_builders = [&] {
        for (DeviceCtx& c : ctx)
            if (c.dict_builder.joinable()) c.dict_builder.join();
    };
    auto destroy_all = [&] {
        join_builders();
        for (DeviceCtx& c : ctx) {
            if (c.dict_dev) fmsi_hip_index_destroy(c.dict_dev);
            if (c.dev) fmsi_hip_index_destroy(c.dev);
        }
    };
    // The accelerators are large: when HBM is short -- the GPU shared with other work -- the table shrinks (first device decides,
    // the others follow) and the optional ones are left out; only a failure of the index proper is an error.
    auto setup = [&](DeviceCtx& c, bool first) -> bool {
        int rc_create = FMSI_HIP_ENOMEM;
        // (a device other than the first starts from what the first one settled on and may go further down by itself: what it streams
        // with and how deep its table is are its own, DeviceCtx::stream)
        int tt = t;
        bool st = stream_kernel;
        for (;;) {
            // (the table's signatures are written beside the first batches: FMSI_HIP_LOAD_ASYNC_TABLE, include/fmsi_hip.h)
            static const bool sync_table = std::getenv("FMSI_HIP_CLI_SYNC_TABLE") != nullptr;   // (A/B runs)
            rc_create = fmsi_hip_index_load(opt.index_prefix.c_str(), (st ? FMSI_HIP_LOAD_KLCP : 0) | (sync_table ? 0 : FMSI_HIP_LOAD_ASYNC_TABLE), c.device, std::max(tt, 0), &c.dev);
            if (rc_create != 0 && rc_create != FMSI_HIP_ENOMEM && st && !load_klcp) {
                // a kLCP file that does not load must not fail a run that never asked for it: the searches of round 3 answer instead
                st = false;
                continue;
            }
            if (rc_create != FMSI_HIP_ENOMEM) break;
            // Not enough device memory.  In this order: the depth a streaming run was raised to goes back to the one chosen before; a run
            // that streams without having been asked to (no -S) lets the kLCP array and the streaming layout go and answers by the plain
            // searches, as round 3 did; then the table shrinks, two levels at a time.
            if (st && tt > t_before_raise) {
                std::cerr << "WARNING: not enough device memory for a prefix table of depth " << tt << ", trying " << t_before_raise << std::endl;
                tt = t_before_raise;
            } else if (st && !load_klcp) {
                std::cerr << "WARNING: not enough device memory to stream the windows through the kLCP kernel; taking the plain searches (same answers)" << std::endl;
                st = false;
            } else if (tt > 0) {
                std::cerr << "WARNING: not enough device memory for a prefix table of depth " << tt << ", trying " << std::max(tt - 2, 0) << std::endl;
                tt = std::max(tt - 2, 0);
            } else {
                break;
            }
        }
        c.stream = st;
        if (first) {
            t = tt;
            stream_kernel = st;
        }
        if (rc_create != 0) {
            c.error = rc_create == FMSI_HIP_EIO ? "index files" : fmsi_hip_last_error();
            return false;
        }
        fmsi_hip_index_info info;
        const bool wide = fmsi_hip_index_get_info(c.dev, &info) == 0 && info.rows64 != 0;
        if (first) rows64 = wide;
        // 2^32 rows and more: the plain search (from a prefix table of its own) -- also for k > 32 and -f, whose kernels are templates over
        // the row width -- and no other accelerator (include/fmsi_hip.h).  -S needs nothing else: the kLCP array is a short cut of the
        // reference's, the answers are the plain search's (and the record loop's strand bookkeeping is replayed on the host as always)
        if (want_dict && !wide) {
            // (compact: 23 <= k <= 32 -- the minimizers are 21 / 23 bases; other k take the k-mer dictionary)
            const uint32_t form = opt.compact_tables && k >= 23 ? FMSI_HIP_DICT_SUPERKMER_ONLY : 0u;
            c.have_dict = fmsi_hip_index_enable_dict(c.dev, form) == 0;
            if (!c.have_dict)
                std::cerr << "WARNING: the k-mer dictionary is off on device " << c.device << " (" << fmsi_hip_last_error() << "); results are unaffected" << std::endl;
        }
        if (!c.stream && !general && !c.have_dict && !wide && fmsi_hip_index_enable_pair_layout(c.dev) != 0) {
            c.error = fmsi_hip_last_error();
            return false;
        }
        const bool want_verify = !wide && !general && opt.candidate_tables && !opt.has_klcp && !output_orders && t > 0 && t < k && k <= 32;
        if (want_verify && fmsi_hip_index_enable_verify(c.dev) != 0)
            std::cerr << "WARNING: text verification is off (" << fmsi_hip_last_error() << "); results are unaffected" << std::endl;
        return true;
    };
    {
        TraceRange accel_range("fmsi: device index + accelerators");
        bool ok = setup(ctx[0], true);
        if (ok && ctx.size() > 1) {   // the other devices side by side
            std::vector<std::thread> pool;
            std::vector<char> oks(ctx.size(), 1);
            for (size_t d = 1; d < ctx.size(); ++d) pool.emplace_back([&, d] { oks[d] = setup(ctx[d], false) ? 1 : 0; });
            for (std::thread& th : pool) th.join();
            ok = std::find(oks.begin(), oks.end(), 0) == oks.end();
        }
        if (!ok) {
            for (const DeviceCtx& c : ctx)
                if (!c.error.empty()) {
                    if (c.error == "index files") {   // malformed payload: what load_index() failing means to ms_query
                        std::cerr << "ERROR: index not correctly loaded. Ensure that you correctly call `fmsi index` before." << std::endl;
                        destroy_all();
                        stop_reader();
                        return usage_query(output_orders);
                    }
                    if (c.error.rfind("-S and k > 32", 0) == 0) std::cerr << "ERROR: " << c.error << std::endl;
                    else std::cerr << "ERROR: cannot create the device index: " << c.error << std::endl;
                    break;
                }
            destroy_all();
            stop_reader();
            return 1;
        }
    }
    auto t_loaded = clock::now();
    if (want_bg_dict && !rows64)
        for (DeviceCtx& c : ctx)
            c.dict_builder = std::thread([&c, prefix = opt.index_prefix] {
                TraceRange range("fmsi: k-mer dictionary beside the queries");
                fmsi_hip_index* h = nullptr;
                // (a failure -- no room on the device -- leaves the run on the path it started on; results are the same)
                if (fmsi_hip_index_load(prefix.c_str(), 0, c.device, 0, &h) != 0) return;
                if (fmsi_hip_index_enable_dict(h, 0) != 0) {
                    fmsi_hip_index_destroy(h);
                    return;
                }
                c.dict_dev = h;
                c.dict_ready.store(true, std::memory_order_release);
            });

    // OR-mode presence output does not depend on the strand order, so the cheaper schedule is exact;
    // everything else replays the predictor from both strands unless --fast.
    const bool need_both = !general && (opt.has_klcp || output_orders || opt.mode == Mode::All);
    const bool strict = need_both && !opt.fast;
    uint32_t flags_all = (opt.mode == Mode::All ? FMSI_HIP_MODE_ALL : FMSI_HIP_MODE_OR);
    if (strict) flags_all |= FMSI_HIP_STRICT;
    if (output_orders) flags_all |= FMSI_HIP_ORDERS;

    int exit_code = 0;
    uint64_t total_windows = 0;
    if (!query_reader.started()) {
        try {
            start_reader();
        } catch (const std::invalid_argument&) {
            destroy_all();
            throw;   // uncaught in the reference too: abort, exit status 134
        }
    }

    // ---- output: a formatter (the team writes the batch's lines into one buffer) and behind it the sink that hands the buffer to
    // stdout with write() while the next batch is formatted.  (Measured and dropped, profiles/README.md r02n: several threads
    // pwrite()-ing slices of a regular file -- buffered writes to one inode serialise on its lock -- and formatting straight into
    // a shared mapping of the grown file -- the team's page faults on one address space cost 2-3x the write() they replace.)
    struct OutBuf {
        std::vector<char> bits;             // presence modes: the whole batch
        std::vector<OrderText> parts;       // lookup: one piece per formatting thread
        std::unique_ptr<Prepared> kmer_batch;   // a k-mer batch: its parts' lines are the output (kept until they are written)
    };
    auto write_all = [](const char* p, size_t n) {
        while (n) {
            const ssize_t w = write(STDOUT_FILENO, p, n);
            if (w < 0 && errno == EINTR) continue;
            if (w <= 0) throw std::runtime_error("write to stdout failed");
            p += w;
            n -= static_cast<size_t>(w);
        }
    };
    BoundedQueue<std::unique_ptr<OutBuf>> sink_q(1), spare_q(3);
    std::exception_ptr sink_err;
    std::thread sink_thread([&] {
        try {
            std::fflush(stdout);   // (nothing of this program's is in stdio's buffer, but whatever is goes first)
            std::unique_ptr<OutBuf> b;
            while (sink_q.pop(b)) {
                out_prealloc.writing_starts();
                auto t1 = clock::now();
                {
                    TraceRange write_range("fmsi: write");
                    if (b->kmer_batch)
                        for (const auto& kp : b->kmer_batch->raw.kparts) write_all(kp->text.get(), kp->text_len);
                    else if (output_orders)
                        for (const OrderText& ot : b->parts) write_all(reinterpret_cast<const char*>(ot.buf.data()), ot.len);
                    else write_all(b->bits.data(), b->bits.size());
                }
                write_s += std::chrono::duration<double>(clock::now() - t1).count();
                b->kmer_batch.reset();   // (its parts go back to the parser's pool)
                (void)spare_q.push(std::move(b));
            }
        } catch (...) {
            sink_err = std::current_exception();
            sink_q.abort();
            spare_q.abort();
        }
    });
    std::thread writer_thread([&] {
        try {
            for (int i = 0; i < 2; ++i) (void)spare_q.push(std::make_unique<OutBuf>());
            std::unique_ptr<Prepared> P;
            while (out_q.pop(P)) {
                const size_t n = output_orders || P->raw.kmer_k ? 0 : P->rec_out[P->raw.recs.size()];
                std::unique_ptr<OutBuf> b;
                if (!spare_q.pop(b)) break;   // the sink failed
                auto t0 = clock::now();
                trace_push("fmsi: format");
                if (P->raw.kmer_k) {
                    patch_kmer_lines(*P, team);
                    b->kmer_batch = std::move(P);
                } else if (output_orders) {
                    format_orders(*P, team, b->parts, rows64);
                } else {
                    b->bits.resize(n);
                    format_bits(*P, team, b->bits.data());
                }
                trace_pop();
                format_s += std::chrono::duration<double>(clock::now() - t0).count();
                P.reset();
                if (!sink_q.push(std::move(b))) break;
            }
            sink_q.close();
        } catch (...) {
            writer_err = std::current_exception();
            sink_q.abort();
        }
        out_q.abort();   // (on every way out: the hand-over loop must not block on a writer that is gone)
    });

    // ---- device workers: one per handle.  A worker takes the next raw batch (its number = its place in the output), packs it,
    // runs it on its device and files it under that number; this thread takes the batches in order for the sequential part -- the
    // replay of the strand predictor -- and hands them to the writer.  With one device this still overlaps a batch's packing and
    // device time with the previous batch's replay.
    std::mutex done_m;
    std::condition_variable done_cv;
    std::map<uint64_t, std::unique_ptr<Prepared>> done;   // finished batches waiting for their turn
    uint64_t next_seq = 0;                                // the batch this thread hands on next
    // two workers per handle: while one has the device, the other packs the next batch
    constexpr unsigned kWorkersPerDevice = 2;
    unsigned workers_left = static_cast<unsigned>(ctx.size()) * kWorkersPerDevice;
    bool stop = false;
    std::exception_ptr worker_err;
    std::mutex stats_m;
    const unsigned worker_team = std::max(1u, team / static_cast<unsigned>(ctx.size()));
    std::vector<std::thread> workers;
    for (unsigned wi = 0; wi < ctx.size() * kWorkersPerDevice; ++wi)
        workers.emplace_back([&, cp = &ctx[wi / kWorkersPerDevice]] {
            try {
                std::unique_ptr<RawBatch> raw;
                uint64_t seq = 0;
                while (raw_q.pop_seq(raw, seq)) {
                    auto P = std::make_unique<Prepared>(std::move(raw));
                    auto t0 = clock::now();
                    trace_push("fmsi: pack batch");
                    const bool kmer_batch = P->raw.kmer_k != 0;
                    if (kmer_batch) {   // the parser packed the k-mers: the parts' words side by side in page-locked memory, nothing else
                        P->windows = P->raw.kmer_recs;
                        P->packed.resize(P->windows + 1);
                        std::vector<uint64_t> part_at(P->raw.kparts.size() + 1, 0);
                        for (size_t i = 0; i < P->raw.kparts.size(); ++i) part_at[i + 1] = part_at[i] + P->raw.kparts[i]->n;
                        const unsigned copiers = static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>(worker_team, P->raw.kparts.size())));
                        parallel_for(copiers, [&](unsigned t) {
                            for (size_t i = t; i < P->raw.kparts.size(); i += copiers)
                                std::memcpy(P->packed.data() + part_at[i], P->raw.kparts[i]->kmers.get(), P->raw.kparts[i]->n * 8);
                        });
                    } else {
                        prepare_batch(*P, k, worker_team, general && k % 2 == 0);
                    }
                    trace_pop();
                    auto t1 = clock::now(), td = t1;
                    const uint64_t nw = P->windows;
                    // nothing to replay: the device hands over one bit per window, an eighth of the bytes to copy and to read here
                    P->packed_bits = !output_orders && !strict;
                    if (output_orders) P->orders.resize(nw * (strict ? 2 : 1));
                    else if (P->packed_bits) P->bits.resize((nw + 63) / 64 + 1);
                    else P->bytes.resize(nw);
                    if (nw) {
                        std::lock_guard<std::mutex> device_turn(*cp->busy);
                        td = clock::now();   // (device time: from the moment the handle is this worker's)
                        TraceRange device_range("fmsi: device (copy in, kernels, copy out)");
                        void* dst = output_orders ? static_cast<void*>(P->orders.data())
                                                  : (P->packed_bits ? static_cast<void*>(P->bits.data()) : static_cast<void*>(P->bytes.data()));
                        const uint32_t flags = flags_all | (P->packed_bits ? FMSI_HIP_OUT_BITS : 0u);
                        // the handle this batch goes to: the dictionary's, once it stands (same answers, so the switch may fall anywhere)
                        const bool bg_dict = !cp->have_dict && cp->dict_ready.load(std::memory_order_acquire);
                        fmsi_hip_index* const handle = bg_dict ? cp->dict_dev : cp->dev;
                        if (bg_dict) ++cp->dict_batches;
                        if (kmer_batch)   // isolated k-mers (query_kmers_single, src/fms_index.h:263-331; under -S a chunk of one window, :181-254: the same)
                            check(fmsi_hip_query_kmers(handle, P->packed.data(), nw, k, flags, dst, nullptr, nullptr), "fmsi_hip_query_kmers");
                        else if (general)   // -f and|xor|r-s: f(ones, total) of both strands (query_kmers<general>, src/fms_index.h:316-327)
                            check(fmsi_hip_query_runs_fn(cp->dev, P->packed.data(), P->lower.empty() ? nullptr : P->lower.data(), P->bases, P->run_start.data(), P->run_windows.data(), P->run_start.size(), nw, k,
                                                         opt.fn, opt.fn_r, opt.fn_s, FMSI_HIP_OUT_BITS, dst, nullptr, nullptr),
                                  "fmsi_hip_query_runs_fn");
                        else if (cp->stream && !cp->have_dict && !bg_dict && !rows64)   // -S: kLCP streaming kernel (query_kmers_streaming, src/fms_index.h:181-254)
                            check(fmsi_hip_query_stream(handle, P->packed.data(), P->bases, P->run_start.data(), P->run_windows.data(),
                                                        P->run_start.size(), nw, k, flags, dst, nullptr, nullptr),
                                  "fmsi_hip_query_stream");
                        else
                            check(fmsi_hip_query_runs(handle, P->packed.data(), P->bases, P->run_start.data(), P->run_windows.data(),
                                                      P->run_start.size(), nw, k, flags, dst, nullptr, nullptr),
                                  "fmsi_hip_query_runs");
                        check(fmsi_hip_sync(handle, nullptr), "fmsi_hip_sync");
                    }
                    auto t2 = clock::now();
                    PinVec<uint64_t>().swap(P->packed);
                    PinVec<uint64_t>().swap(P->lower);
                    {
                        std::lock_guard<std::mutex> lk(stats_m);
                        prep_s += std::chrono::duration<double>(t1 - t0).count();
                        gpu_s += std::chrono::duration<double>(t2 - td).count();
                    }
                    std::unique_lock<std::mutex> lk(done_m);
                    // (a worker runs at most one batch ahead of the hand-over: the map holds no more batches than there are workers)
                    done_cv.wait(lk, [&] { return stop || seq < next_seq + 2 * kWorkersPerDevice * ctx.size(); });
                    if (stop) break;
                    done[seq] = std::move(P);
                    done_cv.notify_all();
                }
            } catch (...) {
                std::lock_guard<std::mutex> lk(done_m);
                if (!worker_err) worker_err = std::current_exception();
                stop = true;
            }
            std::lock_guard<std::mutex> lk(done_m);
            --workers_left;
            done_cv.notify_all();
        });

    try {
        StrandPredictor predictor;
        for (;;) {
            std::unique_ptr<Prepared> P;
            {
                std::unique_lock<std::mutex> lk(done_m);
                done_cv.wait(lk, [&] { return stop || done.count(next_seq) || workers_left == 0; });
                if (stop) break;
                auto it = done.find(next_seq);
                if (it == done.end()) break;   // all workers are gone and the next batch never came: end of input
                P = std::move(it->second);
                done.erase(it);
                ++next_seq;
                done_cv.notify_all();
            }
            auto t2 = clock::now();
            if (strict) {
                TraceRange replay_range("fmsi: predictor replay");
                // (a k-mer batch under -S: every record is one chunk of one window, whose streaming step IS the single step -- the first strand's
                // result, the second's if that does not settle it, log_result of the two: src/fms_index.h:181-254 against :263-331)
                replay_batch(*P, predictor, opt.mode, output_orders, load_klcp && !P->raw.kmer_k, k, team, rows64);
                replay_reruns += P->replay_reruns;
            }
            replay_s += std::chrono::duration<double>(clock::now() - t2).count();
            total_windows += P->windows;
            if (!out_q.push(std::move(P))) break;   // the writer failed
        }
    } catch (const std::exception& e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
        exit_code = 1;
    }
    // whatever ended the loop -- end of input, a device error, a writer that failed (ENOSPC, EPIPE) -- nobody may stay blocked:
    // the reader in raw_q.push(), a worker in raw_q.pop_seq() or waiting for its turn to file a batch
    {
        std::lock_guard<std::mutex> lk(done_m);
        stop = true;
        done_cv.notify_all();
    }
    raw_q.abort();
    out_q.close();
    for (std::thread& th : workers) th.join();
    query_reader.join();
    writer_thread.join();
    sink_thread.join();
    out_prealloc.finish();
    for (const std::exception_ptr& ep : {query_reader.reader_err, writer_err, sink_err, worker_err}) {
        if (!ep) continue;
        try {
            std::rethrow_exception(ep);
        } catch (const std::exception& e) {
            std::cerr << "ERROR: " << e.what() << std::endl;
            exit_code = 1;
        }
    }
    if (opt.stats) {
        const double total = std::chrono::duration<double>(clock::now() - t_start).count();
        const double load = std::chrono::duration<double>(t_loaded - t_start).count();
        fprintf(stderr,
                "{\"windows\": %llu, \"total_seconds\": %.6f, \"load_seconds\": %.6f, \"device_seconds\": %.6f, \"read_seconds\": %.6f, "
                "\"prepare_seconds\": %.6f, \"replay_seconds\": %.6f, \"format_seconds\": %.6f, \"write_seconds\": %.6f, \"threads\": %u, "
                "\"devices\": %zu, \"strict\": %s, \"block_parallel_read_bytes\": %llu, \"parallel_gunzip_bytes\": %llu, \"parallel_gunzip_rounds\": %llu, \"parallel_gunzip_gave_up\": %s, \"dictionary\": \"%s\", \"dictionary_batches\": %llu, \"kmer_batches\": %llu, "
                "\"replay_reruns\": %llu, \"epoch_start\": %.6f, \"epoch_end\": %.6f}\n",
                static_cast<unsigned long long>(total_windows), total, load, gpu_s, query_reader.read_s, prep_s, replay_s, format_s, write_s, team,
                ctx.size(), strict ? "true" : "false", static_cast<unsigned long long>(query_reader.parallel_read_bytes),
                static_cast<unsigned long long>(query_reader.pgz_bytes), static_cast<unsigned long long>(query_reader.pgz_rounds), query_reader.pgz_gave_up ? "true" : "false",
                ctx[0].have_dict ? "before the first batch" : (want_bg_dict && !rows64 ? "beside the queries" : "none"),
                static_cast<unsigned long long>(ctx[0].dict_batches), static_cast<unsigned long long>(query_reader.kmer_batches),
                static_cast<unsigned long long>(replay_reruns),
                // (wall-clock stamps of this function's first and last line: what the process spends before and behind them -- the loader,
                // the runtime's start-up, the teardown of its mappings and of the device context -- is the parent's to see)
                epoch_start, std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count());
    }
    // (the `fmsi` binary exits right behind this call, main.cpp: it neither tears the device down nor waits for a dictionary that is still
    // being built when the input has ended)
    if (!std::getenv("FMSI_HIP_CLI_NO_TEARDOWN")) destroy_all();
    else {
        if (g_exit_hook) g_exit_hook(exit_code);   // (the `fmsi` binary: does not come back)
        for (DeviceCtx& c : ctx)
            if (c.dict_builder.joinable()) {
                c.dict_builder.detach();
                ctx_owner.keep = true;
            }
    }
    return exit_code;
}
