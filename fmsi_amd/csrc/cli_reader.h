// The query side's readers of `fmsi query` / `lookup` (query_cli.cpp): batches of raw records for the packing workers.
//   * a plain file is mapped and parsed in place by several threads (ranges cut at record starts; names and single-line sequences stay
//     views of the mapping),
//   * a block-gzip file has its members inflated side by side, stretch after stretch, each parsed the same way,
//   * an ordinary gzip file is inflated by the team, too (pgunzip.h: chunks of the one deflate stream entered at block starts found in its
//     middle), stretch after stretch, each parsed the same way,
//   * anything else -- stdin, a FASTQ file that cannot be cut safely -- goes through the ordinary reader (fastx.cpp, the
//     reference's kseq rules), which also takes over wherever the three above have to stop.
// Every path yields the sequential parse of the reference (src/kseq.h:183-226, src/parser.h:14-27), batch after batch in file order.
#pragma once

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "cli_host_kernels.h"
#include "fastx.h"
#include "pgunzip.h"

namespace fmsi {
namespace {

// The team: threads that stay.  parallel_for() used to start team - 1 threads per call and join them -- 20 to 30 us each, five such calls per
// batch (parse, copy, replay twice, patch), and a batch of k-mer records is 4 ms of work for sixteen: measured by making the batches four times
// as large, every stage of the k-mer path took half the time or less (read 0.22 -> 0.14 s, format 0.07 -> 0.02 s, replay 0.20 -> 0.07 s per 10^8
// records).  Now the shares of a call are tasks for a pool that lives as long as the process; the caller takes share 0 and then helps with
// whatever is queued until its own shares are done.  Shares are leaves: none of them calls parallel_for().
class TeamPool {
public:
    static TeamPool& get() {
        static TeamPool* p = new TeamPool;   // (never destroyed: the process ends with its threads)
        return *p;
    }
    struct Job {
        std::atomic<unsigned> left{0};
        std::mutex m;
        std::condition_variable cv;
    };
    using Task = std::function<void()>;
    void submit(Task t) {
        {
            std::lock_guard<std::mutex> lk(m_);
            q_.push_back(std::move(t));
            if (idle_ == 0 && threads_ < kMaxThreads) {
                ++threads_;
                try {
                    std::thread([this] { worker(); }).detach();
                } catch (...) {   // (no thread to be had: the callers run the queue themselves, help_until_done)
                    --threads_;
                }
            }
        }
        cv_.notify_one();
    }
    // the caller of parallel_for() takes queued tasks -- its own or anybody's -- while its job has shares outstanding
    void help_until_done(Job& job) {
        for (;;) {
            if (job.left.load(std::memory_order_acquire) == 0) return;
            Task t;
            {
                std::lock_guard<std::mutex> lk(m_);
                if (!q_.empty()) {
                    t = std::move(q_.front());
                    q_.pop_front();
                }
            }
            if (t) {
                t();
                continue;
            }
            // (nothing queued: the caller's outstanding shares are in other threads' hands, and the last of them notifies.  No timed wait: GCC 11's
            // ThreadSanitizer does not know pthread_cond_clockwait and takes the mutex for held across it)
            std::unique_lock<std::mutex> lk(job.m);
            job.cv.wait(lk, [&] { return job.left.load(std::memory_order_acquire) == 0; });
            return;
        }
    }

private:
    static constexpr unsigned kMaxThreads = 64;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Task> q_;
    unsigned threads_ = 0, idle_ = 0;
    void worker() {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(m_);
                ++idle_;
                cv_.wait(lk, [&] { return !q_.empty(); });
                --idle_;
                t = std::move(q_.front());
                q_.pop_front();
            }
            t();
        }
    }
};

inline bool team_spawns_threads() {   // (FMSI_HIP_CLI_SPAWN_THREADS; read first on the main thread: host_kernels_init's rule)
    static const bool v = std::getenv("FMSI_HIP_CLI_SPAWN_THREADS") != nullptr;
    return v;
}

template <class F>
void parallel_for(unsigned team, F&& fn) {   // fn(t) for t in [0, team); t = 0 runs on the caller
    if (team <= 1) {
        fn(0u);
        return;
    }
    const bool spawn = team_spawns_threads();   // (A/B runs: a thread per share and call, as rounds 1-4)
    std::vector<std::exception_ptr> err(team);
    if (spawn) {
        std::vector<std::thread> th;
        th.reserve(team - 1);
        unsigned started = 1;
        try {
            for (; started < team; ++started)
                th.emplace_back([&, started] {
                    try { fn(started); } catch (...) { err[started] = std::current_exception(); }
                });
        } catch (...) {   // a thread that cannot be started: its share and the rest run on the caller
        }
        for (unsigned t = started; t < team; ++t)
            try { fn(t); } catch (...) { err[t] = std::current_exception(); }
        try { fn(0u); } catch (...) { err[0] = std::current_exception(); }
        for (auto& x : th) x.join();
    } else {
        TeamPool& pool = TeamPool::get();
        // (the job's counter and condition variable outlive the call through the shares' references: the last share may still be inside
        // notify_all() when the caller, who saw the count reach zero, is gone; fn and err are the caller's and are done with by then)
        auto job = std::make_shared<TeamPool::Job>();
        job->left.store(team - 1, std::memory_order_relaxed);
        for (unsigned t = 1; t < team; ++t)
            pool.submit([&fn, &err, job, t] {
                try { fn(t); } catch (...) { err[t] = std::current_exception(); }
                std::lock_guard<std::mutex> lk(job->m);
                if (job->left.fetch_sub(1, std::memory_order_acq_rel) == 1) job->cv.notify_all();
            });
        try { fn(0u); } catch (...) { err[0] = std::current_exception(); }
        pool.help_until_done(*job);
    }
    for (auto& e : err)
        if (e) std::rethrow_exception(e);
}

template <class T>
class BoundedQueue {
public:
    explicit BoundedQueue(size_t cap) : cap_(cap) {}
    bool push(T&& v) {   // false: the consumer has gone away
        std::unique_lock<std::mutex> lk(m_);
        not_full_.wait(lk, [&] { return q_.size() < cap_ || closed_; });
        if (closed_) return false;
        q_.push_back(std::move(v));
        not_empty_.notify_one();
        return true;
    }
    bool pop(T& v) {     // false: closed and drained
        std::unique_lock<std::mutex> lk(m_);
        not_empty_.wait(lk, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        v = std::move(q_.front());
        q_.pop_front();
        not_full_.notify_one();
        return true;
    }
    bool pop_seq(T& v, uint64_t& seq) {   // pop() + the item's number in pop order (several consumers, one order)
        std::unique_lock<std::mutex> lk(m_);
        not_empty_.wait(lk, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        v = std::move(q_.front());
        q_.pop_front();
        seq = popped_++;
        not_full_.notify_one();
        return true;
    }
    void set_cap(size_t cap) {   // (the producer's call: a larger bound lets it run further ahead)
        std::lock_guard<std::mutex> lk(m_);
        cap_ = cap;
        not_full_.notify_all();
    }
    void close() {       // producer: no more items; consumer: stop the producer
        std::lock_guard<std::mutex> lk(m_);
        closed_ = true;
        not_full_.notify_all();
        not_empty_.notify_all();
    }
    void abort() {
        std::lock_guard<std::mutex> lk(m_);
        closed_ = true;
        q_.clear();
        not_full_.notify_all();
        not_empty_.notify_all();
    }

private:
    std::mutex m_;
    std::condition_variable not_full_, not_empty_;
    std::deque<T> q_;
    size_t cap_;
    uint64_t popped_ = 0;
    bool closed_ = false;
};

struct RecRef {   // views: into the mapped query file, the batch's arena (multi-line sequences) or the batch's text (ordinary reader)
    const char* name;
    const char* seq;
    uint64_t seq_len;
    uint32_t name_len;
};
// ---- records of ONE k-mer each (`>name\nKMER\n`: k-mer lists, the shape of BASELINE's isolated-query configurations) --------------------
// Such a file is 40 bytes of text per window, where a read file is one: the per-record work decides, so it is done ONCE, by the thread
// that parses the range -- the k-mer is validated and packed into the 8-byte word fmsi_hip_query_kmers takes, and the record's output line
// ("name\t?\n") is written with the answer left open; what remains behind the device is one byte per record to fill in.  No RecRef, no
// run table, no op list.  The buffers are recycled (a fresh 20 MB block per part would be page faults of the whole team on one address space).
struct KmerPart {
    std::unique_ptr<char[]> text;     // the part's output lines
    size_t text_cap = 0, text_len = 0;
    std::unique_ptr<uint64_t[]> kmers;   // record i's k-mer
    std::unique_ptr<uint32_t[]> at;      // ... and the offset of its answer character in `text`
    size_t cap = 0, n = 0;               // (plain arrays: nothing clears or counts them but the parser)
    void reset(size_t text_bytes, size_t recs) {
        if (text_cap < text_bytes) {
            text.reset(new char[text_bytes]);
            text_cap = text_bytes;
        }
        if (cap < recs) {
            kmers.reset(new uint64_t[recs]);
            at.reset(new uint32_t[recs]);
            cap = recs;
        }
        text_len = 0;
        n = 0;
    }
};
class KmerPartPool {
public:
    static std::unique_ptr<KmerPart> get() {
        std::lock_guard<std::mutex> lk(mutex());
        auto& v = parts();
        if (v.empty()) return std::make_unique<KmerPart>();
        auto p = std::move(v.back());
        v.pop_back();
        return p;
    }
    static void put(std::unique_ptr<KmerPart> p) {
        if (!p) return;
        std::lock_guard<std::mutex> lk(mutex());
        if (parts().size() < 512) parts().push_back(std::move(p));
    }

private:
    static std::mutex& mutex() { static std::mutex m; return m; }
    static std::vector<std::unique_ptr<KmerPart>>& parts() { static auto* v = new std::vector<std::unique_ptr<KmerPart>>; return *v; }
};

struct RawBatch {   // never moved or copied: its records point into its own text (a short std::string lives INSIDE the object)
    RawBatch() = default;
    ~RawBatch() {
        for (auto& p : kparts) KmerPartPool::put(std::move(p));
    }
    RawBatch(const RawBatch&) = delete;
    RawBatch& operator=(const RawBatch&) = delete;
    std::string text;                                 // ordinary reader: names and sequences, as read
    std::vector<FastxViewParser::Arena> arenas;       // block-parallel reader: one per part, holding only what is not a view of the file
    std::shared_ptr<char[]> inflated;                 // block-gzip reader: the decompressed bytes the views look into
    std::vector<RecRef> recs;
    uint64_t seq_bytes = 0;
    // a batch of one-k-mer records (kmer_k != 0): the parts in file order instead of `recs`
    int kmer_k = 0;
    uint64_t kmer_recs = 0;
    std::vector<std::unique_ptr<KmerPart>> kparts;
};

// ---- block-parallel reader ---------------------------------------------------------------------------------------
// A plain (not gzipped) FASTA file is mapped and cut at lines that start with '>' into ranges of about a batch; a range
// is cut again the same way into one part per worker, and every worker runs the reader's rules (fastx.cpp, the
// reference's kseq rules) over its part WITHOUT copying: names and single-line sequences stay views of the mapping.  That is the sequential parse exactly, as long as no '+' separator line occurs:
// a line-start '>' is then always a record start (src/kseq.h:183-226: only a FASTQ quality string can hold one that is
// not).  A worker that meets a '+' line says so, and the range is parsed again by ONE reader that then goes on to the
// end of the file -- ranges before it held no '+' line, so its start is a record start of the sequential parse too.
struct MappedFile {
    const unsigned char* data = nullptr;
    size_t size = 0;
    ~MappedFile() {
        if (data) munmap(const_cast<unsigned char*>(data), size);
    }
};
// the query file as a mapping, if it is a regular, uncompressed file that starts with a FASTA or FASTQ header and is worth it
bool map_plain_fasta(const std::string& path, MappedFile& m, bool* fastq) {
    *fastq = false;
    if (path == "-" || std::getenv("FMSI_HIP_SERIAL_READER")) return false;
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    // (FMSI_HIP_READER_MIN_BYTES: the tests send small files through here)
    const char* min_env = std::getenv("FMSI_HIP_READER_MIN_BYTES");
    const long long min_bytes = min_env ? std::max(1ll, atoll(min_env)) : (8ll << 20);
    bool ok = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size >= min_bytes;
    if (ok) {
        void* p = mmap(nullptr, static_cast<size_t>(st.st_size), PROT_READ, MAP_PRIVATE, fd, 0);
        ok = p != MAP_FAILED;
        if (ok) {
            m.data = static_cast<const unsigned char*>(p);
            m.size = static_cast<size_t>(st.st_size);
            (void)madvise(p, m.size, MADV_SEQUENTIAL);
            if (m.data[0] == '@') *fastq = true;
            else if (m.data[0] != '>') ok = false;   // (a gzip stream starts with 0x1f 0x8b)
        }
    }
    close(fd);
    return ok;
}
// FASTQ: first position >= from of a line that starts with '@' and whose next-but-one line starts with '+' -- in a file of 4-line
// records that is a header and nothing else (a quality line may start with '@', but then the line after next is a sequence).
// A wrong guess in an unusual file (multi-line quality) is harmless: the part before it then does not end between two records, its
// reader says so (clean_end), and the ordinary reader takes over from there.
size_t next_fastq_start(const unsigned char* d, size_t from, size_t end) {
    auto next_line = [&](size_t p) -> size_t {
        const void* nl = p < end ? std::memchr(d + p, '\n', end - p) : nullptr;
        return nl ? static_cast<size_t>(static_cast<const unsigned char*>(nl) - d) + 1 : end;
    };
    size_t p = from;
    if (p > 0 && p < end && d[p - 1] != '\n') p = next_line(p);
    while (p < end) {
        if (d[p] == '@') {
            const size_t l2 = next_line(next_line(p));
            if (l2 < end && d[l2] == '+') return p;
        }
        p = next_line(p);
    }
    return end;
}
// first position >= from (and < end) of a '>' that starts a line; `end` if there is none
size_t next_record_start(const unsigned char* d, size_t from, size_t end, bool fastq = false) {
    if (fastq) return next_fastq_start(d, from, end);
    while (from < end) {
        const void* g = std::memchr(d + from, '>', end - from);
        if (!g) return end;
        const size_t at = static_cast<size_t>(static_cast<const unsigned char*>(g) - d);
        if (at == 0 || d[at - 1] == '\n') return at;
        from = at + 1;
    }
    return end;
}
// last position <= len where a record starts (by the rules of next_record_start), 0 if there is none behind the first byte:
// everything in front of it is whole records, what follows is carried over to the next stretch of the stream
size_t last_record_start(const unsigned char* d, size_t len, bool fastq) {
    if (!fastq) {
        for (size_t e = len; e > 0;) {
            const void* g = memrchr(d, '>', e);
            if (!g) return 0;
            const size_t at = static_cast<size_t>(static_cast<const unsigned char*>(g) - d);
            if (at == 0 || d[at - 1] == '\n') return at;
            e = at;
        }
        return 0;
    }
    for (size_t window = size_t(1) << 20;; window <<= 2) {   // a header needs its next two lines in sight: look from a little before the end
        const size_t from = len > window ? len - window : 0;
        size_t c = next_fastq_start(d, from, len);
        if (c < len) {
            for (size_t n; (n = next_fastq_start(d, c + 1, len)) < len;) c = n;
            return c;
        }
        if (from == 0) return 0;
    }
}

// all records of d[lo, hi) (lo at a record start) as views (FastxViewParser: the ordinary reader's rules)
bool parse_range(const unsigned char* d, size_t lo, size_t hi, FastxViewParser::Arena& arena_out, std::vector<RecRef>& recs_out, uint64_t& seq_bytes_out,
                 bool fastq = false) {
    // (the callers' per-part results sit side by side -- vector headers and counters of eight threads on one or two cache lines -- and a
    // record is a push_back and an add: everything stays local until the part is done)
    FastxViewParser r(d + lo, hi - lo);
    FastxViewParser::Arena arena;
    std::vector<RecRef> recs;
    recs.reserve((hi - lo) / 256 + 16);
    uint64_t seq_bytes = 0;
    for (;;) {
        RecRef rr;
        size_t name_len, seq_len;
        if (!r.next(arena, &rr.name, &name_len, &rr.seq, &seq_len)) break;
        rr.name_len = static_cast<uint32_t>(name_len);
        rr.seq_len = seq_len;
        recs.push_back(rr);
        seq_bytes += seq_len;
    }
    arena_out = std::move(arena);
    recs_out = std::move(recs);
    seq_bytes_out += seq_bytes;
    // FASTA file: clean = no '+' line was met (then '>' lines are record starts).  FASTQ file: clean = the part ended between two
    // records -- then the next part starts where the sequential parse would start a record, too.
    return fastq ? r.clean_end() : !r.saw_fastq();
}
// All records of d[lo, hi) (lo at a record start) if every one of them is `>name[ comment]\n` + exactly k characters of ACGTacgt + `\n`
// (the last line of the file may lack its '\n'), followed by the next '>' or the end of the range: false at the first record of any other
// shape (a sequence of another length or on several lines, an invalid character, '\r', a blank line, a FASTQ record) -- the caller then
// parses the range by the ordinary rules.  For the records it accepts this IS the ordinary parse (src/kseq.h:183-226: name = the header up
// to the first white space, the sequence its one line) and the record loop's walk (src/main.cpp:328-373: one chunk of one window; the line
// printed is name, '\t', the answer, '\n').  `limit` = end of the readable bytes behind d (>= hi).
struct KmerPackBmi2 {
    __attribute__((target("bmi2"))) bool operator()(const char* s, int k, uint64_t* w) const { return kmer_word_bmi2(s, k, w); }
};
struct KmerPackPlain {
    bool operator()(const char* s, int k, uint64_t* w) const { return kmer_word_plain(s, k, w); }
};
template <class PackFn>
__attribute__((always_inline)) inline bool parse_kmer_records(const unsigned char* d, size_t lo, size_t hi, size_t limit, int k, KmerPart& part, PackFn pack) {
    const size_t max_recs = (hi - lo) / (static_cast<size_t>(k) + 3) + 1;
    part.reset(hi - lo + 64, max_recs);   // a line out is shorter than its record (+ the slack of the 16-byte name copy)
    uint64_t* kmers_out = part.kmers.get();   // (plain arrays: a push_back is two more branches per record)
    uint32_t* at_out = part.at.get();
    size_t n_recs = 0;
    char* out = part.text.get();
    size_t p = lo;
    const size_t ku = static_cast<size_t>(k);
    while (p < hi) {
        if (d[p] != '>') return false;
        size_t q = p + 1;
        while (q < hi && d[q] > ' ') ++q;   // (white space of isspace(): ' ', \t \n \v \f \r -- all <= ' '; other control characters: the ordinary parse)
        if (q >= hi) return false;
        if (d[q] != ' ' && d[q] != '\n' && d[q] != '\t') return false;
        const size_t name_len = q - (p + 1);
        size_t nl = q;
        if (d[q] != '\n') {
            const void* e = std::memchr(d + q, '\n', hi - q);
            if (!e) return false;
            nl = static_cast<size_t>(static_cast<const unsigned char*>(e) - d);
        }
        const size_t s = nl + 1;
        if (s + ku > hi) return false;
        if (s + ku < hi) {
            if (d[s + ku] != '\n') return false;
            if (s + ku + 1 < hi && d[s + ku + 1] != '>') return false;
        }
        uint64_t w;
        const bool valid = s + 32 <= limit ? pack(reinterpret_cast<const char*>(d + s), k, &w) : kmer_word_tail(reinterpret_cast<const char*>(d + s), k, &w);
        if (!valid) return false;
        // (a name of up to 16 characters -- a read id, a number -- is one fixed 16-byte copy: the bytes behind it are readable, the k-mer's line
        // follows, and writable, the line's own tail and the slack)
        if (name_len <= 16 && s + 32 <= limit) std::memcpy(out, d + p + 1, 16);
        else std::memcpy(out, d + p + 1, name_len);
        out += name_len;
        *out++ = '\t';
        at_out[n_recs] = static_cast<uint32_t>(out - part.text.get());
        *out++ = '0';
        *out++ = '\n';
        kmers_out[n_recs++] = w;
        p = s + ku + 1;
    }
    part.n = n_recs;
    part.text_len = static_cast<size_t>(out - part.text.get());
    return true;
}
__attribute__((target("bmi2"))) bool parse_kmer_range_bmi2(const unsigned char* d, size_t lo, size_t hi, size_t limit, int k, KmerPart& part) {
    return parse_kmer_records(d, lo, hi, limit, k, part, KmerPackBmi2());
}
bool parse_kmer_range_plain(const unsigned char* d, size_t lo, size_t hi, size_t limit, int k, KmerPart& part) {
    return parse_kmer_records(d, lo, hi, limit, k, part, KmerPackPlain());
}
inline bool parse_kmer_range(const unsigned char* d, size_t lo, size_t hi, size_t limit, int k, KmerPart& part) {
    if (hi - lo >= (size_t(1) << 32)) return false;   // (32-bit offsets into the part's text)
    return use_bmi2() ? parse_kmer_range_bmi2(d, lo, hi, limit, k, part) : parse_kmer_range_plain(d, lo, hi, limit, k, part);
}

// a stretch's bytes: from 4 MiB on aligned to 2 MiB and advised for huge pages (the team writes a fresh 64 MB of them per stretch -- inflated
// members, the gzip reader's copies --, which in 4-KiB pages is 16 000 page faults)
inline std::shared_ptr<char[]> alloc_stretch(size_t bytes) {
    constexpr size_t kHuge = size_t(2) << 20;
    if (bytes >= 2 * kHuge) {
        void* p = nullptr;
        const size_t rounded = (bytes + kHuge - 1) / kHuge * kHuge;
        if (posix_memalign(&p, kHuge, rounded) == 0) {
            (void)madvise(p, rounded, MADV_HUGEPAGE);
            return std::shared_ptr<char[]>(static_cast<char*>(p), [](char* q) { std::free(q); });
        }
    }
    return std::shared_ptr<char[]>(new char[bytes]);
}

class QueryReader {
public:
    using clock = std::chrono::steady_clock;
    QueryReader(const std::string& query_fn, uint64_t batch_windows, unsigned team, BoundedQueue<std::unique_ptr<RawBatch>>& raw_q)
        : opt{query_fn, batch_windows}, team(team), raw_q(raw_q) {}
    ~QueryReader() {   // (whoever leaves without draining the queue: the thread must not stay blocked on it)
        raw_q.abort();
        join();
    }
    // opens the file -- throws std::invalid_argument like OpenFile(), src/parser.h:22 -- and starts reading on a thread of its own
    void start() {
    // (every switch of the environment is read HERE, on the caller's thread, before the reader's exists: see host_kernels_init)
    host_kernels_init();
    (void)team_spawns_threads();
    env_serial = std::getenv("FMSI_HIP_SERIAL_READER") != nullptr;
    env_no_stretches = std::getenv("FMSI_HIP_NO_STREAM_STRETCHES") != nullptr;
    if (const char* e = std::getenv("FMSI_HIP_CLI_KMER_RANGE")) kmer_mult = static_cast<uint64_t>(std::max(1, atoi(e)));
    reader = std::make_unique<FastxReader>(opt.query_fn);   // throws std::invalid_argument like OpenFile(), src/parser.h:22
    parallel_read = map_plain_fasta(opt.query_fn, mapped, &parallel_fastq);
    if (!parallel_read && opt.query_fn != "-" && !env_serial) {
        const char* min_env = std::getenv("FMSI_HIP_READER_MIN_BYTES");
        const size_t min_bytes = min_env ? static_cast<size_t>(std::max(1ll, atoll(min_env))) : (size_t(1) << 20);
        (void)bgzf.open(opt.query_fn, min_bytes);
        if (!bgzf && !std::getenv("FMSI_HIP_NO_PARALLEL_GUNZIP") && pgz.open(opt.query_fn, min_bytes))
            if (const char* e = std::getenv("FMSI_HIP_PGZ_CHUNK_BYTES")) pgz.set_chunk_bytes(static_cast<size_t>(std::max(1ll, atoll(e))));   // (tests: many chunks of a small file)
    }
        reader_thread = std::thread([this] { run(); });
    }
    bool started() const { return reader_thread.joinable(); }
    // before start(): records of exactly one k-mer (k <= 32) may be handed on as k-mer batches (RawBatch::kmer_k) -- presence modes only
    void allow_kmer_batches(int k) { kmer_k = k; }
    uint64_t kmer_batches = 0;
    void join() {
        if (reader_thread.joinable()) reader_thread.join();
    }
    double read_s = 0;                  // time spent reading (not waiting for the queue)
    uint64_t parallel_read_bytes = 0;   // (--stats: how far the block-parallel readers got before they had to hand over, if they did)
    uint64_t pgz_bytes = 0, pgz_rounds = 0, pgz_marker_symbols = 0;   // (--stats: what the team's inflate of an ordinary gzip file took)
    bool pgz_gave_up = false;
    std::exception_ptr reader_err;

private:
    struct {
        std::string query_fn;
        uint64_t batch_windows;         // bases of query sequence per batch
    } opt;
    unsigned team;
    BoundedQueue<std::unique_ptr<RawBatch>>& raw_q;
    std::unique_ptr<FastxReader> reader;
    MappedFile mapped;
    BgzfFile bgzf;
    ParallelGunzip pgz;
    bool parallel_read = false, parallel_fastq = false;
    std::thread reader_thread;
    bool env_serial = false, env_no_stretches = false;
    uint64_t kmer_mult = 1;   // (FMSI_HIP_CLI_KMER_RANGE, measurements: ranges of k-mer records this many times the batch size)
    int kmer_k = 0;
    bool kmer_streak = true;   // the last range was k-mer records (or none was tried yet): the next one is cut for the whole team

    // the range d[cuts[0], cuts.back()) as a k-mer batch, if every part of it is one-k-mer records
    bool try_kmer_batch(const unsigned char* d, const std::vector<size_t>& cuts, size_t limit, RawBatch& nb) {
        if (!kmer_k) return false;
        const size_t parts = cuts.size() - 1;
        nb.kparts.resize(parts);
        for (auto& kp : nb.kparts) kp = KmerPartPool::get();
        std::vector<char> ok(parts, 1);
        parallel_for(static_cast<unsigned>(parts), [&](unsigned w) { ok[w] = parse_kmer_range(d, cuts[w], cuts[w + 1], limit, kmer_k, *nb.kparts[w]) ? 1 : 0; });
        kmer_streak = std::find(ok.begin(), ok.end(), 0) == ok.end();
        if (!kmer_streak) {
            for (auto& kp : nb.kparts) KmerPartPool::put(std::move(kp));
            nb.kparts.clear();
            return false;
        }
        nb.kmer_k = kmer_k;
        for (const auto& kp : nb.kparts) nb.kmer_recs += kp->n;
        if (kmer_batches++ == 0) raw_q.set_cap(40);   // 40 bytes of text per window: the parse is the long stage, let it run ahead of the index load
        return true;
    }

    void run() {
    try {
        auto b = std::make_unique<RawBatch>();
        auto t0 = clock::now();
        size_t resume_at = 0;   // block-parallel reader: where the ordinary reader takes over (a '+' line was met)
        if (parallel_read) {
            const unsigned char* d = mapped.data;
            size_t pos = 0;
            bool serial = false;
            while (pos < mapped.size && !serial) {
                const unsigned workers = kmer_k && kmer_streak && !parallel_fastq ? std::max(1u, team) : std::max(1u, std::min(8u, team / 2));
                // a range of about a batch, ending at a record start (or the end of the file)
                size_t end = mapped.size;
                // (a FASTQ file is half qualities; a k-mer list is 40 bytes of text per window: its ranges are larger, so that a batch is not over
                // before the threads that parse it have all started)
                const uint64_t range_bytes = (parallel_fastq ? 2 : 1) * opt.batch_windows * (kmer_k && kmer_streak && kmer_batches ? kmer_mult : 1);
                if (mapped.size - pos > range_bytes + (range_bytes >> 3))
                    end = next_record_start(d, pos + range_bytes, mapped.size, parallel_fastq);
                std::vector<size_t> cuts{pos};
                for (unsigned w = 1; w < workers; ++w) {
                    const size_t c = next_record_start(d, pos + (end - pos) / workers * w, end, parallel_fastq);
                    if (c > cuts.back() && c < end) cuts.push_back(c);
                }
                cuts.push_back(end);
                const size_t parts = cuts.size() - 1;
                std::vector<std::vector<RecRef>> recs(parts);
                std::vector<uint64_t> seq_bytes(parts, 0);
                std::vector<char> clean(parts, 1);
                auto nb = std::make_unique<RawBatch>();
                if (!parallel_fastq && try_kmer_batch(d, cuts, mapped.size, *nb)) {
                    pos = end;
                    parallel_read_bytes = pos;
                    if (nb->kmer_recs == 0) continue;
                    read_s += std::chrono::duration<double>(clock::now() - t0).count();
                    if (!raw_q.push(std::move(nb))) return;
                    t0 = clock::now();
                    continue;
                }
                nb->arenas.resize(parts);
                parallel_for(static_cast<unsigned>(parts), [&](unsigned w) {   // (joins its threads and passes a worker's exception on)
                    clean[w] = parse_range(d, cuts[w], cuts[w + 1], nb->arenas[w], recs[w], seq_bytes[w], parallel_fastq) ? 1 : 0;
                });
                if (std::find(clean.begin(), clean.end(), 0) != clean.end()) {   // a FASTQ record: the ordinary reader from here on
                    serial = true;
                    resume_at = pos;
                    break;
                }
                size_t total = 0;
                for (size_t w = 0; w < parts; ++w) total += recs[w].size();
                nb->recs.reserve(total);
                for (size_t w = 0; w < parts; ++w) {
                    nb->recs.insert(nb->recs.end(), recs[w].begin(), recs[w].end());
                    nb->seq_bytes += seq_bytes[w];
                }
                pos = end;
                parallel_read_bytes = pos;
                if (nb->recs.empty()) continue;
                // (the stages behind count records in 32 bits per batch like the ordinary reader's batches: at most 2^23)
                if (nb->recs.size() > (1u << 23)) {
                    serial = true;
                    resume_at = cuts[0];
                    break;
                }
                read_s += std::chrono::duration<double>(clock::now() - t0).count();
                if (!raw_q.push(std::move(nb))) return;
                t0 = clock::now();
            }
            if (!serial) {
                read_s += std::chrono::duration<double>(clock::now() - t0).count();
                raw_q.close();
                return;
            }
            reader = std::make_unique<FastxReader>(mapped.data + resume_at, mapped.size - resume_at);
        }
        // ---- block-gzip (BGZF) file: the members are found by their headers and inflated side by side into one stretch of the
        // stream after the other; a stretch ends where its last record starts (the rest is carried into the next one) and is
        // parsed like a range of a mapped plain file.  Anything unusual -- a member that is not BGZF, a CRC that does not match, a
        // part that does not end between two records -- hands the stream to the ordinary reader at the last byte given out.
        // ---- any other stream (an ordinary gzip file, stdin, a small file): the same stretches, their bytes read by the ordinary
        // reader's own zlib handle (one thread inflates -- that is gzip -- but the team parses; a stretch that is not plain records
        // goes back into the reader and the ordinary rules take the stream from its first byte).
        const bool stream_stretches = !parallel_read && !bgzf && !env_serial && !env_no_stretches;
        if (bgzf || stream_stretches) {
            size_t at = 0;             // (block-gzip) compressed offset of the next member
            uint64_t consumed = 0;     // uncompressed bytes handed on in batches
            std::string carry;
            bool eof = false, serial = false, first = true, fastq = false;
            uint64_t grow = 1;   // a record longer than a stretch: the next stretch is twice as long (so that carrying it along stays linear)
            std::shared_ptr<char[]> seg;
            size_t len = 0;
            while (!serial && !(eof && carry.empty())) {
                const uint64_t range_bytes = (fastq ? 2 : 1) * opt.batch_windows * grow;
                if (bgzf) {
                    std::vector<BgzfFile::Member> ms;
                    std::vector<size_t> offs;
                    size_t bytes = 0;
                    while (!eof && bytes < range_bytes) {
                        if (at >= bgzf.size()) {
                            eof = true;
                            break;
                        }
                        BgzfFile::Member m;
                        if (!bgzf.member(at, &m)) {
                            serial = true;
                            break;
                        }
                        offs.push_back(bytes);
                        ms.push_back(m);
                        bytes += m.isize;
                        at = m.next;
                    }
                    if (serial) break;
                    len = carry.size() + bytes;
                    seg = alloc_stretch(len + 1);
                    std::memcpy(seg.get(), carry.data(), carry.size());
                    // (inflating is by far the longest piece of work on this path: the whole team takes it)
                    const unsigned inflaters = static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>(team, ms.size())));
                    std::vector<char> ok(inflaters, 1);
                    parallel_for(inflaters, [&](unsigned t) {
                        for (size_t i = t; i < ms.size(); i += inflaters)
                            if (!bgzf.inflate(ms[i], seg.get() + carry.size() + offs[i])) ok[t] = 0;
                    });
                    if (std::find(ok.begin(), ok.end(), 0) != ok.end()) {
                        serial = true;
                        break;
                    }
                } else {
                    seg = alloc_stretch(carry.size() + range_bytes + 1);
                    std::memcpy(seg.get(), carry.data(), carry.size());
                    // (an ordinary gzip file: the team inflates it, pgunzip.h; where that reader gives up -- a stream it does not take, a
                    // trailer that does not match -- zlib goes on from the byte it got to, as it would have from the start)
                    size_t got = 0;
                    if (pgz && !pgz.failed()) {
                        got = pgz.read(seg.get() + carry.size(), range_bytes, team,
                                       [](unsigned n, const std::function<void(unsigned)>& fn) { parallel_for(n, fn); });
                        pgz_bytes = pgz.delivered();
                        pgz_rounds = pgz.stats.rounds;
                        pgz_marker_symbols = pgz.stats.marker_symbols;
                        if (pgz.failed()) {
                            pgz_gave_up = true;
                            reader->skip(pgz.delivered());
                        }
                    }
                    if (!pgz || pgz.failed()) got += reader->read_raw(seg.get() + carry.size() + got, range_bytes - got);
                    eof = got < range_bytes;
                    len = carry.size() + got;
                }
                // (from here on `seg[0, len)` = the carried tail + the new bytes; whatever ends the loop with serial = true before the
                // stretch's records were handed on leaves all of it to the ordinary reader)
                const unsigned char* d = reinterpret_cast<const unsigned char*>(seg.get());
                if (first && len) {
                    if (d[0] == '@') fastq = true;
                    else if (d[0] != '>') serial = true;   // (leading junk: the ordinary reader knows what to do with it)
                    first = false;
                    if (serial) break;
                }
                const size_t cut = eof ? len : last_record_start(d, len, fastq);
                if (cut == 0 && !eof) {   // one record longer than the stretch: read on
                    carry.assign(seg.get(), len);
                    grow *= 2;
                    len = 0;
                    continue;
                }
                grow = 1;
                const unsigned workers = kmer_k && kmer_streak && !fastq ? std::max(1u, team) : std::max(1u, std::min(8u, team / 2));
                std::vector<size_t> cuts{0};
                for (unsigned w = 1; w < workers; ++w) {
                    const size_t c = next_record_start(d, cut / workers * w, cut, fastq);
                    if (c > cuts.back() && c < cut) cuts.push_back(c);
                }
                cuts.push_back(cut);
                const size_t parts = cuts.size() - 1;
                std::vector<std::vector<RecRef>> recs(parts);
                std::vector<uint64_t> seq_bytes(parts, 0);
                std::vector<char> clean(parts, 1);
                auto nb = std::make_unique<RawBatch>();
                if (!fastq && try_kmer_batch(d, cuts, len, *nb)) {   // (the lines are copies: the stretch's bytes are not needed behind this point)
                    carry.assign(seg.get() + cut, len - cut);
                    consumed += cut;
                    parallel_read_bytes = consumed;
                    len = 0;
                    if (nb->kmer_recs == 0) continue;
                    read_s += std::chrono::duration<double>(clock::now() - t0).count();
                    if (!raw_q.push(std::move(nb))) return;
                    t0 = clock::now();
                    continue;
                }
                nb->arenas.resize(parts);
                nb->inflated = seg;
                parallel_for(static_cast<unsigned>(parts), [&](unsigned w) {
                    clean[w] = parse_range(d, cuts[w], cuts[w + 1], nb->arenas[w], recs[w], seq_bytes[w], fastq) ? 1 : 0;
                });
                size_t total = 0;
                for (size_t w = 0; w < parts; ++w) total += recs[w].size();
                if (std::find(clean.begin(), clean.end(), 0) != clean.end() || total > (1u << 23)) {
                    serial = true;
                    break;
                }
                nb->recs.reserve(total);
                for (size_t w = 0; w < parts; ++w) {
                    nb->recs.insert(nb->recs.end(), recs[w].begin(), recs[w].end());
                    nb->seq_bytes += seq_bytes[w];
                }
                carry.assign(seg.get() + cut, len - cut);
                consumed += cut;
                parallel_read_bytes = consumed;
                len = 0;
                if (nb->recs.empty()) continue;
                read_s += std::chrono::duration<double>(clock::now() - t0).count();
                if (!raw_q.push(std::move(nb))) return;
                t0 = clock::now();
            }
            if (!serial) {
                read_s += std::chrono::duration<double>(clock::now() - t0).count();
                raw_q.close();
                return;
            }
            if (bgzf) {
                reader->skip(consumed);
            } else if (len) {
                if (pgz && !pgz.failed()) reader->skip(pgz.delivered());   // (the ordinary reader has not read a byte yet)
                reader->unread(seg.get(), len);       // (the stretch that was not plain records, its carried tail in front)
            } else {
                if (pgz && !pgz.failed()) reader->skip(pgz.delivered());
                reader->unread(carry.data(), carry.size());
            }
        }
        // ordinary reader: names and sequences are appended to the batch's text, which moves as it grows -- the records hold
        // offsets until the batch is complete
        std::vector<std::pair<size_t, size_t>> offs;   // (name, sequence) offsets of the batch's records
        auto finish = [&offs](RawBatch& rb) {
            const char* base = rb.text.data();
            for (size_t r = 0; r < rb.recs.size(); ++r) {
                rb.recs[r].name = base + offs[r].first;
                rb.recs[r].seq = base + offs[r].second;
            }
            offs.clear();
        };
        for (;;) {
            RecRef rr;
            const size_t before = b->text.size();
            size_t name_off, name_len, seq_off, seq_len;
            if (!reader->next_append(b->text, &name_off, &name_len, &seq_off, &seq_len)) {
                b->text.resize(before);   // a partial record (truncated FASTQ) ends the stream, as in the reference
                break;
            }
            rr.name = rr.seq = nullptr;
            rr.name_len = static_cast<uint32_t>(name_len);
            rr.seq_len = seq_len;
            b->recs.push_back(rr);
            offs.emplace_back(name_off, seq_off);
            b->seq_bytes += rr.seq_len;
            if (b->seq_bytes >= opt.batch_windows || b->recs.size() >= (1u << 23)) {
                finish(*b);
                read_s += std::chrono::duration<double>(clock::now() - t0).count();
                if (!raw_q.push(std::move(b))) return;
                b = std::make_unique<RawBatch>();
                t0 = clock::now();
            }
        }
        finish(*b);
        read_s += std::chrono::duration<double>(clock::now() - t0).count();
        if (!b->recs.empty()) raw_q.push(std::move(b));
    } catch (...) {
        reader_err = std::current_exception();
    }
    raw_q.close();
    }
};

}  // namespace
}  // namespace fmsi
