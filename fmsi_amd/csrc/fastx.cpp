#include "fastx.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace fmsi {

namespace {
constexpr size_t kBufSize = 1 << 18;
}

FastxReader::FastxReader(const unsigned char* data, size_t len) : buf_(data), begin_(0), end_(len) {}

FastxReader::FastxReader(const std::string& path) : own_(kBufSize) {
    buf_ = own_.data();
    if (path == "-") {
        fp_ = gzdopen(fileno(stdin), "r");
    } else {
        FILE* f = std::fopen(path.c_str(), "r");
        if (!f) throw std::invalid_argument("couldn't open file " + path);
        fp_ = gzdopen(fileno(f), "r");   // plain files are read transparently, as in the reference
    }
    if (!fp_) throw std::invalid_argument("couldn't open file " + path);
    gzbuffer(fp_, 1 << 20);
}

FastxReader::~FastxReader() {
    if (fp_) gzclose(fp_);
}

void FastxReader::skip(uint64_t bytes) {
    if (fp_ && bytes) (void)gzseek(fp_, static_cast<z_off_t>(bytes), SEEK_SET);   // (forward from 0: zlib inflates and discards)
    begin_ = end_ = 0;
}

size_t FastxReader::read_raw(char* dst, size_t n) {
    size_t got = 0;
    if (begin_ < end_) {   // (what unread() put back)
        got = std::min(n, end_ - begin_);
        std::memcpy(dst, buf_ + begin_, got);
        begin_ += got;
    }
    while (fp_ && got < n && !eof_) {
        const int r = gzread(fp_, dst + got, static_cast<unsigned>(std::min<size_t>(n - got, size_t(1) << 30)));
        if (r <= 0) {
            eof_ = true;
            break;
        }
        got += static_cast<size_t>(r);
    }
    return got;
}

void FastxReader::unread(const char* data, size_t n) {
    if (!fp_) return;
    std::vector<unsigned char> held(buf_ + begin_, buf_ + end_);   // (bytes not yet consumed stay behind the returned ones)
    own_.resize(std::max<size_t>(kBufSize, n + held.size()));
    std::memcpy(own_.data(), data, n);
    if (!held.empty()) std::memcpy(own_.data() + n, held.data(), held.size());
    buf_ = own_.data();
    begin_ = 0;
    end_ = n + held.size();
    if (end_) eof_ = false;   // (the stream may have ended while read_raw() was at it: what came back is still to be read; fill_() finds the end again)
}

BgzfFile::~BgzfFile() {
    if (data_) munmap(const_cast<unsigned char*>(data_), size_);
}

bool BgzfFile::open(const std::string& path, size_t min_bytes) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    bool ok = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && static_cast<size_t>(st.st_size) >= std::max<size_t>(min_bytes, 28);
    if (ok) {
        void* p = mmap(nullptr, static_cast<size_t>(st.st_size), PROT_READ, MAP_PRIVATE, fd, 0);
        ok = p != MAP_FAILED;
        if (ok) {
            data_ = static_cast<const unsigned char*>(p);
            size_ = static_cast<size_t>(st.st_size);
            (void)madvise(p, size_, MADV_SEQUENTIAL);
            Member m;
            if (!member(0, &m)) {
                munmap(p, size_);
                data_ = nullptr;
                size_ = 0;
                ok = false;
            }
        }
    }
    ::close(fd);
    return ok;
}

bool BgzfFile::member(size_t at, Member* m) const {
    // RFC 1952 header with FLG.FEXTRA: ID1 ID2 CM FLG MTIME(4) XFL OS XLEN(2) | subfields: SI1 SI2 SLEN(2) data
    if (at + 18 > size_) return false;
    const unsigned char* h = data_ + at;
    if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || h[3] != 4) return false;   // (exactly FEXTRA: what BGZF writers emit)
    const size_t xlen = h[10] | (static_cast<size_t>(h[11]) << 8);
    if (at + 12 + xlen + 8 > size_) return false;
    size_t bsize = 0;
    for (size_t x = 0; x + 4 <= xlen;) {
        const unsigned char* sf = h + 12 + x;
        const size_t slen = sf[2] | (static_cast<size_t>(sf[3]) << 8);
        if (x + 4 + slen > xlen) return false;
        if (sf[0] == 'B' && sf[1] == 'C' && slen == 2) bsize = (sf[4] | (static_cast<size_t>(sf[5]) << 8)) + 1;   // total block size
        x += 4 + slen;
    }
    if (bsize < 12 + xlen + 8 || at + bsize > size_) return false;
    m->data_off = at + 12 + xlen;
    m->data_len = bsize - (12 + xlen) - 8;
    const unsigned char* t = data_ + at + bsize - 8;
    m->crc = t[0] | (static_cast<uint32_t>(t[1]) << 8) | (static_cast<uint32_t>(t[2]) << 16) | (static_cast<uint32_t>(t[3]) << 24);
    m->isize = t[4] | (static_cast<uint32_t>(t[5]) << 8) | (static_cast<uint32_t>(t[6]) << 16) | (static_cast<uint32_t>(t[7]) << 24);
    m->next = at + bsize;
    return m->isize <= 65536;   // (a BGZF member holds at most 64 KiB; anything else is left to zlib)
}

bool BgzfFile::inflate(const Member& m, char* dst) const {
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;   // raw deflate
    zs.next_in = const_cast<Bytef*>(data_ + m.data_off);
    zs.avail_in = static_cast<uInt>(m.data_len);
    char none = 0;   // (an empty member -- BGZF's end-of-file marker -- still wants somewhere to write)
    zs.next_out = reinterpret_cast<Bytef*>(m.isize ? dst : &none);
    zs.avail_out = m.isize ? m.isize : 1;
    const int rc = ::inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.total_out == m.isize && zs.avail_in == 0;
    inflateEnd(&zs);
    return ok && crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const Bytef*>(dst), m.isize) == m.crc;
}

bool FastxReader::fill_() {
    if (eof_) return false;
    if (!fp_) {   // memory mode: the range is all there is
        begin_ = end_ = 0;
        eof_ = true;
        return false;
    }
    begin_ = 0;
    int n = gzread(fp_, own_.data(), static_cast<unsigned>(own_.size()));
    if (n <= 0) {
        end_ = 0;
        eof_ = true;
        return false;
    }
    end_ = static_cast<size_t>(n);
    return true;
}

int FastxReader::getc_() {
    if (begin_ >= end_ && !fill_()) return -1;
    return buf_[begin_++];
}

bool FastxReader::until_(bool line, std::string* dst, int* delim, size_t* counted, size_t floor) {
    bool gotany = false;
    if (delim) *delim = 0;
    for (;;) {
        if (begin_ >= end_ && !fill_()) break;
        size_t i;
        if (line) {
            const void* nl = std::memchr(buf_ + begin_, '\n', end_ - begin_);
            i = nl ? static_cast<size_t>(static_cast<const unsigned char*>(nl) - buf_) : end_;
        } else {
            for (i = begin_; i < end_; ++i)
                if (std::isspace(buf_[i])) break;
        }
        gotany = true;
        if (dst) dst->append(reinterpret_cast<const char*>(buf_ + begin_), i - begin_);
        if (counted) *counted += i - begin_;
        const size_t at = i;
        begin_ = i + 1;
        if (at < end_) {
            if (delim) *delim = buf_[at];
            break;
        }
    }
    if (!gotany) return false;
    if (line && dst && dst->size() - floor > 1 && dst->back() == '\r') dst->pop_back();   // src/kseq.h:146
    return true;
}

bool FastxReader::next() {
    int c;
    if (last_char_ == 0) {   // jump to the next header line
        while ((c = getc_()) >= 0 && c != '>' && c != '@') {}
        if (c < 0) {
            clean_end_ = true;
            return false;
        }
        last_char_ = c;
    }
    name.clear();
    seq.clear();
    if (!until_(false, &name, &c, nullptr)) return false;
    if (c != '\n') until_(true, nullptr, nullptr, nullptr);   // comment
    while ((c = getc_()) >= 0 && c != '>' && c != '+' && c != '@') {
        if (c == '\n') continue;
        seq.push_back(static_cast<char>(c));
        until_(true, &seq, nullptr, nullptr);
    }
    if (c == '>' || c == '@') last_char_ = c;
    if (c != '+') return true;   // FASTA
    fastq_seen_ = true;
    while ((c = getc_()) >= 0 && c != '\n') {}
    if (c < 0) return false;     // no quality string (-2 in the reference)
    std::string qual;
    while (until_(true, &qual, nullptr, nullptr) && qual.size() < seq.size()) {}
    last_char_ = 0;
    return qual.size() == seq.size();
}

bool FastxReader::next_append(std::string& text, size_t* name_off, size_t* name_len, size_t* seq_off, size_t* seq_len) {
    // next() with both strings living in `text`; every rule of next() is kept (the CR rule looks at the sequence only)
    int c;
    if (last_char_ == 0) {
        while ((c = getc_()) >= 0 && c != '>' && c != '@') {}
        if (c < 0) {
            clean_end_ = true;
            return false;
        }
        last_char_ = c;
    }
    *name_off = text.size();
    if (!until_(false, &text, &c, nullptr, *name_off)) return false;
    *name_len = text.size() - *name_off;
    if (c != '\n') until_(true, nullptr, nullptr, nullptr);   // comment
    *seq_off = text.size();
    while ((c = getc_()) >= 0 && c != '>' && c != '+' && c != '@') {
        if (c == '\n') continue;
        text.push_back(static_cast<char>(c));
        until_(true, &text, nullptr, nullptr, *seq_off);
    }
    *seq_len = text.size() - *seq_off;
    if (c == '>' || c == '@') last_char_ = c;
    if (c != '+') return true;   // FASTA
    fastq_seen_ = true;
    while ((c = getc_()) >= 0 && c != '\n') {}
    if (c < 0) return false;     // no quality string (-2 in the reference)
    qual_.clear();
    while (until_(true, &qual_, nullptr, nullptr) && qual_.size() < *seq_len) {}
    last_char_ = 0;
    return qual_.size() == *seq_len;
}

char* FastxViewParser::Arena::take(size_t n) {
    if (blocks.empty() || cap - used < n) {
        cap = std::max<size_t>(n, size_t(1) << 22);
        blocks.emplace_back(new char[cap]);
        used = 0;
    }
    char* p = blocks.back().get() + used;
    used += n;
    return p;
}

bool FastxViewParser::next(Arena& arena, const char** name, size_t* name_len, const char** seq, size_t* seq_len) {
    // FastxReader::next_append() step by step, on positions instead of copies
    const unsigned char* d = d_;
    auto line_end = [&](size_t from) -> size_t {   // position of the next '\n' at or behind `from`, or the end
        const void* nl = from < end_ ? std::memchr(d + from, '\n', end_ - from) : nullptr;
        return nl ? static_cast<size_t>(static_cast<const unsigned char*>(nl) - d) : end_;
    };
    if (last_char_ == 0) {   // jump to the next '>' or '@', wherever it is (src/kseq.h:186-189)
        while (p_ < end_ && d[p_] != '>' && d[p_] != '@') ++p_;
        if (p_ >= end_) {
            clean_end_ = true;
            return false;
        }
        last_char_ = d[p_++];
    }
    if (p_ >= end_) return false;   // a header character at the very end: no name
    size_t q = p_;
    while (q < end_ && !std::isspace(d[q])) ++q;
    *name = reinterpret_cast<const char*>(d + p_);
    *name_len = q - p_;
    const int delim = q < end_ ? d[q] : 0;
    p_ = q < end_ ? q + 1 : end_;
    if (delim != '\n' && p_ < end_) p_ = std::min(end_, line_end(p_) + 1);   // comment
    lines_.clear();
    size_t total = 0;
    int c = -1;
    while (p_ < end_) {
        c = d[p_++];
        if (c == '>' || c == '+' || c == '@') break;
        if (c == '\n') {
            c = -1;
            continue;
        }
        const bool last_byte = p_ >= end_;   // (the reader's line read finds nothing then and returns before the CR rule)
        size_t from = p_ - 1, to = line_end(p_);
        p_ = std::min(end_, to + 1);
        total += to - from;
        if (!last_byte && total > 1 && d[to - 1] == '\r') {   // the trailing-CR rule looks at the sequence so far (src/kseq.h:146)
            --to;
            --total;
        }
        lines_.emplace_back(from, to);
        c = -1;
    }
    if (lines_.size() <= 1) {
        *seq = lines_.empty() ? reinterpret_cast<const char*>(d + p_) : reinterpret_cast<const char*>(d + lines_[0].first);
    } else {
        char* dst = arena.take(total);
        *seq = dst;
        for (const auto& [from, to] : lines_) {
            std::memcpy(dst, d + from, to - from);
            dst += to - from;
        }
    }
    *seq_len = total;
    if (c == '>' || c == '@') last_char_ = c;
    if (c != '+') return true;   // FASTA
    fastq_seen_ = true;
    {   // rest of the '+' line; none: no quality string
        const size_t nl = line_end(p_);
        if (nl >= end_) {
            p_ = end_;
            return false;
        }
        p_ = nl + 1;
    }
    // quality lines until they are as long as the sequence (at least one line is read); the CR rule looks at the whole string read
    // so far, so its lines are kept as ranges
    size_t qual = 0;
    lines_.clear();
    while (p_ < end_) {
        const size_t to = line_end(p_);
        if (to > p_) lines_.emplace_back(p_, to);
        qual += to - p_;
        p_ = std::min(end_, to + 1);
        if (qual > 1 && d[lines_.back().second - 1] == '\r') {
            --qual;
            if (--lines_.back().second == lines_.back().first) lines_.pop_back();
        }
        if (qual >= total) break;
    }
    last_char_ = 0;
    return qual == total;
}

}  // namespace fmsi
