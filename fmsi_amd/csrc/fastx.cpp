#include "fastx.h"

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

}  // namespace fmsi
