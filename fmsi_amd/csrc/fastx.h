// FASTA/FASTQ (plain or gzip) record reader for the query side of the boundary.
// Behaviour follows the reference's reader, klib kseq over zlib (src/kseq.h:183-226, src/parser.h:14-27):
// name = header up to the first whitespace, multi-line sequences, '>'/'@' record starts, FASTQ quality
// skipped, a truncated FASTQ record ends the stream.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include <zlib.h>

namespace fmsi {

class FastxReader {
public:
    // path "-" = stdin.  Throws std::invalid_argument when the file cannot be opened, like
    // OpenFile() in the reference (src/parser.h:22).
    explicit FastxReader(const std::string& path);
    // The same reader over bytes already in memory (a range of a mapped file that starts at a record): the block-parallel
    // reader of the CLI runs one per range.  saw_fastq() tells whether a '+' separator line was met -- then `>` at a line
    // start is not necessarily a record start and the file must not be cut at such lines.
    FastxReader(const unsigned char* data, size_t len);
    bool saw_fastq() const { return fastq_seen_; }
    // after next() / next_append() returned false: did the input end between two records (true), or inside one -- a truncated or
    // malformed record, which ends the stream (false)?  The block-parallel reader cuts FASTQ files only where every part ends clean.
    bool clean_end() const { return clean_end_; }
    ~FastxReader();
    FastxReader(const FastxReader&) = delete;
    FastxReader& operator=(const FastxReader&) = delete;

    // Reads the next record into name/seq.  Returns false at end of input (or on the reference's
    // error returns -2/-3, which also end its loop, src/main.cpp:328).
    bool next();

    // Same, but the name and the sequence are APPENDED to `text` (first the name, then the sequence: the caller's batch
    // buffer), saving the copy out of name / seq.  Offsets and lengths are returned; on false `text` may hold a partial
    // record past its old size, which the caller truncates.
    bool next_append(std::string& text, size_t* name_off, size_t* name_len, size_t* seq_off, size_t* seq_len);

    // file mode, before the first record is read: go on from `bytes` into the UNCOMPRESSED stream (what a reader that took the
    // first part of the stream itself has consumed)
    void skip(uint64_t bytes);
    // file mode, before the first record is read: the next bytes of the UNCOMPRESSED stream as they are, up to n of them (fewer: the stream
    // ended) -- and back again: bytes handed out that their taker could not use are read by next() / next_append() before anything else.
    // (How the CLI parses gzip files and stdin: stretches of the stream cut at record starts and parsed by several threads, like a
    // block-gzip file's; a stretch that is not plain records goes back and the ordinary rules take the stream from there.)
    size_t read_raw(char* dst, size_t n);
    void unread(const char* data, size_t n);

    std::string name;
    std::string seq;

private:
    int getc_();
    // append bytes up to (not including) the next '\n' / whitespace; returns false at EOF with nothing read
    // `floor`: size of *dst below which the trailing-CR rule must not look (the start of the string being built)
    bool until_(bool line, std::string* dst, int* delim, size_t* counted, size_t floor = 0);
    bool fill_();

    gzFile fp_ = nullptr;
    std::vector<unsigned char> own_;      // gz mode: the read buffer
    const unsigned char* buf_ = nullptr;  // own_.data(), or the caller's bytes (memory mode: fp_ == nullptr)
    size_t begin_ = 0, end_ = 0;
    bool eof_ = false;
    bool fastq_seen_ = false;
    bool clean_end_ = false;
    int last_char_ = 0;
    std::string qual_;
};

// The same rules over bytes in memory WITHOUT copying them: a record's name and -- when its sequence sits on one line, the common
// shape of read files -- its sequence are views into the caller's bytes; only a multi-line sequence is joined into `arena` (whose
// blocks never move: views into it stay valid while the arena lives).  Record by record this yields exactly what FastxReader over the
// same bytes yields (tests/native/fastx_view_fuzz.cpp holds the two against each other), including saw_fastq() and clean_end().
class FastxViewParser {
public:
    struct Arena {
        std::vector<std::unique_ptr<char[]>> blocks;
        size_t used = 0, cap = 0;
        char* take(size_t n);   // n contiguous bytes
    };
    FastxViewParser(const unsigned char* data, size_t len) : d_(data), end_(len) {}
    // false at the end of the input (or a truncated / malformed FASTQ record, which ends the stream as in the reference)
    bool next(Arena& arena, const char** name, size_t* name_len, const char** seq, size_t* seq_len);
    bool saw_fastq() const { return fastq_seen_; }
    bool clean_end() const { return clean_end_; }

private:
    const unsigned char* d_;
    size_t p_ = 0, end_;
    bool fastq_seen_ = false, clean_end_ = false;
    int last_char_ = 0;
    std::vector<std::pair<size_t, size_t>> lines_;   // the current record's sequence lines [from, to)
};

// A block-gzip file (BGZF: htslib's bgzip, Illumina's converters -- a series of gzip members of at most 64 KiB each whose header
// says how long the member is, RFC 1952 extra subfield 'B' 'C'): the members can be found without inflating anything and inflated
// independently, which is what lets several threads share the decompression.  To zlib -- and so to the reference's reader,
// src/parser.h:14-27 -- such a file is an ordinary multi-member gzip stream.
class BgzfFile {
public:
    struct Member {
        size_t data_off, data_len;   // the raw deflate stream
        uint32_t isize, crc;         // uncompressed length and CRC-32 (the member's trailer)
        size_t next;                 // offset of the member behind it
    };
    BgzfFile() = default;
    ~BgzfFile();
    BgzfFile(const BgzfFile&) = delete;
    BgzfFile& operator=(const BgzfFile&) = delete;
    // maps the file if it is a regular file of at least min_bytes that starts with a BGZF member
    bool open(const std::string& path, size_t min_bytes);
    // the member at compressed offset `at`; false: no (complete) BGZF member there
    bool member(size_t at, Member* m) const;
    // inflates one member into dst[0, isize) and checks length and CRC
    bool inflate(const Member& m, char* dst) const;
    size_t size() const { return size_; }
    explicit operator bool() const { return data_ != nullptr; }

private:
    const unsigned char* data_ = nullptr;
    size_t size_ = 0;
};

}  // namespace fmsi
