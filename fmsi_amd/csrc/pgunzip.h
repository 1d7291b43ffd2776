// An ordinary gzip file -- ONE member of many deflate blocks, what `gzip reads.fa` writes -- inflated by several threads.
//
// The reference reads its query file through zlib's gzread() (src/parser.h:14-27, src/kseq.h:75-93): one thread, about 0.35 GB/s of text, which
// is ten times what everything behind it takes in this program.  A deflate stream cannot be entered in the middle by its own rules -- a block
// starts at any BIT, and its matches point up to 32 KiB back into text that the entering decoder has not seen -- so this reader does what pugz /
// rapidgzip do:
//   * the compressed bytes are cut into chunks; every chunk but a round's first LOOKS for a block start at or behind its first bit (a dynamic
//     block's header is redundant enough to be told from noise: code counts in range, a complete code-length code, complete literal / distance
//     codes, an end-of-block symbol -- and the block behind it must decode);
//   * from there the chunk is decoded with its 32 KiB of history UNKNOWN: the output is 16-bit symbols, a byte or "byte i of the window I was
//     not given"; once the last 32 KiB of output hold no such marker the history is known again and zlib itself takes over (inflatePrime +
//     inflateSetDictionary at a block boundary, Z_BLOCK to stop at one), which is where nearly all bytes are decoded;
//   * a chunk ends at the block boundary where the next chunk began -- EXACTLY there, bit for bit, or the next chunk's start was no block start
//     and the chunk decodes on through it; the chain of equal positions from the round's known first bit is what makes the result the stream's;
//   * the markers are replaced once the chunk in front is complete, and every member's CRC-32 and length are checked against its trailer.
// Anything else -- a stream that does not decode, a trailer that does not match, bytes that are not gzip behind a member -- ends the parallel
// reading BEFORE the round in question is handed out: failed() is set, delivered() says how many bytes of the uncompressed stream the caller
// got, and the ordinary reader (zlib, one thread, the reference's behaviour to the letter) goes on from there.
#pragma once

#include <cstdint>
#include <deque>
#include <functional>
#include <memory>
#include <string>
#include <vector>

namespace fmsi {

// The reader's buffers: not zero-filled, and from 4 MiB on aligned to 2 MiB and advised for huge pages -- a round's chunks write 2.4 GB per GB of
// text for the first time (16-bit symbols, then bytes), which in 4-KiB pages is 600 000 page faults from sixteen threads of one process.
class PgzBuf {
public:
    PgzBuf() = default;
    ~PgzBuf();
    PgzBuf(PgzBuf&& o) noexcept : p_(o.p_), cap_(o.cap_) { o.p_ = nullptr; o.cap_ = 0; }
    PgzBuf& operator=(PgzBuf&& o) noexcept;
    PgzBuf(const PgzBuf&) = delete;
    PgzBuf& operator=(const PgzBuf&) = delete;
    unsigned char* data() const { return static_cast<unsigned char*>(p_); }
    size_t cap() const { return cap_; }
    void grow(size_t bytes, size_t keep);   // at least `bytes` of room, the first `keep` bytes kept
private:
    void* p_ = nullptr;
    size_t cap_ = 0;
};

class ParallelGunzip {
public:
    // runs fn(0) ... fn(n - 1), on as many threads as the caller has; shares do not wait for one another
    using Runner = std::function<void(unsigned n, const std::function<void(unsigned)>& fn)>;

    ParallelGunzip() = default;
    ~ParallelGunzip();
    ParallelGunzip(const ParallelGunzip&) = delete;
    ParallelGunzip& operator=(const ParallelGunzip&) = delete;

    // maps the file if it is a regular file of at least min_bytes that starts with a gzip member header (deflate)
    bool open(const std::string& path, size_t min_bytes);
    // the same over bytes in memory (tests)
    bool open_memory(const unsigned char* data, size_t size);
    explicit operator bool() const { return data_ != nullptr; }

    // the next bytes of the uncompressed stream, up to n of them; fewer: the stream ended -- or failed() (see above)
    size_t read(char* dst, size_t n, unsigned team, const Runner& run);
    bool failed() const { return failed_; }
    bool eof() const { return done_ && ready_.empty(); }
    uint64_t delivered() const { return delivered_; }

    // what it did (--stats, tests)
    struct Stats {
        uint64_t rounds = 0, chunks = 0, chunks_passed_over = 0, chunks_without_start = 0;
        uint64_t marker_symbols = 0;   // output decoded with an unknown window
        uint64_t plain_bytes = 0;      // output decoded by zlib
        uint64_t members = 0;
    } stats;
    // (tests: compressed bytes per chunk; 0 = the schedule of its own: 1 MiB per thread in the first round, then four chunks of 2 MiB per thread)
    void set_chunk_bytes(size_t b) { chunk_bytes_ = b; }

private:
    struct Piece {
        PgzBuf buf;
        size_t len = 0, from = 0;
    };
    // buffers that go round: a round's chunks take the symbol arrays and the byte buffers of the rounds before
    std::vector<PgzBuf> sym_pool_, byte_pool_;
    bool round(unsigned team, const Runner& run);

    const unsigned char* data_ = nullptr;
    size_t size_ = 0;
    bool mapped_ = false;
    size_t chunk_bytes_ = 0;
    uint64_t next_bit_ = 0;                  // where the next round starts: a block boundary (or a member's first block)
    std::vector<unsigned char> window_;      // the up to 32 KiB of the member's text in front of next_bit_
    uint32_t crc_ = 0;                       // the current member's CRC-32 and length so far
    uint64_t member_len_ = 0;
    std::deque<Piece> ready_;                // decoded, not yet handed out
    uint64_t delivered_ = 0;
    bool done_ = false, failed_ = false;
};

}  // namespace fmsi
