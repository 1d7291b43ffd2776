/* fmsi_hip.h -- C-ABI of libfmsi_hip.so: the MI355X (gfx950) drop-in for the `fmsi query` hot path.
 *
 * The reference (OndrejSladky/fmsi, single-threaded C++) has no FFI; the path sits behind one C++
 * call, `query_kmers<mode>(fms_index&, char* seq, size_t len, int k, bool has_klcp, std::ostream&,
 * bool output_orders, f)` (src/fms_index.h:333-342, called per chunk from src/main.cpp:344-350), fed
 * by `load_index(prefix, use_klcp)` (src/fms_index.h:502-526).  This header is what a maintainer of
 * the reference would bind instead (see INTEGRATION.md for the stub): plain pointers and sizes, no
 * C++ types, no exceptions, no torch types.  All citations are relative to the reference root.
 *
 * Conventions
 *   - every function returns 0 on success or a negative FMSI_HIP_E* code; the message is available
 *     from fmsi_hip_last_error() (thread-local);
 *   - the library owns device memory behind opaque handles; the caller owns every buffer it passes;
 *   - calls on one handle are not re-entrant; different handles may be driven from different threads;
 *   - query calls are asynchronous on `stream` (a hipStream_t passed as void*, NULL = default
 *     stream); `out` is valid after the stream is synchronised (fmsi_hip_sync);
 *   - there is NO CPU fallback: without a usable HIP device every call fails with FMSI_HIP_ENODEV.
 */
#ifndef FMSI_HIP_H
#define FMSI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMSI_HIP_ABI_VERSION 8   /* 8: FMSI_HIP_LOAD_ASYNC_TABLE, fmsi_hip_index_wait_background, fmsi_hip_index_release_scratch; 7: FMSI_HIP_KERNEL_SORTED; 6: FMSI_HIP_DICT_SUPERKMER + fmsi_hip_index_info.sk_*, fmsi_hip_index_enable_stream_layout, FMSI_HIP_NO_DICT_CHAIN; 5: fmsi_hip_keep_scratch; 4: fmsi_hip_query_runs_fn (`-f and|xor|r-s`); 2: fmsi_hip_index_info.candidate_entry_bytes, fmsi_hip_comm_*, fmsi_hip_gather; 3: FMSI_HIP_OUT_BITS, fmsi_hip_index_load, fmsi_hip_trim_scratch, fmsi_hip_index_enable_dict + fmsi_hip_index_info.dict_*, fmsi_hip_index_build_synthetic_spec */

/* error codes */
#define FMSI_HIP_OK 0
#define FMSI_HIP_EINVAL (-1)   /* bad argument */
#define FMSI_HIP_ENODEV (-2)   /* no HIP device / device index out of range */
#define FMSI_HIP_EHIP (-3)     /* a HIP runtime call failed */
#define FMSI_HIP_ENOMEM (-4)
#define FMSI_HIP_EIO (-5)      /* index files missing / malformed */
#define FMSI_HIP_ERANGE (-6)   /* not built for this size: an accelerator on an index of >= 2^32 rows, or k unsupported */
#define FMSI_HIP_ECOMM (-7)    /* RCCL failure */

/* ---- query flags ------------------------------------------------------------------------------ */
/* demasking mode: exactly one of OR / ALL.
 *   OR  = query_mode::orr  (default `fmsi query`): k-mer present iff any occurrence (either strand) is ON.
 *   ALL = query_mode::all  (`fmsi query -O`): first row of the SA interval decides (src/fms_index.h:129-131). */
#define FMSI_HIP_MODE_OR 0x0u
#define FMSI_HIP_MODE_ALL 0x1u
/* STRICT: search BOTH strands of every window and return both ternary results so that the host can
 * replay the reference's strand predictor (src/fms_index.h:18-49, 268, 312-315) and reproduce its
 * output even when `-O` is used on a mask that is not max-ones.  Without STRICT the forward strand is
 * searched first and the reverse complement only when needed (OR: result != 1; ALL: result == -1),
 * which equals the reference's output for OR always and for ALL under its documented precondition
 * (src/main.cpp:111-112). */
#define FMSI_HIP_STRICT 0x2u
/* ORDERS: `fmsi lookup` -- per window the rank of the ON occurrence (src/fms_index.h:146-156), -1 if absent.
 * Output element is int64_t (STRICT: two int64_t, forward then reverse-complement strand). */
#define FMSI_HIP_ORDERS 0x4u
/* query/out/counters pointers are DEVICE pointers (already resident in HBM); default: host pointers,
 * staged through library-owned device buffers. */
#define FMSI_HIP_DEVICE_PTRS 0x8u
/* kernel selection (diagnostics / A-B runs); default = FMSI_HIP_KERNEL_AUTO */
#define FMSI_HIP_KERNEL_AUTO 0x00u
#define FMSI_HIP_KERNEL_LOCKSTEP 0x10u   /* one lane per k-mer, wave walks the k LF steps in lock-step */
#define FMSI_HIP_KERNEL_REFILL 0x20u     /* persistent lanes: a lane that finishes pulls the next k-mer (ballot/mbcnt); with fmsi_hip_query_stream:
                                            the first streaming kernel (64-byte blocks + separate kLCP words) instead of the S-block one */
#define FMSI_HIP_KERNEL_PAIR 0x30u       /* persistent lanes + two-base LF steps over the 128-B pair layout (needs
                                            fmsi_hip_index_enable_pair_layout); AUTO picks it when the layout exists */
#define FMSI_HIP_KERNEL_PHASED 0x40u     /* pair layout + prefix table: the table lookups run in a kernel of their own and only the
                                            searches they leave alive go to the stepping kernel; AUTO picks it when both exist;
                                            without a usable table it is KERNEL_PAIR */
#define FMSI_HIP_KERNEL_DICT 0x50u       /* the k-mer dictionary (fmsi_hip_index_enable_dict): one bucket read per k-mer; AUTO picks it when the handle
                                            has one and the call is a presence query with the index's k; without it this is AUTO */
#define FMSI_HIP_KERNEL_SORTED 0x60u     /* the table-less LF walk for a batch (fmsi_sorted.hip): both strands' searches sorted by the bases the
                                            backward search consumes first, one lane per search in lock-step, two-base steps (needs the pair
                                            layout; presence modes, k <= 32 -- otherwise the call takes what AUTO takes); reads no table */
#define FMSI_HIP_KERNEL_MASK 0xF0u
/* do not use the private accelerators -- prefix table, candidate tables, k-mer dictionary -- even if the handle has them: the LF walk
 * itself (canonical-step accounting runs, the table-less legs of the benchmark) */
#define FMSI_HIP_NO_PREFIX_TABLE 0x100u

/* bit-packed results (non-ORDERS calls): `out` is an 8-byte aligned array of u64 words, LSB first like the reference's
 * sdsl bit-vectors -- window w is bit (w & 63) of word w >> 6 (what the reference prints as '0' / '1', src/fms_index.h:305-309);
 * with STRICT, nibble (w & 15) of word w >> 4 holding the STRICT byte below.  ceil(n / 64) resp. ceil(n / 16) words; the unused
 * high bits of the last word are zero.  One eighth (half) of the bytes over PCIe and over xGMI (fmsi_hip_gather). */
#define FMSI_HIP_OUT_BITS 0x200u
/* fmsi_hip_query_runs on a handle whose dictionary has the chained lookups' extras: take the plain lookups (one bucket read per window)
 * for this call -- A/B runs of the two forms on one handle */
#define FMSI_HIP_NO_DICT_CHAIN 0x400u
/* ... and not the super-k-mer dictionary either */
#define FMSI_HIP_NO_SUPERKMER 0x800u

/* ---- index creation flags --------------------------------------------------------------------- */
/* build a 4^t-entry prefix table that jumps the first t LF steps (private optimisation: the final SA
 * interval is unchanged).  t is given by fmsi_index_view.prefix_len (0 = none). */

/* Output encoding of the non-ORDERS query calls: one byte per window.
 *   non-STRICT: 0 / 1  (what the reference prints: '0' / '1')
 *   STRICT    : (fwd + 1) | ((rc + 1) << 2), fwd/rc in {-1, 0, 1} = infer_presence() of each strand. */

/* ---- host-side view of a loaded index (what load_index() holds, src/fms_index.h:52-66) -------- */
typedef struct {
    const uint64_t* ac_gt; uint64_t ac_gt_bits;   /* .fmsi.ac_gt payload: LSB-first u64 words (sdsl int_vector<1>) */
    const uint64_t* ac;    uint64_t ac_bits;      /* .fmsi.ac */
    const uint64_t* gt;    uint64_t gt_bits;      /* .fmsi.gt */
    const uint64_t* mask;  uint64_t mask_bits;    /* SA-transformed mask, DECODED from the rrr_vector<63> to plain bits (n bits) */
    const uint64_t* klcp;  uint64_t klcp_bits;    /* .fmsi.klcp, optional (NULL / 0) */
    uint64_t counts[4];                           /* .fmsi.misc lines 2-5 */
    uint64_t dollar_position;                     /* .fmsi.misc line 1 */
    int32_t k;                                    /* .fmsi.misc line 6 */
    int32_t prefix_len;                           /* private prefix table length t (0 = none, <= 16: 4^t entries of 8 bytes) */
} fmsi_index_view;

typedef struct fmsi_hip_index fmsi_hip_index;     /* device-resident index (opaque) */
typedef struct fmsi_host_index fmsi_host_index;   /* host-resident parsed .fmsi.* files (opaque) */

typedef struct {
    uint64_t n;                 /* BWT rows = |MS| + 1 */
    int32_t k;
    int32_t has_klcp;
    int32_t device;
    int32_t prefix_len;
    uint64_t block_bytes;       /* bytes per interleaved block */
    uint64_t block_symbols;     /* BWT positions per block */
    uint64_t n_blocks;
    uint64_t device_bytes;      /* total HBM footprint of the handle */
    uint64_t counts[4];         /* C array of the index ({1, #A+1, #A+#C+1, #A+#C+#G+1}, src/fms_index.h:451) */
    uint64_t dollar_position;
    uint64_t candidate_entry_bytes; /* 0 = no candidate tables (fmsi_hip_index_enable_verify), else 16 or 32 bytes per table entry */
    uint64_t dict_bytes;        /* 0 = no k-mer dictionary (fmsi_hip_index_enable_dict), else its HBM footprint */
    uint64_t dict_keys[4];      /* distinct canonical k-mers handed to each of its levels (level 0: all of them; further: the overflow) */
    uint64_t dict_ambiguous;    /* keys whose `-O` answer was taken from the index's own search (occurrences with differing mask bits) */
    uint32_t dict_levels, dict_log2_buckets;   /* levels in use; log2 of level 0's bucket count */
    uint64_t dict_chain_bytes;  /* 0, or the bytes of the chained lookups' extras (FMSI_HIP_DICT_CHAIN; included in dict_bytes) */
    uint64_t rows64;            /* 1: the handle keeps 64-bit rows (n >= 2^32 - 128; src/fms_index.h:98-103 uses size_t): plain search only --
                                   a prefix table of 16-byte entries (at most 14 deep), any k and the demasking functions, but no pair layout, candidate tables or dictionary
                                   (FMSI_HIP_ERANGE); export works, and
                                   fmsi_hip_query_stream is served by the plain search (no kLCP kernel; the same answers);
                                   loaded from files or built on the device (fmsi_hip_index_build*, len > 2^32 - 8192) */
    uint64_t sk_bytes;          /* 0 = no super-k-mer dictionary (FMSI_HIP_DICT_SUPERKMER), else its HBM footprint incl. the aux dictionary */
    uint64_t sk_records, sk_heavy_minimizers, sk_overflowed_buckets, sk_aux_keys;
    uint32_t sk_log2_buckets, sk_minimizer_len;
} fmsi_hip_index_info;

/* counters written by query calls when `counters` != NULL (8 x u64, accumulated, caller zeroes):
 *   [0] LF steps executed (update_range with i != j, src/fms_index.h:98-103)
 *   [1] interleaved blocks read (64-B units)          [2] mask probes / strand searches ending non-empty
 *   [3] strand searches started                       [4] prefix-table lookups   [5] kLCP words scanned
 *   [6] windows answered                              [7] text-verification reads (position + text) */
#define FMSI_HIP_N_COUNTERS 8

/* ---- device side ------------------------------------------------------------------------------ */
int fmsi_hip_device_count(void);
int fmsi_hip_index_create(const fmsi_index_view* view, int device, fmsi_hip_index** out);
void fmsi_hip_index_destroy(fmsi_hip_index* idx);
int fmsi_hip_index_get_info(const fmsi_hip_index* idx, fmsi_hip_index_info* info);
/* Returns the handle's staging and scratch buffers to the driver (the copies of host-pointer calls' inputs and outputs, the window and
 * task arrays, the batch-sorted walk's sort buffers -- about 66 bytes per k-mer of its sub-batch: 4.5 GB when FMSI_HIP_KERNEL_AUTO chose
 * it, 17-18 GB for an explicit FMSI_HIP_KERNEL_SORTED on 2^28 k-mers and more).  They are grow-only otherwise and stay until the handle is
 * destroyed; the next call allocates what it needs again.  Waits for the device. */
int fmsi_hip_index_release_scratch(fmsi_hip_index* idx);

/* load_index() (src/fms_index.h:502-526) straight into HBM: the files' payloads are uploaded as they are and the device decodes the
 * rrr_vector<63> mask and interleaves the blocks itself -- the same handle as fmsi_host_index_load + fmsi_hip_index_create give, dword
 * for dword, without the host-side decode and layout pass (what the CLI's load time mostly was).  use_klcp: take <prefix>.fmsi.klcp if
 * it exists.  Malformed / missing files: FMSI_HIP_EIO.  Indexes of 2^32 rows and more take the host path internally.
 * `use_klcp` is a set of flags: bit 0 as above; FMSI_HIP_LOAD_ASYNC_TABLE -- the handle is returned as soon as the prefix table's
 * intervals stand, and the signatures of its entries of 1..3 rows (the shortcut that ends a search four bases early, the streaming
 * kernel's absence certificates) are written on a stream of the library's own BESIDE the caller's queries: two thirds of the
 * table's construction (0.11 of 0.17 s at human scale, t = 15).  The pass starts once it can pay for itself: after 4 x 10^9 windows
 * streamed through fmsi_hip_query_stream (its certificates save 0.04 s per 10^9 windows of a mixed read file, the pass costs 0.17 s of
 * device time beside which the queries run at half their rate) or 2.5 x 10^8 k-mers / windows through the other query calls; a shorter
 * job never runs it, and fmsi_hip_index_wait_background runs it at once.  Results never depend on it -- an entry is rewritten by one aligned
 * 8-byte store and both of its forms give the same interval; queries in that time are just not yet at full speed.  Calls that read the
 * table as a whole (fmsi_hip_index_enable_verify / _dict, fmsi_hip_index_destroy) wait for the pass; so does fmsi_hip_index_wait_background. */
#define FMSI_HIP_LOAD_KLCP 0x1
#define FMSI_HIP_LOAD_ASYNC_TABLE 0x2
int fmsi_hip_index_load(const char* prefix, int use_klcp, int device, int prefix_len, fmsi_hip_index** out);
int fmsi_hip_index_wait_background(fmsi_hip_index* idx);

/* On-device construction -- counterpart of construct<T>() (src/fms_index.h:397-460) producing the device layout
 * directly: 2-bit text -> GPU suffix sort -> BWT / SA-transformed mask / kLCP -> interleaved blocks.
 * `ms` is the masked superstring (upper case = mask 1, src/kmers.h:71-73), host memory of any kind (it goes up in slices through page-locked
 * staging buffers; a character outside ACGTacgt: FMSI_HIP_EINVAL with its position).  with_klcp needs k <= 64 (as the reference,
 * src/main.cpp:223-226).  len > 2^32 - 8192 (where the device sort of the 32-bit builder ends, a little before 32-bit rows do): the builder
 * for 64-bit rows -- suffixes dealt into 16 buckets by their first two bases, a sort per bucket with 64-bit positions, tied suffixes ordered
 * group by group or, where groups of more than 64 suffixes agree on 29 bases (repetitive texts), by prefix doubling with 64-bit ranks -- and a
 * handle with rows64 = 1;
 * with_klcp gives it the kLCP array, too (what `fmsi index` writes as .klcp; the handle's own search does not read it). */
int fmsi_hip_index_build(const char* ms, uint64_t len, int k, int with_klcp, int device, int prefix_len, fmsi_hip_index** out);

/* Synthetic workloads (no network on the GPU box, so no real genomes): the superstring is
 * generated in HBM from a counter-based PRNG -- word w (bases 32w .. 32w+31, base p in bits [2(p%32), +1]) is
 * fmsi_hip_synthetic_text_word(seed, w) -- with every k-mer ON (positions 0..len-k upper case: a valid max-ones
 * masked superstring).  The text stays resident so that query k-mers can be drawn from it on the device. */
int fmsi_hip_index_build_synthetic(uint64_t len, uint64_t seed, int k, int with_klcp, int device, int prefix_len, fmsi_hip_index** out);
uint64_t fmsi_hip_synthetic_text_word(uint64_t seed, uint64_t w);
/* The same with repeat families, for a text whose k-mer spectrum is skewed like a genome's: each 2048-base segment is with
 * probability repeat_permille / 1000 a copy of one of `families` consensus segments in which every base is substituted with
 * probability divergence_permille / 1000 (else uniform as above).  Many distinct k-mers then share their last t bases --
 * prefix-table keys with tens of occurrences, which uniform text does not have.  Still every k-mer ON. */
int fmsi_hip_index_build_synthetic_repeats(uint64_t len, uint64_t seed, int k, int with_klcp, int device, int prefix_len,
                                           uint32_t repeat_permille, uint32_t families, uint32_t divergence_permille, fmsi_hip_index** out);
uint64_t fmsi_hip_synthetic_text_word_repeats(uint64_t seed, uint64_t w, uint32_t repeat_permille, uint32_t families, uint32_t divergence_permille);
/* The general form: the text as above, and a MASK that need not be max-ones -- position p <= len - k is ON with probability
 * mask_permille / 1000 (fmsi_hip_synthetic_mask_bit(seed, p, mask_permille); 1000 = every k-mer ON), the last k - 1 positions
 * OFF.  With a mixed mask `-O` answers depend on which occurrence comes first in the suffix array and the default (OR) mode
 * on all of them (src/fms_index.h:126-144): what the at-scale parity tests exercise. */
typedef struct {
    uint64_t len, seed;
    int32_t k, with_klcp, prefix_len;
    uint32_t repeat_permille, families, divergence_permille;
    uint32_t mask_permille;
} fmsi_hip_synthetic_spec;
int fmsi_hip_index_build_synthetic_spec(const fmsi_hip_synthetic_spec* spec, int device, fmsi_hip_index** out);
uint32_t fmsi_hip_synthetic_mask_bit(uint64_t seed, uint64_t p, uint32_t mask_permille);
/* Fills d_kmers[0..n) (DEVICE pointer) with queries first..first+n-1 of the seeded stream: with probability 1/2 a
 * k-mer of the superstring (half of those reverse-complemented), else a uniform random k-mer. */
int fmsi_hip_synthetic_queries(fmsi_hip_index* idx, int k, uint64_t seed, uint64_t first, uint64_t n, uint64_t* d_kmers, void* stream);
/* Fills d_packed (DEVICE pointer, n_reads*read_len/32 + 2 words) with n_reads reads of read_len bases, 2-bit packed back
 * to back: read r (global number first_read + r) is with probability 1/2 a substring of the superstring (half of those
 * reverse-complemented, each base substituted with probability subst_per_million / 10^6), else uniform random. */
int fmsi_hip_synthetic_reads(fmsi_hip_index* idx, uint64_t read_len, uint64_t seed, uint64_t first_read, uint64_t n_reads,
                             uint32_t subst_per_million, uint64_t* d_packed, void* stream);
/* 2-bit text kept by the builder -> host (len/32 + 2 words) */
int fmsi_hip_index_download_text(const fmsi_hip_index* idx, uint64_t* words);

/* Copies the device layout back (n_blocks*16 dwords, n_blocks+1 dwords) -- verification aid. */
int fmsi_hip_index_download_layout(const fmsi_hip_index* idx, uint32_t* blocks, uint32_t* mrank);

/* Adds the optional "pair layout" to a handle (built on the device from the single-step layout, so it works for
 * handles made by fmsi_hip_index_create and by fmsi_hip_index_build*): 128-byte blocks holding, per row, both
 * preceding text symbols and cumulative counts of the 16 symbol pairs, so that ONE block read performs TWO
 * update_range() steps (src/fms_index.h:98-103).  Final intervals and results are unchanged; the number of random
 * block reads per k-mer roughly halves.  Costs 1 byte per row of HBM.  Idempotent. */
int fmsi_hip_index_enable_pair_layout(fmsi_hip_index* idx);

/* Writes the handle's index in the reference's on-disk format, <prefix>.fmsi.{ac_gt,ac,gt,mask,misc} (+ .klcp when
 * the handle has a kLCP array): the writer half of dump_index() (src/fms_index.h:484-500), byte-identical files. */
int fmsi_hip_index_write_files(const fmsi_hip_index* idx, const char* prefix);
/* Same from host bit-vectors (mask = plain bits; it is RRR-63 encoded here). */
int fmsi_host_index_write(const char* prefix, const fmsi_index_view* view);
/* The `fmsi index` command (ms_index, src/main.cpp:172-236) with the suffix sort on the GPU; argv as the reference's
 * sub-command sees it (argv[0] = "index").  Returns the process exit code. */
int fmsi_cli_index(int argc, char** argv);

/* Adds the candidate tables to a handle that has a prefix table (private optimisation, results unchanged): the lookup
 * kernel of the phased path gets its own copy of the table entries -- 32 bytes each when k - t <= 15 and the device has
 * the room (4^t x 32 bytes + 16 GB to spare), else 16 bytes + a 4-byte side array; fmsi_hip_index_info.candidate_entry_bytes tells -- in which
 * every occurrence of a key -- on both strands when the key and its reverse complement occur at most seven (16-byte
 * entries: three) times together, else the forward ones of an interval of up to three rows -- is represented by the
 * k - t bases in front of it and the mask bit of the k-mer it would complete.  A search that comes out of the table with such an entry is settled by comparing its remaining bases with the
 * candidates (one request) instead of (k - t) / 2 dependent block reads: the candidates that match are the rows of the
 * final interval(s), in order.  For k - t > 15 the entries hold text positions and the bases are read from the 2-bit
 * superstring.  Everything comes from the LF walk of fmsi_hip_index_export, so any handle can add it.  Used for the
 * presence modes (not ORDERS).  4^t x 32 (or 20) bytes + n / 4 bytes of HBM.  Idempotent.
 * Environment: FMSI_HIP_NARROW_CANDIDATES forces the 16-byte entries, FMSI_HIP_LOOKUP_PAIRS additionally the lane-pair
 * lookup kernel (A/B measurements). */
int fmsi_hip_index_enable_verify(fmsi_hip_index* idx);

/* Adds the k-mer dictionary to a handle (private optimisation, results unchanged): every distinct canonical k-mer of the
 * superstring with the answers get_range_with_pattern() + infer_presence() (src/fms_index.h:117-144) give for it and for its reverse
 * complement, in buckets of 128 bytes addressed by a hash of the k-mer -- a presence query (either mode, STRICT or not, isolated
 * k-mers, windows, runs) is then ONE random 128-byte read whatever the text's repeats and whether or not the k-mer is present.
 * About 12 bytes of HBM per distinct k-mer (FMSI_HIP_DICT_SPARSE: twice that, no second-level reads), for the index's k only; the
 * superstring and its mask come from the LF walk of fmsi_hip_index_export, answers that depend on the suffix order from the index's
 * own search, so any handle can add it.  ORDERS queries and other k take the other kernels.  Idempotent. */
#define FMSI_HIP_DICT_SPARSE 0x1u
/* ... and the extras of CHAINED lookups (+ about 6.5 bytes per k-mer: a text position per slot, an answer nibble per text position, the
 * 2-bit text): fmsi_hip_query_runs over reads of one length then gives a lane 32 consecutive windows -- it looks the first one up and
 * follows the superstring from its occurrence, a base compare and a nibble per further window; windows that leave the text take the
 * ordinary lookups.  Same answers; several times fewer memory requests on reads that come from the indexed sequences. */
#define FMSI_HIP_DICT_CHAIN 0x2u
/* ... and the SUPER-K-MER dictionary (23 <= k <= 32): the text's k-mers filed by their minimizer (21 or 23 bases, both strands), one 16-byte
 * record per super-k-mer -- the bases around one occurrence of a minimizer and an answer nibble for each of the up to 11 k-mers that share
 * it -- in buckets of 128 bytes addressed by the minimizer.  fmsi_hip_query_runs over reads of one length then fetches a bucket when a
 * lane's minimizer CHANGES, about every sixth window, whether the read is in the text or not; about 5 bytes of HBM per k-mer.  What the
 * scheme does not hold (minimizers with more than four distinct neighbourhoods, overflowing buckets) lives in a small k-mer dictionary
 * beside it.  Same answers.  _ONLY: the k-mer dictionary that the construction goes through is released afterwards (isolated k-mers and
 * runs of differing lengths then take the index's own search). */
#define FMSI_HIP_DICT_SUPERKMER 0x4u
#define FMSI_HIP_DICT_SUPERKMER_ONLY 0x8u
int fmsi_hip_index_enable_dict(fmsi_hip_index* idx, uint32_t flags);

/* Replaces export_ms() (src/fms_index.h:462-473): the masked superstring of the index (upper case = mask 1), written
 * to out[0 .. *out_len) without a terminator.  The reference's single LF chain of |MS| steps is cut into about a million
 * chains walked concurrently on the device.  *out_len is set even when out_capacity is too small (FMSI_HIP_EINVAL). */
int fmsi_hip_index_export(const fmsi_hip_index* idx, char* out, uint64_t out_capacity, uint64_t* out_len);
/* The `fmsi export` command (ms_export, src/main.cpp:508-541); argv as the sub-command sees it. */
int fmsi_cli_export(int argc, char** argv);

/* Copies the handle's kLCP bit-vector back (ceil(n/64) words); fails when the handle has none. */
int fmsi_hip_index_download_klcp(const fmsi_hip_index* idx, uint64_t* words);

/* Replaces query_kmers<orr|all>() without kLCP (query_kmers_single, src/fms_index.h:263-331) for
 * isolated k-mers, k <= 32.  kmers[q] holds k-mer q 2-bit packed (A=0,C=1,G=2,T=3 as in src/kmers.h:3-20),
 * base t (0 = leftmost) in bits [2t, 2t+1]. */
int fmsi_hip_query_kmers(fmsi_hip_index* idx, const uint64_t* kmers, uint64_t n_kmers, int k, uint32_t flags,
                         void* out, uint64_t* counters, void* stream);

/* Same, for windows of a 2-bit packed sequence (base p in bits [2(p%32), 2(p%32)+1] of word p/32; the
 * buffer holds seq_bases / 32 + 2 words: the library copies / reads that many, i.e. up to two words past the last base's).  Window w is the k-mer starting at base
 * window_start[w]; window_start == NULL means windows 0 .. n_windows-1 start at bases 0 .. n_windows-1. */
int fmsi_hip_query_windows(fmsi_hip_index* idx, const uint64_t* packed_seq, uint64_t seq_bases,
                           const uint64_t* window_start, uint64_t n_windows, int k, uint32_t flags,
                           void* out, uint64_t* counters, void* stream);

/* Same windows, given as RUNS of consecutive windows (what a chunk of query_kmers() is, src/main.cpp:337-354): run r
 * covers the windows starting at bases run_start[r] .. run_start[r] + run_windows[r] - 1; results are written
 * consecutively in run order.  run_start / run_windows are HOST arrays in every mode (12 bytes per run instead of
 * 8 bytes per window); the windows are cut on the device (by the lookup kernel itself on the phased path, by a pass
 * of its own before the other kernels).  With the candidate tables this is also the quickest way to the presence
 * answers of `-S` batches, which do not depend on the kLCP array. */
int fmsi_hip_query_runs(fmsi_hip_index* idx, const uint64_t* packed_seq, uint64_t seq_bases,
                        const uint64_t* run_start, const uint32_t* run_windows, uint64_t n_runs, uint64_t n_windows,
                        int k, uint32_t flags, void* out, uint64_t* counters, void* stream);

/* f-masked superstrings, `fmsi query -f and|xor|r-s`: replaces query_kmers<query_mode::general>() (src/fms_index.h:333-342 ->
 * query_kmers_single<general>, :316-327) with the demasking functions of src/functions.h:8-19.  Per window: total = occurrences of
 * the k-mer and of its reverse complement in the superstring (single_query_general(), :171-179, of both strands; a self-complementary
 * k-mer counted once, :319), ones = how many of them have their mask bit ON; the answer is
 *   FMSI_HIP_FN_AND    total != 0 && ones == total          (f_and)
 *   FMSI_HIP_FN_XOR    ones odd                             (f_xor)
 *   FMSI_HIP_FN_RANGE  r <= ones <= s                       (f_r_to_s; r, s as `-f r-s` parses them)
 * with ones and total passed through int as the reference's std::function<bool(int, int)> wrappers pass them.
 * `lower_case`: one bit per base of the sequence (bit p & 63 of word p >> 6, seq_bases / 64 + 1 words; host or device like packed_seq)
 * that is set where the query character is lower case, or NULL for a sequence in one case.  It matters to self-complementary
 * k-mers only: the reference recognises them by comparing the query's characters with their case-preserving reverse complement
 * (AreStringsEqual, src/kmers.h:61-68), so "acGT" is not equal to its reverse complement "ACgt" and is counted TWICE.  Runs, buffers and
 * result layout as fmsi_hip_query_runs (one byte 0 / 1 per window, or FMSI_HIP_OUT_BITS); of `flags` only FMSI_HIP_DEVICE_PTRS,
 * FMSI_HIP_OUT_BITS and FMSI_HIP_NO_PREFIX_TABLE are looked at.  The strand predictor plays no part in this mode (it is asked
 * but never told, and both strands are added up), and `-S` changes nothing (:336).  Any k; not for indexes of 2^32 rows or more. */
#define FMSI_HIP_FN_AND 1
#define FMSI_HIP_FN_XOR 2
#define FMSI_HIP_FN_RANGE 3
int fmsi_hip_query_runs_fn(fmsi_hip_index* idx, const uint64_t* packed_seq, const uint64_t* lower_case, uint64_t seq_bases,
                           const uint64_t* run_start, const uint32_t* run_windows, uint64_t n_runs, uint64_t n_windows,
                           int k, int fn, uint64_t r, uint64_t s, uint32_t flags, void* out, uint64_t* counters, void* stream);

/* Replaces query_kmers_streaming<>() (`-S`, src/fms_index.h:181-254): runs of consecutive windows walked
 * right-to-left with the kLCP extension (src/fms_index.h:106-115).  Run r covers windows starting at bases
 * run_start[r] .. run_start[r] + run_windows[r] - 1; results are written consecutively in run order.
 * Without FMSI_HIP_STRICT the strand order is chosen per task of <= 64 consecutive windows, the way the reference's
 * strand predictor chooses it per chunk: forward strand first, unless the task's last window (where a forward pass starts)
 * is absent on the forward strand and its first window (where a reverse-complement pass starts) is present as reverse
 * complement -- then the reverse-complement strand goes first.  As with the reference's predictor this cannot change an
 * OR-mode answer, nor an ALL-mode answer under the max-ones precondition.
 * FMSI_HIP_ORDERS (`lookup -S`) and 33 <= k <= 64 (the reference's __uint128_t k-mers, src/main.cpp:223-230) are served by the
 * same kernel.  A fresh search that follows a failed window is preceded by a prefix-table request for a (t + 4)-mer of the
 * window that its next k - t - 4 neighbours share: a string the text does not hold settles all of them (exact: a pattern with
 * an absent substring has an empty range, src/fms_index.h:117-124). */
int fmsi_hip_query_stream(fmsi_hip_index* idx, const uint64_t* packed_seq, uint64_t seq_bases,
                          const uint64_t* run_start, const uint32_t* run_windows, uint64_t n_runs,
                          uint64_t n_windows, int k, uint32_t flags, void* out, uint64_t* counters, void* stream);

/* The layout fmsi_hip_query_stream's kernel reads -- 128-byte blocks of 192 rows that carry the kLCP bits of
 * their rows beside the wavelet-tree and mask planes, so that a streamed window is one memory request (4/3 byte per row of HBM) -- is
 * built from the handle's blocks and kLCP array on the first such call; a host that wants the cost at load time calls this.  Idempotent;
 * needs a handle with a kLCP array. */
int fmsi_hip_index_enable_stream_layout(fmsi_hip_index* idx);

int fmsi_hip_sync(fmsi_hip_index* idx, void* stream);
/* The library keeps large freed device blocks (temporaries of the builders) for reuse -- freeing and re-allocating them is what an index
 * load mostly waits for on this platform -- and returns them to the driver when an allocation of its own fails or a handle
 * is destroyed.  A host that needs the HBM for itself in between calls this. */
int fmsi_hip_trim_scratch(int device);
/* By default every builder call (index build / load, pair layout, candidate tables, dictionary, export, file writer) returns that cache to
 * the driver when it ends, so that nothing the handle's info does not count stays on the device.  A host that runs a CHAIN of builders
 * turns that off for the chain -- they then hand each other their temporaries instead of waiting for the allocator -- and trims once:
 *   fmsi_hip_keep_scratch(1); load / build; enable_pair_layout; enable_dict; fmsi_hip_keep_scratch(0); fmsi_hip_trim_scratch(device);
 * Process-wide. */
void fmsi_hip_keep_scratch(int on);
/* bytes of freed blocks the library currently keeps for reuse on `device` (not part of any handle's device_bytes) */
uint64_t fmsi_hip_scratch_bytes(int device);
/* Page-locked host memory for the caller's batch buffers (hipHostMalloc / hipHostFree).  Any host memory works with the
 * query calls; from page-locked memory the copies to and from the device run at the link's rate instead of through the
 * driver's staging (the CLI's device stage: 0.4-0.6 s -> see DESIGN.md 8 per 10^9 windows).  NULL when it cannot be had. */
void* fmsi_hip_host_alloc(uint64_t bytes);
void fmsi_hip_host_free(void* p);

/* per-handle kernel timing with HIP events recorded on the launch stream around every kernel launch */
int fmsi_hip_profile_enable(fmsi_hip_index* idx, int on);
int fmsi_hip_profile_read(fmsi_hip_index* idx, uint64_t* n_launches, double* total_ms, int reset);

const char* fmsi_hip_last_error(void);
int fmsi_hip_abi_version(void);

/* ---- multi-GPU: the result gather -------------------------------------------- */
/* One process per GPU; the index is replicated and each rank answers a contiguous, order-preserving range of the
 * windows (the reference is single-threaded: src/main.cpp:328-373 is one loop; nothing replaces it but the order of
 * its output).  There is no exchange inside the search; the one collective of the path brings the per-rank result
 * buffers to the root over RCCL (xGMI), rank order = output order.
 * fmsi_hip_comm_unique_id: called on ONE rank; the 128 bytes reach the others by the launcher's means (MPI, a file,
 * torch.distributed's store).  fmsi_hip_comm_create: collective over all `world` ranks, each with its own device.
 * RCCL is dlopen()ed on first use: FMSI_HIP_ECOMM when it is absent or reports an error. */
typedef struct fmsi_hip_comm fmsi_hip_comm;
#define FMSI_HIP_COMM_ID_BYTES 128
int fmsi_hip_comm_unique_id(uint8_t* id /* FMSI_HIP_COMM_ID_BYTES */);
int fmsi_hip_comm_create(const uint8_t* id, int world, int rank, int device, fmsi_hip_comm** out);
void fmsi_hip_comm_destroy(fmsi_hip_comm* comm);
/* Rank r's d_send[0, bytes_per_rank[r]) (DEVICE memory) lands at d_recv + sum(bytes_per_rank[0..r)) on `root`
 * (d_recv: device memory of the root, ignored elsewhere).  bytes_per_rank: HOST array of `world` entries, the same on
 * every rank; send_bytes must equal this rank's entry.  Asynchronous on `stream`; sizes may differ and may be 0. */
int fmsi_hip_gather(fmsi_hip_comm* comm, const void* d_send, uint64_t send_bytes, void* d_recv, const uint64_t* bytes_per_rank,
                    int root, void* stream);

/* ---- host side: the loader half of load_index() (src/fms_index.h:502-526) --------------------- */
/* Parses <prefix>.fmsi.{ac_gt,ac,gt,mask,misc} (+ .klcp iff use_klcp and the file exists), decodes the
 * rrr_vector<63> mask to plain bits.  Pure host code (no HIP). */
int fmsi_host_index_load(const char* prefix, int use_klcp, fmsi_host_index** out);
void fmsi_host_index_free(fmsi_host_index* h);
int fmsi_host_index_view(const fmsi_host_index* h, fmsi_index_view* view);
/* rows, k and whether <prefix>.fmsi.klcp exists, from the files' headers alone (what a caller checks before fmsi_hip_index_load) */
int fmsi_host_index_peek(const char* prefix, uint64_t* n, int32_t* k, int32_t* has_klcp_file);

/* The interleaved device layout built on the host (what fmsi_hip_index_create uploads): n_blocks blocks of
 * 16 dwords (64 B) covering 128 rows each -- dw0-3 cumulative #A,#C,#G,#T before the block, dw4-7 hi plane
 * (ac_gt), dw8-11 lo plane (ac / gt bit scattered back to row order), dw12-15 mask plane -- and n_blocks+1
 * cumulative mask popcounts.  Exposed so that the layout can be verified without a GPU. */
uint64_t fmsi_host_layout_blocks(const fmsi_index_view* view);
int fmsi_host_layout_build(const fmsi_index_view* view, uint32_t* blocks /* n_blocks*16 */, uint32_t* mrank /* n_blocks+1 */);

/* 2-bit pack `len` ACGT/acgt characters (src/kmers.h:3-20 codes) into words (caller provides
 * len/32 + 2 words).  Returns the index of the first invalid character or len. */
uint64_t fmsi_host_pack_2bit(const char* seq, uint64_t len, uint64_t* words);

/* The query side's parallel reader of ORDINARY gzip files (one member of many deflate blocks; the reference reads them through
 * zlib's gzread on one thread: src/parser.h:14-27, src/kseq.h:75-93) on its own, for tests and measurements: inflates the file at
 * `path` with `threads` threads (pgunzip.h: block starts found in the middle of the stream, chunks decoded with an unknown window,
 * zlib from the first block boundary whose window is known) into out[0, cap) -- bytes beyond cap are decoded, checked and dropped
 * (out may be NULL) -- and sets *out_len to the length of what it decoded.  chunk_bytes: compressed bytes per chunk (0: its own
 * schedule).  stats7 (optional): rounds, chunks, chunks passed over (a start that was none), chunks that found no start, symbols
 * decoded with an unknown window, bytes decoded by zlib, members.  Returns 0, 1 when the reader gave up at *out_len (a stream that is
 * not what it takes: the CLI lets zlib go on from there), -1 when the file is not a gzip file. */
int fmsi_host_gunzip(const char* path, int threads, uint64_t chunk_bytes, unsigned char* out, uint64_t cap, uint64_t* out_len, uint64_t* stats7);

/* The `fmsi query` / `fmsi lookup` command (ms_query, src/main.cpp:238-375) on the GPU; argv as the
 * reference's sub-command sees it (argv[0] = "query").  Returns the process exit code. */
int fmsi_cli_query(int argc, char** argv, int output_orders);
/* A process that ends with the command (the `fmsi` binary) may register what is to happen the moment the command's output is complete --
 * written, stdout still open -- instead of the return: hook(exit_code) is called from fmsi_cli_query's last line BEFORE its mappings,
 * buffers and device handles are released (unmapping a query file of gigabytes and freeing the batches is 0.1-0.3 s that a process about
 * to exit need not spend); a hook that returns lets the call end the ordinary way.  NULL (default): none. */
void fmsi_cli_set_exit_hook(void (*hook)(int exit_code));

#ifdef __cplusplus
}
#endif
#endif /* FMSI_HIP_H */
