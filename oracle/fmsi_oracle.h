/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's `fmsi query` path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load or execute
 * anything under oracle/; the product (fmsi_amd/) never links, imports or calls it.
 *
 * Parity status: PINNED.  The restatement is checked against (a) the reference's own known-answer
 * tables (tests/fms_index_test.h, committed as tests/golden/kat_fms_index.json), (b) the
 * reference's golden CLI outputs (tests/testfiles/result_*_complements*.txt), (c) index bytes and
 * query outputs produced by the real reference built into oracle/_ref (oracle/Makefile.ref).
 *
 * All file:line citations are relative to /root/reference/.
 */
#ifndef FMSI_ORACLE_H
#define FMSI_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* sdsl::bit_vector (int_vector<1>): n bits, LSB-first in little-endian u64 words
 * (src/sdsl-lite/include/sdsl/int_vector.hpp:1441-1447); one zero pad word is always allocated
 * because rank(size) reads it (memory_management.hpp:351-368). */
typedef struct {
    uint64_t n;   /* bits */
    uint64_t* w;  /* (n+64)/64 words allocated */
} orc_bitvec;

/* sdsl::int_vector<0>: n entries of `width` bits packed LSB-first. */
typedef struct {
    uint64_t n;      /* entries */
    uint8_t width;   /* bits per entry */
    uint64_t bits;   /* n*width */
    uint64_t* w;
} orc_intvec;

/* sdsl::rank_support_v5<1> (rank_support_v5.hpp:65-109): interleaved superblock/sub-block counts. */
typedef struct {
    const orc_bitvec* v;
    uint64_t* basic_block;
    uint64_t n_basic;
} orc_rank_v5;

/* sdsl::rrr_vector<63> (rrr_vector.hpp:110-118, 350-373). */
typedef struct {
    uint64_t size;
    orc_intvec bt;
    orc_bitvec btnr;
    orc_intvec btnrp;
    orc_intvec rank;
    orc_bitvec invert;
} orc_rrr63;

/* strand_predictor (src/fms_index.h:18-49). */
typedef struct {
    int score;
    int result_scores[2];
    int previous;
} orc_predictor;

/* fms_index (src/fms_index.h:52-66). */
typedef struct {
    orc_bitvec ac_gt, ac, gt, klcp;
    orc_rank_v5 ac_gt_rank, ac_rank, gt_rank;
    orc_rrr63 mask;
    uint64_t counts[4];
    uint64_t dollar_position;
    int k;
    orc_predictor predictor;
    /* instrumentation for the roofline formula; not part of the reference */
    uint64_t lf_steps;      /* update_range calls with i != j */
    uint64_t mask_probes;   /* rrr operator[] calls */
    uint64_t klcp_bits;     /* klcp bit reads */
    uint64_t strand_searches;
    /* demasking function of query_mode::general (`-f and|xor|r-s`, src/functions.h:8-57); set by orc_set_function */
    int fn_kind;
    uint64_t fn_r, fn_s;
} orc_index;

enum { ORC_MODE_OR = 0, ORC_MODE_ALL = 1, ORC_MODE_GENERAL = 2 };
enum { ORC_FN_AND = 1, ORC_FN_XOR = 2, ORC_FN_RANGE = 3 };
/* mask_function(name) (src/functions.h:23-57): 0 = a general function (kind, r, s filled in), 1 = "or", 2 = "all" (the two
 * that map to the fast paths), -1 = not recognized (the reference throws std::invalid_argument), -2 = a bound std::stoi rejects
 * (the reference dies of the uncaught std::out_of_range) */
int orc_parse_function(const char* name, int* kind, uint64_t* r, uint64_t* s);
void orc_set_function(orc_index* ix, int kind, uint64_t r, uint64_t s);
/* f(ones, total) as query_kmers_single<general> reaches it (src/fms_index.h:317-327): through std::function<bool(int,int)> wrappers,
 * i.e. both counts truncated to int and widened again */
int orc_apply_function(int kind, uint64_t r, uint64_t s, uint64_t ones, uint64_t total);
/* single_query_general of both strands of one k-mer, a self-complementary k-mer counted once (src/fms_index.h:171-179, 317-323) */
void orc_kmer_general(orc_index* ix, const char* kmer, int k, uint64_t* ones, uint64_t* total);

/* growable output buffer standing in for std::ostream */
typedef struct {
    char* p;
    size_t len, cap;
} orc_obuf;
void orc_obuf_init(orc_obuf* o);
void orc_obuf_free(orc_obuf* o);

/* ---- primitives ---- */
uint64_t orc_rank_v5_rank(const orc_rank_v5* r, uint64_t idx);
int orc_rrr_access(const orc_rrr63* v, uint64_t i);
uint64_t orc_rrr_rank(const orc_rrr63* v, uint64_t i);
int orc_rrr_build(orc_rrr63* out, const orc_bitvec* bv);
void orc_rrr_decode_all(const orc_rrr63* v, orc_bitvec* out);

/* ---- index ---- */
orc_index* orc_index_load(const char* prefix, int use_klcp);       /* fms_index.h:502-526 */
int orc_index_dump(const orc_index* ix, const char* prefix);         /* fms_index.h:484-500 */
orc_index* orc_index_construct(const char* ms, uint64_t len, int k, int use_klcp); /* fms_index.h:397-460 */
orc_index* orc_index_from_planes(const uint64_t* hi, const uint64_t* lo, const uint64_t* mask, uint64_t n,
                                 const uint64_t* counts4, uint64_t dollar_position, int k);
orc_index* orc_index_literal(const uint8_t* ac_gt, uint64_t n_ac_gt, const uint8_t* ac, uint64_t n_ac,
                             const uint8_t* gt, uint64_t n_gt, const uint8_t* mask, uint64_t n_mask,
                             const uint64_t* counts4, uint64_t dollar_position,
                             const uint8_t* klcp, uint64_t n_klcp, int k);
void orc_index_free(orc_index* ix);
uint64_t orc_index_n(const orc_index* ix);
int orc_index_k(const orc_index* ix);
int orc_index_has_klcp(const orc_index* ix);
void orc_predictor_reset(orc_index* ix);
void orc_counters_reset(orc_index* ix);
void orc_counters_get(const orc_index* ix, uint64_t out4[4]);
/* raw views for building other layouts from an oracle-loaded index in tests */
const uint64_t* orc_index_words(const orc_index* ix, int which /*0 ac_gt,1 ac,2 gt,3 klcp*/, uint64_t* nbits);
uint64_t orc_index_count(const orc_index* ix, int c);
uint64_t orc_index_dollar(const orc_index* ix);

/* ---- query path ---- */
uint64_t orc_rank(orc_index* ix, uint64_t i, int c);                                  /* fms_index.h:68-86 */
uint64_t orc_rank1(orc_index* ix, int which, uint64_t i);
void orc_update_range(orc_index* ix, uint64_t* i, uint64_t* j, int c);                /* 98-103 */
void orc_extend_range_with_klcp(orc_index* ix, uint64_t* i, uint64_t* j);             /* 106-115 */
void orc_get_range_with_pattern(orc_index* ix, const char* pattern, int k, uint64_t* i, uint64_t* j); /* 117-124 */
int orc_mask_at(orc_index* ix, uint64_t i);
uint64_t orc_mask_rank(orc_index* ix, uint64_t i);
int orc_infer_presence(orc_index* ix, int maximized_ones, uint64_t i, uint64_t j);    /* 126-144 */
int64_t orc_kmer_order_if_present(orc_index* ix, uint64_t i, uint64_t j);             /* 150-156 */
void orc_query_kmers(orc_index* ix, const char* seq, uint64_t len, int k, int has_klcp, int mode,
                     int output_orders, orc_obuf* of);                                  /* 333-342 */
void orc_query_kmers_streaming(orc_index* ix, const char* seq, uint64_t len, int k, int maximized_ones,
                               int output_orders, orc_obuf* of);                        /* 181-254 */
/* both strands' ternary results of one k-mer, independent of the predictor (for strict-mode checks) */
void orc_kmer_ternaries(orc_index* ix, const char* kmer, int k, int maximized_ones, int* fwd, int* rc);

void orc_kmers_ternaries_bulk(orc_index* ix, const char* kmers, uint64_t n, int k, int maximized_ones,
                               int8_t* fwd, int8_t* rc, int64_t* ord_fwd, int64_t* ord_rc);

/* record loop of ms_query (src/main.cpp:328-373) over a FASTA/FASTQ (optionally gz) file */
int64_t orc_query_file(orc_index* ix, const char* query_fn, int k, int has_klcp, int mode,
                       int output_orders, orc_obuf* of, double* query_seconds, uint64_t* windows);
/* same record loop over an in-memory text of records (used for fuzzing the formatter) */
int64_t orc_query_text(orc_index* ix, const char* fasta_text, uint64_t text_len, int k, int has_klcp,
                       int mode, int output_orders, orc_obuf* of);

/* ctypes-friendly wrappers: copy result text to out (cap bytes); return length or -needed */
int64_t orc_query_kmers_buf(orc_index* ix, const char* seq, uint64_t len, int k, int has_klcp, int mode,
                            int output_orders, char* out, uint64_t cap);
int64_t orc_query_kmers_streaming_buf(orc_index* ix, const char* seq, uint64_t len, int k, int maximized_ones,
                                      int output_orders, char* out, uint64_t cap);
int64_t orc_query_file_buf(orc_index* ix, const char* query_fn, int k, int has_klcp, int mode,
                           int output_orders, char* out, uint64_t cap, double* query_seconds);
/* time query_kmers over n isolated k-mers (n*k chars); one '0'/'1' per k-mer into out if non-NULL */
double orc_time_isolated_kmers(orc_index* ix, const char* kmers, uint64_t n, int k, int mode, char* out);

#ifdef __cplusplus
}
#endif
#endif
