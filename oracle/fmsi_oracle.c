/* TEST INFRASTRUCTURE ONLY -- see fmsi_oracle.h.  CPU restatement (plain C99) of the reference's
 * `fmsi query` path: sdsl bit-vector / int-vector files, rank_support_v5<1>, rrr_vector<63> access +
 * rank, the 3-node wavelet-tree rank, LF step, kLCP extension, backward search, presence inference,
 * strand predictor, single + streaming query loops, the ms_query record loop; plus index
 * construction / dump so tests can make indexes without the reference.
 *
 * Every function cites the reference file:line (relative to /root/reference/) it restates.
 * Nothing here is copied text: the algorithms are re-expressed in C over plain structs.
 */
#define _POSIX_C_SOURCE 200809L
#include "fmsi_oracle.h"

#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <zlib.h>

/* ------------------------------------------------------------------------------------------------
 * tables: src/kmers.h:3-38
 * ---------------------------------------------------------------------------------------------- */
static uint8_t NT2INT[256];
static char COMPL[256];
static uint64_t BINOM[65][65];   /* rrr_helper.hpp:173-208 binomial_table<64,uint64_t> */
static uint16_t SPACE63[64];     /* rrr_helper.hpp:247-257 space[k] for n = 63 */
static int tables_ready = 0;

static uint32_t bits_hi(uint64_t x) { /* bits.hpp:392-414: index of highest set bit, hi(0) = 0 */
    uint32_t r = 0;
    while (x >>= 1) ++r;
    return r;
}

static void tables_init(void) {
    if (tables_ready) return;
    memset(NT2INT, 4, sizeof NT2INT);
    NT2INT['A'] = NT2INT['a'] = 0;
    NT2INT['C'] = NT2INT['c'] = 1;
    NT2INT['G'] = NT2INT['g'] = 2;
    NT2INT['T'] = NT2INT['t'] = 3;
    memset(COMPL, 'N', sizeof COMPL);
    COMPL['A'] = 'T'; COMPL['C'] = 'G'; COMPL['G'] = 'C'; COMPL['T'] = 'A';
    COMPL['a'] = 't'; COMPL['c'] = 'g'; COMPL['g'] = 'c'; COMPL['t'] = 'a';
    memset(BINOM, 0, sizeof BINOM);
    for (int nn = 0; nn <= 64; ++nn) BINOM[nn][0] = 1;
    for (int nn = 1; nn <= 64; ++nn)
        for (int k = 1; k <= 64; ++k) BINOM[nn][k] = BINOM[nn - 1][k - 1] + BINOM[nn - 1][k];
    for (int k = 0; k <= 63; ++k) SPACE63[k] = (BINOM[63][k] == 1) ? 0 : (uint16_t)(bits_hi(BINOM[63][k]) + 1);
    tables_ready = 1;
}
__attribute__((constructor)) static void tables_ctor(void) { tables_init(); }

/* bits.hpp:248-265, the branch the as-shipped build takes (no __SSE4_2__): SWAR popcount */
static inline uint64_t bits_cnt(uint64_t x) {
    x = x - ((x >> 1) & 0x5555555555555555ull);
    x = (x & 0x3333333333333333ull) + ((x >> 2) & 0x3333333333333333ull);
    x = (x + (x >> 4)) & 0x0f0f0f0f0f0f0f0full;
    return (0x0101010101010101ull * x) >> 56;
}
static inline uint64_t lo_set(unsigned len) { return len >= 64 ? ~0ull : ((1ull << len) - 1); }

/* bits.hpp:501-511 read_int */
static inline uint64_t read_int(const uint64_t* word, unsigned offset, unsigned len) {
    uint64_t w1 = (*word) >> offset;
    if (offset + len > 64) return w1 | ((word[1] & lo_set((offset + len) & 0x3F)) << (64 - offset));
    return w1 & lo_set(len);
}
static inline void write_int(uint64_t* word, uint64_t x, unsigned offset, unsigned len) {
    if (len == 0) return;
    x &= lo_set(len);
    if (offset + len <= 64) {
        word[0] = (word[0] & ~(lo_set(len) << offset)) | (x << offset);
    } else {
        unsigned lo = 64 - offset;
        word[0] = (word[0] & lo_set(offset)) | (x << offset);
        word[1] = (word[1] & ~lo_set(len - lo)) | (x >> lo);
    }
}

/* ------------------------------------------------------------------------------------------------
 * bit_vector / int_vector<0>: int_vector.hpp:416-419 (capacity), 593-609 (header), 1563-1595 (io)
 * ---------------------------------------------------------------------------------------------- */
static int bv_alloc(orc_bitvec* b, uint64_t n) {
    b->n = n;
    b->w = (uint64_t*)calloc((size_t)((n + 64) >> 6) + 1, 8); /* +pad word: memory_management.hpp:351-368 */
    return b->w ? 0 : -1;
}
static void bv_free(orc_bitvec* b) { free(b->w); b->w = NULL; b->n = 0; }
static inline int bv_get(const orc_bitvec* b, uint64_t i) { return (int)((b->w[i >> 6] >> (i & 63)) & 1); }
static inline void bv_set(orc_bitvec* b, uint64_t i, int v) {
    if (v) b->w[i >> 6] |= 1ull << (i & 63);
    else b->w[i >> 6] &= ~(1ull << (i & 63));
}
static inline uint64_t bv_get_int(const orc_bitvec* b, uint64_t idx, unsigned len) { /* int_vector.hpp:1331-1342 */
    return read_int(b->w + (idx >> 6), (unsigned)(idx & 0x3F), len);
}
static int iv_alloc(orc_intvec* v, uint64_t n, uint8_t width) {
    v->n = n; v->width = width; v->bits = n * width;
    v->w = (uint64_t*)calloc((size_t)((v->bits + 64) >> 6) + 1, 8);
    return v->w ? 0 : -1;
}
static void iv_free(orc_intvec* v) { free(v->w); v->w = NULL; v->n = 0; }
static inline uint64_t iv_get(const orc_intvec* v, uint64_t i) {
    uint64_t p = i * v->width;
    return read_int(v->w + (p >> 6), (unsigned)(p & 63), v->width);
}
static inline void iv_set(orc_intvec* v, uint64_t i, uint64_t x) {
    uint64_t p = i * v->width;
    write_int(v->w + (p >> 6), x, (unsigned)(p & 63), v->width);
}

static int bv_read(orc_bitvec* b, FILE* f) {
    uint64_t n;
    if (fread(&n, 8, 1, f) != 1) return -1;
    if (bv_alloc(b, n)) return -1;
    size_t words = (size_t)((n + 63) >> 6);
    if (words && fread(b->w, 8, words, f) != words) return -1;
    return 0;
}
static int bv_write(const orc_bitvec* b, FILE* f) {
    size_t words = (size_t)((b->n + 63) >> 6);
    if (fwrite(&b->n, 8, 1, f) != 1) return -1;
    if (words && fwrite(b->w, 8, words, f) != words) return -1;
    return 0;
}
static int iv_read(orc_intvec* v, FILE* f) {
    uint64_t bits; uint8_t width;
    if (fread(&bits, 8, 1, f) != 1 || fread(&width, 1, 1, f) != 1) return -1;
    if (width == 0) return -1;
    if (iv_alloc(v, bits / width, width)) return -1;
    v->bits = bits;
    size_t words = (size_t)((bits + 63) >> 6);
    if (words && fread(v->w, 8, words, f) != words) return -1;
    return 0;
}
static int iv_write(const orc_intvec* v, FILE* f) {
    size_t words = (size_t)((v->bits + 63) >> 6);
    if (fwrite(&v->bits, 8, 1, f) != 1 || fwrite(&v->width, 1, 1, f) != 1) return -1;
    if (words && fwrite(v->w, 8, words, f) != words) return -1;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * rank_support_v5<1>: rank_support_v5.hpp:65-109 (ctor), 116-134 (rank)
 * ---------------------------------------------------------------------------------------------- */
static int rank_v5_build(orc_rank_v5* r, const orc_bitvec* v) {
    r->v = v;
    if (v->n == 0) { /* :69-71 */
        r->n_basic = 2;
        r->basic_block = (uint64_t*)calloc(2, 8);
        return r->basic_block ? 0 : -1;
    }
    uint64_t capacity = ((v->n + 63) >> 6) << 6;
    uint64_t nb = ((capacity >> 11) + 1) << 1;      /* :73 */
    uint64_t* bb = (uint64_t*)calloc((size_t)nb + 2, 8);
    if (!bb) return -1;
    r->basic_block = bb; r->n_basic = nb;
    const uint64_t* data = v->w;
    uint64_t i, j = 0;
    bb[0] = bb[1] = 0;
    uint64_t sum = bits_cnt(*data);
    uint64_t second_level_cnt = 0;
    uint64_t cnt_words = 1;
    for (i = 1; i < (capacity >> 6); ++i, ++cnt_words) {   /* :84-95 */
        if (cnt_words == 32) {
            j += 2;
            bb[j - 1] = second_level_cnt;
            bb[j] = bb[j - 2] + sum;
            second_level_cnt = sum = cnt_words = 0;
        } else if ((cnt_words % 6) == 0) {
            second_level_cnt |= sum << (60 - 12 * (cnt_words / 6));
        }
        sum += bits_cnt(*(++data));
    }
    if ((cnt_words % 6) == 0) second_level_cnt |= sum << (60 - 12 * (cnt_words / 6));  /* :97-99 */
    if (cnt_words == 32) {                                                            /* :100-107 */
        j += 2;
        bb[j - 1] = second_level_cnt;
        bb[j] = bb[j - 2] + sum;
        bb[j + 1] = 0;
    } else {
        bb[j + 1] = second_level_cnt;
    }
    return 0;
}

uint64_t orc_rank_v5_rank(const orc_rank_v5* r, uint64_t idx) { /* rank_support_v5.hpp:116-134 */
    const uint64_t* data = r->v->w;
    const uint64_t* p = r->basic_block + ((idx >> 10) & 0xFFFFFFFFFFFFFFFEull);
    uint64_t result = *p + ((*(p + 1) >> (60 - 12 * ((idx & 0x7FF) / (64 * 6)))) & 0x7FFull)
                      + bits_cnt(data[idx >> 6] & lo_set((unsigned)(idx & 0x3F)));   /* word_rank, rank_support.hpp:149-152 */
    idx -= (idx & 0x3F);
    uint8_t to_do = (uint8_t)(((idx >> 6) & 0x1Full) % 6);
    --idx;
    while (to_do) {
        result += bits_cnt(data[idx >> 6]);   /* full_word_rank, rank_support.hpp:154-157 */
        --to_do;
        idx -= 64;
    }
    return result;
}

/* ------------------------------------------------------------------------------------------------
 * rrr_vector<63> + rrr_helper<63> + rank_support_rrr<1,63>
 * ---------------------------------------------------------------------------------------------- */
#define RRR_BS 63
#define RRR_K 32
#define RRR_BIN_SEARCH_THRESHOLD 10 /* n / MAX_LOG = 63 / 6, rrr_helper.hpp:242 */

static uint64_t rrr_bin_to_nr(uint64_t bin) { /* rrr_helper.hpp:304-320 */
    if (bin == 0 || bin == lo_set(RRR_BS)) return 0;
    uint64_t nr = 0;
    unsigned k = (unsigned)bits_cnt(bin);
    unsigned nn = RRR_BS;
    while (bin != 0) {
        if (bin & 1ull) { nr += BINOM[nn - 1][k]; --k; }
        bin >>= 1;
        --nn;
    }
    return nr;
}

static int rrr_decode_bit(unsigned k, uint64_t nr, unsigned off) { /* rrr_helper.hpp:323-371 */
    const unsigned n = RRR_BS;
    if (k == n) return 1;
    if (k == 0) return 0;
    if (k == 1) return (n - nr - 1) == off;
    unsigned nn = n;
    if (k + 1 < RRR_BIN_SEARCH_THRESHOLD + 1) {
        while (k > 1) {
            unsigned nn_lb = k, nn_rb = nn + 1;
            while (nn_lb < nn_rb) {
                unsigned nn_mid = (nn_lb + nn_rb) / 2;
                if (nr >= BINOM[nn_mid - 1][k]) nn_lb = nn_mid + 1;
                else nn_rb = nn_mid;
            }
            nn = nn_lb - 1;
            if (n - nn >= off) return (n - nn) == off;
            nr -= BINOM[nn - 1][k];
            --k; --nn;
        }
    } else {
        unsigned i = 0;
        while (k > 1) {
            if (i > off) return 0;
            if (nr >= BINOM[nn - 1][k]) {
                nr -= BINOM[nn - 1][k];
                --k;
                if (i == off) return 1;
            }
            --nn; ++i;
        }
    }
    return (n - nr - 1) == off;
}

static unsigned rrr_decode_popcount(unsigned k, uint64_t nr, unsigned off) { /* rrr_helper.hpp:411-460 */
    const unsigned n = RRR_BS;
    if (k == n) return off;
    if (k == 0) return 0;
    if (k == 1) return (n - nr - 1) < off;
    unsigned result = 0, nn = n;
    if (k + 1 < RRR_BIN_SEARCH_THRESHOLD + 1) {
        while (k > 1) {
            unsigned nn_lb = k, nn_rb = nn + 1;
            while (nn_lb < nn_rb) {
                unsigned nn_mid = (nn_lb + nn_rb) / 2;
                if (nr >= BINOM[nn_mid - 1][k]) nn_lb = nn_mid + 1;
                else nn_rb = nn_mid;
            }
            nn = nn_lb - 1;
            if (n - nn >= off) return result;
            ++result;
            nr -= BINOM[nn - 1][k];
            --k; --nn;
        }
    } else {
        unsigned i = 0;
        while (k > 1) {
            if (i >= off) return result;
            if (nr >= BINOM[nn - 1][k]) { nr -= BINOM[nn - 1][k]; --k; ++result; }
            --nn; ++i;
        }
    }
    return result + ((n - nr - 1) < off);
}

int orc_rrr_access(const orc_rrr63* v, uint64_t i) { /* rrr_vector.hpp:257-277 */
    uint64_t bt_idx = i / RRR_BS;
    unsigned bt = (unsigned)iv_get(&v->bt, bt_idx);
    uint64_t sample_pos = bt_idx / RRR_K;
    if (bv_get(&v->invert, sample_pos)) bt = RRR_BS - bt;
    if (bt == 0 || bt == RRR_BS) return bt > 0;
    unsigned off = (unsigned)(i % RRR_BS);
    uint64_t btnrp = iv_get(&v->btnrp, sample_pos);
    for (uint64_t j = sample_pos * RRR_K; j < bt_idx; ++j) btnrp += SPACE63[iv_get(&v->bt, j)];
    unsigned btnrlen = SPACE63[bt];
    uint64_t btnr = bv_get_int(&v->btnr, btnrp, btnrlen);
    return rrr_decode_bit(bt, btnr, off);
}

uint64_t orc_rrr_rank(const orc_rrr63* v, uint64_t i) { /* rrr_vector.hpp:445-481 (t_b = 1: adjust_rank is identity) */
    uint64_t bt_idx = i / RRR_BS;
    uint64_t sample_pos = bt_idx / RRR_K;
    uint64_t btnrp = iv_get(&v->btnrp, sample_pos);
    uint64_t rank = iv_get(&v->rank, sample_pos);
    if (sample_pos + 1 < v->rank.n) {
        uint64_t diff_rank = iv_get(&v->rank, sample_pos + 1) - rank;
        if (diff_rank == 0) return rank;
        if (diff_rank == (uint64_t)RRR_BS * RRR_K) return rank + i - sample_pos * RRR_K * RRR_BS;
    }
    int inv = bv_get(&v->invert, sample_pos);
    for (uint64_t j = sample_pos * RRR_K; j < bt_idx; ++j) {
        unsigned r = (unsigned)iv_get(&v->bt, j);
        rank += inv ? RRR_BS - r : r;
        btnrp += SPACE63[r];
    }
    unsigned off = (unsigned)(i % RRR_BS);
    if (!off) return rank;
    unsigned bt = (unsigned)iv_get(&v->bt, bt_idx);
    if (inv) bt = RRR_BS - bt;
    unsigned btnrlen = SPACE63[bt];
    uint64_t btnr = bv_get_int(&v->btnr, btnrp, btnrlen);
    return rank + rrr_decode_popcount(bt, btnr, off);
}

int orc_rrr_build(orc_rrr63* out, const orc_bitvec* bv) { /* rrr_vector.hpp:145-238 */
    memset(out, 0, sizeof *out);
    out->size = bv->n;
    uint64_t m_size = bv->n;
    uint64_t n_bt = (m_size + RRR_BS) / RRR_BS;             /* :148 incl. dummy block when n % 63 == 0 */
    orc_intvec bt_array;
    if (iv_alloc(&bt_array, n_bt, (uint8_t)(bits_hi(RRR_BS) + 1))) return -1;
    uint64_t pos = 0, i = 0, x = 0, btnr_pos = 0, sum_rank = 0;
    while (pos + RRR_BS <= m_size) {                        /* :158-163 */
        x = bits_cnt(bv_get_int(bv, pos, RRR_BS));
        iv_set(&bt_array, i++, x);
        sum_rank += x;
        btnr_pos += SPACE63[x];
        pos += RRR_BS;
    }
    if (pos < m_size) {                                     /* :164-168 */
        x = bits_cnt(bv_get_int(bv, pos, (unsigned)(m_size - pos)));
        iv_set(&bt_array, i++, x);
        sum_rank += x;
        btnr_pos += SPACE63[x];
    }
    uint64_t n_super = (n_bt + RRR_K - 1) / RRR_K;
    if (bv_alloc(&out->btnr, btnr_pos > 64 ? btnr_pos : 64)) return -1;                      /* :169 */
    if (iv_alloc(&out->btnrp, n_super, (uint8_t)(bits_hi(btnr_pos) + 1))) return -1;          /* :170 */
    if (iv_alloc(&out->rank, n_super + ((m_size % (RRR_K * RRR_BS)) > 0), (uint8_t)(bits_hi(sum_rank) + 1))) return -1; /* :171 */
    if (bv_alloc(&out->invert, n_super)) return -1;                                          /* :174 */
    pos = 0; i = 0; btnr_pos = 0; sum_rank = 0;
    int invert = 0;
    while (pos + RRR_BS <= m_size) {                        /* :180-213 */
        if ((i % RRR_K) == 0) {
            iv_set(&out->btnrp, i / RRR_K, btnr_pos);
            iv_set(&out->rank, i / RRR_K, sum_rank);
            if (i + RRR_K <= n_bt) {
                uint64_t gt_half = 0;
                for (uint64_t j = i; j < i + RRR_K; ++j)
                    if (iv_get(&bt_array, j) > RRR_BS / 2) ++gt_half;
                if (gt_half > (RRR_K / 2)) {
                    bv_set(&out->invert, i / RRR_K, 1);
                    for (uint64_t j = i; j < i + RRR_K; ++j) iv_set(&bt_array, j, RRR_BS - iv_get(&bt_array, j));
                    invert = 1;
                } else invert = 0;
            } else invert = 0;
        }
        x = iv_get(&bt_array, i++);
        unsigned space = SPACE63[x];
        sum_rank += invert ? (RRR_BS - x) : x;
        if (space) {
            uint64_t bin = bv_get_int(bv, pos, RRR_BS);
            write_int(out->btnr.w + (btnr_pos >> 6), rrr_bin_to_nr(bin), (unsigned)(btnr_pos & 63), space);
        }
        btnr_pos += space;
        pos += RRR_BS;
    }
    if (pos < m_size) {                                     /* :214-231 */
        if ((i % RRR_K) == 0) {
            iv_set(&out->btnrp, i / RRR_K, btnr_pos);
            iv_set(&out->rank, i / RRR_K, sum_rank);
            bv_set(&out->invert, i / RRR_K, 0);
            invert = 0;
        }
        x = iv_get(&bt_array, i++);
        unsigned space = SPACE63[x];
        sum_rank += invert ? (RRR_BS - x) : x;
        if (space) {
            uint64_t bin = bv_get_int(bv, pos, (unsigned)(m_size - pos));
            write_int(out->btnr.w + (btnr_pos >> 6), rrr_bin_to_nr(bin), (unsigned)(btnr_pos & 63), space);
        }
        btnr_pos += space;
    }
    iv_set(&out->rank, out->rank.n - 1, sum_rank);           /* :236 */
    out->bt = bt_array;
    return 0;
}

void orc_rrr_decode_all(const orc_rrr63* v, orc_bitvec* out) {
    bv_alloc(out, v->size);
    for (uint64_t i = 0; i < v->size; ++i)
        if (orc_rrr_access(v, i)) bv_set(out, i, 1);
}

static int rrr_read(orc_rrr63* v, FILE* f) { /* rrr_vector.hpp:365-373 */
    if (fread(&v->size, 8, 1, f) != 1) return -1;
    if (iv_read(&v->bt, f) || bv_read(&v->btnr, f) || iv_read(&v->btnrp, f) || iv_read(&v->rank, f) ||
        bv_read(&v->invert, f)) return -1;
    return 0;
}
static int rrr_write(const orc_rrr63* v, FILE* f) { /* rrr_vector.hpp:350-362 */
    if (fwrite(&v->size, 8, 1, f) != 1) return -1;
    if (iv_write(&v->bt, f) || bv_write(&v->btnr, f) || iv_write(&v->btnrp, f) || iv_write(&v->rank, f) ||
        bv_write(&v->invert, f)) return -1;
    return 0;
}
static void rrr_free(orc_rrr63* v) {
    iv_free(&v->bt); bv_free(&v->btnr); iv_free(&v->btnrp); iv_free(&v->rank); bv_free(&v->invert);
}

/* ------------------------------------------------------------------------------------------------
 * output buffer
 * ---------------------------------------------------------------------------------------------- */
void orc_obuf_init(orc_obuf* o) { o->p = NULL; o->len = o->cap = 0; }
void orc_obuf_free(orc_obuf* o) { free(o->p); o->p = NULL; o->len = o->cap = 0; }
static void ob_reserve(orc_obuf* o, size_t extra) {
    if (o->len + extra + 1 > o->cap) {
        size_t nc = o->cap ? o->cap * 2 : 4096;
        while (nc < o->len + extra + 1) nc *= 2;
        o->p = (char*)realloc(o->p, nc);
        o->cap = nc;
    }
}
static inline void ob_putc(orc_obuf* o, char c) { ob_reserve(o, 1); o->p[o->len++] = c; }
static void ob_puts(orc_obuf* o, const char* s) { size_t n = strlen(s); ob_reserve(o, n); memcpy(o->p + o->len, s, n); o->len += n; }
static void ob_puti(orc_obuf* o, int64_t v) { char t[32]; snprintf(t, sizeof t, "%lld", (long long)v); ob_puts(o, t); }

/* ------------------------------------------------------------------------------------------------
 * index: load / dump / literal / free
 * ---------------------------------------------------------------------------------------------- */
static void build_ranks(orc_index* ix) {
    rank_v5_build(&ix->ac_gt_rank, &ix->ac_gt);
    rank_v5_build(&ix->ac_rank, &ix->ac);
    rank_v5_build(&ix->gt_rank, &ix->gt);
}

void orc_predictor_reset(orc_index* ix) {
    ix->predictor.score = 0; ix->predictor.result_scores[0] = ix->predictor.result_scores[1] = 0;
    ix->predictor.previous = -1;
}
void orc_counters_reset(orc_index* ix) { ix->lf_steps = ix->mask_probes = ix->klcp_bits = ix->strand_searches = 0; }
void orc_counters_get(const orc_index* ix, uint64_t out4[4]) {
    out4[0] = ix->lf_steps; out4[1] = ix->mask_probes; out4[2] = ix->klcp_bits; out4[3] = ix->strand_searches;
}

static FILE* open_part(const char* prefix, const char* ext, const char* mode) {
    size_t n = strlen(prefix) + strlen(ext) + 8;
    char* p = (char*)malloc(n);
    snprintf(p, n, "%s.fmsi.%s", prefix, ext);
    FILE* f = fopen(p, mode);
    free(p);
    return f;
}

orc_index* orc_index_load(const char* prefix, int use_klcp) { /* fms_index.h:502-526 */
    tables_init();
    orc_index* ix = (orc_index*)calloc(1, sizeof *ix);
    FILE* f;
    int ok = 1;
    if ((f = open_part(prefix, "ac_gt", "rb"))) { ok &= !bv_read(&ix->ac_gt, f); fclose(f); } else ok = 0;
    if ((f = open_part(prefix, "ac", "rb"))) { ok &= !bv_read(&ix->ac, f); fclose(f); } else ok = 0;
    if ((f = open_part(prefix, "gt", "rb"))) { ok &= !bv_read(&ix->gt, f); fclose(f); } else ok = 0;
    if ((f = open_part(prefix, "mask", "rb"))) { ok &= !rrr_read(&ix->mask, f); fclose(f); } else ok = 0;
    if (use_klcp && (f = open_part(prefix, "klcp", "rb"))) { ok &= !bv_read(&ix->klcp, f); fclose(f); }
    if ((f = open_part(prefix, "misc", "r"))) {
        unsigned long long d, c[4]; int k;
        if (fscanf(f, "%llu %llu %llu %llu %llu %d", &d, &c[0], &c[1], &c[2], &c[3], &k) == 6) {
            ix->dollar_position = d;
            for (int i = 0; i < 4; ++i) ix->counts[i] = c[i];
            ix->k = k;
        } else ok = 0;
        fclose(f);
    } else ok = 0;
    if (!ok) { orc_index_free(ix); return NULL; }
    if (!ix->klcp.w) bv_alloc(&ix->klcp, 0);
    build_ranks(ix);
    orc_predictor_reset(ix);
    return ix;
}

int orc_index_dump(const orc_index* ix, const char* prefix) { /* fms_index.h:484-500 */
    FILE* f;
    if (!(f = open_part(prefix, "ac_gt", "wb"))) return -1;
    bv_write(&ix->ac_gt, f); fclose(f);
    if (!(f = open_part(prefix, "ac", "wb"))) return -1;
    bv_write(&ix->ac, f); fclose(f);
    if (!(f = open_part(prefix, "gt", "wb"))) return -1;
    bv_write(&ix->gt, f); fclose(f);
    if (!(f = open_part(prefix, "mask", "wb"))) return -1;
    rrr_write(&ix->mask, f); fclose(f);
    if (ix->klcp.n > 0) {
        if (!(f = open_part(prefix, "klcp", "wb"))) return -1;
        bv_write(&ix->klcp, f); fclose(f);
    }
    if (!(f = open_part(prefix, "misc", "w"))) return -1;
    fprintf(f, "%llu\n", (unsigned long long)ix->dollar_position);
    for (int i = 0; i < 4; ++i) fprintf(f, "%llu\n", (unsigned long long)ix->counts[i]);
    fprintf(f, "%d\n", ix->k);
    fclose(f);
    return 0;
}

static void bv_from_bytes(orc_bitvec* b, const uint8_t* bytes, uint64_t n) {
    bv_alloc(b, n);
    for (uint64_t i = 0; i < n; ++i) if (bytes[i]) bv_set(b, i, 1);
}

orc_index* orc_index_literal(const uint8_t* ac_gt, uint64_t n_ac_gt, const uint8_t* ac, uint64_t n_ac,
                             const uint8_t* gt, uint64_t n_gt, const uint8_t* mask, uint64_t n_mask,
                             const uint64_t* counts4, uint64_t dollar_position,
                             const uint8_t* klcp, uint64_t n_klcp, int k) {
    tables_init();
    orc_index* ix = (orc_index*)calloc(1, sizeof *ix);
    bv_from_bytes(&ix->ac_gt, ac_gt, n_ac_gt);
    bv_from_bytes(&ix->ac, ac, n_ac);
    bv_from_bytes(&ix->gt, gt, n_gt);
    orc_bitvec m;
    bv_from_bytes(&m, mask, n_mask);
    orc_rrr_build(&ix->mask, &m);
    bv_free(&m);
    bv_from_bytes(&ix->klcp, klcp, n_klcp);
    for (int i = 0; i < 4; ++i) ix->counts[i] = counts4[i];
    ix->dollar_position = dollar_position;
    ix->k = k;
    build_ranks(ix);
    orc_predictor_reset(ix);
    return ix;
}

/* Index from row-order bit planes (hi = ac_gt bit, lo = second wavelet-tree level bit of the row, mask =
 * SA-transformed mask), n rows each, given as u64 words.  Used by bench.py's cpu_baseline leg to hand the CPU
 * reference the very index the GPU built: ac/gt are the lo bits of the rows with hi = 0 / 1, in row order
 * (the split of src/fms_index.h:427-450). */
orc_index* orc_index_from_planes(const uint64_t* hi, const uint64_t* lo, const uint64_t* mask, uint64_t n,
                                 const uint64_t* counts4, uint64_t dollar_position, int k) {
    tables_init();
    orc_index* ix = (orc_index*)calloc(1, sizeof *ix);
    bv_alloc(&ix->ac_gt, n);
    uint64_t words = (n + 63) >> 6, gt_count = 0;
    for (uint64_t w = 0; w < words; ++w) {
        uint64_t v = hi[w];
        if (w == words - 1 && (n & 63)) v &= lo_set((unsigned)(n & 63));
        ix->ac_gt.w[w] = v;
        gt_count += (uint64_t)__builtin_popcountll(v);
    }
    bv_alloc(&ix->ac, n - gt_count);
    bv_alloc(&ix->gt, gt_count);
    uint64_t ai = 0, gi = 0;
    for (uint64_t r = 0; r < n; ++r) {
        int h = (int)((hi[r >> 6] >> (r & 63)) & 1), l = (int)((lo[r >> 6] >> (r & 63)) & 1);
        if (h) { if (l) ix->gt.w[gi >> 6] |= 1ull << (gi & 63); ++gi; }
        else { if (l) ix->ac.w[ai >> 6] |= 1ull << (ai & 63); ++ai; }
    }
    orc_bitvec m;
    bv_alloc(&m, n);
    for (uint64_t w = 0; w < words; ++w) m.w[w] = mask[w];
    if (n & 63) m.w[words - 1] &= lo_set((unsigned)(n & 63));
    orc_rrr_build(&ix->mask, &m);
    bv_free(&m);
    bv_alloc(&ix->klcp, 0);
    for (int i = 0; i < 4; ++i) ix->counts[i] = counts4[i];
    ix->dollar_position = dollar_position;
    ix->k = k;
    build_ranks(ix);
    orc_predictor_reset(ix);
    return ix;
}

/* attach a kLCP bit-vector (n bits as u64 words) to an index made by orc_index_from_planes */
void orc_index_set_klcp(orc_index* ix, const uint64_t* klcp, uint64_t n) {
    bv_free(&ix->klcp);
    bv_alloc(&ix->klcp, n);
    uint64_t words = (n + 63) >> 6;
    for (uint64_t w = 0; w < words; ++w) ix->klcp.w[w] = klcp[w];
    if (n & 63) ix->klcp.w[words - 1] &= lo_set((unsigned)(n & 63));
}

void orc_index_free(orc_index* ix) {
    if (!ix) return;
    bv_free(&ix->ac_gt); bv_free(&ix->ac); bv_free(&ix->gt); bv_free(&ix->klcp);
    free(ix->ac_gt_rank.basic_block); free(ix->ac_rank.basic_block); free(ix->gt_rank.basic_block);
    rrr_free(&ix->mask);
    free(ix);
}
uint64_t orc_index_n(const orc_index* ix) { return ix->mask.size; }
int orc_index_k(const orc_index* ix) { return ix->k; }
int orc_index_has_klcp(const orc_index* ix) { return ix->klcp.n > 0; }
const uint64_t* orc_index_words(const orc_index* ix, int which, uint64_t* nbits) {
    const orc_bitvec* b = which == 0 ? &ix->ac_gt : which == 1 ? &ix->ac : which == 2 ? &ix->gt : &ix->klcp;
    *nbits = b->n;
    return b->w;
}
uint64_t orc_index_count(const orc_index* ix, int c) { return ix->counts[c]; }
uint64_t orc_index_dollar(const orc_index* ix) { return ix->dollar_position; }

/* ------------------------------------------------------------------------------------------------
 * construction: fms_index.h:357-460.  The suffix array is unique ('$' sorts first), so any correct
 * suffix sort yields the reference's bytes; the reference uses Larsson-Sadakane (QSufSort.c), the
 * oracle a plain comparison sort (fine for test sizes).
 * ---------------------------------------------------------------------------------------------- */
static const uint8_t* g_sa_text;
static uint64_t g_sa_len;
static int suffix_cmp(const void* a, const void* b) {
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    if (x == y) return 0;
    uint64_t lx = g_sa_len - x, ly = g_sa_len - y;
    uint64_t m = lx < ly ? lx : ly;
    int c = memcmp(g_sa_text + x, g_sa_text + y, (size_t)m);
    if (c) return c;
    return lx < ly ? -1 : 1;   /* the shorter suffix hits '$' first and '$' is the smallest symbol */
}

orc_index* orc_index_construct(const char* ms, uint64_t len, int k, int use_klcp) {
    tables_init();
    orc_index* ix = (orc_index*)calloc(1, sizeof *ix);
    uint64_t n = len + 1;
    uint8_t* codes = (uint8_t*)malloc((size_t)len + 1);
    for (uint64_t i = 0; i < len; ++i) codes[i] = NT2INT[(uint8_t)ms[i]];   /* convert_superstring :389-395 */
    uint64_t* sa = (uint64_t*)malloc((size_t)n * 8);
    for (uint64_t i = 0; i < n; ++i) sa[i] = i;
    g_sa_text = codes; g_sa_len = len;
    qsort(sa, (size_t)n, 8, suffix_cmp);    /* sa[0] == len (the empty suffix / '$') */

    if (use_klcp) {                         /* construct_klcp :357-385 */
        uint64_t km1 = (uint64_t)(k - 1);
        bv_alloc(&ix->klcp, n);
        for (uint64_t i = 0; i < len; ++i) {
            uint64_t a = sa[i], b = sa[i + 1];
            if (km1 > len || a > len - km1 || b > len - km1) continue;   /* :379-381 */
            if (memcmp(codes + a, codes + b, (size_t)km1) == 0) bv_set(&ix->klcp, i, 1);
        }
    } else {
        bv_alloc(&ix->klcp, 0);
    }

    orc_bitvec sa_mask;
    bv_alloc(&sa_mask, n);
    uint8_t* bwt = (uint8_t*)calloc((size_t)n, 1);
    for (uint64_t i = 0; i <= len; ++i) {   /* :413-422 */
        if (sa[i] == 0) ix->dollar_position = i;
        else bwt[i] = codes[sa[i] - 1];
        if (sa[i] != len) {
            char ch = ms[sa[i]];
            if (ch >= 'A' && ch <= 'Z') bv_set(&sa_mask, i, 1);
        }
    }
    free(sa);
    orc_rrr_build(&ix->mask, &sa_mask);
    bv_free(&sa_mask);

    bv_alloc(&ix->ac_gt, n);                /* :427-450 */
    uint64_t gt_count = 0;
    for (uint64_t i = 0; i < n; ++i) {
        int is_gt = bwt[i] >= 2;
        gt_count += (uint64_t)is_gt;
        if (is_gt) bv_set(&ix->ac_gt, i, 1);
    }
    uint64_t ac_count = n - gt_count;
    bv_alloc(&ix->ac, ac_count);
    bv_alloc(&ix->gt, gt_count);
    uint64_t a_count = 0, g_count = 0, ac_index = 0, gt_index = 0;
    for (uint64_t i = 0; i < n; ++i) {
        int is_one = bwt[i] & 1;
        if (!bv_get(&ix->ac_gt, i)) {
            if (is_one) bv_set(&ix->ac, ac_index, 1);
            ac_index++;
            a_count += (uint64_t)!is_one;
        } else {
            if (is_one) bv_set(&ix->gt, gt_index, 1);
            gt_index++;
            g_count += (uint64_t)!is_one;
        }
    }
    ix->counts[0] = 1; ix->counts[1] = a_count; ix->counts[2] = ac_count; ix->counts[3] = ac_count + g_count; /* :451 */
    ix->k = k;
    free(bwt); free(codes);
    build_ranks(ix);
    orc_predictor_reset(ix);
    return ix;
}

/* ------------------------------------------------------------------------------------------------
 * query path: src/fms_index.h:18-342
 * ---------------------------------------------------------------------------------------------- */
static inline int clipped(int x) { return x < -7 ? -7 : (x > 7 ? 7 : x); } /* :23-25 */

static void predictor_log_result(orc_predictor* p, int fwd, int rev) { /* :27-38 */
    int difference = fwd - rev;
    p->score = clipped(p->score + difference);
    if (p->previous != -1) p->result_scores[p->previous] = clipped(p->result_scores[p->previous] + difference);
    p->previous = fwd > rev;
}
static int predictor_predict_swap(const orc_predictor* p) { /* :41-48 */
    if (p->previous != -1 && p->result_scores[p->previous] != 0) return p->result_scores[p->previous] < 0;
    return p->score < 0;
}

uint64_t orc_rank1(orc_index* ix, int which, uint64_t i) {
    return orc_rank_v5_rank(which == 0 ? &ix->ac_gt_rank : which == 1 ? &ix->ac_rank : &ix->gt_rank, i);
}

uint64_t orc_rank(orc_index* ix, uint64_t i, int c) { /* :68-86 */
    uint64_t gt_position = orc_rank_v5_rank(&ix->ac_gt_rank, i);
    if (c >= 2) {
        uint64_t t_position = orc_rank_v5_rank(&ix->gt_rank, gt_position);
        return c == 2 ? gt_position - t_position : t_position;
    }
    uint64_t c_position = orc_rank_v5_rank(&ix->ac_rank, i - gt_position);
    if (c == 0) return i - gt_position - c_position - (uint64_t)(i >= ix->dollar_position + 1);
    return c_position;
}

void orc_update_range(orc_index* ix, uint64_t* i, uint64_t* j, int c) { /* :98-103 */
    if (*j == *i) return;
    ix->lf_steps++;
    uint64_t count = ix->counts[c];
    *i = count + orc_rank(ix, *i, c);
    *j = count + orc_rank(ix, *j, c);
}

void orc_extend_range_with_klcp(orc_index* ix, uint64_t* i, uint64_t* j) { /* :106-115 */
    while (bv_get(&ix->klcp, *j - 1)) { (*j)++; ix->klcp_bits++; }
    while (bv_get(&ix->klcp, *i - 1)) { (*i)--; ix->klcp_bits++; }
    ix->klcp_bits += 2;
}

void orc_get_range_with_pattern(orc_index* ix, const char* pattern, int k, uint64_t* sa_start, uint64_t* sa_end) { /* :117-124 */
    *sa_start = 0;
    *sa_end = ix->mask.size;
    ix->strand_searches++;
    for (int i = k - 1; i >= 0 && *sa_start != *sa_end; --i)
        orc_update_range(ix, sa_start, sa_end, NT2INT[(uint8_t)pattern[i]]);
}

int orc_mask_at(orc_index* ix, uint64_t i) { ix->mask_probes++; return orc_rrr_access(&ix->mask, i); }
uint64_t orc_mask_rank(orc_index* ix, uint64_t i) { return orc_rrr_rank(&ix->mask, i); }

int orc_infer_presence(orc_index* ix, int maximized_ones, uint64_t sa_start, uint64_t sa_end) { /* :126-144 */
    if (maximized_ones) {
        if (sa_start != sa_end) return orc_mask_at(ix, sa_start);
        return -1;
    }
    for (uint64_t i = sa_start; i < sa_end; ++i)
        if (orc_mask_at(ix, i)) return 1;
    return sa_start == sa_end ? -1 : 0;
}

int64_t orc_kmer_order_if_present(orc_index* ix, uint64_t sa_start, uint64_t sa_end) { /* :146-156 */
    if (orc_infer_presence(ix, 0, sa_start, sa_end) == 1) return (int64_t)orc_rrr_rank(&ix->mask, sa_start);
    return -1;
}

static int single_query_or(orc_index* ix, int maximized_ones, const char* pattern, int k) { /* :158-163 */
    uint64_t a, b;
    orc_get_range_with_pattern(ix, pattern, k, &a, &b);
    return orc_infer_presence(ix, maximized_ones, a, b);
}
static int64_t single_query_order(orc_index* ix, const char* pattern, int k) { /* :165-169 */
    uint64_t a, b;
    orc_get_range_with_pattern(ix, pattern, k, &a, &b);
    return orc_kmer_order_if_present(ix, a, b);
}

static char* reverse_complement(const char* s, uint64_t length) { /* kmers.h:53-59 */
    char* r = (char*)malloc((size_t)length + 1);
    for (uint64_t i = 0; i < length; ++i) r[i] = COMPL[(uint8_t)s[length - i - 1]];
    r[length] = 0;
    return r;
}

void orc_kmer_ternaries(orc_index* ix, const char* kmer, int k, int maximized_ones, int* fwd, int* rc) {
    char* r = reverse_complement(kmer, (uint64_t)k);
    *fwd = single_query_or(ix, maximized_ones, kmer, k);
    *rc = single_query_or(ix, maximized_ones, r, k);
    free(r);
}

void orc_kmers_ternaries_bulk(orc_index* ix, const char* kmers, uint64_t n, int k, int maximized_ones,
                               int8_t* fwd, int8_t* rc, int64_t* ord_fwd, int64_t* ord_rc) {
    /* per k-mer, both strands, no predictor: infer_presence (and kmer_order_if_present when ord_* != NULL) */
    for (uint64_t q = 0; q < n; ++q) {
        const char* kmer = kmers + q * (uint64_t)k;
        char* r = reverse_complement(kmer, (uint64_t)k);
        uint64_t a, b;
        orc_get_range_with_pattern(ix, kmer, k, &a, &b);
        fwd[q] = (int8_t)orc_infer_presence(ix, maximized_ones, a, b);
        if (ord_fwd) ord_fwd[q] = orc_kmer_order_if_present(ix, a, b);
        orc_get_range_with_pattern(ix, r, k, &a, &b);
        rc[q] = (int8_t)orc_infer_presence(ix, maximized_ones, a, b);
        if (ord_rc) ord_rc[q] = orc_kmer_order_if_present(ix, a, b);
        free(r);
    }
}

static void query_kmers_streaming(orc_index* ix, const char* sequence, const char* rc_sequence, uint64_t sequence_length,
                                  int k, int maximized_ones, int output_orders, orc_obuf* of) { /* :181-254 */
    uint64_t nres = sequence_length - (uint64_t)k + 1;
    int64_t* result = (int64_t*)malloc((size_t)nres * 8);
    for (uint64_t i = 0; i < nres; ++i) result[i] = -1;
    int should_swap = predictor_predict_swap(&ix->predictor);
    if (should_swap) { const char* t = sequence; sequence = rc_sequence; rc_sequence = t; }
    int forward_predictor_result = 0, backward_predictor_result = 0;
    uint64_t sa_start = (uint64_t)-1, sa_end = (uint64_t)-1;
    for (uint64_t i = 0; i <= sequence_length - (uint64_t)k; ++i) {           /* :192-208 */
        uint64_t i_back = sequence_length - (uint64_t)k - i;
        if (sa_start == sa_end) {
            orc_get_range_with_pattern(ix, sequence + i_back, k, &sa_start, &sa_end);
        } else {
            orc_extend_range_with_klcp(ix, &sa_start, &sa_end);
            orc_update_range(ix, &sa_start, &sa_end, NT2INT[(uint8_t)sequence[i_back]]);
        }
        if (output_orders) {
            result[i_back] = orc_kmer_order_if_present(ix, sa_start, sa_end);
            if (result[i_back] >= 0) forward_predictor_result++;
            else forward_predictor_result--;
        } else {
            result[i_back] = orc_infer_presence(ix, maximized_ones, sa_start, sa_end);
            forward_predictor_result += (int)result[i_back];
        }
    }
    sa_start = sa_end = (uint64_t)-1;
    for (uint64_t i = 0; i <= sequence_length - (uint64_t)k; ++i) {           /* :211-234 */
        if ((result[i] >= 0 && output_orders) || result[i] == 1 || (result[i] == 0 && maximized_ones)) {
            sa_start = sa_end = (uint64_t)-1;
            continue;
        }
        uint64_t i_back = sequence_length - (uint64_t)k - i;
        if (sa_start == sa_end) {
            orc_get_range_with_pattern(ix, rc_sequence + i_back, k, &sa_start, &sa_end);
        } else {
            orc_extend_range_with_klcp(ix, &sa_start, &sa_end);
            orc_update_range(ix, &sa_start, &sa_end, NT2INT[(uint8_t)rc_sequence[i_back]]);
        }
        int64_t res;
        if (output_orders) {
            res = orc_kmer_order_if_present(ix, sa_start, sa_end);
            if (res >= 0) backward_predictor_result++;
            else backward_predictor_result--;
        } else {
            res = orc_infer_presence(ix, maximized_ones, sa_start, sa_end);
            backward_predictor_result += (int)res;
        }
        if (res > result[i]) result[i] = res;
    }
    if (should_swap) {                                                         /* :236-239 */
        for (uint64_t a = 0, b = nres - 1; a < b; ++a, --b) { int64_t t = result[a]; result[a] = result[b]; result[b] = t; }
        int t = forward_predictor_result; forward_predictor_result = backward_predictor_result; backward_predictor_result = t;
    }
    predictor_log_result(&ix->predictor, forward_predictor_result, backward_predictor_result);
    for (uint64_t i = 0; i < nres; ++i) {                                      /* :242-253 */
        int64_t c = result[i];
        if (output_orders) {
            if (i > 0) ob_putc(of, ',');
            ob_puti(of, c);
        } else ob_putc(of, c == 1 ? '1' : '0');
    }
    free(result);
}

/* ---- query_mode::general: f-masked superstrings (src/functions.h, src/fms_index.h:171-179, 317-327) ---- */
int orc_parse_function(const char* name, int* kind, uint64_t* r, uint64_t* s) { /* functions.h:23-57 */
    if (!strcmp(name, "or")) return 1;
    if (!strcmp(name, "and")) { *kind = ORC_FN_AND; return 0; }
    if (!strcmp(name, "xor")) { *kind = ORC_FN_XOR; return 0; }
    if (!strcmp(name, "all")) return 2;
    size_t len = strlen(name);
    for (size_t i = 1; i + 1 < len; ++i) {                       /* :39 */
        if (name[i] != '-') continue;
        int valid = 1;
        for (size_t j = 0; j < i; ++j) if (name[j] < '0' || name[j] > '9') valid = 0;
        for (size_t j = i + 1; j < len; ++j) if (name[j] < '0' || name[j] > '9') valid = 0;
        if (!valid) break;                                          /* :50-51 */
        /* std::stoi: digits only here, so the one failure is a value above INT_MAX */
        uint64_t rv = 0, sv = 0;
        for (size_t j = 0; j < i; ++j) { rv = rv * 10 + (uint64_t)(name[j] - '0'); if (rv > 2147483647ull) return -2; }
        for (size_t j = i + 1; j < len; ++j) { sv = sv * 10 + (uint64_t)(name[j] - '0'); if (sv > 2147483647ull) return -2; }
        *kind = ORC_FN_RANGE; *r = rv; *s = sv;
        return 0;
    }
    return -1;                                                      /* :59 */
}
void orc_set_function(orc_index* ix, int kind, uint64_t r, uint64_t s) { ix->fn_kind = kind; ix->fn_r = r; ix->fn_s = s; }
int orc_apply_function(int kind, uint64_t r, uint64_t s, uint64_t ones, uint64_t total) {
    /* the chain of std::function wrappers between ms_query's f and f_and/f_xor/the lambda converts size_t -> int -> size_t */
    uint64_t o = (uint64_t)(int64_t)(int32_t)(uint32_t)ones, t = (uint64_t)(int64_t)(int32_t)(uint32_t)total;
    if (kind == ORC_FN_AND) return t && o == t;                     /* functions.h:11 */
    if (kind == ORC_FN_XOR) return (int)(o % 2);                    /* :14 */
    return o <= s && o >= r;                                        /* :17-19 */
}
static void single_query_general(orc_index* ix, const char* pattern, int k, uint64_t* ones, uint64_t* total) { /* :171-179 */
    uint64_t sa_start, sa_end;
    orc_get_range_with_pattern(ix, pattern, k, &sa_start, &sa_end);
    uint64_t o = 0;
    for (uint64_t i = sa_start; i < sa_end; ++i) o += (uint64_t)orc_rrr_access(&ix->mask, i);
    *ones = o;
    *total = sa_end - sa_start;
}
void orc_kmer_general(orc_index* ix, const char* kmer, int k, uint64_t* ones, uint64_t* total) { /* :317-323 */
    char* rc = reverse_complement(kmer, (uint64_t)k);
    single_query_general(ix, kmer, k, ones, total);
    if (strncmp(kmer, rc, (size_t)k) != 0) {                        /* AreStringsEqual, kmers.h */
        uint64_t o2, t2;
        single_query_general(ix, rc, k, &o2, &t2);
        *ones += o2;
        *total += t2;
    }
    free(rc);
}

static void query_kmers_single(orc_index* ix, const char* sequence, const char* rc_sequence, uint64_t sequence_length,
                               int k, int mode, int output_orders, orc_obuf* of) { /* :263-331, modes orr / all */
    for (uint64_t i = 0; i <= sequence_length - (uint64_t)k; ++i) {
        const char* kmer = sequence + i;
        const char* rc_kmer = rc_sequence + (sequence_length - (uint64_t)k - i);
        int should_swap = predictor_predict_swap(&ix->predictor);
        if (should_swap) { const char* t = kmer; kmer = rc_kmer; rc_kmer = t; }
        if (mode == ORC_MODE_GENERAL) {                                                /* :316-327; the predictor is not told */
            uint64_t ones, total;
            single_query_general(ix, kmer, k, &ones, &total);
            if (strncmp(kmer, rc_kmer, (size_t)k) != 0) {
                uint64_t o2, t2;
                single_query_general(ix, rc_kmer, k, &o2, &t2);
                ones += o2;
                total += t2;
            }
            ob_putc(of, orc_apply_function(ix->fn_kind, ix->fn_r, ix->fn_s, ones, total) ? '1' : '0');
            continue;
        }
        int forward_predictor_result = 0, backward_predictor_result = 0;
        int64_t got;
        if (output_orders) got = single_query_order(ix, kmer, k);
        else if (mode == ORC_MODE_OR) got = single_query_or(ix, 0, kmer, k);
        else got = single_query_or(ix, 1, kmer, k);
        forward_predictor_result = (int)got;
        if (output_orders) {
            if (forward_predictor_result >= 0) forward_predictor_result = 1;
            else {
                got = single_query_order(ix, rc_kmer, k);
                backward_predictor_result = got >= 0 ? 1 : -1;
            }
        } else if (mode == ORC_MODE_OR) {
            if (got != 1) { got = single_query_or(ix, 0, rc_kmer, k); backward_predictor_result = (int)got; }
        } else {
            if (got == -1) { got = single_query_or(ix, 1, rc_kmer, k); backward_predictor_result = (int)got; }
        }
        if (output_orders) {
            if (i > 0) ob_putc(of, ',');
            ob_puti(of, got);
        } else ob_putc(of, got == 1 ? '1' : '0');
        if (should_swap) { int t = forward_predictor_result; forward_predictor_result = backward_predictor_result; backward_predictor_result = t; }
        predictor_log_result(&ix->predictor, forward_predictor_result, backward_predictor_result);
    }
}

void orc_query_kmers(orc_index* ix, const char* seq, uint64_t len, int k, int has_klcp, int mode,
                     int output_orders, orc_obuf* of) { /* :333-342 */
    char* rc = reverse_complement(seq, len);
    if (has_klcp && mode != ORC_MODE_GENERAL) query_kmers_streaming(ix, seq, rc, len, k, mode == ORC_MODE_ALL, output_orders, of);   /* :336 */
    else query_kmers_single(ix, seq, rc, len, k, mode, output_orders, of);
    free(rc);
}

void orc_query_kmers_streaming(orc_index* ix, const char* seq, uint64_t len, int k, int maximized_ones,
                               int output_orders, orc_obuf* of) {
    char* rc = reverse_complement(seq, len);
    query_kmers_streaming(ix, seq, rc, len, k, maximized_ones, output_orders, of);
    free(rc);
}

/* ------------------------------------------------------------------------------------------------
 * FASTA/FASTQ reader restating kseq_read (src/kseq.h:183-226) over an in-memory text, and the
 * record loop of ms_query (src/main.cpp:328-373).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const char* buf; uint64_t len, pos;
    int last_char;
    char* name; size_t name_l, name_m;
    char* seq; size_t seq_l, seq_m;
    size_t qual_l;
} orc_reader;

static int rd_getc(orc_reader* r) { return r->pos < r->len ? (unsigned char)r->buf[r->pos++] : -1; }
static void str_append(char** s, size_t* l, size_t* m, const char* src, size_t n) {
    if (*l + n + 2 > *m) { size_t nm = *m ? *m : 256; while (nm < *l + n + 2) nm *= 2; *s = (char*)realloc(*s, nm); *m = nm; }
    memcpy(*s + *l, src, n);
    *l += n;
    (*s)[*l] = 0;
}
/* ks_getuntil2 (kseq.h:102-150): delimiter 0 = any isspace, 2 = line.  Returns -1 at EOF with nothing read. */
static int64_t rd_getuntil(orc_reader* r, int line, char** s, size_t* l, size_t* m, int* dret, int append, int count_only) {
    if (dret) *dret = 0;
    if (!append) *l = 0;
    if (r->pos >= r->len) return -1;
    uint64_t i = r->pos;
    if (line) { const char* nl = (const char*)memchr(r->buf + i, '\n', (size_t)(r->len - i)); i = nl ? (uint64_t)(nl - r->buf) : r->len; }
    else { while (i < r->len && !isspace((unsigned char)r->buf[i])) ++i; }
    if (count_only) *l += (size_t)(i - r->pos);
    else str_append(s, l, m, r->buf + r->pos, (size_t)(i - r->pos));
    if (i < r->len && dret) *dret = (unsigned char)r->buf[i];
    uint64_t start = r->pos;
    r->pos = i + 1 > r->len ? r->len : i + 1;
    if (line && *l > 1) {
        /* trailing '\r' strip (kseq.h:146) looks at the accumulated string's last char */
        int last_is_cr = count_only ? (i > start ? r->buf[i - 1] == '\r' : 0) : ((*s)[*l - 1] == '\r');
        if (last_is_cr) { --*l; if (!count_only) (*s)[*l] = 0; }
    }
    return (int64_t)*l;
}

static int64_t rd_read(orc_reader* r) { /* kseq_read, kseq.h:183-226 */
    int c;
    if (r->last_char == 0) {
        while ((c = rd_getc(r)) >= 0 && c != '>' && c != '@') {}
        if (c < 0) return c;
        r->last_char = c;
    }
    r->seq_l = 0; r->qual_l = 0;
    if (rd_getuntil(r, 0, &r->name, &r->name_l, &r->name_m, &c, 0, 0) < 0) return -1;
    if (c != '\n') { size_t dl = 0; rd_getuntil(r, 1, NULL, &dl, NULL, NULL, 0, 1); }
    if (!r->seq) { r->seq_m = 256; r->seq = (char*)malloc(r->seq_m); r->seq[0] = 0; }
    while ((c = rd_getc(r)) >= 0 && c != '>' && c != '+' && c != '@') {
        if (c == '\n') continue;
        char ch = (char)c;
        str_append(&r->seq, &r->seq_l, &r->seq_m, &ch, 1);
        rd_getuntil(r, 1, &r->seq, &r->seq_l, &r->seq_m, NULL, 1, 0);
    }
    if (c == '>' || c == '@') r->last_char = c;
    r->seq[r->seq_l] = 0;
    if (c != '+') return (int64_t)r->seq_l;
    while ((c = rd_getc(r)) >= 0 && c != '\n') {}
    if (c == -1) return -2;
    while (rd_getuntil(r, 1, NULL, &r->qual_l, NULL, NULL, 1, 1) >= 0 && r->qual_l < r->seq_l) {}
    r->last_char = 0;
    if (r->seq_l != r->qual_l) return -2;
    return (int64_t)r->seq_l;
}

static uint64_t next_invalid_character_or_end(const char* sequence, uint64_t length) { /* parser.h:57-64 */
    for (uint64_t i = 0; i < length; ++i)
        if (NT2INT[(uint8_t)sequence[i]] == 4) return i;
    return length;
}

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

static int64_t run_records(orc_index* ix, const char* text, uint64_t text_len, int k, int has_klcp, int mode,
                           int output_orders, orc_obuf* of, double* query_seconds, uint64_t* windows_out) {
    orc_reader r;
    memset(&r, 0, sizeof r);
    r.buf = text; r.len = text_len;
    int64_t sequence_length;
    uint64_t windows = 0;
    double qsec = 0;
    while ((sequence_length = rd_read(&r)) >= 0) {                                  /* main.cpp:328 */
        int64_t max_chunk = 400;
        int64_t sq = 2 * (int64_t)sqrt((double)sequence_length);
        if (sq < max_chunk) max_chunk = sq;
        if (max_chunk < 10) max_chunk = 10;
        max_chunk += k;                                                                /* :330-331 */
        ob_puts(of, r.name ? r.name : "");
        ob_putc(of, '\t');                                                             /* :333 */
        const char* sequence = r.seq;
        int output_comma = 0;
        while (sequence_length > 0) {                                                  /* :337 */
            int64_t current_length = (int64_t)next_invalid_character_or_end(sequence, (uint64_t)sequence_length);
            while (current_length >= k) {                                              /* :340-354 */
                if (output_orders && output_comma) ob_putc(of, ',');
                output_comma = 1;
                int64_t chunk_length = current_length < max_chunk ? current_length : max_chunk;
                double t0 = query_seconds ? now_s() : 0;
                orc_query_kmers(ix, sequence, (uint64_t)chunk_length, k, has_klcp, mode, output_orders, of);
                if (query_seconds) qsec += now_s() - t0;
                windows += (uint64_t)(chunk_length - k + 1);
                sequence += chunk_length - k + 1;
                current_length -= chunk_length - k + 1;
                sequence_length -= chunk_length - k + 1;
            }
            sequence_length -= current_length + 1;                                     /* :356-357 */
            sequence += current_length + 1;
            if (sequence_length >= 0) {                                                /* :359-370 */
                int64_t lim = current_length + 1 < k ? current_length + 1 : k;
                for (int64_t i = 0; i < lim; ++i) {
                    if (output_orders) {
                        if (output_comma) ob_putc(of, ',');
                        output_comma = 1;
                        ob_puts(of, "-1");
                    } else ob_putc(of, '0');
                }
            }
        }
        ob_putc(of, '\n');                                                             /* :372 */
    }
    free(r.name); free(r.seq);
    if (query_seconds) *query_seconds = qsec;
    if (windows_out) *windows_out = windows;
    return (int64_t)of->len;
}

static char* slurp_gz(const char* path, uint64_t* len_out) { /* parser.h:14-27: gzdopen reads plain or gzip */
    gzFile fp = strcmp(path, "-") == 0 ? gzdopen(0, "r") : gzopen(path, "r");
    if (!fp) return NULL;
    size_t cap = 1 << 20, len = 0;
    char* buf = (char*)malloc(cap);
    for (;;) {
        if (cap - len < (1 << 16)) { cap *= 2; buf = (char*)realloc(buf, cap); }
        int n = gzread(fp, buf + len, (unsigned)(cap - len > (1u << 30) ? (1u << 30) : cap - len));
        if (n <= 0) break;
        len += (size_t)n;
    }
    gzclose(fp);
    *len_out = len;
    return buf;
}

int64_t orc_query_file(orc_index* ix, const char* query_fn, int k, int has_klcp, int mode, int output_orders,
                       orc_obuf* of, double* query_seconds, uint64_t* windows) {
    uint64_t len;
    char* text = slurp_gz(query_fn, &len);
    if (!text) return -1;
    int64_t r = run_records(ix, text, len, k, has_klcp, mode, output_orders, of, query_seconds, windows);
    free(text);
    return r;
}

int64_t orc_query_text(orc_index* ix, const char* fasta_text, uint64_t text_len, int k, int has_klcp, int mode,
                       int output_orders, orc_obuf* of) {
    return run_records(ix, fasta_text, text_len, k, has_klcp, mode, output_orders, of, NULL, NULL);
}

static int64_t copy_out(orc_obuf* o, char* out, uint64_t cap) {
    int64_t n = (int64_t)o->len;
    if ((uint64_t)n > cap) { orc_obuf_free(o); return -n; }
    if (n) memcpy(out, o->p, (size_t)n);
    orc_obuf_free(o);
    return n;
}
int64_t orc_query_kmers_buf(orc_index* ix, const char* seq, uint64_t len, int k, int has_klcp, int mode,
                            int output_orders, char* out, uint64_t cap) {
    orc_obuf o; orc_obuf_init(&o);
    orc_query_kmers(ix, seq, len, k, has_klcp, mode, output_orders, &o);
    return copy_out(&o, out, cap);
}
int64_t orc_query_kmers_streaming_buf(orc_index* ix, const char* seq, uint64_t len, int k, int maximized_ones,
                                      int output_orders, char* out, uint64_t cap) {
    orc_obuf o; orc_obuf_init(&o);
    orc_query_kmers_streaming(ix, seq, len, k, maximized_ones, output_orders, &o);
    return copy_out(&o, out, cap);
}
int64_t orc_query_file_buf(orc_index* ix, const char* query_fn, int k, int has_klcp, int mode, int output_orders,
                           char* out, uint64_t cap, double* query_seconds) {
    orc_obuf o; orc_obuf_init(&o);
    if (orc_query_file(ix, query_fn, k, has_klcp, mode, output_orders, &o, query_seconds, NULL) < 0) { orc_obuf_free(&o); return INT64_MIN; }
    return copy_out(&o, out, cap);
}

double orc_time_isolated_kmers(orc_index* ix, const char* kmers, uint64_t n, int k, int mode, char* out) {
    orc_obuf o; orc_obuf_init(&o);
    double t0 = now_s();
    for (uint64_t q = 0; q < n; ++q) orc_query_kmers(ix, kmers + q * (uint64_t)k, (uint64_t)k, k, 0, mode, 0, &o);
    double s = now_s() - t0;
    if (out && o.len >= n) memcpy(out, o.p, (size_t)n);
    orc_obuf_free(&o);
    return s;
}

/* ------------------------------------------------------------------------------------------------
 * CLI: fmsi_oracle index [-k K] [-x] <ms.fa> | fmsi_oracle query|lookup [-q F] [-k K] [-S] [-O] [-T] <prefix>
 * ---------------------------------------------------------------------------------------------- */
#ifdef ORACLE_MAIN
#include <unistd.h>
static int cli_index(int argc, char** argv) {
    const char* fn = argv[argc - 1];
    argc--;
    int k = 0, c, no_streaming = 0;
    while ((c = getopt(argc, argv, "k:x")) >= 0) {
        if (c == 'k') k = atoi(optarg);
        else if (c == 'x') no_streaming = 1;
        else return 1;
    }
    uint64_t len;
    char* text = slurp_gz(fn, &len);
    if (!text) { fprintf(stderr, "cannot open %s\n", fn); return 1; }
    orc_reader r; memset(&r, 0, sizeof r); r.buf = text; r.len = len;
    int64_t l = rd_read(&r);                       /* read_masked_superstring, parser.h:41-55 */
    if (l <= 0) { fprintf(stderr, "bad masked superstring file\n"); return 1; }
    int inferred_k = 1;                            /* infer_k, parser.h:31-37 */
    while (!(r.seq[l - inferred_k] >= 'A' && r.seq[l - inferred_k] <= 'Z')) inferred_k++;
    if (k == 0) k = inferred_k;
    if (k > 64) no_streaming = 1;
    orc_index* ix = orc_index_construct(r.seq, (uint64_t)l, k, !no_streaming);
    int rc = orc_index_dump(ix, fn);
    orc_index_free(ix);
    free(text);
    return rc ? 1 : 0;
}
static int cli_query(int argc, char** argv, int output_orders) {
    const char* fn = argv[argc - 1];
    argc--;
    const char* query_fn = "-";
    int k = 0, c, has_klcp = 0, mode = ORC_MODE_OR, timing = 0, fn_kind = 0;
    uint64_t fn_r = 0, fn_s = 0;
    while ((c = getopt(argc, argv, "f:q:k:OST")) >= 0) {
        switch (c) {
        case 'f': {                                                                    /* main.cpp:255-263 */
            int p = orc_parse_function(optarg, &fn_kind, &fn_r, &fn_s);
            if (p == -2) abort();
            if (p < 0) { fprintf(stderr, "ERROR: Function '%s' not recognized.\n", optarg); return 1; }
            mode = p == 1 ? ORC_MODE_OR : (p == 2 ? ORC_MODE_ALL : ORC_MODE_GENERAL);
            break;
        }
        case 'q': query_fn = optarg; break;
        case 'k': k = atoi(optarg); break;
        case 'O':                                                                      /* main.cpp:273-280 */
            if (mode != ORC_MODE_OR) fprintf(stderr, "WARNING: Parameter -O is ignored when parameter -f is specified.\n");
            else mode = ORC_MODE_ALL;
            break;
        case 'S': has_klcp = 1; break;
        case 'T': timing = 1; break;
        default: return 1;
        }
    }
    if (output_orders && mode == ORC_MODE_GENERAL) {                                   /* main.cpp:297-300 */
        fprintf(stderr, "ERROR: FMSI as Minimum Perfect Hash Function is not allowed with f-masked superstrings in the current version.\n");
        return 1;
    }
    double t0 = now_s();
    orc_index* ix = orc_index_load(fn, has_klcp);
    double load_s = now_s() - t0;
    if (ix) orc_set_function(ix, fn_kind, fn_r, fn_s);
    if (!ix || orc_index_n(ix) == 0) { fprintf(stderr, "ERROR: index not correctly loaded\n"); return 1; }
    if (has_klcp != orc_index_has_klcp(ix)) { fprintf(stderr, "ERROR: kLCP mismatch\n"); return 1; }
    if (k != 0 && k != ix->k) { fprintf(stderr, "ERROR: k mismatch\n"); return 1; }
    if (k == 0) k = ix->k;
    orc_obuf o; orc_obuf_init(&o);
    double qsec = 0; uint64_t windows = 0;
    double t1 = now_s();
    if (orc_query_file(ix, query_fn, k, has_klcp, mode, output_orders, &o, &qsec, &windows) < 0) {
        fprintf(stderr, "cannot open %s\n", query_fn);
        return 1;
    }
    double wall = now_s() - t1;
    if (timing) {
        uint64_t cn[4]; orc_counters_get(ix, cn);
        fprintf(stderr, "{\"windows\": %llu, \"query_seconds\": %.6f, \"loop_wall_seconds\": %.6f, \"load_seconds\": %.6f, "
                        "\"lf_steps\": %llu, \"mask_probes\": %llu, \"klcp_bits\": %llu, \"strand_searches\": %llu}\n",
                (unsigned long long)windows, qsec, wall, load_s, (unsigned long long)cn[0], (unsigned long long)cn[1],
                (unsigned long long)cn[2], (unsigned long long)cn[3]);
    } else {
        fwrite(o.p, 1, o.len, stdout);
    }
    orc_obuf_free(&o);
    orc_index_free(ix);
    return 0;
}
int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: fmsi_oracle index|query|lookup ...\n"); return 1; }
    if (!strcmp(argv[1], "index")) return cli_index(argc - 1, argv + 1);
    if (!strcmp(argv[1], "query")) return cli_query(argc - 1, argv + 1, 0);
    if (!strcmp(argv[1], "lookup")) return cli_query(argc - 1, argv + 1, 1);
    return 1;
}
#endif
