/* oracle/salt_oracle.c -- CPU restatement of salt's single-end per-read alignment path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (see salt_oracle.h).  Parity: PINNED against the real
 * reference binary and the committed golden vectors.
 *
 * This is a from-scratch restatement of the algorithm, organised by stage; each function cites
 * the reference code (under /root/reference/) whose behaviour it follows.
 */
#include "salt_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <pthread.h>

/* ------------------------------------------------------------------------------------------ */
/* index                                                                                      */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int64_t offset; int32_t len, n_ambs; char *name; } so_ann_t;
typedef struct { int64_t offset; int32_t len; char amb; } so_amb_t;

struct so_index {
    /* C index: BWA 0.5/0.6 32-bit layout (Align_src/bwt.h:40-64, bwtio.c:30-71) */
    uint32_t c_primary, c_L2[5], c_seq_len, c_bwt_size;
    uint32_t *c_bwt;
    uint32_t c_sa_intv, c_n_sa;
    uint32_t *c_sa;
    /* 12-mer table (Align_src/lookup.c:47-65) */
    uint32_t lkt_len, lkt_n;
    uint32_t *lkt;
    /* R index, backward only (Align_src/rbwt.c:258-297,477-508,558-574) */
    uint32_t r_text_len, r_inv_sa0, r_cum[6], r_bwt_words;
    uint32_t *r_bwt;
    uint32_t r_occ_words, r_major_words;
    uint32_t *r_occ, *r_major;
    uint32_t r_n_sa;
    uint32_t *r_sa;
    /* mixRef (Align_src/metaref.c:61-93) */
    uint32_t ref_len;
    uint32_t *ref;
    /* bntseq + pac (Align_src/bntseq.c:88-139, indexio.h:45-56) */
    int64_t l_pac;
    int32_t n_seqs, n_holes;
    so_ann_t *anns;
    so_amb_t *ambs;
    uint8_t *pac;
    int32_t seed_len;
};

static void *xcalloc(size_t n, size_t s)
{
    void *p = calloc(n ? n : 1, s);
    if (!p) { fprintf(stderr, "[salt_oracle] out of memory\n"); exit(1); }
    return p;
}

static FILE *open_sfx(const char *prefix, const char *sfx, const char *mode, int must)
{
    char fn[2048];
    snprintf(fn, sizeof fn, "%s%s", prefix, sfx);
    FILE *fp = fopen(fn, mode);
    if (!fp && must) fprintf(stderr, "[salt_oracle] cannot open %s\n", fn);
    return fp;
}

static int rd(void *dst, size_t sz, size_t n, FILE *fp)
{
    return fread(dst, sz, n, fp) == n ? 0 : -1;
}

/* LKT content when the 64 MiB file is absent: counts of every 12-mer start in the 2-bit pac plus
 * the 12 A-padded tail suffixes, then prefix-summed (Index_src/LookUpTable.c:70-150). */
static uint32_t *lkt_from_pac(const uint8_t *pac, uint32_t l_ref, int len, uint32_t *n_item_out)
{
    uint32_t n_item = (1u << (2 * len)) + 1;
    uint32_t *item = xcalloc(n_item, 4);
    uint32_t mask = n_item - 2;
    uint32_t x = 0, i;
    for (i = 0; i < l_ref; ++i) {
        uint32_t c = (pac[i >> 2] >> ((~i & 3) << 1)) & 3;
        x = ((x << 2) & mask) | c;
        if (i + 1 >= (uint32_t)len) ++item[x + 1];
    }
    for (i = 0; i < (uint32_t)len; ++i) { x = (x << 2) & mask; ++item[x + 1]; }
    for (i = 1; i < n_item; ++i) item[i] += item[i - 1];
    *n_item_out = n_item;
    return item;
}

so_index_t *so_index_load(const char *prefix)
{
    so_index_t *ix = xcalloc(1, sizeof *ix);
    FILE *fp;
    long sz;
    /* .R.seedLen (aln.c:215-224) */
    if (!(fp = open_sfx(prefix, ".R.seedLen", "rb", 1))) goto fail;
    if (rd(&ix->seed_len, 4, 1, fp)) goto fail_fp;
    fclose(fp);
    /* .C.bwt (bwtio.c:52-71) */
    if (!(fp = open_sfx(prefix, ".C.bwt", "rb", 1))) goto fail;
    fseek(fp, 0, SEEK_END); sz = ftell(fp); fseek(fp, 0, SEEK_SET);
    ix->c_bwt_size = (uint32_t)((sz - 20) >> 2);
    ix->c_bwt = xcalloc(ix->c_bwt_size, 4);
    if (rd(&ix->c_primary, 4, 1, fp) || rd(ix->c_L2 + 1, 4, 4, fp) ||
        rd(ix->c_bwt, 4, ix->c_bwt_size, fp)) goto fail_fp;
    ix->c_seq_len = ix->c_L2[4];
    fclose(fp);
    /* .C.sa (bwtio.c:30-50) */
    if (!(fp = open_sfx(prefix, ".C.sa", "rb", 1))) goto fail;
    {
        uint32_t hdr[7];
        if (rd(hdr, 4, 7, fp)) goto fail_fp;
        if (hdr[0] != ix->c_primary || hdr[6] != ix->c_seq_len) {
            fprintf(stderr, "[salt_oracle] SA-BWT inconsistency\n"); goto fail_fp;
        }
        ix->c_sa_intv = hdr[5];
        ix->c_n_sa = (ix->c_seq_len + ix->c_sa_intv) / ix->c_sa_intv;
        ix->c_sa = xcalloc(ix->c_n_sa, 4);
        ix->c_sa[0] = (uint32_t)-1;
        if (rd(ix->c_sa + 1, 4, ix->c_n_sa - 1, fp)) goto fail_fp;
    }
    fclose(fp);
    /* .R.backward.bwt (rbwt.c:258-279) */
    if (!(fp = open_sfx(prefix, ".R.backward.bwt", "rb", 1))) goto fail;
    if (rd(&ix->r_text_len, 4, 1, fp) || rd(&ix->r_inv_sa0, 4, 1, fp) ||
        rd(ix->r_cum + 1, 4, 5, fp) || rd(&ix->r_bwt_words, 4, 1, fp)) goto fail_fp;
    {
        /* same over-allocation as the reference so that backward counting from the last
         * checkpoint never leaves the buffer */
        uint32_t alloc = (ix->r_bwt_words * 8 + 256) / 256 * 256 / 8 + 1;
        ix->r_bwt = xcalloc(alloc, 4);
        if (rd(ix->r_bwt, 4, ix->r_bwt_words, fp)) goto fail_fp;
    }
    fclose(fp);
    /* .R.backward.occ (rbwt.c:280-297) */
    if (!(fp = open_sfx(prefix, ".R.backward.occ", "rb", 1))) goto fail;
    if (rd(&ix->r_occ_words, 4, 1, fp)) goto fail_fp;
    ix->r_occ = xcalloc(ix->r_occ_words, 4);
    if (rd(ix->r_occ, 4, ix->r_occ_words, fp) || rd(&ix->r_major_words, 4, 1, fp)) goto fail_fp;
    ix->r_major = xcalloc(ix->r_major_words, 4);
    if (rd(ix->r_major, 4, ix->r_major_words, fp)) goto fail_fp;
    fclose(fp);
    /* .R.backward.sa (rbwt.c:558-574) */
    if (!(fp = open_sfx(prefix, ".R.backward.sa", "rb", 1))) goto fail;
    if (rd(&ix->r_n_sa, 4, 1, fp)) goto fail_fp;
    ix->r_sa = xcalloc(ix->r_n_sa, 4);
    if (rd(ix->r_sa, 4, ix->r_n_sa, fp)) goto fail_fp;
    fclose(fp);
    /* .ref (metaref.c:61-93) */
    if (!(fp = open_sfx(prefix, ".ref", "rb", 1))) goto fail;
    if (rd(&ix->ref_len, 4, 1, fp)) goto fail_fp;
    ix->ref = xcalloc((ix->ref_len + 7) / 8 + 2, 4);
    if (rd(ix->ref, 4, (ix->ref_len + 7) / 8, fp)) goto fail_fp;
    fclose(fp);
    /* .C.ann / .C.amb (bntseq.c:88-139) */
    if (!(fp = open_sfx(prefix, ".C.ann", "r", 1))) goto fail;
    {
        long long xx; unsigned seed; int i;
        if (fscanf(fp, "%lld%d%u", &xx, &ix->n_seqs, &seed) != 3) goto fail_fp;
        ix->l_pac = xx;
        ix->anns = xcalloc(ix->n_seqs, sizeof(so_ann_t));
        for (i = 0; i < ix->n_seqs; ++i) {
            unsigned gi; char str[1024]; int c;
            if (fscanf(fp, "%u%1023s", &gi, str) != 2) goto fail_fp;
            ix->anns[i].name = strdup(str);
            while ((c = fgetc(fp)) != '\n' && c != EOF) { }
            if (fscanf(fp, "%lld%d%d", &xx, &ix->anns[i].len, &ix->anns[i].n_ambs) != 3) goto fail_fp;
            ix->anns[i].offset = xx;
        }
    }
    fclose(fp);
    if (!(fp = open_sfx(prefix, ".C.amb", "r", 1))) goto fail;
    {
        long long xx; int n_seqs, i;
        if (fscanf(fp, "%lld%d%d", &xx, &n_seqs, &ix->n_holes) != 3) goto fail_fp;
        ix->ambs = xcalloc(ix->n_holes, sizeof(so_amb_t));
        for (i = 0; i < ix->n_holes; ++i) {
            char str[64];
            if (fscanf(fp, "%lld%d%63s", &xx, &ix->ambs[i].len, str) != 3) goto fail_fp;
            ix->ambs[i].offset = xx; ix->ambs[i].amb = str[0];
        }
    }
    fclose(fp);
    /* .C.pac (indexio.h:45-56) */
    if (!(fp = open_sfx(prefix, ".C.pac", "rb", 1))) goto fail;
    ix->pac = xcalloc((size_t)ix->l_pac / 4 + 2, 1);
    if (fread(ix->pac, 1, (size_t)ix->l_pac / 4 + 2, fp) == 0) goto fail_fp;
    fclose(fp);
    /* .C.lkt (lookup.c:47-65); rebuilt from the pac when the 64 MiB file is not there */
    if ((fp = open_sfx(prefix, ".C.lkt", "rb", 0))) {
        int32_t len;
        if (rd(&len, 4, 1, fp)) goto fail_fp;
        ix->lkt_len = (uint32_t)len;
        ix->lkt_n = (1u << (2 * len)) + 1;
        ix->lkt = xcalloc(ix->lkt_n, 4);
        if (rd(ix->lkt, 4, ix->lkt_n, fp)) goto fail_fp;
        fclose(fp);
    } else {
        ix->lkt_len = 12;
        ix->lkt = lkt_from_pac(ix->pac, (uint32_t)ix->l_pac, 12, &ix->lkt_n);
    }
    return ix;
fail_fp:
    fprintf(stderr, "[salt_oracle] short or malformed index file under prefix %s\n", prefix);
    fclose(fp);
fail:
    so_index_free(ix);
    return NULL;
}

void so_index_free(so_index_t *ix)
{
    int i;
    if (!ix) return;
    free(ix->c_bwt); free(ix->c_sa); free(ix->lkt); free(ix->r_bwt); free(ix->r_occ);
    free(ix->r_major); free(ix->r_sa); free(ix->ref); free(ix->pac); free(ix->ambs);
    if (ix->anns) for (i = 0; i < ix->n_seqs; ++i) free(ix->anns[i].name);
    free(ix->anns);
    free(ix);
}

int so_index_seed_len(const so_index_t *ix) { return ix->seed_len; }

void so_opt_default(const so_index_t *ix, so_opt_t *o)
{
    memset(o, 0, sizeof *o);
    o->l_seed = ix->seed_len;       /* aln.c:215-224 */
    o->l_overlap = ix->seed_len;    /* aln.c:223 */
    o->max_seed = 50;               /* aln.c:46 */
    o->max_locate = 1000;           /* aln.c:47 */
    o->max_hits = 5;                /* aln.h:133 */
}

void so_index_arrays(const so_index_t *ix, so_arrays_t *a)
{
    memset(a, 0, sizeof *a);
    a->c_primary = ix->c_primary; memcpy(a->c_L2, ix->c_L2, sizeof a->c_L2);
    a->c_seq_len = ix->c_seq_len; a->c_bwt_size = ix->c_bwt_size; a->c_bwt = ix->c_bwt;
    a->c_sa_intv = ix->c_sa_intv; a->c_n_sa = ix->c_n_sa; a->c_sa = ix->c_sa;
    a->lkt_len = ix->lkt_len; a->lkt_n = ix->lkt_n; a->lkt = ix->lkt;
    a->r_text_len = ix->r_text_len; a->r_inv_sa0 = ix->r_inv_sa0;
    memcpy(a->r_cum, ix->r_cum, sizeof a->r_cum);
    a->r_bwt_words = ix->r_bwt_words; a->r_bwt = ix->r_bwt;
    a->r_occ_words = ix->r_occ_words; a->r_occ = ix->r_occ;
    a->r_major_words = ix->r_major_words; a->r_major = ix->r_major;
    a->r_n_sa = ix->r_n_sa; a->r_sa = ix->r_sa;
    a->ref_len = ix->ref_len; a->ref = ix->ref;
    a->l_pac = (uint64_t)ix->l_pac; a->pac = ix->pac;
}

/* ------------------------------------------------------------------------------------------ */
/* C index: Occ / backward step / SA walk                                                     */
/* ------------------------------------------------------------------------------------------ */
#define C_OCC_INTV 128u

/* symbol k of the $-removed BWT string (bwt.h:57-64) */
static inline uint32_t c_sym(const so_index_t *ix, uint32_t k)
{
    uint32_t w = ix->c_bwt[k / C_OCC_INTV * 12 + 4 + (k % C_OCC_INTV) / 16];
    return (w >> ((~k & 15u) << 1)) & 3u;
}

/* how many of the first m (1..16) symbols of a 16-symbol word equal c */
static inline uint32_t cnt_word2(uint32_t w, uint32_t c, uint32_t m)
{
    uint32_t y = ~(w ^ (c * 0x55555555u));
    y = y & (y >> 1) & 0x55555555u;
    if (m < 16) y &= ~((1u << (2 * (16 - m))) - 1u);
    return (uint32_t)__builtin_popcount(y);
}

/* Occ(k, c): occurrences of c in BWT rows [0, k] (bwt.c:113-138) */
static uint32_t c_occ(const so_index_t *ix, uint32_t k, uint32_t c, so_counters_t *ctr)
{
    if (k == ix->c_seq_len) return ix->c_L2[c + 1] - ix->c_L2[c];
    if (k == (uint32_t)-1) return 0;
    if (ctr) ++ctr->n_occC;
    if (k >= ix->c_primary) --k;           /* '$' is not stored (bwt.c:120) */
    const uint32_t *p = ix->c_bwt + k / C_OCC_INTV * 12;
    uint32_t n = p[c];
    uint32_t in_blk = k % C_OCC_INTV;      /* rows blockstart..k inclusive */
    uint32_t full = in_blk / 16, w;
    p += 4;
    for (w = 0; w < full; ++w) n += cnt_word2(p[w], c, 16);
    n += cnt_word2(p[full], c, in_blk % 16 + 1);
    return n;
}

/* one backward-search step on [k, l] (bwt.c:281-309 body) ; returns 0 when the interval dies */
static inline int c_step(const so_index_t *ix, uint32_t c, uint32_t *k, uint32_t *l, so_counters_t *ctr)
{
    uint32_t ok = c_occ(ix, *k - 1, c, ctr), ol = c_occ(ix, *l, c, ctr);
    *k = ix->c_L2[c] + ok + 1;
    *l = ix->c_L2[c] + ol;
    return *k <= *l;
}

/* bwt_match_exact_alt (bwt.c:281-309): extend [k0,l0] by str[len-1..0]; untouched on failure */
static int c_match_exact(const so_index_t *ix, int len, const uint8_t *str, uint32_t *k0, uint32_t *l0,
                         so_counters_t *ctr)
{
    uint32_t k = *k0, l = *l0;
    int i;
    for (i = len - 1; i >= 0; --i) {
        if (str[i] > 3) return 0;
        if (!c_step(ix, str[i], &k, &l, ctr)) return 0;
    }
    *k0 = k; *l0 = l;
    return (int)(l - k + 1);
}

/* bwt_sa (bwt.c:89-102) with bwt_invPsi (bwt.h:67-71) */
static uint32_t c_sa(const so_index_t *ix, uint32_t k, so_counters_t *ctr)
{
    uint32_t steps = 0;
    if (ctr) ++ctr->n_saC;
    while (k % ix->c_sa_intv != 0) {
        ++steps;
        if (k == ix->c_primary) k = 0;
        else {
            uint32_t c = c_sym(ix, k < ix->c_primary ? k : k - 1);
            k = ix->c_L2[c] + c_occ(ix, k, c, ctr);
        }
    }
    return steps + ix->c_sa[k / ix->c_sa_intv];
}

/* ------------------------------------------------------------------------------------------ */
/* 12-mer table (lookup.h:39-53, lookup.c:163-177)                                            */
/* ------------------------------------------------------------------------------------------ */
static void lkt_lookup(const so_index_t *ix, const uint8_t *seq, int from, int to, uint32_t *k, uint32_t *l,
                       so_counters_t *ctr)
{
    uint32_t x = 0;
    int i;
    for (i = from; i <= to; ++i) {
        if (seq[i] > 3) { *k = 1; *l = 0; return; }
        x = (x << 2) | seq[i];
    }
    if (ctr) ++ctr->n_lkt;
    *k = ix->lkt[x];
    *l = ix->lkt[x + 1] - 1;
}

/* ------------------------------------------------------------------------------------------ */
/* R index                                                                                    */
/* ------------------------------------------------------------------------------------------ */
#define R_OCC_INTV 256u
#define R_OCC_MAJOR 65536u
#define R_SHARP 4u

static inline uint32_t r_nib(const so_index_t *ix, uint32_t i)
{
    return (ix->r_bwt[i >> 3] >> ((7u - (i & 7u)) * 4u)) & 15u;
}

/* explicit checkpoint value (rbwt.c:40-80) */
static inline uint32_t r_explicit(const so_index_t *ix, uint32_t e, uint32_t c)
{
    uint32_t major = ix->r_major[(e * R_OCC_INTV / R_OCC_MAJOR) * 5 + c];
    uint32_t w = ix->r_occ[e / 2 * 5 + c];
    return major + ((e & 1u) ? (w & 0xFFFFu) : (w >> 16));
}

/* Rbwt_BWTOccValue (rbwt.c:159-191): # of c among the first `index` stored symbols, evaluated
 * from the nearest 256-symbol checkpoint, counting forwards or backwards (rbwt.c:81-147) */
static uint32_t r_occ(const so_index_t *ix, uint32_t index, uint32_t c, so_counters_t *ctr)
{
    if (index > ix->r_inv_sa0) --index;       /* '$' not stored (rbwt.c:165) */
    uint32_t e = (index + R_OCC_INTV / 2 - 1) / R_OCC_INTV;
    uint32_t at = e * R_OCC_INTV, v = r_explicit(ix, e, c), i, n = 0;
    if (ctr) { ++ctr->n_occR; ctr->n_occR_syms += at > index ? at - index : index - at; }
    if (at == index) return v;
    if (at < index) {
        for (i = at; i < index; ++i) n += r_nib(ix, i) == c;
        return v + n;
    }
    for (i = index; i < at; ++i) n += r_nib(ix, i) == c;
    return v - n;
}

/* Rbwt_bwt2nt (rbwt.h:103-122): the '$' row reads as '#' */
static inline uint32_t r_bwt2nt(const so_index_t *ix, uint32_t pos)
{
    if (pos == ix->r_inv_sa0) return R_SHARP;
    if (pos > ix->r_inv_sa0) --pos;
    return r_nib(ix, pos);
}

/* Rbwt_exact_match_backward (rbwt.c:619-648) */
static int r_match_backward(const so_index_t *ix, const uint8_t *q, int qlen, uint32_t *k, uint32_t *l,
                            so_counters_t *ctr)
{
    uint32_t k0 = *k, l0 = *l;
    int step = 0;
    if (qlen <= 0) return 0;
    while (k0 <= l0 && step < qlen) {
        uint32_t c = q[qlen - step - 1];
        if (c > 3) return 0;
        k0 = ix->r_cum[c] + r_occ(ix, k0, c, ctr) + 1;
        l0 = ix->r_cum[c] + r_occ(ix, l0 + 1, c, ctr);
        ++step;
    }
    *k = k0; *l = l0;
    return l0 >= k0;
}

/* Rbwt_back_bwt_sa (rbwt.c:316-333): LF-walk to the preceding '#' */
static uint32_t r_back_sa(const so_index_t *ix, uint32_t sa_index, so_counters_t *ctr)
{
    uint32_t step = 0;
    if (ctr) ++ctr->n_saR;
    while (sa_index <= ix->r_cum[R_SHARP]) {
        uint32_t c = r_bwt2nt(ix, sa_index);
        if (ctr) ++ctr->n_bwt2nt;
        sa_index = ix->r_cum[c] + r_occ(ix, sa_index, c, ctr) + 1;
        ++step;
    }
    return ix->r_sa[sa_index - ix->r_cum[R_SHARP] - 1] + step - 1;
}

/* ------------------------------------------------------------------------------------------ */
/* klib introsort, restated as a generic routine over an index-comparable array               */
/* (ksort.h:159-228; the instability decides which seed is located first under the cap)       */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint32_t sp, ep, offset; } so_sai_t;
#define SAI_LT(a, b) ((a).ep - (a).sp < (b).ep - (b).sp)      /* alnse.c:35 */

static void sai_insertsort(so_sai_t *s, so_sai_t *t)
{
    so_sai_t *i, *j, tmp;
    for (i = s + 1; i < t; ++i)
        for (j = i; j > s && SAI_LT(*j, *(j - 1)); --j) { tmp = *j; *j = *(j - 1); *(j - 1) = tmp; }
}

static void sai_combsort(size_t n, so_sai_t *a)
{
    const double shrink = 1.2473309501039786540366528676643;
    int do_swap;
    size_t gap = n;
    so_sai_t tmp, *i, *j;
    do {
        if (gap > 2) {
            gap = (size_t)(gap / shrink);
            if (gap == 9 || gap == 10) gap = 11;
        }
        do_swap = 0;
        for (i = a; i < a + n - gap; ++i) {
            j = i + gap;
            if (SAI_LT(*j, *i)) { tmp = *i; *i = *j; *j = tmp; do_swap = 1; }
        }
    } while (do_swap || gap > 2);
    if (gap != 1) sai_insertsort(a, a + n);
}

static void sai_introsort(size_t n, so_sai_t *a)
{
    struct { so_sai_t *left, *right; int depth; } stack[8 * sizeof(size_t) + 2], *top = stack;
    so_sai_t rp, tmp, *s, *t, *i, *j, *k;
    int d;
    if (n < 1) return;
    if (n == 2) { if (SAI_LT(a[1], a[0])) { tmp = a[0]; a[0] = a[1]; a[1] = tmp; } return; }
    for (d = 2; (1ul << d) < n; ++d) { }
    s = a; t = a + (n - 1); d <<= 1;
    for (;;) {
        if (s < t) {
            if (--d == 0) { sai_combsort((size_t)(t - s) + 1, s); t = s; continue; }
            i = s; j = t; k = i + ((j - i) >> 1) + 1;
            if (SAI_LT(*k, *i)) { if (SAI_LT(*k, *j)) k = j; }
            else k = SAI_LT(*j, *i) ? i : j;
            rp = *k;
            if (k != t) { tmp = *k; *k = *t; *t = tmp; }
            for (;;) {
                do ++i; while (SAI_LT(*i, rp));
                do --j; while (i <= j && SAI_LT(rp, *j));
                if (j <= i) break;
                tmp = *i; *i = *j; *j = tmp;
            }
            tmp = *i; *i = *t; *t = tmp;
            if (i - s > t - i) {
                if (i - s > 16) { top->left = s; top->right = i - 1; top->depth = d; ++top; }
                s = t - i > 16 ? i + 1 : t;
            } else {
                if (t - i > 16) { top->left = i + 1; top->right = t; top->depth = d; ++top; }
                t = i - s > 16 ? i - 1 : s;
            }
        } else {
            if (top == stack) { sai_insertsort(a, a + n); return; }
            --top; s = top->left; t = top->right; d = top->depth;
        }
    }
}

static int cmp_u32(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

/* ------------------------------------------------------------------------------------------ */
/* per-thread scratch (aln.h:99-109)                                                          */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    so_sai_t *sai_c, *sai_r;
    int n_c, n_r, cap_sai;
    uint32_t *loci; uint32_t n_loci, cap_loci;
    so_hit_t *hits; uint32_t n_hits, cap_hits;
} so_aux_t;

static void aux_reserve(so_aux_t *a, int n_sai, uint32_t n_loci)
{
    if (n_sai > a->cap_sai) {
        a->sai_c = realloc(a->sai_c, sizeof(so_sai_t) * n_sai);
        a->sai_r = realloc(a->sai_r, sizeof(so_sai_t) * n_sai);
        a->cap_sai = n_sai;
    }
    if (n_loci > a->cap_loci) {
        a->loci = realloc(a->loci, 4 * (size_t)n_loci);
        a->hits = realloc(a->hits, sizeof(so_hit_t) * (size_t)n_loci);
        a->cap_loci = a->cap_hits = n_loci;
    }
}

static void aux_free(so_aux_t *a) { free(a->sai_c); free(a->sai_r); free(a->loci); free(a->hits); }

/* ------------------------------------------------------------------------------------------ */
/* S1: seeding (alnse.c:199-312)                                                              */
/* ------------------------------------------------------------------------------------------ */
static void seed_overlap(const so_index_t *ix, const so_opt_t *o, const uint8_t *seq, int l_seq,
                         so_aux_t *aux, so_counters_t *ctr)
{
    const int ks = o->l_seed, lk = (int)ix->lkt_len;
    int s, n_c = 0, n_r = 0;
    for (s = 0; s + ks <= l_seq; ++s) {
        if (s % o->l_overlap != 0) continue;
        const int e = s + ks - 1;
        uint32_t k = 1, l = ix->c_seq_len;
        /* plain-genome seed: 12-mer table on the tail, backward search on the head */
        lkt_lookup(ix, seq, e - lk + 1, e, &k, &l, ctr);
        if (k <= l && c_match_exact(ix, ks - lk, seq + s, &k, &l, ctr) > 0) {
            int ext = 0;
            while (l - k > o->max_seed && ext < s) {          /* alnse.c:248-258 */
                uint32_t c = seq[s - ext - 1], ok, ol;
                if (c > 3) break;
                ok = c_occ(ix, k - 1, c, ctr); ol = c_occ(ix, l, c, ctr);
                if (ok + 1 > ol) break;
                k = ix->c_L2[c] + ok + 1; l = ix->c_L2[c] + ol;
                ++ext;
                if (l - k <= o->max_seed) break;
            }
            aux->sai_c[n_c].sp = k; aux->sai_c[n_c].ep = l; aux->sai_c[n_c].offset = (uint32_t)(s - ext);
            ++n_c;
        }
        if (o->seed_only_ref) continue;                        /* alnse.c:272 */
        /* SNP-aware seed on the local-pattern index */
        k = 0; l = ix->r_text_len;
        if (r_match_backward(ix, seq + s, ks, &k, &l, ctr) > 0) {
            int ext = 0;
            while (l - k > o->max_seed && ext < s) {          /* alnse.c:280-291: no N guard */
                uint32_t c = seq[s - ext - 1], ok, ol;
                ok = r_occ(ix, k, c, ctr); ol = r_occ(ix, l + 1, c, ctr);
                if (ok + 1 > ol) break;
                k = ix->r_cum[c] + ok + 1; l = ix->r_cum[c] + ol;
                ++ext;
                if (l - k <= o->max_seed) break;
            }
            aux->sai_r[n_r].sp = k; aux->sai_r[n_r].ep = l; aux->sai_r[n_r].offset = (uint32_t)(s - ext);
            ++n_r;
        }
    }
    aux->n_c = n_c; aux->n_r = n_r;
    sai_introsort((size_t)n_c, aux->sai_c);                    /* alnse.c:307-308 */
    sai_introsort((size_t)n_r, aux->sai_r);
}

/* ------------------------------------------------------------------------------------------ */
/* L1: locate (alnse.c:633-731)                                                               */
/* ------------------------------------------------------------------------------------------ */
static void locate_alt(const so_index_t *ix, const so_opt_t *o, uint32_t l_seq, so_aux_t *aux,
                       so_counters_t *ctr)
{
    const uint32_t l_ref = ix->ref_len;
    int i;
    uint32_t j;
    aux->n_loci = 0;
    for (i = 0; i < aux->n_c; ++i) {
        const so_sai_t *c = aux->sai_c + i;
        for (j = c->sp; j <= c->ep; ++j) {
            uint32_t pos = c_sa(ix, j, ctr) - c->offset;
            if (pos + l_seq > l_ref) continue;                /* u32 wrap kept as is (alnse.c:672-673) */
            aux->loci[aux->n_loci++] = pos;
            if (aux->n_loci == o->max_locate) goto done;
        }
    }
    for (i = 0; i < aux->n_r; ++i) {
        const so_sai_t *r = aux->sai_r + i;
        uint32_t skip = (r->ep + 1 - r->sp) / 0x40000u;        /* alnse.c:707-708 */
        if ((int)skip <= 0) skip = 1;
        for (j = r->sp; j <= r->ep; j += skip) {
            uint32_t pos = r_back_sa(ix, j, ctr) - r->offset;
            if (pos > l_ref || pos + l_seq > l_ref) continue;
            aux->loci[aux->n_loci++] = pos;
            if (aux->n_loci == o->max_locate) goto done;
        }
    }
done:
    qsort(aux->loci, aux->n_loci, 4, cmp_u32);                /* ks_introsort(uint32_t): any sort */
}

/* ------------------------------------------------------------------------------------------ */
/* V2: masked Hamming (editdistance.c:88-163)                                                 */
/* ------------------------------------------------------------------------------------------ */
static const uint8_t NT2BIT[5] = { 1, 2, 4, 8, 15 };          /* editdistance.c:40 */

static inline uint32_t ref_nib(const uint32_t *ref, uint32_t i) { return (ref[i >> 3] >> (4 * (i & 7))) & 15u; }

int so_ed_mismatch(const uint32_t *ref, uint32_t pos, const uint8_t *seq, uint32_t L, int max_err)
{
    int n = 0;
    uint32_t i;
    for (i = 0; i < L; ++i)
        if ((ref_nib(ref, pos + i) & NT2BIT[seq[i]]) == 0 && ++n > max_err) return -1;
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* V4 / C1: Landau-Vishkin on masks (LandauVishkin.c:19-122, 176-470; editdistance.c:174-284)  */
/* ------------------------------------------------------------------------------------------ */
#define LV_MAXK 31
typedef struct { uint8_t *t, *p; int tlen, plen; } so_lvbuf_t;

/* unpack text masks and one-hot pattern into zero-padded byte buffers (editdistance.c:183-227) */
static void lv_unpack(const uint32_t *ref, uint32_t pos, uint32_t l_ref, const uint8_t *seq, uint32_t L,
                      so_lvbuf_t *b)
{
    uint32_t i;
    size_t tcap = (l_ref + 15) / 8 * 8, pcap = (L + 15) / 8 * 8;
    /* generous extra zero padding: the reference may peek a few bytes past its own buffers for
     * long reads (undefined there); zeros are what its calloc'ed heap normally holds */
    b->t = xcalloc(tcap + 64, 1); b->p = xcalloc(pcap + 64, 1);
    for (i = 0; i < l_ref; ++i) b->t[i] = (uint8_t)ref_nib(ref, pos + i);
    for (i = 0; i < L; ++i) b->p[i] = seq[i] > 3 ? 15 : (uint8_t)(1u << seq[i]);
    b->tlen = (int)l_ref; b->plen = (int)L;
}

/* length of the matching run that starts at pattern offset `from` on diagonal d, capped at end */
static inline int lv_run(const so_lvbuf_t *b, int d, int from, int end)
{
    int i = from;
    if (i >= end) return end;
    while (i < end && (b->p[i] & b->t[d + i]) != 0) ++i;
    return i;
}

static int lv_distance(const so_lvbuf_t *b, int k)
{
    short L[LV_MAXK + 1][2 * LV_MAXK + 1];
    int e, d, i, j;
    for (i = 0; i <= LV_MAXK; ++i) for (j = 0; j <= 2 * LV_MAXK; ++j) L[i][j] = -2;
    if (k > LV_MAXK - 1) k = LV_MAXK - 1;
    int end0 = b->plen < b->tlen ? b->plen : b->tlen;
    L[0][LV_MAXK] = (short)lv_run(b, 0, 0, end0);
    if (L[0][LV_MAXK] == end0) return b->plen > end0 ? b->plen - end0 : 0;
    for (e = 1; e <= k; ++e) {
        for (d = 0; d != e + 1; d = (d > 0 ? -d : -d + 1)) {   /* 0, 1, -1, 2, -2 ... */
            int best = L[e - 1][LV_MAXK + d] + 1;
            int left = L[e - 1][LV_MAXK + d - 1];
            int right = L[e - 1][LV_MAXK + d + 1] + 1;
            if (left > best) best = left;
            if (right > best) best = right;
            if (b->p[best] == b->t[d + best]) {                /* equality gate (LandauVishkin.c:79) */
                int end = b->plen < b->tlen - d ? b->plen : b->tlen - d;
                best = lv_run(b, d, best, end);
            }
            if (best == b->plen) return e;
            L[e][LV_MAXK + d] = (short)best;
        }
    }
    return -1;
}

static int cig_put(char **o, int *cap, int count, char code)   /* writeCigar, COMPACT_CIGAR_STRING */
{
    int w;
    if (count <= 0) return 1;
    if (*cap == 0) { *(*o - 1) = 0; return 0; }
    w = snprintf(*o, (size_t)*cap, "%d%c", count, code);
    if (w > *cap - 1) return 0;
    *o += w; *cap -= w;
    return 1;
}

/* useM = 1, COMPACT string (LandauVishkin.c:176-470) */
static int lv_cigar(const so_lvbuf_t *b, int k, char *out, int cap)
{
    short L[LV_MAXK + 1][2 * LV_MAXK + 1];
    char A[LV_MAXK + 1][2 * LV_MAXK + 1];
    char act[LV_MAXK + 1];
    int matched[LV_MAXK + 1];
    int e, d, i, j;
    if (k >= LV_MAXK) return -3;                              /* reference asserts (LandauVishkin.c:183) */
    for (i = 0; i <= LV_MAXK; ++i) for (j = 0; j <= 2 * LV_MAXK; ++j) L[i][j] = -2;
    int end0 = b->plen < b->tlen ? b->plen : b->tlen;
    L[0][LV_MAXK] = (short)lv_run(b, 0, 0, end0);
    if (L[0][LV_MAXK] == end0) {
        if (!cig_put(&out, &cap, b->plen, 'M')) return -2;
        return 0;
    }
    for (e = 1; e <= k; ++e) {
        for (d = 0; d != -(e + 1); d = (d >= 0 ? -(d + 1) : -d)) {   /* 0, -1, 1, -2, 2 ... */
            int best = L[e - 1][LV_MAXK + d] + 1;
            int left = L[e - 1][LV_MAXK + d - 1];
            int right = L[e - 1][LV_MAXK + d + 1] + 1;
            A[e][LV_MAXK + d] = 'X';
            if (left > best) { best = left; A[e][LV_MAXK + d] = 'D'; }
            if (right > best) { best = right; A[e][LV_MAXK + d] = 'I'; }
            if (b->p[best] == b->t[d + best]) {
                int end = b->plen < b->tlen - d ? b->plen : b->tlen - d;
                best = lv_run(b, d, best, end);
            }
            L[e][LV_MAXK + d] = (short)best;
            if (best != b->plen) continue;
            /* trace back, then emit forward, merging =/X into M */
            int cd = d, ce;
            for (ce = e; ce >= 1; --ce) {
                act[ce] = A[ce][LV_MAXK + cd];
                if (act[ce] == 'I') {
                    matched[ce] = L[ce][LV_MAXK + cd] - L[ce - 1][LV_MAXK + cd + 1] - 1; cd += 1;
                } else if (act[ce] == 'D') {
                    matched[ce] = L[ce][LV_MAXK + cd] - L[ce - 1][LV_MAXK + cd - 1]; cd -= 1;
                } else {
                    matched[ce] = L[ce][LV_MAXK + cd] - L[ce - 1][LV_MAXK + cd] - 1;
                }
            }
            int acc = L[0][LV_MAXK];
            ce = 1;
            while (ce <= e) {
                char a = act[ce];
                int cnt = 1;
                while (ce + 1 <= e && matched[ce] == 0 && act[ce + 1] == a) { ++cnt; ++ce; }
                if (a == 'X') acc += cnt;
                else {
                    if (acc != 0) { if (!cig_put(&out, &cap, acc, 'M')) return -2; acc = 0; }
                    if (!cig_put(&out, &cap, cnt, a)) return -2;
                }
                if (matched[ce] > 0) acc += matched[ce];
                ++ce;
            }
            if (acc != 0 && !cig_put(&out, &cap, acc, 'M')) return -2;
            *(out - (cap == 0 ? 1 : 0)) = 0;
            return e;
        }
    }
    *(out - (cap == 0 ? 1 : 0)) = 0;
    return -1;
}

int so_ed_diff(const uint32_t *ref, uint32_t l_mref, uint32_t pos, uint32_t l_ref, const uint8_t *seq,
               uint32_t L, int k)
{
    so_lvbuf_t b;
    int r;
    if (pos > l_mref || pos + l_ref > l_mref) return -1;      /* editdistance.c:178 */
    lv_unpack(ref, pos, l_ref, seq, L, &b);
    r = lv_distance(&b, k);
    free(b.t); free(b.p);
    return r;
}

int so_ed_diff_cigar(const uint32_t *ref, uint32_t pos, uint32_t l_ref, const uint8_t *seq, uint32_t L,
                     int k, char *cigar, int cap)
{
    so_lvbuf_t b;
    int r;
    lv_unpack(ref, pos, l_ref, seq, L, &b);
    r = lv_cigar(&b, k, cigar, cap);
    free(b.t); free(b.p);
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* V1 / V3: candidate checks with the best/first-hit rule (alnse.c:348-393,734-782,871-901)    */
/* ------------------------------------------------------------------------------------------ */
#define NO_MATCH (-1)

static int check_nogap(const so_index_t *ix, so_result_t *q, const uint8_t *seq, uint32_t l_seq,
                       int max_diff, int strand, so_aux_t *aux, so_counters_t *ctr)
{
    int found = 0;
    uint32_t i, prev = (uint32_t)-1;
    for (i = 0; i < aux->n_loci; ++i) {
        uint32_t pos = aux->loci[i];
        int n;
        if (pos == prev || pos >= ix->ref_len) continue;
        if (ctr) { ++ctr->n_verify; ctr->n_verify_words += (pos % 8 + l_seq + 7) / 8; }
        n = so_ed_mismatch(ix->ref, pos, seq, l_seq, max_diff);
        if (n >= 0) {
            if (n < max_diff || !found) {
                max_diff = n;
                q->is_gap = 0; q->n_diff = (uint8_t)n; q->strand = strand; q->pos = pos;
            }
            found = 1;
            so_hit_t *h = aux->hits + aux->n_hits++;
            h->is_gap = 0; h->n_diff = (uint8_t)n; h->pos = pos; h->strand = (uint16_t)strand;
        }
        prev = pos;
    }
    return found ? max_diff : NO_MATCH;
}

static int check_withgap(const so_index_t *ix, so_result_t *q, const uint8_t *seq, uint32_t l_seq,
                         int max_diff, int strand, so_aux_t *aux, so_counters_t *ctr)
{
    int found = 0;
    uint32_t i, prev = (uint32_t)-1;
    for (i = 0; i < aux->n_loci; ++i) {
        uint32_t pos = aux->loci[i];
        int n;
        if (pos == prev || pos + l_seq + 4 >= ix->ref_len) continue;
        if (ctr) ++ctr->n_lv;
        n = so_ed_diff(ix->ref, ix->ref_len, pos, l_seq + 4, seq, l_seq, max_diff);
        if (n >= 0) {
            if (n < max_diff || !found) {
                max_diff = n;
                q->is_gap = 1; q->n_diff = (uint8_t)n; q->strand = strand; q->pos = pos;
            }
            found = 1;
            so_hit_t *h = aux->hits + aux->n_hits++;
            h->is_gap = 1; h->n_diff = (uint8_t)n; h->pos = pos; h->strand = (uint16_t)strand;
        }
        prev = pos;
    }
    return found ? max_diff : NO_MATCH;
}

/* H2 (query.c:270-281) */
static uint32_t gen_mapq(uint32_t b0, uint32_t b1)
{
    if (b0 == 0) return 0;
    double a = 255.0;
    uint32_t mapq = (uint32_t)(a * ((double)abs((int)(b0 - b1)) / (double)b0));
    return mapq < 254 ? mapq : 254;
}

/* H1 (query.c:297-333), including the a[0].n_diff and stale last_pos behaviour */
static void set_hits(so_result_t *q, int max_hits, so_aux_t *aux[2])
{
    int s, tot = 0;
    uint32_t primary = q->pos;
    q->b0 = q->n_diff;
    q->b1 = 100000;
    for (s = 0; s < 2; ++s) {
        const so_hit_t *a = aux[s]->hits;
        uint32_t j;
        for (j = 0; j < aux[s]->n_hits; ++j) {
            if (a[j].pos == (uint32_t)-1 || a[j].pos == primary) continue;
            if (a[0].n_diff <= q->n_diff) {
                if (a[0].n_diff <= q->b1) q->b1 = a[0].n_diff;
                if (q->n_hits[s] < SO_MAX_HITS) q->hits[s][q->n_hits[s]] = a[j];
                ++q->n_hits[s];
                ++tot;
            }
            if (tot == max_hits) goto end;
        }
    }
end:
    q->mapq = (uint8_t)gen_mapq((uint32_t)q->b0, (uint32_t)q->b1);
}

/* C1 (query.c:282-296) */
static void gen_cigar(const so_index_t *ix, so_result_t *q, const uint8_t *seq, const uint8_t *rseq, int l_seq)
{
    q->seq_start = 0; q->seq_end = (uint32_t)l_seq - 1;
    if (q->pos == 0xFFFFFFFFu) return;
    if (q->is_gap)
        so_ed_diff_cigar(ix->ref, q->pos, (uint32_t)l_seq + 4, q->strand == 0 ? seq : rseq, (uint32_t)l_seq,
                         q->n_diff, q->cigar, SO_CIGAR_MAX);
    else snprintf(q->cigar, SO_CIGAR_MAX, "%dM", l_seq);
}

static void revcomp(const uint8_t *seq, int n, uint8_t *out)   /* query.c:46-71 */
{
    int i;
    for (i = 0; i < n; ++i) { uint8_t c = seq[n - 1 - i]; out[i] = c < 4 ? (uint8_t)(3 - c) : c; }
}

static void result_init(so_result_t *q, int l_seq)             /* query.c:201-206 */
{
    memset(q, 0, sizeof *q);
    q->pos = 0xFFFFFFFFu; q->n_diff = 255; q->is_gap = 255; q->strand = 3; q->b0 = -1; q->b1 = -1;
    q->seq_start = 0; q->seq_end = (uint32_t)l_seq - 1;
}

/* alnse_overlap_alt + query_gen_cigar (alnse.c:1045-1104, 1316-1352) */
static void align_se1(const so_index_t *ix, const so_opt_t *o, const uint8_t *seq, int l_seq, so_result_t *q,
                      so_aux_t *aux[2], uint8_t *rseq, so_counters_t *ctr)
{
    int n_amb = 0, i, n0, n1, max_diff;
    result_init(q, l_seq);
    for (i = 0; i < l_seq; ++i) n_amb += seq[i] > 3;
    if (ctr) ++ctr->n_reads;
    if (n_amb > 200) return;                                   /* alnse.c:1281,1328 */
    revcomp(seq, l_seq, rseq);
    int n_sai = l_seq - o->l_seed + 1;
    for (i = 0; i < 2; ++i) {
        aux_reserve(aux[i], n_sai > 1 ? n_sai : 1, o->max_locate + 1);
        aux[i]->n_c = aux[i]->n_r = 0; aux[i]->n_loci = 0; aux[i]->n_hits = 0;
    }
    for (i = 0; i < 2; ++i) {
        so_counters_t before;
        if (ctr) before = *ctr;
        seed_overlap(ix, o, i == 0 ? seq : rseq, l_seq, aux[i], ctr);
        if (ctr) {
            ctr->n_occC_seed += ctr->n_occC - before.n_occC;
            ctr->n_occR_seed += ctr->n_occR - before.n_occR;
            ctr->n_occR_syms_seed += ctr->n_occR_syms - before.n_occR_syms;
        }
        locate_alt(ix, o, (uint32_t)l_seq, aux[i], ctr);
    }
    max_diff = 3;                                              /* alnse.c:1079 */
    n0 = check_nogap(ix, q, seq, (uint32_t)l_seq, max_diff, 0, aux[0], ctr);
    if (n0 != NO_MATCH && n0 < max_diff) max_diff = n0;
    n1 = check_nogap(ix, q, rseq, (uint32_t)l_seq, max_diff, 1, aux[1], ctr);
    if (n1 != NO_MATCH && n1 < max_diff) max_diff = n1;
    if (n0 == NO_MATCH && n1 == NO_MATCH) {
        max_diff = l_seq / 10;                                 /* alnse.c:1090 */
        int d0 = check_withgap(ix, q, seq, (uint32_t)l_seq, max_diff, 0, aux[0], ctr);
        if (d0 != NO_MATCH && d0 < max_diff) max_diff = d0;
        (void)check_withgap(ix, q, rseq, (uint32_t)l_seq, max_diff, 1, aux[1], ctr);
    }
    set_hits(q, o->max_hits, aux);
    gen_cigar(ix, q, seq, rseq, l_seq);
    if (ctr) { ctr->n_bases += (uint64_t)l_seq; ctr->n_hits_out += (uint64_t)(q->n_hits[0] + q->n_hits[1]); }
}

void so_align_se1(const so_index_t *ix, const so_opt_t *o, const uint8_t *seq, int l_seq, so_result_t *res,
                  so_counters_t *ctr)
{
    so_aux_t a0, a1, *aux[2] = { &a0, &a1 };
    uint8_t *rseq = xcalloc((size_t)l_seq + 1, 1);
    memset(&a0, 0, sizeof a0); memset(&a1, 0, sizeof a1);
    align_se1(ix, o, seq, l_seq, res, aux, rseq, ctr);
    aux_free(&a0); aux_free(&a1); free(rseq);
}

typedef struct {
    const so_index_t *ix; const so_opt_t *o; int n, tid, nt;
    const uint8_t *seqs; const uint32_t *offs; so_result_t *res; so_counters_t ctr; int want_ctr;
} so_job_t;

static void *batch_worker(void *p)
{
    so_job_t *j = p;
    so_aux_t a0, a1, *aux[2] = { &a0, &a1 };
    uint8_t *rseq = NULL; size_t rcap = 0;
    int i;
    memset(&a0, 0, sizeof a0); memset(&a1, 0, sizeof a1);
    for (i = j->tid; i < j->n; i += j->nt) {                  /* alnse.c:1321 static interleave */
        int l = (int)(j->offs[i + 1] - j->offs[i]);
        if ((size_t)l + 1 > rcap) { rcap = (size_t)l + 64; rseq = realloc(rseq, rcap); }
        align_se1(j->ix, j->o, j->seqs + j->offs[i], l, j->res + i, aux, rseq, j->want_ctr ? &j->ctr : NULL);
    }
    aux_free(&a0); aux_free(&a1); free(rseq);
    return NULL;
}

void so_align_se_batch(const so_index_t *ix, const so_opt_t *o, int n, const uint8_t *seqs, const uint32_t *offs,
                       so_result_t *res, int n_threads, so_counters_t *ctr)
{
    int t;
    if (n_threads < 1) n_threads = 1;
    so_job_t *jobs = xcalloc((size_t)n_threads, sizeof *jobs);
    pthread_t *th = xcalloc((size_t)n_threads, sizeof *th);
    for (t = 0; t < n_threads; ++t) {
        jobs[t].ix = ix; jobs[t].o = o; jobs[t].n = n; jobs[t].tid = t; jobs[t].nt = n_threads;
        jobs[t].seqs = seqs; jobs[t].offs = offs; jobs[t].res = res; jobs[t].want_ctr = ctr != NULL;
        if (n_threads == 1) batch_worker(jobs + t);
        else pthread_create(th + t, NULL, batch_worker, jobs + t);
    }
    for (t = 0; t < n_threads; ++t) {
        if (n_threads > 1) pthread_join(th[t], NULL);
        if (ctr) {
            uint64_t *d = (uint64_t *)ctr; const uint64_t *s = (const uint64_t *)&jobs[t].ctr;
            size_t k;
            for (k = 0; k < sizeof(so_counters_t) / 8; ++k) d[k] += s[k];
        }
    }
    free(jobs); free(th);
}

/* ------------------------------------------------------------------------------------------ */
/* O1: SAM text (sam.c:56-328)                                                                */
/* ------------------------------------------------------------------------------------------ */
typedef struct { char *s; size_t l, cap; int ovf; } so_str_t;

static void sput(so_str_t *s, const char *fmt, ...)
{
    va_list ap;
    if (s->ovf) return;
    va_start(ap, fmt);
    int w = vsnprintf(s->s + s->l, s->cap - s->l, fmt, ap);
    va_end(ap);
    if (w < 0 || (size_t)w >= s->cap - s->l) { s->ovf = 1; return; }
    s->l += (size_t)w;
}

/* sequence id for a pac coordinate (bntseq.c:269-289) */
static int coor_rid(const so_index_t *ix, int64_t pac_coor)
{
    int left = 0, mid = 0, right = ix->n_seqs;
    while (left < right) {
        mid = (left + right) >> 1;
        if (pac_coor >= ix->anns[mid].offset) {
            if (mid == ix->n_seqs - 1) break;
            if (pac_coor < ix->anns[mid + 1].offset) break;
            left = mid + 1;
        } else right = mid;
    }
    return mid;
}

static inline uint32_t pac_base(const uint8_t *pac, uint32_t l) { return (pac[l >> 2] >> ((~l & 3) << 1)) & 3u; }

int so_sam_header(const so_index_t *ix, const so_opt_t *o, char *buf, size_t cap)
{
    so_str_t s = { buf, 0, cap, 0 };
    int i;
    sput(&s, "@HD\tVN:ec1fec2\tSO:unsorted\n");
    for (i = 0; i < ix->n_seqs; ++i) sput(&s, "@SQ\tSN:%s\tLN:%d\n", ix->anns[i].name, ix->anns[i].len);
    sput(&s, "@RG\tID:%s\n", o->rg_id ? o->rg_id : "(null)");
    return s.ovf ? -1 : (int)s.l;
}

/* XA (sam.c:186-240) */
static void sam_xa(so_str_t *s, const so_index_t *ix, const so_opt_t *o, const uint8_t *seq, const uint8_t *rseq,
                   int l_seq, const so_result_t *q)
{
    int strand, i, first = 1;
    for (strand = 0; strand < 2; ++strand) {
        for (i = 0; i < q->n_hits[strand] && i < SO_MAX_HITS; ++i) {
            const so_hit_t *h = &q->hits[strand][i];
            if (h->pos == q->pos) continue;
            if (first) { sput(s, "\tXA:Z:"); first = 0; }
            int rid = coor_rid(ix, h->pos);
            sput(s, "%s,", ix->anns[rid].name);
            sput(s, "%c%lu,", "+-"[strand], (unsigned long)((int64_t)h->pos - ix->anns[rid].offset + 1));
            if (o->print_xa_cigar) {
                if (h->is_gap) {
                    char cig[256];
                    memset(cig, 0, sizeof cig);
                    so_ed_diff_cigar(ix->ref, h->pos, (uint32_t)l_seq + 4, strand == 0 ? seq : rseq, (uint32_t)l_seq,
                                     h->n_diff, cig, 256);
                    sput(s, "%s,", cig);
                } else sput(s, "%dM,", l_seq);
            } else sput(s, "*,");
            sput(s, "%u;", (unsigned)h->n_diff);
        }
    }
}

/* MD / NM / XV (sam.c:246-328) */
static void sam_md_nm(so_str_t *s, const so_index_t *ix, const uint8_t *seq0, int l_seq, const so_result_t *q)
{
    int i, nm = 0, n_match = 0, n_rs = 0, rs[64];
    uint32_t ref_pos = q->pos;
    const uint8_t *seq = seq0 + q->seq_start;
    const char *cig = q->cigar;
    (void)l_seq;
    sput(s, "\tMD:Z:");
    while (*cig) {
        char *endp;
        long n = strtol(cig, &endp, 10);
        char op = *endp;
        cig = endp;
        if (op == 'M') {
            for (i = 0; i < n; ++i) {
                uint32_t bt = pac_base(ix->pac, ref_pos);
                if (bt == *seq) ++n_match;
                else {
                    uint32_t meta = ref_nib(ix->ref, ref_pos);
                    if ((meta & (1u << *seq)) != 0 && n_rs < 64) rs[n_rs++] = (int)(seq - (seq0 + q->seq_start));
                    ++nm;
                    if (n_match != 0) sput(s, "%d", n_match);
                    n_match = 0;
                    sput(s, "%c", "ACGTN"[bt]);
                }
                ++ref_pos; ++seq;
            }
        } else if (op == 'I') { nm += (int)n; seq += n; }
        else if (op == 'D') {
            if (n_match != 0) sput(s, "%d", n_match);
            n_match = 0; nm += (int)n;
            sput(s, "^");
            for (i = 0; i < n; ++i) { sput(s, "%c", "ACGTN"[pac_base(ix->pac, ref_pos)]); ++ref_pos; }
        }
        if (*cig) ++cig;
    }
    if (n_match != 0) sput(s, "%d", n_match);
    sput(s, "\tNM:i:%u", (unsigned)nm);
    if (n_rs > 0) {
        sput(s, "\tXV:i:");
        for (i = 0; i < n_rs; ++i) sput(s, i ? ",%d" : "%d", rs[i]);
    }
}

int so_sam_se(const so_index_t *ix, const so_opt_t *o, const char *name, const uint8_t *seq, int l_seq,
              const char *qual, const so_result_t *q, char *buf, size_t cap)
{
    so_str_t s = { buf, 0, cap, 0 };
    int i, n_amb = 0;
    uint8_t *rseq;
    if (cap) buf[0] = 0;
    for (i = 0; i < l_seq; ++i) n_amb += seq[i] > 3;
    if (n_amb > 200) return 0;                                 /* record never built (alnse.c:1328) */
    if (q->pos == 0xFFFFFFFFu) {                               /* sam.c:105-125 */
        sput(&s, "%s\t%u\t*\t0\t0\t*\t*\t0\t0\t", name, 4u);
        for (i = 0; i < l_seq; ++i) sput(&s, "%c", "ACGTN"[seq[i]]);
        if (qual) sput(&s, "\t%s", qual); else sput(&s, "\t*");
        return s.ovf ? -1 : (int)s.l;
    }
    rseq = xcalloc((size_t)l_seq + 1, 1);
    revcomp(seq, l_seq, rseq);
    int rid = coor_rid(ix, q->pos);
    sput(&s, "%s\t%u\t%s\t%lu\t%u\t%s\t*\t0\t0\t", name, q->strand ? 16u : 0u, ix->anns[rid].name,
         (unsigned long)((int64_t)q->pos - ix->anns[rid].offset + 1), (unsigned)q->mapq, q->cigar);
    if (q->strand) {
        for (i = 0; i < l_seq; ++i) sput(&s, "%c", "ACGTN"[rseq[i]]);
        sput(&s, "\t");
        if (qual) for (i = l_seq - 1; i >= 0; --i) sput(&s, "%c", qual[i]);
        else sput(&s, "*");
    } else {
        for (i = 0; i < l_seq; ++i) sput(&s, "%c", "ACGTN"[seq[i]]);
        sput(&s, "\t");
        if (qual && qual[0]) sput(&s, "%s", qual); else sput(&s, "*");
    }
    sam_xa(&s, ix, o, seq, rseq, l_seq, q);
    if (o->print_nm_md) sam_md_nm(&s, ix, q->strand == 0 ? seq : rseq, l_seq, q);
    if (o->rg_id) sput(&s, "\tRG:Z:%s", o->rg_id);
    free(rseq);
    return s.ovf ? -1 : (int)s.l;
}
