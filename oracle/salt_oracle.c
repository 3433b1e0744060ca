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
typedef struct { uint8_t *t, *p; int tlen, plen; int eq; } so_lvbuf_t;     /* eq: a base matches when the bytes are EQUAL (polish's stock LV) instead of mask & one-hot != 0 */

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
    b->tlen = (int)l_ref; b->plen = (int)L; b->eq = 0;
}

/* length of the matching run that starts at pattern offset `from` on diagonal d, capped at end */
static inline int lv_run(const so_lvbuf_t *b, int d, int from, int end)
{
    int i = from;
    if (i >= end) return end;
    if (b->eq) while (i < end && b->p[i] == b->t[d + i]) ++i;
    else while (i < end && (b->p[i] & b->t[d + i]) != 0) ++i;
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
            if (b->eq) {                                       /* the stock LV first tries e plain mismatches on diagonal 0 (Polish_src/lv.c:274-296; commented out in Align_src) */
                int straight = 0, q;
                for (q = 0; q < end0; ++q) if (b->p[q] != b->t[q]) ++straight;
                straight += b->plen - end0;
                if (straight == e) { if (!cig_put(&out, &cap, b->plen, 'M')) return -2; return e; }
            }
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

/* ========================================================================================== */
/* Paired end (Align_src/alnpe.c, ssw.c, sam.c:331-457)                                        */
/* ========================================================================================== */
#include <time.h>

/* L2: alnse_locate (alnse.c:501-629): per-interval cap, 0x40000 global cap, rand() subsampling of big R
 * intervals (srand(time(0)) as there: outputs are only defined when no R interval exceeds max_locate) */
static void locate_pe(const so_index_t *ix, const so_opt_t *o, uint32_t l_seq, so_aux_t *aux, so_counters_t *ctr)
{
    const uint32_t l_ref = ix->ref_len, MAX_LOC_POS = 0x40000u;
    int i; uint32_t j;
    aux->n_loci = 0;
#define PUSH_LOCUS(p) do { if (aux->n_loci == aux->cap_loci) { aux->cap_loci = aux->cap_loci ? aux->cap_loci * 2 : 1024; \
        aux->loci = realloc(aux->loci, 4 * (size_t)aux->cap_loci); } aux->loci[aux->n_loci++] = (p); } while (0)
    for (i = 0; i < aux->n_c; ++i) {
        const so_sai_t *c = aux->sai_c + i;
        for (j = c->sp; j <= c->ep && j - c->sp <= o->max_locate; ++j) {
            uint32_t pos = c_sa(ix, j, ctr) - c->offset;
            if (pos + l_seq > l_ref) continue;
            PUSH_LOCUS(pos);
            if (aux->n_loci == MAX_LOC_POS) goto done;
        }
    }
    srand((unsigned)time(0));
    for (i = 0; i < aux->n_r; ++i) {
        const so_sai_t *r = aux->sai_r + i;
        if (r->ep - r->sp > o->max_locate) {
            uint32_t range = (r->ep - r->sp) / o->max_locate, pick = (uint32_t)-1;
            for (j = r->sp; j <= r->ep; ++j) {
                uint32_t iter = j - r->sp;
                if (iter % range == 0) pick = (uint32_t)rand() % range;
                if (iter % range != pick) continue;
                uint32_t pos = r_back_sa(ix, j, ctr) - r->offset;
                if (pos > l_ref || pos + l_seq > l_ref) continue;
                PUSH_LOCUS(pos);
                if (aux->n_loci == MAX_LOC_POS) goto done;
            }
        } else {
            for (j = r->sp; j <= r->ep; ++j) {
                uint32_t pos = r_back_sa(ix, j, ctr) - r->offset;
                if (pos > l_ref || pos + l_seq > l_ref) continue;
                PUSH_LOCUS(pos);
                if (aux->n_loci == MAX_LOC_POS) goto done;
            }
        }
    }
done:
    qsort(aux->loci, aux->n_loci, 4, cmp_u32);
    if (aux->n_loci > aux->cap_hits) { aux->hits = realloc(aux->hits, sizeof(so_hit_t) * (size_t)aux->n_loci); aux->cap_hits = aux->n_loci; }
}

/* one mate: alnse_overlap (alnse.c:985-1044): gap-free bound 3, gapped bound stays 3 */
static void align_pe_mate(const so_index_t *ix, const so_opt_t *o, const uint8_t *seq, int l_seq, so_result_t *q,
                          so_aux_t *aux[2], uint8_t *rseq, so_counters_t *ctr)
{
    int n_amb = 0, i, n0, n1, max_diff;
    result_init(q, l_seq);
    for (i = 0; i < l_seq; ++i) n_amb += seq[i] > 3;
    if (n_amb > 5) return;                                     /* alnpe.c:481,495 */
    revcomp(seq, l_seq, rseq);
    int n_sai = l_seq - o->l_seed + 1;
    for (i = 0; i < 2; ++i) {
        aux_reserve(aux[i], n_sai > 1 ? n_sai : 1, 1024);
        aux[i]->n_c = aux[i]->n_r = 0; aux[i]->n_loci = 0; aux[i]->n_hits = 0;
    }
    seed_overlap(ix, o, seq, l_seq, aux[0], ctr);
    locate_pe(ix, o, (uint32_t)l_seq, aux[0], ctr);
    seed_overlap(ix, o, rseq, l_seq, aux[1], ctr);
    locate_pe(ix, o, (uint32_t)l_seq, aux[1], ctr);
    max_diff = 3;
    n0 = check_nogap(ix, q, seq, (uint32_t)l_seq, max_diff, 0, aux[0], ctr);
    if (n0 != NO_MATCH && n0 < max_diff) max_diff = n0;
    n1 = check_nogap(ix, q, rseq, (uint32_t)l_seq, max_diff, 1, aux[1], ctr);
    if (n1 != NO_MATCH && n1 < max_diff) max_diff = n1;
    if (n0 == NO_MATCH && n1 == NO_MATCH) {
        int d0 = check_withgap(ix, q, seq, (uint32_t)l_seq, max_diff, 0, aux[0], ctr);
        if (d0 != NO_MATCH && d0 < max_diff) max_diff = d0;
        (void)check_withgap(ix, q, rseq, (uint32_t)l_seq, max_diff, 1, aux[1], ctr);
    }
    set_hits(q, o->max_hits, aux);
}

/* ---- SSW 0.1.4 word kernel, emulated lane by lane (ssw.c:347-547) ---- */
typedef struct { int score, ref, read; } so_aend_t;
static inline int16_t sat_adds16(int a, int b) { int v = a + b; return (int16_t)(v > 32767 ? 32767 : v < -32768 ? -32768 : v); }
static inline int16_t sat_subu16(int16_t a, int b) { unsigned x = (uint16_t)a; return (int16_t)(uint16_t)(x > (unsigned)b ? x - (unsigned)b : 0); }
static inline int16_t max16(int16_t a, int16_t b) { return a > b ? a : b; }

static int16_t *ssw_profile(const int8_t *read, const int8_t *mat, int readLen, int n)        /* qP_word */
{
    int segLen = (readLen + 7) / 8, nt, i, seg;
    int16_t *t0 = xcalloc((size_t)n * segLen * 8, 2), *t = t0;
    for (nt = 0; nt < n; ++nt)
        for (i = 0; i < segLen; ++i) {
            int j = i;
            for (seg = 0; seg < 8; ++seg) { *t++ = j >= readLen ? 0 : mat[nt * n + read[j]]; j += segLen; }
        }
    return t0;
}

static void ssw_word(const int8_t *ref, int ref_dir, int refLen, int readLen, int go, int ge, const int16_t *prof,
                     uint16_t terminate, int maskLen, so_aend_t bests[2])
{
    int segLen = (readLen + 7) / 8, i, j, k, l, begin = 0, end = refLen, step = 1, edge;
    uint16_t max = 0; int end_read = readLen - 1, end_ref = 0;
    uint16_t *maxColumn = xcalloc((size_t)refLen, 2);
    int16_t (*Hs)[8] = xcalloc((size_t)segLen, 16), (*Hl)[8] = xcalloc((size_t)segLen, 16), (*E)[8] = xcalloc((size_t)segLen, 16),
            (*Hmax)[8] = xcalloc((size_t)segLen, 16);
    int16_t vMaxScore[8] = { 0 }, vMaxMark[8] = { 0 };
    if (ref_dir == 1) { begin = refLen - 1; end = -1; step = -1; }
    for (i = begin; i != end; i += step) {
        int16_t e[8], vF[8] = { 0 }, vH[8], vMaxColumn[8] = { 0 };
        int16_t (*pv)[8];
        const int16_t *vP = prof + (size_t)ref[i] * segLen * 8;
        vH[0] = 0; for (l = 1; l < 8; ++l) vH[l] = Hs[segLen - 1][l - 1];
        pv = Hl; Hl = Hs; Hs = pv;
        for (j = 0; j < segLen; ++j) {
            for (l = 0; l < 8; ++l) {
                int16_t h = sat_adds16(vH[l], vP[j * 8 + l]);
                e[l] = E[j][l];
                h = max16(h, e[l]); h = max16(h, vF[l]);
                vMaxColumn[l] = max16(vMaxColumn[l], h);
                Hs[j][l] = h;
                h = sat_subu16(h, go);
                e[l] = sat_subu16(e[l], ge); e[l] = max16(e[l], h); E[j][l] = e[l];
                vF[l] = sat_subu16(vF[l], ge); vF[l] = max16(vF[l], h);
                vH[l] = Hl[j][l];
            }
        }
        for (k = 0; k < 8; ++k) {                                   /* lazy F */
            int16_t t7[8];
            t7[0] = 0; for (l = 1; l < 8; ++l) t7[l] = vF[l - 1];
            memcpy(vF, t7, sizeof vF);
            for (j = 0; j < segLen; ++j) {
                int any = 0;
                for (l = 0; l < 8; ++l) {
                    int16_t h = max16(Hs[j][l], vF[l]);
                    Hs[j][l] = h;
                    h = sat_subu16(h, go);
                    vF[l] = sat_subu16(vF[l], ge);
                    if (vF[l] > h) any = 1;
                }
                if (!any) goto lazy_done;
            }
        }
lazy_done:
        {
            int diff = 0;
            for (l = 0; l < 8; ++l) { vMaxScore[l] = max16(vMaxScore[l], vMaxColumn[l]); if (vMaxMark[l] != vMaxScore[l]) diff = 1; }
            if (diff) {
                uint16_t temp = 0; int16_t m = vMaxScore[0];
                memcpy(vMaxMark, vMaxScore, sizeof vMaxMark);
                for (l = 1; l < 8; ++l) m = max16(m, vMaxScore[l]);
                temp = (uint16_t)m;
                if (temp > max) { max = temp; end_ref = i; memcpy(Hmax, Hs, (size_t)segLen * 16); }
            }
        }
        { int16_t m = vMaxColumn[0]; for (l = 1; l < 8; ++l) m = max16(m, vMaxColumn[l]); maxColumn[i] = (uint16_t)m; }
        if (maxColumn[i] == terminate) break;
    }
    for (i = 0; i < segLen * 8; ++i) {
        if ((uint16_t)Hmax[i / 8][i % 8] == max) { int temp = i / 8 + i % 8 * segLen; if (temp < end_read) end_read = temp; }
    }
    bests[0].score = max; bests[0].ref = end_ref; bests[0].read = end_read;
    bests[1].score = 0; bests[1].ref = 0; bests[1].read = 0;
    edge = (end_ref - maskLen) > 0 ? (end_ref - maskLen) : 0;
    for (i = 0; i < edge; ++i) if (maxColumn[i] > bests[1].score) { bests[1].score = maxColumn[i]; bests[1].ref = i; }
    edge = (end_ref + maskLen) > refLen ? refLen : (end_ref + maskLen);
    for (i = edge; i < refLen; ++i) if (maxColumn[i] > bests[1].score) { bests[1].score = maxColumn[i]; bests[1].ref = i; }
    free(maxColumn); free(Hs); free(Hl); free(E); free(Hmax);
}

/* banded_sw (ssw.c:549-727): returns ops (len<<4|op) in out, count as return value (0 on traceback error) */
static int ssw_banded(const int8_t *ref, const int8_t *read, int refLen, int readLen, int score, int go, int ge, int band_width,
                      const int8_t *mat, int n, uint32_t *out, int out_cap)
{
#define SET_U(u, w, i, j) { int x_ = (i) - (w); x_ = x_ > 0 ? x_ : 0; (u) = (j) - x_ + 1; }
#define SET_D(u, w, i, j, p) { int x_ = (i) - (w); x_ = x_ > 0 ? x_ : 0; x_ = (j) - x_; (u) = x_ * 3 + p; }
    int32_t i, j, e, f, temp1, temp2, l, max = 0, width, width_d;
    int32_t *h_b = NULL, *e_b = NULL, *h_c = NULL; int8_t *direction = NULL, *direction_line;
    uint32_t c[256];
    do {
        width = band_width * 2 + 3; width_d = band_width * 2 + 1;
        free(h_b); free(e_b); free(h_c); free(direction);
        h_b = xcalloc((size_t)width + 8, 4); e_b = xcalloc((size_t)width + 8, 4); h_c = xcalloc((size_t)width + 8, 4);
        direction = xcalloc((size_t)width_d * readLen * 3 + 64, 1);
        max = 0;                                                  /* the reference keeps max across retries: it only grows */
        for (j = 1; j < width - 1; ++j) h_b[j] = 0;
        for (i = 0; i < readLen; ++i) {
            int32_t beg = 0, end = refLen - 1, u = 0, edge;
            j = i - band_width; beg = beg > j ? beg : j;
            j = i + band_width; end = end < j ? end : j;
            edge = end + 1 < width - 1 ? end + 1 : width - 1;
            f = h_b[0] = e_b[0] = h_b[edge] = e_b[edge] = h_c[0] = 0;
            direction_line = direction + width_d * i * 3;
            for (j = beg; j <= end; ++j) {
                int32_t b, e1, f1, d, de, df, dh;
                SET_U(u, band_width, i, j); SET_U(e, band_width, i - 1, j);
                SET_U(b, band_width, i, j - 1); SET_U(d, band_width, i - 1, j - 1);
                SET_D(de, band_width, i, j, 0); SET_D(df, band_width, i, j, 1); SET_D(dh, band_width, i, j, 2);
                temp1 = i == 0 ? -go : h_b[e] - go;
                temp2 = i == 0 ? -ge : e_b[e] - ge;
                e_b[u] = temp1 > temp2 ? temp1 : temp2;
                direction_line[de] = temp1 > temp2 ? 3 : 2;
                temp1 = h_c[b] - go; temp2 = f - ge;
                f = temp1 > temp2 ? temp1 : temp2;
                direction_line[df] = temp1 > temp2 ? 5 : 4;
                e1 = e_b[u] > 0 ? e_b[u] : 0; f1 = f > 0 ? f : 0;
                temp1 = e1 > f1 ? e1 : f1;
                temp2 = h_b[d] + mat[ref[j] * n + read[i]];
                h_c[u] = temp1 > temp2 ? temp1 : temp2;
                if (h_c[u] > max) max = h_c[u];
                if (temp1 <= temp2) direction_line[dh] = 1;
                else direction_line[dh] = e1 > f1 ? direction_line[de] : direction_line[df];
            }
            for (j = 1; j <= u; ++j) h_b[j] = h_c[j];
        }
        band_width *= 2;
    } while (max < score);
    band_width /= 2;
    i = readLen - 1; j = refLen - 1; e = 0; l = 0; f = max = 0; temp2 = 2;
    direction_line = direction + width_d * (readLen - 1) * 3;
    while (i > 0) {
        SET_D(temp1, band_width, i, j, temp2);
        switch (direction_line[temp1]) {
        case 1: --i; --j; temp2 = 2; direction_line -= width_d * 3; f = 0; break;
        case 2: --i; temp2 = 0; direction_line -= width_d * 3; f = 1; break;
        case 3: --i; temp2 = 2; direction_line -= width_d * 3; f = 1; break;
        case 4: --j; temp2 = 1; f = 2; break;
        case 5: --j; temp2 = 2; f = 2; break;
        default: free(h_b); free(e_b); free(h_c); free(direction); return 0;
        }
        if (f == max) ++e;
        else { ++l; if (l < 256) c[l - 1] = (uint32_t)e << 4 | (uint32_t)max; max = f; e = 1; }
    }
    if (f == 0) { ++l; if (l <= 256) c[l - 1] = (uint32_t)(e + 1) << 4; }
    else { l += 2; if (l <= 256) { c[l - 2] = (uint32_t)e << 4 | (uint32_t)f; c[l - 1] = 16; } }
    if (l > out_cap) l = out_cap;
    for (i = 0; i < l; ++i) out[i] = c[l - 1 - i];
    free(h_b); free(e_b); free(h_c); free(direction);
    return l;
}

typedef struct { int score1, score2, ref_begin1, ref_end1, read_begin1, read_end1, n_cigar; uint32_t cigar[256]; } so_ssw_t;

/* ssw_align with flag = 2, filters = 0 (ssw.c:771-856) */
static void ssw_align2(const int8_t *read, int readLen, const int8_t *mat, int n, const int8_t *ref, int refLen, int go, int ge,
                       int maskLen, so_ssw_t *r)
{
    so_aend_t b[2];
    int16_t *prof = ssw_profile(read, mat, readLen, n);
    memset(r, 0, sizeof *r); r->ref_begin1 = -1; r->read_begin1 = -1;
    ssw_word(ref, 0, refLen, readLen, go, ge, prof, (uint16_t)-1, maskLen, b);
    free(prof);
    r->score1 = b[0].score; r->ref_end1 = b[0].ref; r->read_end1 = b[0].read;
    if (maskLen >= 15) r->score2 = b[1].score; else r->score2 = 0;
    {
        int rl = r->read_end1 + 1, i;
        int8_t *rev = xcalloc((size_t)rl + 1, 1);
        for (i = 0; i < rl; ++i) rev[i] = read[r->read_end1 - i];
        prof = ssw_profile(rev, mat, rl, n);
        ssw_word(ref, 1, r->ref_end1 + 1, rl, go, ge, prof, (uint16_t)r->score1, maskLen, b);
        free(prof); free(rev);
        r->ref_begin1 = b[0].ref; r->read_begin1 = r->read_end1 - b[0].read;
    }
    {
        int rfl = r->ref_end1 - r->ref_begin1 + 1, rdl = r->read_end1 - r->read_begin1 + 1;
        int bw = abs(rfl - rdl) + 1;
        r->n_cigar = ssw_banded(ref + r->ref_begin1, read + r->read_begin1, rfl, rdl, r->score1, go, ge, bw, mat, n, r->cigar, 256);
    }
}

static const int8_t SCORE_MAT[25] = { 1, -3, -3, -3, -1,  -3, 1, -3, -3, -1,  -3, -3, 1, -3, -1,  -3, -3, -3, 1, -1,  -1, -1, -1, -1, -1 };
static int8_t SCORE_MAT2[256 + 16];                           /* alnpe.c:58-73 (+16: the N read code indexes one row further) */
static void score_mat2_init(void)
{
    static int done = 0; int r, c;
    if (done) return;
    for (r = 0; r < 16; ++r) for (c = 0; c < 16; ++c) {
        int v = -3;
        if (r == 1 && (c & 1)) v = 1;
        if (r == 2 && (c & 2)) v = 1;
        if (r == 4 && (c & 4)) v = 1;
        if (r == 8 && (c & 8)) v = 1;
        SCORE_MAT2[r * 16 + c] = (int8_t)v;
    }
    for (c = 0; c < 16; ++c) SCORE_MAT2[256 + c] = -3;         /* past the array in the reference: undefined there */
    done = 1;
}

typedef struct { so_result_t r; const uint8_t *seq; uint8_t *rseq; int l_seq; } so_mate_t;

/* snpaln_sw_snpaware / snpaln_sw (alnpe.c:261-393); aware: 4-bit ref + score_mat2, else 2-bit pac + score_mat */
static int pe_sw(const so_index_t *ix, uint32_t start, uint32_t end, so_mate_t *m, int strand, int aware)
{
    const uint8_t *seq = strand ? m->rseq : m->seq;
    int l_seq = m->l_seq, i, l_ref = (int)(end - start + 1), ok = 0;
    so_ssw_t res;
    if ((int64_t)start >= ix->l_pac) { fprintf(stderr, "[salt_oracle] SW window starts past the genome\n"); return 0; }
    if (l_ref <= 0) return 0;
    int8_t *ref = xcalloc((size_t)l_ref + 1, 1), *read = xcalloc((size_t)l_seq + 1, 1);
    if (aware) {
        score_mat2_init();
        for (i = 0; i < l_ref; ++i) ref[i] = (int8_t)ref_nib(ix->ref, start + (uint32_t)i);
        for (i = 0; i < l_seq; ++i) read[i] = (int8_t)(1 << seq[i]);
        ssw_align2(read, l_seq, SCORE_MAT2, 16, ref, l_ref, 3, 1, l_seq / 2, &res);
    } else {
        for (i = 0; i < l_ref; ++i) ref[i] = (int8_t)pac_base(ix->pac, start + (uint32_t)i);
        for (i = 0; i < l_seq; ++i) read[i] = (int8_t)seq[i];
        ssw_align2(read, l_seq, SCORE_MAT, 5, ref, l_ref, 3, 1, l_seq / 2, &res);
    }
    if (res.score1 >= 0 && res.read_end1 - res.read_begin1 + 1 >= 20) {
        so_result_t *q = &m->r; char *o = q->cigar; int cap = SO_CIGAR_MAX, j;
        q->b0 = res.score1; q->b1 = res.score2; q->mapq = (uint8_t)gen_mapq((uint32_t)q->b0, (uint32_t)q->b1);
        q->pos = (uint32_t)res.ref_begin1 + start; q->strand = strand;
        q->seq_start = (uint32_t)res.read_begin1; q->seq_end = (uint32_t)res.read_end1;
        q->cigar[0] = 0;
        for (j = 0; j < res.n_cigar; ++j) {
            int w = snprintf(o, (size_t)cap, "%u%c", res.cigar[j] >> 4, "MID"[res.cigar[j] & 15]);
            if (w >= cap) break;
            o += w; cap -= w;
        }
        ok = 1;
    }
    free(ref); free(read);
    return ok;
}

static void pe_gen_cigar(const so_index_t *ix, so_mate_t *m) { gen_cigar(ix, &m->r, m->seq, m->rseq, m->l_seq); }

static int in_range(uint32_t a, uint32_t b, uint32_t small, uint32_t large)        /* CHECK_IN_RANGE: -1 low, 0 in, 1 high */
{
    uint32_t r = a < b ? b - a : a - b;
    if (a > b || r < small) return -1;
    return r > large ? 1 : 0;
}

/* pairing2 (alnpe.c:94-258) */
static void pairing2(const so_index_t *ix, so_mate_t *m0, so_mate_t *m1, uint32_t min_tlen, uint32_t max_tlen)
{
    so_result_t *q0 = &m0->r, *q1 = &m1->r;
    uint32_t l2 = (uint32_t)(m0->l_seq + m1->l_seq);
    uint32_t min_isize = min_tlen > l2 ? min_tlen - l2 : 0, max_isize = max_tlen > l2 ? max_tlen - l2 : 0;
    uint32_t min_erros = (uint32_t)-1, l_pac = (uint32_t)ix->l_pac, start, end;
    so_hit_t b0, b1; int pass;
    memset(&b0, 0, sizeof b0); memset(&b1, 0, sizeof b1);
    if (q0->strand == 0 && q1->strand == 1 && q0->pos < q1->pos) {
        if (in_range(q0->pos + (uint32_t)m0->l_seq, q1->pos, min_isize, max_isize) == 0) { pe_gen_cigar(ix, m0); pe_gen_cigar(ix, m1); return; }
    } else if (q1->strand == 0 && q0->strand == 1 && q1->pos < q0->pos) {
        if (in_range(q1->pos + (uint32_t)m1->l_seq, q0->pos, min_isize, max_isize) == 0) { pe_gen_cigar(ix, m0); pe_gen_cigar(ix, m1); return; }
    }
    for (pass = 0; pass < 2; ++pass) {                          /* alternative hits only; `j == jj;` is a no-op there */
        so_result_t *qf = pass == 0 ? q0 : q1, *qb = pass == 0 ? q1 : q0;
        uint32_t nf = (uint32_t)qf->n_hits[0], nb = (uint32_t)qb->n_hits[1], lf = (uint32_t)(pass == 0 ? m0->l_seq : m1->l_seq), i, jj;
        if (nf > SO_MAX_HITS) nf = SO_MAX_HITS;
        if (nb > SO_MAX_HITS) nb = SO_MAX_HITS;
        if (!(nf > 0 && nb > 0)) continue;
        for (i = 0; i < nf; ++i)
            for (jj = 0; jj < nb; ++jj) {
                int rg = in_range(qf->hits[0][i].pos + lf, qb->hits[1][jj].pos, min_isize, max_isize);
                if (rg == 0) {
                    uint32_t e = (uint32_t)qf->hits[0][i].n_diff + qb->hits[1][jj].n_diff;
                    if (e < min_erros) { min_erros = e; if (pass == 0) { b0 = qf->hits[0][i]; b1 = qb->hits[1][jj]; } else { b1 = qf->hits[0][i]; b0 = qb->hits[1][jj]; } }
                } else if (rg == 1) break;
            }
    }
    if (min_erros != (uint32_t)-1) {
        q0->pos = b0.pos; q0->strand = b0.strand; q0->n_diff = b0.n_diff; q0->is_gap = b0.is_gap;
        q1->pos = b1.pos; q1->strand = b1.strand; q1->n_diff = b1.n_diff; q1->is_gap = b1.is_gap;
        pe_gen_cigar(ix, m0); pe_gen_cigar(ix, m1);
        return;
    }
    /* mate rescue by SNP-aware SW, q0 then q1 as the anchor */
    if (q0->strand == 0) {
        start = q0->pos + min_isize + (uint32_t)m0->l_seq; end = q0->pos + max_isize + (uint32_t)m0->l_seq + (uint32_t)m1->l_seq; end = end >= l_pac ? l_pac : end;
        if (pe_sw(ix, start, end, m1, 1, 1)) { pe_gen_cigar(ix, m0); return; }
    } else {
        start = q0->pos > max_isize + (uint32_t)m1->l_seq ? q0->pos - max_isize - (uint32_t)m1->l_seq : 0;
        end = q0->pos > min_isize ? q0->pos - min_isize : 0; end = end >= l_pac ? l_pac : end;
        if (pe_sw(ix, start, end, m1, 0, 1)) { pe_gen_cigar(ix, m0); return; }
    }
    if (q1->strand == 0) {
        start = q1->pos + min_isize + (uint32_t)m1->l_seq; end = q1->pos + max_isize + (uint32_t)m1->l_seq + (uint32_t)m0->l_seq; end = end >= l_pac ? l_pac : end;
        if (pe_sw(ix, start, end, m0, 1, 1)) { pe_gen_cigar(ix, m1); return; }
    } else {
        start = q1->pos > max_isize + (uint32_t)m0->l_seq ? q1->pos - max_isize - (uint32_t)m0->l_seq : 0;
        end = q1->pos > min_isize ? q1->pos - min_isize : 0; end = end >= l_pac ? l_pac : end;
        if (pe_sw(ix, start, end, m0, 0, 1)) { pe_gen_cigar(ix, m1); return; }
    }
    if (q0->pos != 0xFFFFFFFFu) pe_gen_cigar(ix, m0);
    if (q1->pos != 0xFFFFFFFFu) pe_gen_cigar(ix, m1);
}

/* pairing_singleton (alnpe.c:395-480) */
static void pairing_singleton(const so_index_t *ix, so_mate_t *m0, so_mate_t *m1, uint32_t min_tlen, uint32_t max_tlen)
{
    so_result_t *q0 = &m0->r, *q1 = &m1->r;
    uint32_t l2 = (uint32_t)(m0->l_seq + m1->l_seq), lim = (uint32_t)ix->l_pac - 1, start, end;
    uint32_t min_isize = min_tlen > l2 ? min_tlen - l2 : 0, max_isize = max_tlen > l2 ? max_tlen - l2 : 0;
#define UMIN(a, b) ((a) < (b) ? (a) : (b))
    if (q0->pos != 0xFFFFFFFFu) {
        if (q0->strand == 0) {
            start = UMIN(q0->pos + min_isize + (uint32_t)m0->l_seq, lim); end = UMIN(q0->pos + max_isize + (uint32_t)m0->l_seq + (uint32_t)m1->l_seq, lim);
            if (pe_sw(ix, start, end, m1, 1, 0)) { pe_gen_cigar(ix, m0); return; }
        } else {
            start = q0->pos > max_isize + (uint32_t)m1->l_seq ? q0->pos - max_isize - (uint32_t)m1->l_seq : 0; start = UMIN(start, lim);
            end = q0->pos > min_isize ? q0->pos - min_isize : 0; end = UMIN(end, lim);
            if (pe_sw(ix, start, end, m1, 0, 0)) { pe_gen_cigar(ix, m0); return; }
        }
    }
    if (q1->pos != 0xFFFFFFFFu) {
        if (q1->strand == 0) {
            start = UMIN(q1->pos + min_isize + (uint32_t)m1->l_seq, lim); end = UMIN(q1->pos + max_isize + (uint32_t)m1->l_seq + (uint32_t)m0->l_seq, lim);
            if (pe_sw(ix, start, end, m0, 1, 0)) { pe_gen_cigar(ix, m1); return; }
        } else {
            start = q1->pos > max_isize + (uint32_t)m0->l_seq ? q1->pos - max_isize - (uint32_t)m0->l_seq : 0; start = UMIN(start, lim);
            end = q1->pos > min_isize ? q1->pos - min_isize : 0; end = UMIN(end, lim);
            if (pe_sw(ix, start, end, m0, 0, 0)) { pe_gen_cigar(ix, m1); return; }
        }
    }
    if (q0->pos != 0xFFFFFFFFu) pe_gen_cigar(ix, m0);
    if (q1->pos != 0xFFFFFFFFu) pe_gen_cigar(ix, m1);
}

/* alnpe_core1 for one pair (alnpe.c:482-528) */
void so_align_pe1(const so_index_t *ix, const so_opt_t *o, uint32_t min_tlen, uint32_t max_tlen, const uint8_t *seq0, int l0,
                  const uint8_t *seq1, int l1, so_result_t res[2])
{
    so_aux_t a0, a1, *aux[2] = { &a0, &a1 };
    so_mate_t m[2];
    int k;
    memset(&a0, 0, sizeof a0); memset(&a1, 0, sizeof a1);
    m[0].seq = seq0; m[0].l_seq = l0; m[1].seq = seq1; m[1].l_seq = l1;
    for (k = 0; k < 2; ++k) {
        m[k].rseq = xcalloc((size_t)m[k].l_seq + 1, 1);
        revcomp(m[k].seq, m[k].l_seq, m[k].rseq);
        align_pe_mate(ix, o, m[k].seq, m[k].l_seq, &m[k].r, aux, m[k].rseq, NULL);
    }
    if (m[0].r.pos != 0xFFFFFFFFu && m[1].r.pos != 0xFFFFFFFFu) pairing2(ix, &m[0], &m[1], min_tlen, max_tlen);
    else if (m[0].r.pos != 0xFFFFFFFFu || m[1].r.pos != 0xFFFFFFFFu) pairing_singleton(ix, &m[0], &m[1], min_tlen, max_tlen);
    res[0] = m[0].r; res[1] = m[1].r;
    free(m[0].rseq); free(m[1].rseq); aux_free(&a0); aux_free(&a1);
}

/* pairs interleaved (pair i = reads 2i, 2i+1); static interleave over threads like alnpe_core1's workers (alnpe.c:490) */
typedef struct { const so_index_t *ix; const so_opt_t *o; uint32_t lo, hi; int n, tid, nt; const uint8_t *seqs; const uint32_t *offs; so_result_t *res; } so_pejob_t;
static void *pe_worker(void *p)
{
    so_pejob_t *j = p; int i;
    for (i = j->tid; i < j->n; i += j->nt) {
        const uint32_t *f = j->offs + 2 * i;
        so_align_pe1(j->ix, j->o, j->lo, j->hi, j->seqs + f[0], (int)(f[1] - f[0]), j->seqs + f[1], (int)(f[2] - f[1]), j->res + 2 * i);
    }
    return NULL;
}
void so_align_pe_batch(const so_index_t *ix, const so_opt_t *o, uint32_t min_tlen, uint32_t max_tlen, int n_pairs, const uint8_t *seqs,
                       const uint32_t *offs, so_result_t *res, int n_threads)
{
    int t;
    if (n_threads < 1) n_threads = 1;
    so_pejob_t *jobs = xcalloc((size_t)n_threads, sizeof *jobs);
    pthread_t *th = xcalloc((size_t)n_threads, sizeof *th);
    score_mat2_init();                                        /* once, before the workers read it */
    for (t = 0; t < n_threads; ++t) {
        so_pejob_t jb = { ix, o, min_tlen, max_tlen, n_pairs, t, n_threads, seqs, offs, res };
        jobs[t] = jb;
        if (n_threads == 1) pe_worker(jobs + t); else pthread_create(th + t, NULL, pe_worker, jobs + t);
    }
    for (t = 0; t < n_threads && n_threads > 1; ++t) pthread_join(th[t], NULL);
    free(jobs); free(th);
}

/* alnpe_sam (sam.c:331-457): both records of a pair, each followed by '\n' (the driver's printf adds another) */
int so_sam_pe(const so_index_t *ix, const so_opt_t *o, uint32_t min_tlen, uint32_t max_tlen, const char *const name[2],
              const uint8_t *const seq[2], const int l_seq[2], const char *const qual[2], const so_result_t q[2], char *buf, size_t cap)
{
    so_str_t s = { buf, 0, cap, 0 };
    int rid[2] = { -1, -1 }, is_map[2] = { 0, 0 }, i, j, tlen = 0;
    uint32_t pos[2] = { 0, 0 };
    for (i = 0; i < 2; ++i) if (q[i].pos != 0xFFFFFFFFu) { is_map[i] = 1; rid[i] = coor_rid(ix, q[i].pos); pos[i] = q[i].pos - (uint32_t)ix->anns[rid[i]].offset + 1; }
    if (is_map[0] && is_map[1]) {
        if (rid[0] != rid[1]) tlen = 0;
        else if (pos[0] < pos[1]) tlen = (int)(pos[1] + q[1].seq_end - q[1].seq_start + 1 - pos[0]);
        else tlen = (int)(pos[0] + q[0].seq_end - q[1].seq_start + 1 - pos[1]);            /* sam.c:355-356 uses q[1].seq_start twice */
        if ((uint32_t)tlen > max_tlen || (uint32_t)tlen < min_tlen) tlen = 0;
    }
    for (i = 0; i < 2; ++i) {
        unsigned flag = 0x1;
        uint8_t *rseq = xcalloc((size_t)l_seq[i] + 1, 1);
        revcomp(seq[i], l_seq[i], rseq);
        sput(&s, "%s\t", name[i]);
        if (!is_map[i]) flag |= 0x4;
        if (!is_map[1 - i]) flag |= 0x8;
        if (q[i].strand == 1) flag |= 0x10;
        if (q[1 - i].strand == 1) flag |= 0x20;
        if (tlen != 0) flag |= 0x2;
        flag |= i == 0 ? 0x40 : 0x80;
        sput(&s, "%u\t", flag);
        if (is_map[i]) {
            sput(&s, "%s\t%lu\t%u\t", ix->anns[rid[i]].name, (unsigned long)pos[i], (unsigned)q[i].mapq);
            if (q[i].seq_start != 0) sput(&s, "%dS", (int)q[i].seq_start);
            sput(&s, "%s", q[i].cigar);
            if (q[i].seq_end != (uint32_t)l_seq[i] - 1) sput(&s, "%dS", l_seq[i] - (int)q[i].seq_end - 1);
            sput(&s, "\t");
        } else if (is_map[1 - i]) sput(&s, "%s\t%lu\t255\t*\t", ix->anns[rid[1 - i]].name, (unsigned long)pos[1 - i]);
        else sput(&s, "*\t0\t255\t*\t");
        if (is_map[1 - i]) {
            if (rid[i] == rid[1 - i] || is_map[i] != 1) sput(&s, "=\t"); else sput(&s, "%s\t", ix->anns[rid[1 - i]].name);
            sput(&s, "%lu\t", (unsigned long)pos[1 - i]);
        } else sput(&s, "*\t0\t");
        if (tlen != 0) { if (q[i].pos >= q[1 - i].pos) sput(&s, "-%d\t", tlen); else sput(&s, "%d\t", tlen); }
        else sput(&s, "0\t");
        if (q[i].strand == 1) {
            for (j = 0; j < l_seq[i]; ++j) sput(&s, "%c", "ACGTN"[rseq[j]]);
            sput(&s, "\t");
            if (qual[i] && qual[i][0]) for (j = l_seq[i] - 1; j >= 0; --j) sput(&s, "%c", qual[i][j]); else sput(&s, "*");
        } else {
            for (j = 0; j < l_seq[i]; ++j) sput(&s, "%c", "ACGTN"[seq[i][j]]);
            sput(&s, "\t");
            if (qual[i] && qual[i][0]) sput(&s, "%s", qual[i]); else sput(&s, "*");
        }
        sam_xa(&s, ix, o, seq[i], rseq, l_seq[i], &q[i]);
        if (o->print_nm_md && q[i].pos != 0xFFFFFFFFu) sam_md_nm(&s, ix, q[i].strand == 0 ? seq[i] : rseq, l_seq[i], &q[i]);
        if (o->rg_id) sput(&s, "\tRG:Z:%s", o->rg_id);
        sput(&s, "\n\n");
        free(rseq);
    }
    return s.ovf ? -1 : (int)s.l;
}

/* unit entry for tests/golden/ssw_vectors.txt: ssw_init + ssw_align as snpaln_sw[_snpaware] call them */
int so_ssw_unit(int aware, const uint8_t *ref_syms, int refLen, const uint8_t *codes, int L, int out6[6], char *cigar, int cap)
{
    so_ssw_t res; int i, j; char *o = cigar;
    int8_t *ref = xcalloc((size_t)refLen + 1, 1), *read = xcalloc((size_t)L + 1, 1);
    score_mat2_init();
    for (i = 0; i < refLen; ++i) ref[i] = (int8_t)ref_syms[i];
    for (i = 0; i < L; ++i) read[i] = aware ? (int8_t)(1 << codes[i]) : (int8_t)codes[i];
    ssw_align2(read, L, aware ? SCORE_MAT2 : SCORE_MAT, aware ? 16 : 5, ref, refLen, 3, 1, L / 2, &res);
    out6[0] = res.score1; out6[1] = res.score2; out6[2] = res.ref_begin1; out6[3] = res.ref_end1; out6[4] = res.read_begin1; out6[5] = res.read_end1;
    cigar[0] = 0;
    for (j = 0; j < res.n_cigar; ++j) { int w = snprintf(o, (size_t)cap, "%u%c", res.cigar[j] >> 4, "MID"[res.cigar[j] & 15]); if (w >= cap) break; o += w; cap -= w; }
    free(ref); free(read);
    return res.n_cigar;
}

/* ---- N3: insert-size window inferred from the first batch -----------------------------------------------------------------
 * The reference has no such function: `-b 0` makes alnpe_core print "infer isize func haven't been implemented" and stop
 * (Align_src/alnpe.c:586-589).  DEFINITION (ours, deterministic, integer arithmetic only; restated independently in
 * salt_amd/host/salt_host.cc):
 *   pairs used   the mates of the first batch are aligned one by one as single-end reads (alnse_overlap_alt + query_set_hits); a pair
 *                counts when both mates are mapped gap-free, neither has an alternative hit, they lie on opposite strands of the same
 *                sequence with the forward mate first, and the template t = pos(reverse) + L(reverse) - pos(forward) is <= 100000
 *   estimate     t sorted; q1 = t[n/4], q3 = t[3n/4], iqr = q3 - q1; over the values inside [q1 - 2 iqr, q3 + 2 iqr]: mean (rounded) and
 *                sd (square root of the integer variance, rounded up); window = [mean - 4 sd, mean + 4 sd], widened to
 *                [q1 - 3 iqr, q3 + 3 iqr] where that reaches further; lower end at least 1.  Fewer than 25 usable pairs: no estimate. */
static int cmp_tlen(const void *a, const void *b) { uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b; return x < y ? -1 : x > y; }
int so_isize_estimate(uint32_t *t, int n, uint32_t *min_tlen, uint32_t *max_tlen)
{
    if (n < 25) return -1;
    qsort(t, (size_t)n, sizeof *t, cmp_tlen);
    const uint64_t q1 = t[n / 4], q3 = t[(size_t)3 * n / 4], iqr = q3 - q1;
    const uint64_t lo = q1 > 2 * iqr ? q1 - 2 * iqr : 0, hi = q3 + 2 * iqr;
    uint64_t m = 0, sum = 0; int i;
    for (i = 0; i < n; ++i) if (t[i] >= lo && t[i] <= hi) { ++m; sum += t[i]; }
    const uint64_t mean = (sum + m / 2) / m;
    uint64_t var = 0;
    for (i = 0; i < n; ++i) if (t[i] >= lo && t[i] <= hi) { const uint64_t d = t[i] > mean ? t[i] - mean : mean - t[i]; var += d * d; }
    var /= m;
    uint64_t sd = 0;
    while ((sd + 1) * (sd + 1) <= var) ++sd;
    if (sd * sd < var) ++sd;
    uint64_t a = mean > 4 * sd ? mean - 4 * sd : 1, b = mean + 4 * sd;
    const uint64_t a2 = q1 > 3 * iqr ? q1 - 3 * iqr : 1, b2 = q3 + 3 * iqr;
    if (a2 < a) a = a2;
    if (b2 > b) b = b2;
    if (a < 1) a = 1;
    *min_tlen = (uint32_t)a; *max_tlen = (uint32_t)b;
    return 0;
}
/* the templates of the usable pairs of a batch (mates interleaved: pair i = reads 2i, 2i+1, already aligned as single-end reads) */
int so_isize_templates(const so_index_t *ix, int n_pairs, const uint32_t *offs, const so_result_t *res, uint32_t *t_out)
{
    int n = 0, i;
    for (i = 0; i < n_pairs; ++i) {
        const so_result_t *a = res + 2 * i, *b = a + 1;
        if (a->pos == 0xFFFFFFFFu || b->pos == 0xFFFFFFFFu || a->is_gap != 0 || b->is_gap != 0) continue;
        if (a->n_hits[0] + a->n_hits[1] + b->n_hits[0] + b->n_hits[1] != 0) continue;
        if (a->strand == b->strand || a->strand > 1 || b->strand > 1) continue;
        const so_result_t *f = a->strand == 0 ? a : b, *r = a->strand == 0 ? b : a;
        const uint32_t lr = r == a ? offs[2 * i + 1] - offs[2 * i] : offs[2 * i + 2] - offs[2 * i + 1];
        if (r->pos < f->pos) continue;
        if (coor_rid(ix, f->pos) != coor_rid(ix, r->pos)) continue;
        const uint64_t t = (uint64_t)r->pos + lr - f->pos;
        if (t > 100000) continue;
        t_out[n++] = (uint32_t)t;
    }
    return n;
}
int so_infer_isize(const so_index_t *ix, const so_opt_t *o, int n_pairs, const uint8_t *seqs, const uint32_t *offs, int n_threads,
                   uint32_t *min_tlen, uint32_t *max_tlen, int *n_used)
{
    so_result_t *res = xcalloc((size_t)2 * n_pairs, sizeof *res);
    uint32_t *t = xcalloc((size_t)n_pairs + 1, sizeof *t);
    so_align_se_batch(ix, o, 2 * n_pairs, seqs, offs, res, n_threads, NULL);
    const int n = so_isize_templates(ix, n_pairs, offs, res, t);
    if (n_used) *n_used = n;
    const int rc = so_isize_estimate(t, n, min_tlen, max_tlen);
    free(res); free(t);
    return rc;
}


/* ------------------------------------------------------------------------------------------ */
/* N4: polish (Polish_src/polish.c:448-762, samParser.c, lv.c = stock SNAP Landau-Vishkin)      */
/* ------------------------------------------------------------------------------------------ */
/* `polish [-s] [-p] <idx> <SAM>`: every record's primary hit and XA hits are re-scored against the 2-bit genome -- plain edit
 * distance (k = 13) or Smith-Waterman (+2 / -2, gaps 3 / 1) --, the best becomes the record's alignment (MAPQ 60 when it is the
 * only scored hit, else 0), its CIGAR is generated, and a bare SAM record is printed.  Quirks kept: strtok field splitting (empty
 * fields vanish); only the FIRST optional field that contains "XA" is read and nothing behind it; the reference window length
 * shrinks for good once a hit is clipped at the genome end and the window buffer keeps the previous hit's bytes there
 * (polish.c:84-92, 461-466); an empty line ends the input (samParser.c:87-90); QUAL is followed by a tab in two of its four cases
 * (polish.c:236-247); isize = |pos0 - pos1|, negative for the mate on the reverse strand; the pair window is fixed at 350..650
 * (polish.c:148-149).  Reads with bases other than A C G T print "ACGT"[4] or memory past it in the reference: undefined, not
 * restated (the fixtures and tests leave such reads out). */
#define PL_UNMAPPED (-100000)
#define PL_MAX_DISTANCE 13
typedef struct { char *chrom; uint32_t pos, offset; int score; } pl_hit_t;
typedef struct { pl_hit_t *a; size_t n, m; } pl_hits_t;
typedef struct {
    char *buf, *name, *qual, *seq; int flag, l_seq, strand, primary, b0, b1;
    uint8_t *nst[2];                         /* [0] the read as sequenced, [1] its reverse complement (after the 0x10 swap) */
    pl_hits_t h[2];
    char cigar[1024];
} pl_sam_t;
static const int8_t PL_SCORE_MAT[25] = { 2, -2, -2, -2, 0,  -2, 2, -2, -2, 0,  -2, -2, 2, -2, 0,  -2, -2, -2, 2, 0,  0, 0, 0, 0, 0 };

static void pl_push(pl_hits_t *h, char *chrom, uint32_t pos)
{
    if (h->n == h->m) { h->m = h->m ? 2 * h->m : 32; h->a = realloc(h->a, h->m * sizeof *h->a); }
    h->a[h->n].chrom = chrom; h->a[h->n].pos = pos; h->a[h->n].offset = 0; h->a[h->n].score = 0; ++h->n;
}
static char *pl_readline(FILE *fp)
{
    size_t l = 0, m = 1024; int c;
    char *line = xcalloc(m, 1);
    while ((c = fgetc(fp)) != EOF && c != '\n') { if (l + 2 > m) { m *= 2; line = realloc(line, m); } line[l++] = (char)c; }
    line[l] = 0;
    return line;
}
static pl_sam_t *pl_read(FILE *fp)                            /* sam_readline (samParser.c:84-190) */
{
    char *line = pl_readline(fp), *save = NULL;
    if (line[0] == 0) { free(line); return NULL; }
    pl_sam_t *s = xcalloc(1, sizeof *s);
    s->buf = line;
    s->name = strtok_r(line, "\t", &save);
    s->flag = atoi(strtok_r(NULL, "\t", &save));
    char *chrom = strtok_r(NULL, "\t", &save);
    uint32_t pos = (uint32_t)strtoul(strtok_r(NULL, "\t", &save), NULL, 10);
    if ((s->flag & 4) == 0 && strcmp(chrom, "*") != 0) pl_push(&s->h[(s->flag & 0x10) ? 1 : 0], chrom, pos);
    strtok_r(NULL, "\t", &save);                              /* MAPQ */
    strtok_r(NULL, "\t", &save);                              /* CIGAR */
    strtok_r(NULL, "\t", &save); strtok_r(NULL, "\t", &save); strtok_r(NULL, "\t", &save);      /* MRNM MPOS ISIZE */
    s->seq = strtok_r(NULL, "\t", &save); s->l_seq = (int)strlen(s->seq);
    s->b0 = s->b1 = PL_UNMAPPED;
    uint8_t *f = xcalloc((size_t)(s->l_seq + 15) / 8 * 8 + 8, 1), *r = xcalloc((size_t)(s->l_seq + 15) / 8 * 8 + 8, 1);
    int i;
    for (i = 0; i < s->l_seq; ++i) { int c = s->seq[i]; f[i] = c == 'A' || c == 'a' ? 0 : c == 'C' || c == 'c' ? 1 : c == 'G' || c == 'g' ? 2 : c == 'T' || c == 't' ? 3 : c == '-' ? 5 : 4; }
    for (i = 0; i < s->l_seq; ++i) r[i] = (uint8_t)(3 - f[s->l_seq - i - 1]);
    if (s->flag & 0x10) { s->nst[0] = r; s->nst[1] = f; } else { s->nst[0] = f; s->nst[1] = r; }
    s->qual = strtok_r(NULL, "\t", &save);
    char *opt = strtok_r(NULL, "\t", &save);
    while (opt) {
        if (strstr(opt, "XA")) {                             /* "XA:Z:chr,+pos,cigar,nd;..." -- nothing behind this field is looked at */
            char *sv2 = NULL, *multi = strtok_r(opt, ":", &sv2);
            multi = strtok_r(NULL, ":", &sv2); multi = strtok_r(NULL, ":", &sv2);
            while (multi && *multi) {
                char *semi = strchr(multi, ';');
                if (semi) *semi = 0;
                size_t l_aln = strlen(multi);
                if (l_aln == 0) break;
                char *sv3 = NULL, *achrom = strtok_r(multi, ",", &sv3), *apos = strtok_r(NULL, ",", &sv3);
                if (apos[0] != '-') pl_push(&s->h[0], achrom, (uint32_t)strtoul(apos, NULL, 10));
                else pl_push(&s->h[1], achrom, (uint32_t)strtoul(apos + 1, NULL, 10));
                if (!semi) break;
                multi += l_aln + 1;
            }
            break;
        }
        opt = strtok_r(NULL, "\t", &save);
    }
    return s;
}
static void pl_free(pl_sam_t *s) { if (!s) return; free(s->h[0].a); free(s->h[1].a); free(s->nst[0]); free(s->nst[1]); free(s->buf); free(s); }
static int pl_cmp_hit(const void *a, const void *b) { uint32_t x = ((const pl_hit_t *)a)->offset, y = ((const pl_hit_t *)b)->offset; return x < y ? -1 : x > y; }
static int pl_tid(const so_index_t *ix, const char *chrom)
{
    int i;
    for (i = 0; i < ix->n_seqs; ++i) if (strcmp(ix->anns[i].name, chrom) == 0) return i;
    fprintf(stderr, "[polish] unknown sequence %s\n", chrom); exit(1);
}
/* __get_refseq (polish.c:84-92): l bases from the 2-bit genome, clipped at its end; bytes behind the clipped length keep what they held */
static int pl_refseq(uint8_t *buf, int l, const so_index_t *ix, uint32_t start)
{
    int i;
    if ((int64_t)start > ix->l_pac) { fprintf(stderr, "[Error]: Out of reference length!\n"); exit(1); }
    if ((int64_t)start + l > ix->l_pac) l = (int)(ix->l_pac - (int64_t)start);
    for (i = 0; i < l; ++i) buf[i] = (uint8_t)pac_base(ix->pac, start + (uint32_t)i);
    return l;
}
/* all hits of one record: offsets, sort, unique, scores (polish.c:461-497 / 690-713) */
static void pl_score_hits(const so_index_t *ix, pl_sam_t *s, int use_sw)
{
    int l_ref = s->l_seq, i; size_t j;
    uint8_t *ref = xcalloc((size_t)(s->l_seq + 15) / 8 * 8 + 64, 1);
    for (i = 0; i < 2; ++i) {
        pl_hits_t *h = &s->h[i];
        for (j = 0; j < h->n; ++j) h->a[j].offset = (uint32_t)ix->anns[pl_tid(ix, h->a[j].chrom)].offset + h->a[j].pos - 1;
        qsort(h->a, h->n, sizeof *h->a, pl_cmp_hit);          /* equal offsets are equal hits: the sort's stability does not matter */
        if (h->n) { size_t n = 1; for (j = 1; j < h->n; ++j) if (h->a[j].offset != h->a[n - 1].offset) h->a[n++] = h->a[j]; h->n = n; }
        for (j = 0; j < h->n; ++j) {
            l_ref = pl_refseq(ref, l_ref, ix, h->a[j].offset);
            if (use_sw) {
                so_aend_t b[2];
                int16_t *prof = ssw_profile((const int8_t *)s->nst[i], PL_SCORE_MAT, s->l_seq, 5);
                ssw_word((const int8_t *)ref, 0, l_ref, s->l_seq, 3, 1, prof, (uint16_t)-1, s->l_seq, b);
                free(prof);
                h->a[j].score = b[0].score;
            } else {
                so_lvbuf_t lb = { ref, s->nst[i], l_ref, s->l_seq, 1 };
                const int d = lv_distance(&lb, PL_MAX_DISTANCE);
                h->a[j].score = d == -1 ? PL_UNMAPPED : -d;
            }
        }
    }
    free(ref);
}
/* best / second best of one record (polish.c:718-737) */
static void pl_pick(pl_sam_t *s)
{
    int best0 = PL_UNMAPPED, best1 = PL_UNMAPPED, i; size_t j;
    s->strand = -1; s->primary = -1;
    for (i = 0; i < 2; ++i)
        for (j = 0; j < s->h[i].n; ++j) {
            const int sc = s->h[i].a[j].score;
            if (sc == PL_UNMAPPED) continue;
            if (sc > best1) { best1 = sc; if (best1 > best0) { int t = best0; best0 = best1; best1 = t; s->strand = i; s->primary = (int)j; } }
        }
    s->b0 = best0; s->b1 = best1;
}
/* gen_cigar (polish.c:190-249) */
static void pl_gen_cigar(const so_index_t *ix, pl_sam_t *s, int use_sw)
{
    uint8_t *ref = xcalloc((size_t)(s->l_seq + 15) / 8 * 8 + 64, 1);
    const int l_ref = pl_refseq(ref, s->l_seq, ix, s->h[s->strand].a[s->primary].offset);
    const int d = s->h[s->strand].a[s->primary].score;
    s->cigar[0] = 0;
    if (use_sw) {
        so_ssw_t r; int j; char *o = s->cigar;
        ssw_align2((const int8_t *)s->nst[s->strand], s->l_seq, PL_SCORE_MAT, 5, (const int8_t *)ref, l_ref, 3, 1, s->l_seq / 2, &r);
        if (r.score1 != d) { fprintf(stderr, "push cigar error!\n"); exit(1); }
        if (r.read_begin1 != 0) o += sprintf(o, "%dS", r.read_begin1);
        for (j = 0; j < r.n_cigar; ++j) o += sprintf(o, "%u%c", r.cigar[j] >> 4, "MID"[r.cigar[j] & 15]);
        if (r.read_end1 + 1 != s->l_seq) o += sprintf(o, "%dS", s->l_seq - r.read_end1 - 1);
    } else if (d == -PL_MAX_DISTANCE) strcpy(s->cigar, "*");
    else {
        so_lvbuf_t lb = { ref, s->nst[s->strand], l_ref, s->l_seq, 1 };
        if (lv_cigar(&lb, -d, s->cigar, (int)sizeof s->cigar) != -d) { fprintf(stderr, "push cigar error!\n"); exit(1); }
    }
    free(ref);
}
static void pl_seq_qual(FILE *out, const pl_sam_t *s)
{
    int i;
    const uint8_t *q = s->strand == 0 ? s->nst[0] : s->nst[1];
    for (i = 0; i < s->l_seq; ++i) fputc("ACGT"[q[i] & 3], out);        /* reads with other bases are undefined in the reference */
    fputc('\t', out);
    const int rev_in = (s->flag & 0x10) != 0;
    if ((rev_in && s->strand == 0) || (!rev_in && s->strand != 0)) for (i = s->l_seq - 1; i >= 0; --i) fputc(s->qual[i], out);
    else fprintf(out, "%s\t", s->qual);
    fputc('\n', out);
}
/* the pair window of polish.c:148-179: two pointers over the forward hits of one mate and the reverse hits of the other */
static unsigned pl_pairing(pl_hits_t *fw, pl_hits_t *bw)
{
    unsigned n = 0; size_t i = 0, j = 0;
    if (fw->n == 0 || bw->n == 0) return 0;
    while (i < fw->n && j < bw->n) {
        const uint32_t a = fw->a[i].offset, b = bw->a[j].offset, r = a > b ? a - b : b - a;
        if (a > b || r < 350) ++j;
        else if (r > 650) ++i;
        else { pl_hit_t t = fw->a[n]; fw->a[n] = fw->a[i]; fw->a[i] = t; t = bw->a[n]; bw->a[n] = bw->a[j]; bw->a[j] = t; ++i; ++j; ++n; }
    }
    return n;
}
static void pl_print_mate(FILE *out, const pl_sam_t *me, const pl_sam_t *mate, int first, int proper)
{
    const int map0 = me->strand != -1, map1 = mate->strand != -1;
    const uint32_t pos0 = map0 ? me->h[me->strand].a[me->primary].pos : 0, pos1 = map1 ? mate->h[mate->strand].a[mate->primary].pos : 0;
    const char *c0 = map0 ? me->h[me->strand].a[me->primary].chrom : NULL, *c1 = map1 ? mate->h[mate->strand].a[mate->primary].chrom : NULL;
    unsigned flag = 1;
    if (proper) flag |= 2;
    if (me->strand == 1) flag |= 0x10;
    if (mate->strand == 1) flag |= 0x20;
    flag |= first ? 0x40 : 0x80;
    if (!map0) flag |= 4;
    if (first) { if (!map1) flag |= 8; } else { if (!map1) flag |= 4; }              /* the second record sets UNMAPPED for either mate (polish.c:383-384) */
    fprintf(out, "%s\t%u\t", first ? me->name : mate->name, flag & 0xFF);          /* both records carry the FIRST mate's name; flag is an unsigned char there */
    if (!map0) fprintf(out, "*\t0\t"); else fprintf(out, "%s\t%u\t", c0, pos0);
    fprintf(out, me->b1 == PL_UNMAPPED && me->b0 != PL_UNMAPPED ? "60\t" : "0\t");
    if (map0) fprintf(out, "%s\t", me->cigar); else fprintf(out, "*\t");
    if (!map1) fprintf(out, "*\t0\t");
    else if (!map0 || strcmp(c0, c1) != 0) fprintf(out, "%s\t%u\t", c1, pos1);
    else fprintf(out, "=\t%u\t", pos1);
    if (map0 && map1) { const int a = (int)(pos0 < pos1 ? pos1 - pos0 : pos0 - pos1); fprintf(out, "%d\t", me->strand == 0 ? a : -a); }
    else fprintf(out, "0\t");
    pl_seq_qual(out, me);
}
int so_polish(const so_index_t *ix, const char *sam_path, int use_sw, int paired, FILE *out)
{
    FILE *fp = fopen(sam_path, "r");
    if (!fp) { fprintf(stderr, "[Error]: Can't open file %s\n", sam_path); return 1; }
    {   /* sam_skipHeader (samParser.c:43-55) */
        char b[1024];
        while (fgets(b, sizeof b, fp)) if (b[0] != '@') { fseek(fp, -(long)strlen(b), SEEK_CUR); break; }
    }
    if (!paired) {
        pl_sam_t *s;
        while ((s = pl_read(fp)) != NULL) {
            pl_score_hits(ix, s, use_sw);
            pl_pick(s);
            if (s->strand != -1) pl_gen_cigar(ix, s, use_sw);
            const int mapped = s->strand != -1;
            unsigned flag = 0x40 | (s->strand == 1 ? 0x10 : 0) | (mapped ? 0 : 4);
            fprintf(out, "%s\t%u\t", s->name, flag);
            if (!mapped) fprintf(out, "*\t0\t"); else fprintf(out, "%s\t%u\t", s->h[s->strand].a[s->primary].chrom, s->h[s->strand].a[s->primary].pos);
            fprintf(out, s->b1 == PL_UNMAPPED && s->b0 != PL_UNMAPPED ? "60\t" : "0\t");
            if (mapped) fprintf(out, "%s\t", s->cigar); else fprintf(out, "*\t");
            fprintf(out, "*\t0\t0\t");
            pl_seq_qual(out, s);
            pl_free(s);
        }
    } else {
        pl_sam_t *s0 = pl_read(fp), *s1 = pl_read(fp);
        while (s0 && s1) {
            pl_score_hits(ix, s0, use_sw); pl_score_hits(ix, s1, use_sw);
            const unsigned n0 = pl_pairing(&s0->h[0], &s1->h[1]), n1 = pl_pairing(&s1->h[0], &s0->h[1]);
            const int proper = n0 + n1 != 0;
            if (!proper) { pl_pick(s0); pl_pick(s1); }
            else {                                            /* best pair (polish.c:600-640) */
                int best0 = PL_UNMAPPED, best1 = PL_UNMAPPED; unsigned i;
                s0->strand = s1->strand = s0->primary = s1->primary = -1;
                for (i = 0; i < n0; ++i) {
                    const int sc = s0->h[0].a[i].score + s1->h[1].a[i].score;
                    if (sc == PL_UNMAPPED) continue;
                    if (sc > best1) { best1 = sc; if (best1 > best0) { int t = best0; best0 = best1; best1 = t; s0->strand = 0; s1->strand = 1; s0->primary = s1->primary = (int)i; } }
                }
                for (i = 0; i < n1; ++i) {
                    const int sc = s0->h[1].a[i].score + s1->h[0].a[i].score;
                    if (sc == PL_UNMAPPED) continue;
                    if (sc > best1) { best1 = sc; if (best1 > best0) { int t = best0; best0 = best1; best1 = t; s0->strand = 1; s1->strand = 0; s0->primary = s1->primary = (int)i; } }
                }
                s0->b0 = s1->b0 = best0; s0->b1 = s1->b1 = best1;
            }
            if (s0->strand != -1 && s0->primary != -1) pl_gen_cigar(ix, s0, use_sw);
            if (s1->strand != -1 && s1->primary != -1) pl_gen_cigar(ix, s1, use_sw);
            pl_print_mate(out, s0, s1, 1, proper);
            pl_print_mate(out, s1, s0, 0, proper);
            pl_free(s0); pl_free(s1);
            s0 = pl_read(fp); s1 = pl_read(fp);
        }
        pl_free(s0); pl_free(s1);
    }
    fclose(fp);
    return 0;
}
