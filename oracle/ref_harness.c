/* oracle/ref_harness.c -- TEST INFRASTRUCTURE (fixture generator), our own source.
 *
 * Links against the reference's LandauVishkin.c + editdistance.c where they lie (see
 * oracle/Makefile target _ref/lvharness) and prints known-answer vectors for
 *   ed_mismatch            (Align_src/editdistance.c:88)
 *   ed_diff                (Align_src/editdistance.c:174)  -> computeEditDistance (LandauVishkin.c:19)
 *   ed_diff_withcigar      (Align_src/editdistance.c:234)  -> computeEditDistanceWithCigar (:176)
 * on a seeded synthetic 4-bit "mixRef".  Output format (text, one record per line):
 *   R <l> <hex words...>                      the mixRef (l bases, 8 per u32, LSB-first nibbles)
 *   V <pos> <L> <kmis> <kdiff> <read codes 0-4 as digits> <mis> <diff> <cigar_ret> <cigar|->
 * where mis = ed_mismatch(ref,pos,seq,L,kmis), diff = ed_diff(ref,l,pos,L+4,seq,L,kdiff) and
 * cigar_ret/cigar = ed_diff_withcigar(ref,pos,L+4,seq,L,diff,buf,128,1,COMPACT) when diff >= 0.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "editdistance.h"

static uint64_t s_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void)
{
    s_state ^= s_state << 13; s_state ^= s_state >> 7; s_state ^= s_state << 17;
    return (uint32_t)(s_state >> 11);
}
static uint32_t rndn(uint32_t n) { return rnd() % n; }

#define REFLEN 6000

int main(int argc, char **argv)
{
    int n_cases = argc > 1 ? atoi(argv[1]) : 3000;
    uint32_t l = REFLEN;
    uint32_t nw = (l + 7) / 8;
    uint32_t *ref = calloc(nw + 4, 4);
    uint8_t *base = calloc(l, 1);
    uint32_t i;
    for (i = 0; i < l; ++i) {
        uint32_t c = rndn(4);
        uint32_t m = 1u << c;
        uint32_t r = rndn(100);
        if (r < 8) m |= 1u << rndn(4);           /* bi-allelic SNP site */
        else if (r < 10) m |= (1u << rndn(4)) | (1u << rndn(4));
        else if (r < 11) m = 0;                  /* reference N */
        base[i] = (uint8_t)c;
        ref[i >> 3] |= m << (4 * (i & 7));
    }
    printf("R %u", l);
    for (i = 0; i < nw; ++i) printf(" %08x", ref[i]);
    printf("\n");

    int t;
    for (t = 0; t < n_cases; ++t) {
        int L = (t % 7 == 0) ? 36 + (int)rndn(200) : 100;
        uint32_t pos;
        int mode = (int)rndn(10);
        if (mode == 0) pos = l - L - rndn(12);            /* at / over the end (LV window L+4) */
        else pos = rndn(l - L - 16);
        uint8_t *seq = calloc(L + 16, 1);
        /* derive the read from the reference with edits */
        int nsub = (int)rndn(100) < 50 ? (int)rndn(3) : (int)rndn(9);
        int nind = (int)rndn(100) < 55 ? 0 : 1 + (int)rndn(3);
        uint8_t tmp[600]; int n = 0; uint32_t p = pos;
        int shift = (int)rndn(100) < 15 ? (int)rndn(4) : 0; /* start a few bases in (leading D) */
        p += shift;
        while (n < L + 12 && p < l) {
            uint32_t m = (ref[p >> 3] >> (4 * (p & 7))) & 15;
            uint8_t c = base[p];
            if (m && rndn(2)) { /* pick any listed allele */
                int tries = 0; uint32_t a;
                do { a = rndn(4); } while (!((m >> a) & 1) && ++tries < 32);
                if ((m >> a) & 1) c = (uint8_t)a;
            }
            tmp[n++] = c; ++p;
        }
        while (n < L + 12) tmp[n++] = (uint8_t)rndn(4);
        int e;
        for (e = 0; e < nsub; ++e) { int q = (int)rndn(L); tmp[q] = (uint8_t)((tmp[q] + 1 + rndn(3)) & 3); }
        for (e = 0; e < nind; ++e) {
            int q = 2 + (int)rndn(L - 4);
            if (rndn(2)) { memmove(tmp + q, tmp + q + 1, n - q - 1); }
            else { memmove(tmp + q + 1, tmp + q, n - q - 1); tmp[q] = (uint8_t)rndn(4); }
        }
        memcpy(seq, tmp, L);
        if (rndn(100) < 6) seq[rndn(L)] = 4;           /* read N */
        int kmis = (int)rndn(4);
        int kdiff = (t % 5 == 0) ? (int)rndn(31) : L / 10;
        int mis = -9;
        if (pos + L <= l) mis = ed_mismatch(ref, pos, seq, L, kmis);
        int diff = ed_diff(ref, l, pos, L + 4, seq, L, kdiff);
        char cig[160]; memset(cig, 0, sizeof cig);
        int cret = -9;
        if (diff >= 0 && diff < 31)
            cret = ed_diff_withcigar(ref, pos, L + 4, seq, L, diff, cig, 128, 1, COMPACT_CIGAR_STRING);
        printf("V %u %d %d %d ", pos, L, kmis, kdiff);
        for (i = 0; i < (uint32_t)L; ++i) putchar('0' + seq[i]);
        printf(" %d %d %d %s\n", mis, diff, cret, cig[0] ? cig : "-");
        free(seq);
    }
    return 0;
}
