/* oracle/ref_harness_ssw.c -- TEST INFRASTRUCTURE (fixture generator), our own source, linked against the
 * reference's ssw.c where it lies (oracle/Makefile target _ref/sswharness).  Prints known answers of
 *   ssw_init(read, L, mat, n, 1) + ssw_align(prof, ref, refLen, 3, 1, 2, 0, 20, L/2)     (Align_src/ssw.c:741-856)
 * exactly as snpaln_sw_snpaware / snpaln_sw call it (alnpe.c:261-393), for seeded random windows.
 * Line format:  S <aware 0|1> <ref symbols as hex digits> <read codes as digits 0-4>
 *                 <score1> <score2> <ref_begin1> <ref_end1> <read_begin1> <read_end1> <cigar text|->
 * aware=1: ref symbols are 4-bit allele masks and the read is encoded 1<<code with the 16x16 matrix score_mat2;
 * aware=0: ref symbols are 0..3 and the read 0..4 with the 5x5 matrix score_mat (both matrices: alnpe.c:52-73). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "ssw.h"

static const int8_t score_mat[25] = { 1, -3, -3, -3, -1,  -3, 1, -3, -3, -1,  -3, -3, 1, -3, -1,  -3, -3, -3, 1, -1,  -1, -1, -1, -1, -1 };
static int8_t score_mat2[256 + 32];
static uint64_t st = 0x2545F4914F6CDD1Dull;
static uint32_t rnd(void) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 11); }

int main(int argc, char **argv)
{
    int n_cases = argc > 1 ? atoi(argv[1]) : 600, t, r, c;
    for (r = 0; r < 16; ++r) for (c = 0; c < 16; ++c) {
        int v = -3;
        if (r == 1 && (c & 1)) v = 1;
        if (r == 2 && (c & 2)) v = 1;
        if (r == 4 && (c & 4)) v = 1;
        if (r == 8 && (c & 8)) v = 1;
        score_mat2[r * 16 + c] = v;
    }
    for (c = 0; c < 32; ++c) score_mat2[256 + c] = -3;
    for (t = 0; t < n_cases; ++t) {
        int aware = t & 1, L = (t % 5 == 0) ? 40 + rnd() % 110 : 100, refLen = 120 + rnd() % 600, i;
        int8_t *ref = calloc(refLen + 8, 1), *read = calloc(L + 8, 1);
        uint8_t *base = calloc(refLen, 1);
        for (i = 0; i < refLen; ++i) {
            base[i] = rnd() & 3;
            if (aware) { int m = 1 << base[i]; if (rnd() % 100 < 6) m |= 1 << (rnd() & 3); if (rnd() % 200 == 0) m = 0; ref[i] = m; }
            else ref[i] = base[i];
        }
        int kind = rnd() % 10, p = rnd() % (refLen - L > 1 ? refLen - L : 1);
        uint8_t code[256];
        if (refLen < L + 2) p = 0;
        for (i = 0; i < L; ++i) code[i] = (p + i < refLen) ? base[p + i] : (rnd() & 3);
        if (kind == 0) for (i = 0; i < L; ++i) code[i] = rnd() & 3;                      /* junk read */
        else {
            int ne = rnd() % 6, e;
            for (e = 0; e < ne; ++e) code[rnd() % L] = rnd() & 3;
            if (kind < 5) {                                                                 /* an indel */
                int q = 10 + rnd() % (L - 20), k = 1 + rnd() % 3;
                if (rnd() & 1) memmove(code + q, code + q + k, L - q - k);
                else { memmove(code + q + k, code + q, L - q - k); for (e = 0; e < k; ++e) code[q + e] = rnd() & 3; }
            }
            if (kind == 5) for (i = 0; i < 12; ++i) code[i] = rnd() & 3;                   /* clipped head */
            if (kind == 6) for (i = L - 15; i < L; ++i) code[i] = rnd() & 3;               /* clipped tail */
            if (rnd() % 10 == 0) code[rnd() % L] = 4;                                      /* N */
        }
        for (i = 0; i < L; ++i) read[i] = aware ? (int8_t)(1 << code[i]) : (int8_t)code[i];
        s_profile *pr = ssw_init(read, L, aware ? score_mat2 : score_mat, aware ? 16 : 5, 1);
        s_align *a = ssw_align(pr, ref, refLen, 3, 1, 2, 0, 20, L / 2);
        printf("S %d ", aware);
        for (i = 0; i < refLen; ++i) printf("%x", ref[i] & 15);
        printf(" ");
        for (i = 0; i < L; ++i) printf("%d", code[i]);
        printf(" %d %d %d %d %d %d ", a->score1, a->score2, a->ref_begin1, a->ref_end1, a->read_begin1, a->read_end1);
        if (a->cigarLen == 0) printf("-");
        for (i = 0; i < a->cigarLen; ++i) printf("%u%c", a->cigar[i] >> 4, "MID"[a->cigar[i] & 15]);
        printf("\n");
        align_destroy(a); init_destroy(pr); free(ref); free(read); free(base);
    }
    return 0;
}
