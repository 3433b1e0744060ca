/* oracle/salt_oracle_main.c -- CLI around the CPU restatement: `salt_oracle [opts] <idx> <reads.fq[.gz]>`.
 * TEST INFRASTRUCTURE (parity checker + CPU baseline), NOT PRODUCT.  Prints the SAM the reference's
 * `salt` prints for single-end input, minus the @PG header line (Align_src/alnse.c:1353-1480, aln.c:102-227).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <getopt.h>
#include <zlib.h>
#include <time.h>
#include "salt_oracle.h"

static unsigned char nt4(int c)
{
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
                 case 'T': case 't': return 3; default: return 4; }
}

static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }

typedef struct { char *name, *qual; uint8_t *seq; int l; } rec_t;

static char *gets_trim(gzFile fp, char *buf, int cap)
{
    if (!gzgets(fp, buf, cap)) return NULL;
    size_t n = strlen(buf);
    while (n && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) buf[--n] = 0;
    return buf;
}

int main(int argc, char **argv)
{
    int c, n_threads = 1, overlap = -1, quiet_sam = 0, pe = 0; unsigned min_tlen = 250, max_tlen = 550;   /* aln.c:43-44 */
    so_opt_t o; memset(&o, 0, sizeof o);
    uint32_t max_seed = 50, max_locate = 1000; int only_ref = 0, xa = 0, md = 0; const char *rg = NULL;
    while ((c = getopt(argc, argv, "t:n:hpa:b:g:em:s:l:cdr:vM:O:E:X:q")) >= 0) {
        switch (c) {
        case 't': n_threads = atoi(optarg); break;
        case 'g': rg = optarg; break;
        case 's': max_seed = (uint32_t)atoi(optarg); break;
        case 'm': max_locate = (uint32_t)atoi(optarg); break;
        case 'c': xa = 1; break;
        case 'd': md = 1; break;
        case 'v': only_ref = 1; break;
        case 'r': overlap = atoi(optarg); break;
        case 'q': quiet_sam = 1; break;   /* timing only */
        case 'p': pe = 1; break;
        case 'a': min_tlen = (unsigned)atoi(optarg); break;
        case 'b': max_tlen = (unsigned)atoi(optarg); break;
        default: break;                    /* -n -e -l -a -b -M -O -E -X parsed and ignored (aln.c:189-197) */
        }
    }
    if (optind + 2 + pe > argc) { fprintf(stderr, "usage: salt_oracle [opts] <idx> <reads.fq> [mates.fq]\n"); return 1; }
    so_index_t *ix = so_index_load(argv[optind]);
    if (!ix) return 1;
    so_opt_default(ix, &o);
    o.max_seed = max_seed; o.max_locate = max_locate; o.seed_only_ref = only_ref; o.print_xa_cigar = xa;
    o.print_nm_md = md; o.rg_id = rg;
    if (overlap > 0) o.l_overlap = overlap;
    if (pe) {                                                   /* alnpe_core (alnpe.c:530-661), one pair at a time */
        gzFile f1 = gzopen(argv[optind + 1], "r"), f2 = gzopen(argv[optind + 2], "r");
        if (!f1 || !f2) { fprintf(stderr, "[salt_oracle] cannot open read files\n"); return 1; }
        static char l1[1 << 16], l2[1 << 16];
        char *hdr = malloc(1 << 20), *sam = malloc(1 << 17);
        if (max_tlen == 0) {                                    /* N3: -b 0 = infer the window from the first batch (N_SEQS / 2 pairs, aln.h:27) */
            gzFile g1 = gzopen(argv[optind + 1], "r"), g2 = gzopen(argv[optind + 2], "r");
            const int MAXP = 50000; int np = 0; size_t cap = 1 << 24, at = 0;
            uint8_t *bs = malloc(cap); uint32_t *bo = malloc(4 * (2 * (size_t)MAXP + 1));
            bo[0] = 0;
            while (np < MAXP) {
                gzFile gg[2] = { g1, g2 }; int k, okp = 1;
                for (k = 0; k < 2 && okp; ++k) {
                    if (!gets_trim(gg[k], l1, 1 << 16) || !gets_trim(gg[k], l1, 1 << 16)) { okp = 0; break; }
                    const size_t L = strlen(l1);
                    if (at + L > cap) { cap = 2 * (at + L); bs = realloc(bs, cap); }
                    for (size_t i = 0; i < L; ++i) bs[at + i] = nt4(l1[i]);
                    at += L; bo[2 * np + k + 1] = (uint32_t)at;
                    if (!gets_trim(gg[k], l1, 1 << 16) || !gets_trim(gg[k], l1, 1 << 16)) { okp = 0; break; }
                }
                if (!okp) break;
                ++np;
            }
            gzclose(g1); gzclose(g2);
            int used = 0;
            if (np == 0 || so_infer_isize(ix, &o, np, bs, bo, n_threads, &min_tlen, &max_tlen, &used) != 0) {
                fprintf(stderr, "[alnpe_core]: cannot infer the insert size: %d usable pairs in the first batch (25 needed); give -a / -b\n", used);
                return 1;
            }
            fprintf(stderr, "[alnpe_core]: insert size window [%u, %u] inferred from %d pairs\n", min_tlen, max_tlen, used);
            free(bs); free(bo);
        }
        if (so_sam_header(ix, &o, hdr, 1 << 20) < 0) return 1;
        fputs(hdr, stdout);
        for (;;) {
            char *nm[2]; char *ql[2]; uint8_t *sq[2]; int ln[2]; gzFile ff[2] = { f1, f2 }; char *lb[2] = { l1, l2 }; int k, okp = 1;
            for (k = 0; k < 2 && okp; ++k) {
                if (!gets_trim(ff[k], lb[k], 1 << 16)) { okp = 0; break; }
                char *p = lb[k] + 1, *q = p; while (*q && !isspace((unsigned char)*q)) ++q; *q = 0;
                size_t n = strlen(p);
                if (n > 2 && p[n - 2] == '/' && isdigit((unsigned char)p[n - 1])) p[n - 2] = 0;
                nm[k] = strdup(p);
                if (!gets_trim(ff[k], lb[k], 1 << 16)) { okp = 0; break; }
                ln[k] = (int)strlen(lb[k]); sq[k] = malloc((size_t)ln[k] + 1);
                for (int i = 0; i < ln[k]; ++i) sq[k][i] = nt4(lb[k][i]);
                if (!gets_trim(ff[k], lb[k], 1 << 16) || !gets_trim(ff[k], lb[k], 1 << 16)) { okp = 0; break; }
                ql[k] = strdup(lb[k]);
            }
            if (!okp) break;
            so_result_t r2[2];
            so_align_pe1(ix, &o, min_tlen, max_tlen, sq[0], ln[0], sq[1], ln[1], r2);
            const char *cn[2] = { nm[0], nm[1] }, *cq[2] = { ql[0], ql[1] }; const uint8_t *cs[2] = { sq[0], sq[1] };
            if (so_sam_pe(ix, &o, min_tlen, max_tlen, cn, cs, ln, cq, r2, sam, 1 << 17) < 0) return 1;
            fputs(sam, stdout);
            for (k = 0; k < 2; ++k) { free(nm[k]); free(ql[k]); free(sq[k]); }
        }
        return 0;
    }
    gzFile fp = gzopen(argv[optind + 1], "r");
    if (!fp) { fprintf(stderr, "[salt_oracle] cannot open %s\n", argv[optind + 1]); return 1; }
    static char line[1 << 16];
    char *hdr = malloc(1 << 20);
    if (so_sam_header(ix, &o, hdr, 1 << 20) < 0) return 1;
    if (!quiet_sam) fputs(hdr, stdout);
    const int BATCH = 100000;                                   /* aln.h:27 */
    rec_t *recs = calloc(BATCH, sizeof *recs);
    so_result_t *res = calloc(BATCH, sizeof *res);
    char *sam = malloc(1 << 16);
    double t_aln = 0; long n_tot = 0;
    for (;;) {
        int n = 0; size_t tot = 0;
        while (n < BATCH && gets_trim(fp, line, sizeof line)) {
            if (line[0] != '@') continue;
            char *p = line + 1; char *q = p; while (*q && !isspace((unsigned char)*q)) ++q; *q = 0;
            size_t ln = strlen(p);
            if (ln > 2 && p[ln - 2] == '/' && isdigit((unsigned char)p[ln - 1])) p[ln - 2] = 0;   /* query.c:139-143 */
            recs[n].name = strdup(p);
            if (!gets_trim(fp, line, sizeof line)) break;
            recs[n].l = (int)strlen(line);
            recs[n].seq = malloc((size_t)recs[n].l + 1);
            for (int i = 0; i < recs[n].l; ++i) recs[n].seq[i] = nt4(line[i]);
            if (!gets_trim(fp, line, sizeof line)) break;
            if (!gets_trim(fp, line, sizeof line)) break;
            recs[n].qual = strdup(line);
            tot += (size_t)recs[n].l; ++n;
        }
        if (n == 0) break;
        uint8_t *seqs = malloc(tot + 1); uint32_t *offs = malloc(4 * ((size_t)n + 1));
        size_t at = 0;
        for (int i = 0; i < n; ++i) { offs[i] = (uint32_t)at; memcpy(seqs + at, recs[i].seq, (size_t)recs[i].l); at += (size_t)recs[i].l; }
        offs[n] = (uint32_t)at;
        double t0 = now();
        so_align_se_batch(ix, &o, n, seqs, offs, res, n_threads, NULL);
        t_aln += now() - t0;
        for (int i = 0; i < n; ++i) {
            if (!quiet_sam) {
                if (so_sam_se(ix, &o, recs[i].name, recs[i].seq, recs[i].l, recs[i].qual, res + i, sam, 1 << 16) < 0) return 1;
                puts(sam);
            }
            free(recs[i].name); free(recs[i].seq); free(recs[i].qual);
        }
        free(seqs); free(offs);
        n_tot += n;
    }
    fprintf(stderr, "[salt_oracle] %ld reads, align %.3f s, %d threads => %.1f Kreads/s\n", n_tot, t_aln, n_threads,
            t_aln > 0 ? n_tot / t_aln / 1e3 : 0.0);
    gzclose(fp);
    so_index_free(ix);
    return 0;
}
