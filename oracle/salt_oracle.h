/* oracle/salt_oracle.h -- CPU restatement of salt's single-end per-read alignment path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.  The product (salt_amd/, include/) never links,
 * imports or executes anything in oracle/.
 *
 * Parity status: PINNED -- byte-identical SAM against the real reference (oracle/_ref/salt,
 * compiled from /root/reference by oracle/Makefile) on tests/golden/lambda/expect_se_*.sam and
 * against tests/golden/lv_vectors.txt (see tests/test_oracle_golden.py).
 *
 * Every function in salt_oracle.c cites the reference file:line it restates.
 */
#ifndef SALT_ORACLE_H
#define SALT_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct so_index so_index_t;

/* effective options of the live SE path (Align_src/aln.h:63-89, aln.c:28-56) */
typedef struct {
    int32_t  l_seed;          /* from <idx>.R.seedLen (aln.c:215-224) */
    int32_t  l_overlap;       /* -r, defaults to l_seed (aln.c:223) */
    uint32_t max_seed;        /* -s, 50 */
    uint32_t max_locate;      /* -m, 1000 */
    int32_t  max_hits;        /* fixed 5 (aln.h:133) */
    int32_t  seed_only_ref;   /* -v */
    int32_t  print_xa_cigar;  /* -c */
    int32_t  print_nm_md;     /* -d */
    const char *rg_id;        /* -g or NULL */
} so_opt_t;

#define SO_MAX_HITS 5
#define SO_CIGAR_MAX 128

typedef struct {
    uint32_t pos;
    uint8_t  n_diff, is_gap;
    uint16_t strand;
} so_hit_t;

/* result fields the reference leaves in query_t (Align_src/query.h:37-63) */
typedef struct {
    uint32_t pos;             /* 0xFFFFFFFF = unmapped */
    int32_t  strand;          /* 0 fwd, 1 rev, 3 unset */
    uint8_t  n_diff, is_gap, mapq, pad;
    int32_t  b0, b1;
    uint32_t seq_start, seq_end;
    int32_t  n_hits[2];
    so_hit_t hits[2][SO_MAX_HITS];
    char     cigar[SO_CIGAR_MAX];
} so_result_t;

/* per-read logical access counters */
typedef struct {
    uint64_t n_lkt, n_occC, n_occR, n_occR_syms, n_saC, n_saR, n_bwt2nt, n_verify,
             n_verify_words, n_lv, n_reads,
             n_occC_seed, n_occR_seed, n_occR_syms_seed,   /* the part of the three Occ counters spent in seeding */
             n_bases, n_hits_out;
} so_counters_t;

so_index_t *so_index_load(const char *prefix);      /* NULL on failure (message on stderr) */
void        so_index_free(so_index_t *);
int         so_index_seed_len(const so_index_t *);
void        so_opt_default(const so_index_t *, so_opt_t *);

/* seq: l_seq codes 0..4 (A C G T N).  Fills *res.  ctr may be NULL. */
void so_align_se1(const so_index_t *, const so_opt_t *, const uint8_t *seq, int l_seq,
                  so_result_t *res, so_counters_t *ctr);
/* batch: seqs concatenated, offs[i]..offs[i+1]; n_threads>=1 (static interleave like alnse_core1) */
void so_align_se_batch(const so_index_t *, const so_opt_t *, int n, const uint8_t *seqs,
                       const uint32_t *offs, so_result_t *res, int n_threads, so_counters_t *ctr);

/* SAM text (no trailing newline) exactly as aln_samse builds it (sam.c:87-182).
 * Returns bytes written (excl. NUL) or -1 if cap too small. */
int so_sam_se(const so_index_t *, const so_opt_t *, const char *name, const uint8_t *seq, int l_seq,
              const char *qual, const so_result_t *res, char *buf, size_t cap);
/* header without the @PG line (sam.c:56-84) */
int so_sam_header(const so_index_t *, const so_opt_t *, char *buf, size_t cap);

/* paired end: one pair through alnpe_core1 (alnpe.c:482-528); res[0], res[1] = the two mates.
 * so_sam_pe writes both records, each followed by the empty line the reference prints (sam.c:450, alnpe.c:620). */
void so_align_pe1(const so_index_t *, const so_opt_t *, uint32_t min_tlen, uint32_t max_tlen, const uint8_t *seq0, int l0,
                  const uint8_t *seq1, int l1, so_result_t res[2]);
int so_sam_pe(const so_index_t *, const so_opt_t *, uint32_t min_tlen, uint32_t max_tlen, const char *const name[2],
              const uint8_t *const seq[2], const int l_seq[2], const char *const qual[2], const so_result_t q[2], char *buf, size_t cap);

int so_ssw_unit(int aware, const uint8_t *ref_syms, int refLen, const uint8_t *codes, int L, int out6[6], char *cigar, int cap);

/* unit entry points for the golden vectors (editdistance.c:88,174,234) */
int so_ed_mismatch(const uint32_t *mixref, uint32_t pos, const uint8_t *seq, uint32_t L, int max_err);
int so_ed_diff(const uint32_t *mixref, uint32_t l_mref, uint32_t pos, uint32_t l_ref,
               const uint8_t *seq, uint32_t L, int k);
int so_ed_diff_cigar(const uint32_t *mixref, uint32_t pos, uint32_t l_ref, const uint8_t *seq,
                     uint32_t L, int k, char *cigar, int cap);

/* raw array access so tests can hand the same host arrays to the product's C-ABI */
typedef struct {
    uint32_t c_primary, c_L2[5], c_seq_len, c_bwt_size; const uint32_t *c_bwt;
    uint32_t c_sa_intv, c_n_sa; const uint32_t *c_sa;
    uint32_t lkt_len, lkt_n; const uint32_t *lkt;
    uint32_t r_text_len, r_inv_sa0, r_cum[6], r_bwt_words; const uint32_t *r_bwt;
    uint32_t r_occ_words; const uint32_t *r_occ; uint32_t r_major_words; const uint32_t *r_major;
    uint32_t r_n_sa; const uint32_t *r_sa;
    uint32_t ref_len; const uint32_t *ref;
    uint64_t l_pac; const uint8_t *pac;
} so_arrays_t;
void so_align_pe_batch(const so_index_t *, const so_opt_t *, uint32_t min_tlen, uint32_t max_tlen, int n_pairs, const uint8_t *seqs,
                       const uint32_t *offs, so_result_t *res, int n_threads);
void so_index_arrays(const so_index_t *, so_arrays_t *);

/* N3: insert-size window from the first batch (definition in salt_oracle.c; the reference prints "not implemented", alnpe.c:586-589) */
int so_isize_estimate(uint32_t *t, int n, uint32_t *min_tlen, uint32_t *max_tlen);      /* sorts t; -1 when n < 25 */
int so_isize_templates(const so_index_t *, int n_pairs, const uint32_t *offs, const so_result_t *res, uint32_t *t_out);
int so_infer_isize(const so_index_t *, const so_opt_t *, int n_pairs, const uint8_t *seqs, const uint32_t *offs, int n_threads,
                   uint32_t *min_tlen, uint32_t *max_tlen, int *n_used);

/* N4: the reference's SAM post-processor `polish [-s] [-p] <idx> <SAM>` (Polish_src/polish.c:448-762); records to `out` */
int so_polish(const so_index_t *, const char *sam_path, int use_sw, int paired, FILE *out);

#ifdef __cplusplus
}
#endif
#endif
