/* include/salt_host.h -- C ABI of the host side that surrounds the GPU path: index-file loading
 * (the reference's on-disk formats are the surface) and SAM text formatting.
 *
 * Reference interfaces mirrored (paths under Align_src/):
 *   salt_index_load / _free      alnse_index_reload / alnse_index_destroy        indexio.c:23-60
 *   salt_index_host_view         the arrays of index_t                            indexio.h:26-33
 *   salt_sam_header              aln_samhead (without the dated @PG line)         sam.c:56-84
 *   salt_sam_se                  aln_samse + sam_add_xa + sam_add_md_nm           sam.c:87-328
 *   salt_sam_pe                  alnpe_sam                                        sam.c:331-457
 *   salt_lkt_build               LKT_build_lookuptable                            Index_src/LookUpTable.c:70-150
 */
#ifndef SALT_HOST_H
#define SALT_HOST_H
#include <stdint.h>
#include <stddef.h>
#include "salt_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct salt_index salt_index_t;

/* Loads <prefix>.R.seedLen, .C.bwt, .C.sa, .C.lkt, .C.pac, .C.ann, .C.amb, .R.backward.{bwt,occ,sa},
 * .ref.  If .C.lkt is absent and rebuild_lkt != 0 the table is rebuilt from .C.pac in memory.
 * Returns NULL on failure; salt_host_last_error() has the message (the reference would exit(1)). */
salt_index_t *salt_index_load(const char *prefix, int rebuild_lkt);
void          salt_index_free(salt_index_t *ix);
const salt_host_index_t *salt_index_host_view(const salt_index_t *ix);
int32_t       salt_index_seed_len(const salt_index_t *ix);
int32_t       salt_index_n_seqs(const salt_index_t *ix);
/* sequence i of <P>.C.ann (bntann1_t, bntseq.h): its offset in the concatenated genome, length and name -- what salt_gpu_index_set_contigs takes */
int           salt_index_seq(const salt_index_t *ix, int32_t i, int64_t *offset, int32_t *len, const char **name);
const uint8_t *salt_index_pac(const salt_index_t *ix, uint64_t *l_pac);   /* <P>.C.pac bytes (bntseq.c 2-bit packing) for salt_gpu_index_set_pac */
const char   *salt_host_last_error(void);

/* 12-mer table exactly as salt-idx writes it; out must hold 4^len + 1 entries. */
void salt_lkt_build(const uint8_t *pac, uint32_t l_ref, int len, uint32_t *out);

typedef struct {
    int32_t print_xa_cigar;     /* -c */
    int32_t print_nm_md;        /* -d */
    const char *rg_id;          /* -g, or NULL */
} salt_sam_opt_t;

/* Both return the number of bytes written (no trailing NUL counted), or -1 when cap is too small.
 * salt_sam_se writes one record without the newline; a skipped read (> 200 N) yields 0 bytes,
 * which the reference prints as an empty line (alnse.c:1328,1437). */
int salt_sam_header(const salt_index_t *ix, const salt_sam_opt_t *opt, char *buf, size_t cap);
int salt_sam_se(const salt_index_t *ix, const salt_sam_opt_t *opt, const char *name, const uint8_t *seq,
                int32_t l_seq, const char *qual, const salt_result_t *res, char *buf, size_t cap);

/* alnpe_sam (sam.c:331-457): both records of a pair (q[0], q[1]), each followed by "\n\n" as the reference's
 * driver prints them (alnpe.c:640-648).  Names are the mates' names with a trailing /1 /2 already removed. */
int salt_sam_pe(const salt_index_t *ix, const salt_sam_opt_t *opt, const salt_pe_opt_t *pe, const char *const name[2],
                const uint8_t *const seq[2], const int32_t l_seq[2], const char *const qual[2],
                const salt_result_t *q, char *buf, size_t cap);

/* Index builder (row N1): writes <prefix>.{R.seedLen,C.pac,C.ann,C.amb,C.lkt,C.bwt,C.sa,lp,
 * R.backward.bwt,R.backward.occ,R.backward.sa,ref} in salt-idx's formats from a FASTA (plain or .gz)
 * and salt's 4-column SNP file (chr, 1-based pos, alleles "A/G", ref; no header; grouped by
 * chromosome in FASTA order).  Mirrors index_main (Index_src/index1.c:46-185).  0 on success.
 *
 * The two suffix-array-bound steps go through a backend: NULL = SA-IS on the host (any machine; 64-bit indices beyond
 * 2^31 symbols), or the device builder of libsalt_gpu.so -- { device, salt_gpu_idx_build_c, salt_gpu_idx_build_r,
 * salt_gpu_idx_last_error } (include/salt_gpu.h) -- which is what indexes a GRCh38-sized genome in seconds.  Both give
 * the same bytes.  Nothing is chosen silently: the caller names the backend. */
typedef struct {
    int device;
    int (*build_c)(int device, const uint8_t *text, uint64_t n, uint32_t sa_intv, uint32_t *primary, uint32_t L2[5],
                   uint32_t *bwt, uint32_t *sa, uint32_t *lkt, uint32_t lkt_len);
    int (*build_r)(int device, const uint8_t *rtext, uint64_t n, const uint32_t *sharp_off, const uint32_t *sharp_hdr, uint64_t n_sharp,
                   uint32_t cum4, uint32_t *inv_sa0, uint32_t *code, uint64_t code_words, uint32_t *rsa);
    const char *(*last_error)(void);
} salt_idx_backend_t;
#define SALT_IDX_NO_LP 1            /* flags: do not write <prefix>.lp (the local patterns as text; `salt` never reads it) */
#define SALT_IDX_ALL_FILES 2        /* also write what the reference indexer writes and `salt` never reads: <prefix>.R.pac, .R.rpac, .R.ann, .R.amb,
                                       .R.forward.bwt / .occ / .sa (Index_src/index1.c:150-176): the directory then equals the reference's file for file */
int salt_idx_build(const char *fn_fa, const char *fn_snp, const char *prefix, int l_seed);
int salt_idx_build_ex(const char *fn_fa, const char *fn_snp, const char *prefix, int l_seed, const salt_idx_backend_t *backend, int flags);
/* The same from memory (what bench.py and the tests use for generated genomes): contigs as base letters (as a FASTA would hold
 * them), SNP groups in file order -- group i belongs to contig i, like the reference matches them (mixRef.c:149-152) --
 * with 0-based positions, allele masks (bit b = base b listed) and reference base codes. */
typedef struct { const char *name, *comment; const char *seq; uint64_t len; } salt_idx_contig_t;
typedef struct { const char *chr; const uint32_t *pos; const uint8_t *alleles; const uint8_t *ref; uint32_t n; } salt_idx_snps_t;
int salt_idx_build_mem(const salt_idx_contig_t *contigs, int n_contigs, const salt_idx_snps_t *snps, int n_groups, const char *prefix,
                       int l_seed, const salt_idx_backend_t *backend, int flags);
/* the host suffix sorter alone: sa_out[0..n] for text[0..n) of `bits`-bit symbols (2 or 3), empty suffix first */
int salt_idx_suffix_array_cpu(const uint8_t *text, uint64_t n, int bits, uint32_t *sa_out);
const char *salt_idx_last_error(void);

/* N3 -- insert-size window from the first batch of a paired-end run (`salt -p -b 0`; the reference prints "infer isize func haven't
 * been implemented" there and stops, Align_src/alnpe.c:586-589, so the definition is this build's; oracle/salt_oracle.c states it).
 * res: the 2 * n_pairs mates (pair i = rows 2i, 2i+1) aligned as SINGLE-END reads (salt_gpu_align_se).  A pair counts when both mates
 * map gap-free without alternative hits, on opposite strands of one sequence, forward mate first, template <= 100000; from the
 * sorted templates: quartiles, mean / sd inside [q1 - 2 iqr, q3 + 2 iqr] (integers, sd rounded up), window = mean -+ 4 sd widened
 * to [q1 - 3 iqr, q3 + 3 iqr].  Returns 0, or -1 when fewer than 25 pairs count (*n_used tells how many did). */
int salt_isize_infer(const salt_index_t *ix, uint32_t n_pairs, const uint32_t *offs, const salt_result_t *res,
                     uint32_t *min_tlen, uint32_t *max_tlen, uint32_t *n_used);

/* "<len><op>..." text of a binary CIGAR (ops as in salt_result_t); returns bytes written or -1 */
int salt_cigar_text(const uint16_t *ops, int n_ops, char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
