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
 * chromosome in FASTA order).  Mirrors index_main (Index_src/index1.c:46-185).  0 on success. */
int salt_idx_build(const char *fn_fa, const char *fn_snp, const char *prefix, int l_seed);
const char *salt_idx_last_error(void);

/* "<len><op>..." text of a binary CIGAR (ops as in salt_result_t); returns bytes written or -1 */
int salt_cigar_text(const uint16_t *ops, int n_ops, char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
