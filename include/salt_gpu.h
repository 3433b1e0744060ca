/* include/salt_gpu.h -- C ABI of the MI355X (gfx950) implementation of salt's single-end per-read
 * alignment path.  Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * What it replaces in the reference (weiquan/salt, paths under Align_src/):
 *
 *   salt_gpu_index_attach      one-time upload after alnse_index_reload()        indexio.c:23-49
 *   salt_gpu_index_detach      alnse_index_destroy() for the device copy         indexio.c:50-60
 *   salt_gpu_align_se          the per-batch call that stands where the pthread  alnse.c:1419-1429
 *                              fan-out over alnse_core1() is: for every read      alnse.c:1316-1352
 *                              alnse_overlap_alt + query_gen_cigar                alnse.c:1045-1104
 *                                                                                 query.c:282-333
 *   salt_gpu_align_se_resident same work on buffers that already live in HBM (what bench.py times)
 *
 * The reference has no FFI of its own (pure C, one process); INTEGRATION.md shows the ~40-line
 * patch to alnse.c that binds these entry points in place of alnse_core1().
 *
 * Error model: the reference reports nothing upward (any failure is exit(1)); here every entry
 * point returns 0 on success or a negative SALT_E_* code, and salt_gpu_last_error() gives text.
 */
#ifndef SALT_GPU_H
#define SALT_GPU_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SALT_OK              0
#define SALT_E_INVAL        -1   /* bad argument / unsupported option value */
#define SALT_E_HIP          -2   /* a HIP runtime call failed */
#define SALT_E_NOMEM        -3
#define SALT_E_INDEX        -4   /* malformed index arrays */
#define SALT_E_CAPACITY     -5   /* batch larger than the workspace was created for */

#define SALT_MAX_READ_LEN   512  /* bases per read handled by the kernels */
#define SALT_MAX_HITS       5    /* aln.h:133 */
#define SALT_MAX_LOCATE     1024 /* located rows per strand kept in LDS (reference default -m 1000); -m up to 262144 is accepted, larger lists go to global memory */
#define SALT_MAX_SEED_SLOTS 512  /* seeds per strand, ceil((L-k+1)/overlap): any stride down to -r 1 on reads of SALT_MAX_READ_LEN bases */
#define SALT_MAX_CIGAR_OPS  64

/* Host-side view of index_t (indexio.h:26-33): the arrays exactly as the index files hold them. */
typedef struct {
    /* <P>.C.bwt / <P>.C.sa : bwt_t (bwt.h:40-55, bwtio.c:30-71) */
    uint32_t c_primary, c_L2[5], c_seq_len, c_bwt_size;
    const uint32_t *c_bwt;
    uint32_t c_sa_intv, c_n_sa;
    const uint32_t *c_sa;                       /* c_sa[0] == 0xFFFFFFFF */
    /* <P>.C.lkt : lookupTable_t (lookup.h:21-25) */
    uint32_t lkt_len, lkt_n;
    const uint32_t *lkt;
    /* <P>.R.backward.{bwt,occ,sa} : rbwt_t (rbwt.h:60-80) */
    uint32_t r_text_len, r_inv_sa0, r_cum[6], r_bwt_words;
    const uint32_t *r_bwt;
    uint32_t r_occ_words;   const uint32_t *r_occ;
    uint32_t r_major_words; const uint32_t *r_major;
    uint32_t r_n_sa;        const uint32_t *r_sa;
    /* <P>.ref : mixRef_t (metaref.h) */
    uint32_t ref_len;
    const uint32_t *ref;
    /* <P>.R.seedLen (aln.c:215-224); 0 = unknown.  Bounds the width of the device-side k-mer tables. */
    int32_t  l_seed;
} salt_host_index_t;

/* The fields of aln_opt_t (aln.h:63-89) the single-end path reads. */
typedef struct {
    int32_t  l_seed;         /* <P>.R.seedLen */
    int32_t  l_overlap;      /* -r (defaults to l_seed) */
    uint32_t max_seed;       /* -s, 50 */
    uint32_t max_locate;     /* -m, 1000 */
    int32_t  max_hits;       /* 5 */
    int32_t  seed_only_ref;  /* -v */
    int32_t  collect_counters; /* accumulate the logical-access counters below */
    int32_t  reserved;
} salt_aln_opt_t;

/* hit_t (query.h:28-33) */
typedef struct {
    uint32_t pos;
    uint8_t  n_diff, is_gap;
    uint16_t strand;
} salt_hit_t;

/* The result fields alnse_core1 leaves in query_t (query.h:37-63).  CIGARs are binary:
 * (len << 4) | op with op 0=M 1=I 2=D, in the order the text CIGAR lists them. */
typedef struct {
    uint32_t pos;                               /* 0xFFFFFFFF = unmapped */
    uint8_t  strand, n_diff, is_gap, mapq;      /* strand 3 / n_diff 255 / is_gap 255 when unset */
    int32_t  b0, b1;
    uint16_t seq_start, seq_end;
    uint8_t  n_hits[2];                         /* alternative hits kept per strand (<= 5 in all) */
    uint8_t  n_cigar;                           /* ops in cigar[] (0 when unmapped) */
    uint8_t  skipped;                           /* 1: > 200 N, read left untouched (alnse.c:1328) */
    salt_hit_t hits[2][SALT_MAX_HITS];
    uint8_t  hit_n_cigar[SALT_MAX_HITS];        /* ops of hit_cigar[h]; h counts strand-0 hits, then */
    uint8_t  pad[3];                            /* strand-1 hits; 0 for gap-free hits ("<L>M")        */
    uint16_t cigar[SALT_MAX_CIGAR_OPS];
    uint16_t hit_cigar[SALT_MAX_HITS][SALT_MAX_CIGAR_OPS];
} salt_result_t;                                /* 880 bytes */

/* access counters summed over a batch (only with collect_counters): SA / verify / LV / loci entries count what the align kernels do
 * per located row and candidate; LKT / OCC_C / OCC_R count k_seed's DEVICE accesses (W-mer gathers, Occ blocks fetched) */
enum { SALT_CTR_LKT, SALT_CTR_OCC_C, SALT_CTR_OCC_R, SALT_CTR_SA_C, SALT_CTR_SA_R, SALT_CTR_VERIFY,
       SALT_CTR_VERIFY_WORDS, SALT_CTR_LV, SALT_CTR_READS, SALT_CTR_BASES, SALT_CTR_LOCI,
       /* diagnostics: shader-clock cycles k_heavy waves spent per phase (only with collect_counters) */
       SALT_CTR_T_LOAD, SALT_CTR_T_GATHER, SALT_CTR_T_LOCATE, SALT_CTR_T_SORT, SALT_CTR_T_DEDUP, SALT_CTR_T_VERIFY,
       SALT_CTR_T_SCAN, SALT_CTR_T_GAP, SALT_CTR_T_TAIL, SALT_CTR_HEAVY_READS, SALT_CTR_X0, SALT_CTR_X1, SALT_CTR_X2, SALT_CTR_X3,
       SALT_CTR_LT_SEEDS, SALT_CTR_LT_LOCATE, SALT_CTR_LT_SORT, SALT_CTR_LT_VERIFY, SALT_CTR_LT_OUT, SALT_CTR_LT_SAMPLES,   /* k_light: s_memtime ticks of every 128th read */
       SALT_CTR_MAX_HEAVY, SALT_CTR_MAX_GAPFIN,   /* slowest read of k_heavy / k_gapfin: (s_memrealtime ticks << 32) | read index */
       /* DEVICE-layout accesses (what the kernels' own structures move; bench.py's roofline): k_seed -- 16-byte W-mer table gathers,
        * 32-byte C / 64-byte R Occ blocks fetched, suffix-array loads and 8-byte text loads of the one-row resolve; k_light / k_heavy --
        * 4-byte suffix-array / R-position loads, 16-byte verify lane-loads, result bytes stored */
       SALT_CTR_D_WLKT, SALT_CTR_D_COCC_SEED, SALT_CTR_D_ROCC_SEED, SALT_CTR_D_SA_SEED, SALT_CTR_D_TEXT_SEED,
       SALT_CTR_D_SA_LIGHT, SALT_CTR_D_VERIFY_LIGHT, SALT_CTR_D_OUT_LIGHT, SALT_CTR_D_SA_HEAVY, SALT_CTR_D_VERIFY_HEAVY, SALT_CTR_D_OUT_HEAVY,
       /* context table: 16-byte records read in place of 4-byte suffix-array loads, and the located rows it ruled out (windows never read) */
       SALT_CTR_D_CTX_ROWS, SALT_CTR_D_CTX_REJECTED,
       SALT_CTR_N };

typedef struct salt_gpu_index salt_gpu_index_t;
typedef struct salt_gpu_ws    salt_gpu_ws_t;

/* ---- device index ------------------------------------------------------------------------- */
/* Re-packs the host arrays into the device layout (DESIGN.md "HBM layout") on `device`. */
int  salt_gpu_index_attach(const salt_host_index_t *host, int device, salt_gpu_index_t **out);
void salt_gpu_index_detach(salt_gpu_index_t *ix);
/* The packed device image is one contiguous, position-independent allocation so that a multi-GPU
 * driver can broadcast it (RCCL) instead of re-packing on every rank. */
int  salt_gpu_index_image(const salt_gpu_index_t *ix, void **dev_ptr, uint64_t *bytes);
int  salt_gpu_index_attach_image(void *dev_ptr, uint64_t bytes, int device, salt_gpu_index_t **out);
/* The image ends with the W-mer table (16 B x 4^W: 64 GiB at W = 16, the default on a free MI355X), a pure function of what
 * precedes it.  The part before it is the COMPACT image -- what a multi-GPU driver needs to move: broadcast
 * [dev_ptr, dev_ptr + bytes) of salt_gpu_index_image_compact, then salt_gpu_index_attach_compact on each receiver
 * allocates the full image, copies the compact part in and tabulates the W-mer table there (it owns the result;
 * the broadcast buffer may be freed afterwards). */
int  salt_gpu_index_image_compact(const salt_gpu_index_t *ix, void **dev_ptr, uint64_t *bytes);
int  salt_gpu_index_attach_compact(const void *dev_ptr, uint64_t bytes, int device, salt_gpu_index_t **out);
/* Replicates the image of `src` (attached on devices[0]) onto devices[1..n): one RCCL broadcast of the compact part
 * over xGMI inside this process (ncclCommInitAll + grouped ncclBroadcast); each device then builds its W-mer table.
 * out[0] = src; out[i] owns its copy and is released with salt_gpu_index_detach. */
int  salt_gpu_index_replicate(salt_gpu_index_t *src, const int *devices, int n, salt_gpu_index_t **out);
/* device-to-device copy into a caller-owned buffer (e.g. a broadcast buffer): the full image when dst_bytes >= its
 * size, else its compact part (dst_bytes >= the compact size) */
int  salt_gpu_index_image_copy(const salt_gpu_index_t *ix, void *dst_dev_ptr, uint64_t dst_bytes);

/* ---- per-batch work ----------------------------------------------------------------------- */
int  salt_gpu_ws_create(salt_gpu_index_t *ix, uint32_t max_reads, uint64_t max_bases, salt_gpu_ws_t **out);
void salt_gpu_ws_destroy(salt_gpu_ws_t *ws);

/* Host buffers in, host results out; synchronous.  seqs: base codes 0..4 (A C G T N, query.c:177-183),
 * read i at seqs[offs[i] .. offs[i+1]).  Fills results[0..n_reads). */
int  salt_gpu_align_se(salt_gpu_ws_t *ws, const salt_aln_opt_t *opt, uint32_t n_reads,
                       const uint8_t *seqs, const uint32_t *offs, salt_result_t *results);

/* Paired end: the per-batch call that stands where alnpe_core1 runs (Align_src/alnpe.c:482-528, 596-606).
 * Mates are interleaved as query_read_multiPairedSeqs lays them out (query.c:252-268): pair i = reads 2i, 2i+1 of
 * seqs/offs; results[2i], results[2i+1].  Per mate: alnse_overlap (alnse.c:985-1044); per pair: pairing2 /
 * pairing_singleton with Smith-Waterman mate rescue (alnpe.c:94-480, ssw.c); CIGARs by query_gen_cigar or from the
 * rescue.  A rescued mate has seq_start / seq_end set (soft clips), b0 / b1 = SW scores.  `pac`: the 2-bit genome
 * (<P>.C.pac bytes, needed by the singleton rescue), l_pac bases. */
typedef struct { uint32_t min_tlen, max_tlen; } salt_pe_opt_t;      /* -a 250, -b 550 (aln.c:43-44) */
int  salt_gpu_index_set_pac(salt_gpu_index_t *ix, const uint8_t *pac, uint64_t l_pac);
int  salt_gpu_align_pe(salt_gpu_ws_t *ws, const salt_aln_opt_t *opt, const salt_pe_opt_t *pe, uint32_t n_pairs,
                       const uint8_t *seqs, const uint32_t *offs, salt_result_t *results);
/* Same on device-resident buffers (2 * n_pairs reads); only enqueues on `hip_stream`. */
int  salt_gpu_align_pe_resident(salt_gpu_ws_t *ws, const salt_aln_opt_t *opt, const salt_pe_opt_t *pe, uint32_t n_pairs,
                                uint32_t max_read_len, const void *d_seqs, const void *d_offs, void *d_results, void *hip_stream);
/* Mate rescues of the LAST paired batch that could not be finished as the reference would (their banded traceback needs
 * a band wider than the build's row buffers: possible only for reads longer than ~270 bases with a gap of that size).
 * salt_gpu_align_pe fails with SALT_E_CAPACITY when this is non-zero; callers of the resident entry check it themselves
 * (it synchronises the device). */
int  salt_gpu_ws_pe_overflow(salt_gpu_ws_t *ws, uint32_t *n);
/* Counters of the LAST paired batch (diagnostics): out[0] = Smith-Waterman rescue requests, out[2] = CIGAR items queued by
 * k_pe_final, out[4] = the overflow count above; the rest is internal. */
int  salt_gpu_ws_pe_counts(salt_gpu_ws_t *ws, uint32_t out[8]);

/* ---- FASTQ text in, SAM text out ------------------------------------------------------------------------------------
 * query_read_seq (query.c:146-239) + alnse_core1 + aln_samse / sam_add_xa / sam_add_md_nm (sam.c:87-328) for one block of whole,
 * strict 4-line FASTQ records: the block is parsed, aligned and formatted on the device; the host only hands over the text and
 * gets the SAM records of the block back (one line per read, input order, a skipped read = an empty line).  `fastq` should be
 * page-locked (salt_gpu_host_alloc) and must end with a newline; *sam points into a page-locked buffer the workspace owns and
 * stays valid until the next call on it.  Needs the contig table (bntann1_t offset + name per sequence, bntseq.h) for RNAME / POS.
 * Multi-line FASTQ records are not read here (SALT_E_INVAL names the record); callers fall back to their host parser. */
typedef struct { int32_t print_xa_cigar, print_nm_md; const char *rg_id; } salt_text_opt_t;      /* -c, -d, -g */
int  salt_gpu_index_set_contigs(salt_gpu_index_t *ix, int32_t n, const int64_t *offsets, const char *const *names);
/* Paired end (alnpe_core1 + alnpe_sam, sam.c:331-457): fastq1 / fastq2 hold the same number of whole 4-line records, mate 1 and mate 2 of
 * pair p at the same record index; the SAM block holds both records of every pair in order, each followed by the reference's empty
 * line (alnpe.c:640-648).  Needs salt_gpu_index_set_pac and the contig table. */
int  salt_gpu_align_pe_text(salt_gpu_ws_t *ws, const salt_aln_opt_t *opt, const salt_pe_opt_t *pe, const salt_text_opt_t *topt,
                            const char *fastq1, uint64_t n1_bytes, const char *fastq2, uint64_t n2_bytes,
                            const char **sam, uint64_t *sam_bytes, uint32_t *n_pairs);
/* optional: size the workspace's text buffers ahead of the first call (blocks of up to max_block_bytes with about est_reads reads of
 * up to max_read_len bases, about est_sam_bytes of SAM); what a later block needs beyond that is grown then.  host_sam (may be NULL):
 * page-locked memory of the caller's (salt_gpu_host_alloc, e.g. allocated while the index was loading) the SAM text is returned in as
 * long as it suffices; it stays the caller's to free, after the workspace is destroyed. */
int  salt_gpu_ws_reserve_text(salt_gpu_ws_t *ws, const salt_aln_opt_t *opt, uint64_t max_block_bytes, uint32_t est_reads, uint32_t max_read_len,
                              uint64_t est_sam_bytes, void *host_sam, uint64_t host_sam_bytes);
int  salt_gpu_align_se_text(salt_gpu_ws_t *ws, const salt_aln_opt_t *opt, const salt_text_opt_t *topt, const char *fastq, uint64_t n_bytes,
                            const char **sam, uint64_t *sam_bytes, uint32_t *n_reads);
int  salt_gpu_host_alloc(uint64_t bytes, void **ptr);      /* page-locked host memory for the text buffers */
void salt_gpu_host_free(void *ptr);
/* Multi-GPU drivers (`salt --gpus N`, which stands where alnse_core's pthread fan-out is, alnse.c:1419-1429): the host NUMA node of a device
 * (-1 = unknown), so that a device's worker threads and their page-locked buffers can be kept on the socket it is attached to. */
int  salt_gpu_device_numa_node(int device, int *node);
int  salt_gpu_device_count(int *n);

/* Same work on device-resident buffers; only enqueues on `hip_stream` (a hipStream_t, NULL = default). */
int  salt_gpu_align_se_resident(salt_gpu_ws_t *ws, const salt_aln_opt_t *opt, uint32_t n_reads,
                                uint32_t max_read_len, const void *d_seqs, const void *d_offs,
                                void *d_results, void *hip_stream);

/* Per-kernel device time.  After salt_gpu_ws_timing(ws, 1) every resident/host align call brackets its
 * kernels with HIP events on the stream it launches on; salt_gpu_ws_kernel_ms synchronises, adds the
 * elapsed times of all calls since the last read into ms[0] k_pack, ms[1] k_seed, ms[2] k_light, ms[3] k_heavy,
 * ms[4] k_gap, ms[5] k_gapfin, ms[6] k_cigar, and for paired-end calls ms[7] k_pair, ms[8] k_sw (the Smith-Waterman kernels k_swf, k_swf1, k_swr, k_swtb together), ms[9] k_pe_final (with the k_cigar pass
 * behind it); returns the number of calls in *n_calls and resets.  At most 256 calls are kept. */
#define SALT_N_KERNELS 10
int  salt_gpu_ws_timing(salt_gpu_ws_t *ws, int enable);
int  salt_gpu_ws_kernel_ms(salt_gpu_ws_t *ws, double ms[SALT_N_KERNELS], uint32_t *n_calls);

/* The reads of the LAST batch that k_light handed to k_heavy (diagnostics / per-kernel accounting):
 * *n = how many; ids[0..min(*n,cap)) = their indices in the batch, in queue order. */
int  salt_gpu_ws_heavy_reads(salt_gpu_ws_t *ws, uint32_t *ids, uint32_t cap, uint32_t *n);

/* Plain device buffers for callers without a HIP runtime of their own (resident inputs / results, image copies). */
int  salt_gpu_buffer_alloc(int device, uint64_t bytes, void **dev_ptr);
int  salt_gpu_buffer_free(int device, void *dev_ptr);
/* *equal = 1 iff the two device buffers hold the same bytes (bytes % 4 == 0); tests compare index images with it */
int  salt_gpu_buffer_equal(int device, const void *a, const void *b, uint64_t bytes, int *equal);

/* Unit entry of the candidate rule the kernels evaluate on UNSORTED lists: case i has candidates
 * pos/val[offs[i]..offs[i+1]) (val > vmax = no candidate) and the incoming bound bound_in[i].  mode 0: gap-free rule
 * (code_kmismatch, alnse.c:348-369; distances 0..3, range filter pos < ref_len) by rule_unsorted; 1: the same by
 * rule_sparse; 2: gapped rule (code_kdiff, alnse.c:371-393; distances 0..12, filter !(pos + L + 4 >= ref_len)).
 * out[i][18] = found, best_pos, best_dist, n_hits, a0, bound_out, then 6 x (hit_pos, hit_dist) in position order. */
int  salt_gpu_diag_rule(uint32_t n_cases, const uint32_t *pos, const uint8_t *val, const uint32_t *offs, const uint32_t *bound_in,
                        uint32_t L, uint32_t ref_len, int mode, uint32_t *out);

/* Unit entry of the candidate verifiers (tests): case i = read seqs[offs[i]..offs[i+1]) against candidates
 * cand[cand_offs[i]..cand_offs[i+1]) (<= 256) on the 4-bit mixRef `ref_words`; mode 0 = one candidate per lane, 1 / 2 = four / eight lanes
 * per candidate (reads <= 120 / 248 bases), 3 / 4 = the two-strand variants of 1 / 2.  out[j] = masked Hamming distance 0..3 of candidate
 * j or 255 (more, or a candidate at or beyond ref_len -- a locate that wrapped below 0, alnse.c:672-673,762 -- which must never be
 * dereferenced).  Mirrors ed_mismatch (editdistance.c:88-163) under code_kmismatch's cap (alnse.c:348-369). */
int  salt_gpu_diag_verify(const uint32_t *ref_words, uint32_t ref_len, uint32_t n_cases, const uint8_t *seqs, const uint32_t *offs,
                          const uint32_t *cand, const uint32_t *cand_offs, int mode, uint8_t *out);

/* Work-queue counters of the LAST batch (diagnostics): [0] reads k_light handed to k_heavy, [2] reads whose gapped pass
 * was deferred, [5] k_gap items (32 candidates each), [6] k_cigar items; the odd entries are the consumers' heads. */
int  salt_gpu_ws_queue_counts(salt_gpu_ws_t *ws, uint32_t out[8]);

/* counters of the last batch(es) since the previous call; resets them */
int  salt_gpu_ws_counters(salt_gpu_ws_t *ws, uint64_t out[SALT_CTR_N]);

/* Unit access (tests): for each case c -- a read seqs[offs[c]..offs[c+1]) against the 4-bit mixRef
 * `ref_words` at pos[c] -- out4[4c+0] = masked Hamming distance capped at 3 (255 = more; -2 = window
 * past the end), out4[4c+1] = LV distance with bound kdiff[c] from the diagonal-per-lane kernel, out4[4c+2]
 * = the same from the candidate-per-lane kernel (-2 when kdiff/L are outside its range), out4[4c+3] =
 * number of CIGAR ops written to cigars[64c..] (-2 when there is no alignment).  Mirrors ed_mismatch,
 * ed_diff, ed_diff_withcigar (editdistance.c:88,174,234). */
int  salt_gpu_diag_lv(const uint32_t *ref_words, uint32_t ref_len, uint32_t n_cases, const uint32_t *pos,
                      const uint32_t *kdiff, const uint8_t *seqs, const uint32_t *offs, int32_t *out4, uint16_t *cigars);

/* Unit entry of the Smith-Waterman mate-rescue kernel (ssw_init + ssw_align as snpaln_sw[_snpaware] call them,
 * alnpe.c:260-393; ssw.c): case i aligns codes[read_offs[i]..read_offs[i+1]) against reference symbols
 * ref_syms[ref_offs[i]..ref_offs[i+1]) -- 4-bit allele masks when aware[i], bases 0..3 otherwise.
 * out6[6i..]: score1, score2, ref_begin1, ref_end1, read_begin1, read_end1; cigars[i][SALT_MAX_CIGAR_OPS], n_cigar[i]. */
int  salt_gpu_diag_ssw(uint32_t n_cases, const uint8_t *aware, const uint8_t *ref_syms, const uint32_t *ref_offs,
                       const uint8_t *codes, const uint32_t *read_offs, int32_t *out6, uint16_t *cigars, uint16_t *n_cigar);

/* ---- polish (row N4): the re-scoring step of the reference's SAM post-processor (Polish_src/polish.c:461-497 scores, 190-249 CIGAR) ---
 * For every item = (read, strand, offset into the 2-bit genome): the plain edit distance between the read (strand 1: its reverse
 * complement) and the `tlen` bases at `offset`, by stock Landau-Vishkin with bound k (computeEditDistance, Polish_src/lv.c; -1 = more than
 * k), and with want_cigar the CIGAR of computeEditDistanceWithCigar (useM; binary ops as in salt_result_t).  `pool`: explicit windows
 * (pool_stride code bytes each) for the few items whose window the reference clips at the genome end and pads with what its buffer
 * held before (polish.c:84-92); item.pool = 0xFFFFFFFF for everything else.  salt_amd/host/polish_main.cc is the caller. */
typedef struct { uint32_t read, offset, pool; uint16_t tlen; uint8_t strand, k; } salt_polish_item_t;
typedef struct salt_gpu_polish salt_gpu_polish_t;
int  salt_gpu_polish_open(int device, const uint8_t *pac, uint64_t l_pac, salt_gpu_polish_t **out);
void salt_gpu_polish_close(salt_gpu_polish_t *p);
int  salt_gpu_polish_lv(salt_gpu_polish_t *p, const uint8_t *codes, const uint32_t *offs, uint32_t n_reads, const salt_polish_item_t *items,
                        uint32_t n_items, const uint8_t *pool, uint32_t pool_stride, uint32_t n_pool, int want_cigar,
                        int32_t *dist, uint16_t *cigars, uint8_t *n_cigar);

/* polish -s: the same items by Smith-Waterman (ssw_init + ssw_align over polish's own matrix: +2 / -2, N 0, gaps 3 / 1; polish.c:48-52,
 * 209-222, 509-510).  score[i] = score1; with want_cigar also read_span[2i], [2i+1] = read_begin1, read_end1 (the soft clips) and the
 * banded traceback's CIGAR.  item.tlen is the (possibly clipped) window length; item.pool and item.k are not used. */
int  salt_gpu_polish_sw(salt_gpu_polish_t *p, const uint8_t *codes, const uint32_t *offs, uint32_t n_reads, const salt_polish_item_t *items,
                        uint32_t n_items, int want_cigar, int32_t *score, int32_t *read_span, uint16_t *cigars, uint16_t *n_cigar);

/* ---- index construction on the device (salt-idx's heavy steps; row N1) ----------------------------------------------
 * Replaces, for texts of any length the 32-bit formats admit (n < 2^32 - 16): bwt_bwtgen / Rbwt_bwt_bwtgen (Index_src/bwt_gen.c,
 * 4bit_bwt_gen.c:1044-1130), bwt_bwtupdate_core (bwtmisc.c:121-143), bwt_cal_sa (bwt.c:48-68), LKT_build_lookuptable
 * (LookUpTable.c:70-150), Rbwt_gen_sa (rbwt.c:424-475).  Host buffers in and out; the suffix sorter (prefix doubling over a
 * radix sort) and every derived array run on `device`.  salt_amd/host/salt_idx.cc is the caller.
 *
 * salt_gpu_suffix_array: sa_out[0..n] = suffix array of text[0..n) with the empty suffix first (sa_out[0] = n); text holds one symbol
 *     per byte, values < 2^bits, bits = 2 or 3.
 * salt_gpu_idx_build_c: text = n base codes 0..3.  Writes the words of <P>.C.bwt behind its 5-word header (salt_gpu_idx_c_bwt_words(n)
 *     of them: 2-bit BWT with the interleaved Occ counts, bwt.h:57-64), primary and L2[0..4], the (n + sa_intv) / sa_intv suffix-array
 *     samples (sa[0] = 0xFFFFFFFF) and the 4^lkt_len + 1 items of the k-mer table.
 * salt_gpu_idx_build_r: rtext = n symbols 0..4 ('#' = 4) of the local-pattern text; sharp_off[j] = offset of the j-th '#', sharp_hdr[j]
 *     = header value of its record (localPattern.c:269-271, 304-307), cum4 = number of symbols < 4.  Writes inverseSa0, the
 *     code_words words of the 4-bit BWT (zero padded) and the n_sharp + 1 entries of saValueSharp (the last one is the caller's). */
int  salt_gpu_suffix_array(int device, const uint8_t *text, uint64_t n, int bits, uint32_t *sa_out);
uint64_t salt_gpu_idx_c_bwt_words(uint64_t n);
int  salt_gpu_idx_build_c(int device, const uint8_t *text, uint64_t n, uint32_t sa_intv, uint32_t *primary, uint32_t L2[5],
                          uint32_t *bwt, uint32_t *sa, uint32_t *lkt, uint32_t lkt_len);
int  salt_gpu_idx_build_r(int device, const uint8_t *rtext, uint64_t n, const uint32_t *sharp_off, const uint32_t *sharp_hdr, uint64_t n_sharp,
                          uint32_t cum4, uint32_t *inv_sa0, uint32_t *code, uint64_t code_words, uint32_t *rsa);
const char *salt_gpu_idx_last_error(void);

const char *salt_gpu_last_error(void);
uint32_t    salt_gpu_result_size(void);         /* sizeof(salt_result_t), for bindings */

#ifdef __cplusplus
}
#endif
#endif
