// salt_amd/csrc/salt_kernels.h -- host-visible declarations of the kernel launchers.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/salt_gpu.h"
#include "salt_device.h"

// Diagnostics that CHANGE results (phase timing by leaving kernels early) exist only in -DSALT_DIAG builds (`make DIAG=1`): the release
// library has neither the environment switches nor the code paths behind them (tests/test_release_build.py greps for the names).
#ifdef SALT_DIAG
#define SALT_DIAG_VAL(x) (x)
#else
#define SALT_DIAG_VAL(x) 0
#endif

namespace salt {

// Per-batch packed copies of the reads (k_pack), fixed stride per read so that no kernel waits for offs[]:
//   pm record (pm_stride words): one-hot nibble words (nt2bit, editdistance.c:40), 8 bases per word, base i in bits
//       4*(i%8)..+3 (the mixRef layout): words [0, nw8) forward strand, [nw8, 2*nw8) reverse complement, [2*nw8] = L
//   tb record (tb_stride words): 2-bit codes, 16 bases per word, base i in bits 30-2*(i%16) (first base highest, so a
//       k-mer is a funnel shift): [0, nw16) forward, [nw16, 2*nw16) reverse complement; then 'is N' bits, 32 bases
//       per word, base i in bit 31-(i%32): [2*nw16, +nw32) forward, [.., +nw32) reverse; then L
struct PackGeom {
    uint32_t nw8, nw16, nw32, pm_stride, tb_stride;
    __host__ __device__ static PackGeom make(uint32_t max_len)
    {
        if (max_len < 1) max_len = 1;
        PackGeom g; g.nw8 = (max_len + 7) / 8; g.nw16 = (max_len + 15) / 16; g.nw32 = (max_len + 31) / 32;
        g.pm_stride = (2 * g.nw8 + 1 + 3) & ~3u; g.tb_stride = (2 * g.nw16 + 2 * g.nw32 + 1 + 3) & ~3u;
        return g;
    }
};

struct SeedParams {
    uint32_t n_reads, spr;          // spr: seed slots per (read, strand) = ceil((Lmax-k+1)/overlap)
    int32_t  l_seed, l_overlap;
    uint32_t max_seed;
    int32_t  seed_only_ref;
    int32_t  resolve_unique;        // finish one-row C intervals against the genome text (k_seed)
    PackGeom pg;
};

struct AlignParams {
    uint32_t n_reads, spr;
    int32_t  l_seed;
    uint32_t max_locate;
    int32_t  max_hits;
    PackGeom pg;
    int32_t  dbg_stop;              // debug/A-B: k_light leaves after phase dbg_stop (1..4); results are then garbage
    int32_t  heavy_stop;            // diagnostics: k_heavy leaves a read after its locate (1), verify (2) or rule (3) passes; results are then garbage
    int32_t  all_heavy;             // debug/A-B: skip k_light, k_heavy walks reads 0..n_reads-1
    int32_t  pe;                    // 1: mates of a paired-end batch -- alnse_overlap semantics (alnse.c:985-1044, 501-629):
                                    //    per-interval locate cap, gapped bound stays 3, > 5 N skips the mate
    uint32_t max_amb;               // reads with more N than this are left untouched (200 SE / 5 PE)
};
static const uint32_t PE_LOCI_CAP = 0x40000;    // loci per strand a PE mate may enumerate = MAX_LOC_POS (alnse.c:42,533); global scratch

void launch_pack(const PackGeom &pg, uint32_t n_reads, const uint8_t *seqs, const uint32_t *offs, uint32_t *pm, uint32_t *tb, hipStream_t st);
// k_seed (W-mer gather, in-register resolves, walks queued) + k_seed_walk (the queued walks, one per lane).  wq: seed_wq_words(items) words,
// wq_cnt: seed_wq_cnt_words() words (zeroed inside); walk_blocks: 256-lane blocks of the walk kernel (what fills the device: CUs x 8)
void launch_seed(const IndexView &ix, const SeedParams &sp, const uint32_t *tb, uint4 *sai_c, uint4 *sai_r, uint4 *wq, uint32_t *wq_cnt,
                 uint32_t walk_blocks, unsigned long long *ctr, hipStream_t st);
size_t seed_wq_words(uint64_t items);
uint32_t seed_wq_cnt_words();
void launch_light(const IndexView &ix, const AlignParams &ap, const uint32_t *pm, const uint8_t *seqs, const uint32_t *offs, const uint4 *sai_c,
                  const uint4 *sai_r, salt_result_t *results, uint32_t *queue, uint32_t *qctl, uint32_t *qseg, uint32_t *qsub, unsigned long long *ctr, hipStream_t st);
size_t queue_words(uint32_t max_reads);                      // d_queue: the flat queue, k_heavy's overflow queue, k_light's segments
uint32_t queue_heads_offset();                              // k_heavy's queue heads inside the same array: word offset of head 0 (zeroed with the counters)
uint32_t queue_sub_words();                                 // the segments' counters
// Control word k of a workspace (qctl / GapBufs::gctl) lives at word QC(k): a cache line each.  Atomics on one line are served one after the
// other (~14 ns each) however many waves issue them, and k_heavy's gapped reads push through three of these words.
#define QC(k) ((k) * 32u)
static const uint32_t QCTL_WORDS = 16 * 32;
// Deferred gapped passes (k_heavy -> k_gap -> k_gapfin -> k_cigar), `cap` slots; gctl = the workspace's qctl
struct GapBufs {
    uint32_t *gq;        // [cap] read index of the slot (0xFFFFFFFF: not used after all)
    uint32_t *gn;        // [cap][2] located rows per strand
    uint32_t *goff;      // [cap][2] where they start in the pool
    uint32_t *gloci;     // [pool] located positions, unsorted, duplicates included
    uint8_t  *ge;        // [pool] their Landau-Vishkin distances (255 = none within the bound)
    uint32_t *gitems;    // k_gap items: (slot << 11) | (strand << 10) | chunk of 32 candidates
    uint32_t *cq;        // k_cigar items: (read << 3) | which (0 = the alignment, 1 + i = alternative hit i)
    uint32_t *gctl;
    uint32_t cap, pool, items_cap;
    uint32_t *ovq;       // reads k_heavy's small shape could not finish (launch_heavy); count gctl[9], head gctl[10]
    uint32_t *qheads;    // k_heavy's queue heads (launch_heavy): one per range of the queue, 256 bytes apart, zero at launch
};
GapBufs gap_bufs_layout(uint8_t *base, uint32_t cap, uint32_t *gctl, size_t *bytes);   // base = nullptr: size only
void launch_heavy(const IndexView &ix, const AlignParams &ap, const uint32_t *pm, const uint4 *sai_c,
                  const uint4 *sai_r, salt_result_t *results, const uint32_t *queue, unsigned long long *ctr,
                  uint32_t n_blocks, uint32_t gap_blocks, void *lvtab, const GapBufs &g, uint32_t *ovq, uint32_t *qheads, uint8_t *pe_scr, hipEvent_t *ev3, hipStream_t st);
size_t lv_table_bytes();                // per-block LV traceback table (global memory)

// ---- paired end (salt_pe.hip) ----
static const uint32_t SW_MAX_SEG = 64;              // stripes: reads up to 512 bases
static const uint32_t SW_BAND_W = 1100;             // ints per banded-SW row buffer (band width <= 548)
// Scratch of one Smith-Waterman launch, sized per call from the batch (sw_geom): k_sw's 8-lane groups keep the per-column maxima of the
// longest rescue window the insert-size options allow (2 B per column); k_swtb's groups three band rows and one direction byte per cell for
// alignments whose band outgrows its LDS (bands up to SW_BAND_W cells).
struct SwGeom {
    uint32_t maxcol_bytes, n_blocks;           // k_sw
    uint32_t tb_group_bytes, tb_blocks;        // k_swtb
};
SwGeom sw_geom(uint32_t max_len, uint64_t max_window, uint32_t cus);
void sw_geom_limit(SwGeom &g, uint32_t blocks);            // small batches: no more blocks than that (and their scratch)
uint64_t sw_scratch_bytes(const SwGeom &g);
struct PeSwReq { uint32_t start, end, mate; uint8_t strand, aware; uint16_t pad; };          // mate: index of the rescued mate (2p or 2p+1);
                                                                                            // aware: 0 plain, 1 SNP-aware, 2 polish matrix; pad bit 0: score only, bit 1: mate rescue (no CIGAR for spans under 20 bases)
struct PeSwRes { int32_t score1, score2, ref_begin, ref_end, read_begin, read_end; uint32_t start, strand; uint16_t n_cigar, ok; uint16_t cigar[SALT_MAX_CIGAR_OPS]; };
struct PePair { uint32_t req0; uint8_t n_req; uint8_t rescued[2]; uint8_t pad; };             // requests req0 .. req0+n_req-1, in the order tried
void launch_pair(uint32_t n_pairs, uint32_t min_tlen, uint32_t max_tlen, uint32_t l_pac, const uint32_t *offs, salt_result_t *res,
                 PePair *pairs, PeSwReq *req, uint32_t *pctl, hipStream_t st);
// k_swf (+ k_swf1: forward pass: scores, end point), k_swr (reverse pass: begin point), k_swtb (banded traceback -> CIGAR); heads[0..3]: the
// four kernels' queue heads, heads[4]: the number of requests k_swf leaves to k_swf1 (five words zeroed by the caller).
// overflow: counts rescues this build cannot finish as the reference would (window beyond the scratch, band beyond SW_BAND_W, more than
// SALT_MAX_CIGAR_OPS operations); the caller turns a non-zero count into an error instead of returning rows that differ from the reference's
void launch_sw(const IndexView &ix, const uint8_t *pac, const uint8_t *seqs, const uint32_t *offs, const PeSwReq *req, const uint32_t *pctl,
               PeSwRes *res, uint32_t *heads, uint32_t *overflow, uint8_t *scratch, SwGeom g, uint32_t max_len, hipStream_t st);
static const uint32_t SW_MAX_BLOCKS_PER_CU = 16;   // one-wave blocks per CU at most
static const uint64_t SW_SCRATCH_TOTAL = 2ull << 30;   // k_swtb's grid shrinks before its groups' global scratch passes 2 GiB
uint32_t sw_blocks_per_cu(uint32_t max_len);
void launch_pe_final(const IndexView &ix, const PackGeom &pg, uint32_t n_pairs, const uint32_t *pm, salt_result_t *res, const PePair *pairs,
                     const PeSwRes *sw, void *lvtab, uint32_t *citems, uint32_t *cctl, uint32_t n_blocks, hipStream_t st);   // cctl[0] count, cctl[1] head

uint32_t heavy_blocks_per_cu();
void launch_polish(const uint8_t *pac, const uint8_t *codes, const uint32_t *offs, const salt_polish_item_t *items, uint32_t n_items, const uint8_t *pool,
                   uint32_t pool_stride, int want_cigar, int32_t *dist, uint16_t *cigars, uint8_t *n_cigar, void *tabs, uint32_t n_blocks, hipStream_t st);
void launch_heads(const salt_result_t *res, uint32_t n, uint8_t *heads, hipStream_t st);   // heads[i] = first 128 bytes of res[i]
void launch_diag_rule(uint32_t n_cases, const uint32_t *pos, const uint8_t *val, const uint32_t *offs, const uint32_t *bound_in,
                      uint32_t L, uint32_t ref_len, int mode, uint32_t *out, hipStream_t st);
uint32_t diag_rule_words();
void launch_diag_verify(const uint32_t *ref, uint32_t ref_len, uint32_t n_cases, const uint8_t *seqs, const uint32_t *offs, const uint32_t *cand,
                        const uint32_t *coffs, int mode, uint8_t *out, hipStream_t st);
void launch_diag_lv(const IndexView &ix, uint32_t n, const uint32_t *pos, const uint32_t *kdiff, const uint8_t *seqs,
                    const uint32_t *offs, int32_t *out, uint16_t *cig, void *lvtab, hipStream_t st);

// ---- FASTQ text in, SAM text out (salt_text.hip) ----
struct FqRec { uint32_t name_off, name_len, seq_off, len, qual_off; };        // one 4-line record: offsets into the raw block
// per record, between k_sam_len and k_sam_write: the formatted head (flag ... the tab in front of SEQ) and tail (the tags) and their lengths
static const uint32_t SAM_HEAD_CAP = 96, SAM_TAIL_CAP = 224, SAM_SLOT = SAM_HEAD_CAP + SAM_TAIL_CAP;
struct SamSeg { uint16_t head_len, tail_len; uint8_t what, strand, over, pad; };      // what: 0 record, 1 empty line, 2 record without tail; over: head or tail outgrew the slot
struct SamDev {                                                               // what the SAM kernels read (by value)
    const uint8_t *raw; const FqRec *rec; const uint8_t *seqs; const uint32_t *offs; const salt_result_t *res;
    const int64_t *c_off; const uint32_t *c_name_off; const char *c_names; int32_t n_contigs;      // contigs (bntann1_t: offset, name)
    const uint32_t *text, *ref;                                                // 2-bit genome, mixRef
    int32_t xa_cigar, nm_md; const char *rg; int32_t rg_len;
    int32_t pe; uint32_t min_tlen, max_tlen;                                   // pe: records 2p, 2p + 1 are the mates of pair p (alnpe_sam)
    char *slot; SamSeg *seg;                                                   // [n] x SAM_SLOT bytes, [n]: written by k_sam_len, read by k_sam_write
    const uint32_t *tb; PackGeom pg;                                           // k_pack's 2-bit records of the reads (MD / NM / XV by words), or null
};
hipError_t text_warm();                                     // forces the load of the text kernels' code object
size_t text_scan_bytes(uint64_t max_items);
hipError_t launch_fq_count(const uint8_t *raw, uint64_t n, uint32_t *tile_cnt, void *tmp, size_t tmp_bytes, hipStream_t st);
hipError_t launch_fq_lines(const uint8_t *raw, uint64_t n, const uint32_t *tile_off, uint32_t *line_start, hipStream_t st);
hipError_t launch_fq_ctl_init(uint32_t *ctl, hipStream_t st);                       // ctl = { 0, 0, 0xFFFFFFFF, 0 }
hipError_t launch_fq_parse(const uint8_t *raw, const uint32_t *line_start, uint32_t n_rec, FqRec *rec, uint32_t *offs, uint32_t *ctl,
                           void *tmp, size_t tmp_bytes, hipStream_t st);
// one file of a pair: the records of the block at raw + base become rec[2 i + which] (offsets relative to raw), their lengths len[2 i + which];
// the caller initialises ctl ({ 0, 0, 0xFFFFFFFF, 0 }) and scans len afterwards (launch_text_scan)
hipError_t launch_fq_parse_mate(const uint8_t *raw, uint32_t base, const uint32_t *line_start, uint32_t n_rec, uint32_t which, FqRec *rec, uint32_t *len,
                                uint32_t *ctl, hipStream_t st);
hipError_t launch_text_scan(uint32_t *v, uint32_t n_plus_1, void *tmp, size_t tmp_bytes, hipStream_t st);       // exclusive scan in place
hipError_t launch_fq_codes(const uint8_t *raw, const FqRec *rec, const uint32_t *offs, uint32_t n_rec, uint8_t *seqs, hipStream_t st);
hipError_t launch_sam_len(const SamDev &d, uint32_t n, uint32_t *off, unsigned long long *total64, void *tmp, size_t tmp_bytes, hipStream_t st);   // *total64: all bytes, 64-bit
hipError_t launch_sam_write(const SamDev &d, uint32_t n, const uint32_t *off, char *out, hipStream_t st);
static const uint32_t FQ_TILE = 1024;                                          // bytes per newline-count tile (k_fq_count)

// attach-time re-packing + expansion kernels (salt_index.hip)
void launch_pack_c_occ(const uint32_t *bwt, uint64_t bwt_words, uint32_t seq_len, uint64_t n_blocks, COcc *out, uint32_t *err, hipStream_t st);
void launch_pack_r_occ(const uint32_t *code, uint64_t code_words, const uint32_t *minor, uint64_t minor_words, const uint32_t *major, uint64_t major_words,
                       uint32_t text_len, uint64_t n_blocks, ROcc *out, uint32_t *err, hipStream_t st);
void launch_build_c_ctx(const IndexView &ix, uint32_t ctx_k, uint4 *out, hipStream_t st);
void launch_build_r_ctx(const IndexView &ix, uint32_t ctx_k, uint4 *out, hipStream_t st);
void launch_build_c_sa(const IndexView &ix, const uint32_t *sa_sampled, uint32_t sa_intv, uint32_t *out, hipStream_t st);
void launch_build_r_pos(const IndexView &ix, const uint32_t *r_sa, uint32_t *out, hipStream_t st);
void launch_build_text(const IndexView &ix, uint32_t *out, hipStream_t st);
void launch_build_wlkt(const IndexView &ix, uint32_t len, uint4 *out, hipStream_t st);

} // namespace salt
