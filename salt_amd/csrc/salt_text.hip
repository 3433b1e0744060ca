// salt_amd/csrc/salt_text.hip -- FASTQ text in, SAM text out, on the device (gfx950).
//
// The reference parses FASTQ with kseq (Align_src/query.c:146-239, kseq.h) and formats SAM with ksprintf one character at a time
// (sam.c:87-328) on the host; at the rate the align kernels run that is two orders of magnitude too slow (SURVEY 7 "host I/O at
// target rate", row N2).  Here both ends are kernels, and the host only moves bytes:
//   k_fq_count / k_fq_lines   newline scan of the raw block -> start offset of every line
//   k_fq_parse                one thread per 4-line record: name (up to the first white space, a trailing /1 /2 trimmed: query.c:139-143),
//                             sequence and quality spans (CR trimmed), checks; read lengths for the offsets scan
//   k_fq_codes                bases -> codes A0 C1 G2 T3 other 4 (nst_nt4_table), the aligner's input layout
//   k_sam_len / k_sam_write   aln_samse (sam.c:87-182) with sam_add_xa (186-240) and sam_add_md_nm (246-328): a thread per read formats the small fields, eight lanes per read copy name / SEQ / QUAL:
//                             the exact record length first, an exclusive scan for the offsets, then the bytes -- the batch's SAM text
//                             comes out contiguous and in input order.
// Strict 4-line FASTQ only (what sequencers write); salt's host path keeps reading multi-line records the way kseq does.
#include <hip/hip_runtime.h>
#include <string.h>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>
#include "salt_device.h"
#include "salt_kernels.h"

namespace salt {

static inline uint32_t tgrid(uint64_t n) { uint64_t b = (n + 255) / 256; if (b > (1u << 18)) b = 1u << 18; return b ? (uint32_t)b : 1u; }
#define TSTRIDE(i, n) for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, s_ = (uint64_t)gridDim.x * blockDim.x; i < (n); i += s_)

static constexpr uint32_t TILE = 1024;          // bytes per newline-count tile

__global__ void __launch_bounds__(256) k_fq_count(const uint8_t *__restrict__ raw, uint64_t n, uint32_t *__restrict__ tile_cnt)
{
    const uint64_t n_tiles = (n + TILE - 1) / TILE;
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t p = t * TILE + threadIdx.x * 4;
        uint32_t c = 0;
        for (uint32_t q = 0; q < 4; ++q) if (p + q < n && raw[p + q] == '\n') ++c;
        for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
        __shared__ uint32_t part[4];
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) tile_cnt[t] = part[0] + part[1] + part[2] + part[3];
        __syncthreads();
    }
}
// line_start[k + 1] = position behind newline k (line_start[0] = 0 is set by the caller); tile_off = exclusive scan of tile_cnt
__global__ void __launch_bounds__(256) k_fq_lines(const uint8_t *__restrict__ raw, uint64_t n, const uint32_t *__restrict__ tile_off, uint32_t *__restrict__ line_start)
{
    const uint64_t n_tiles = (n + TILE - 1) / TILE;
    __shared__ uint32_t wave_base[4];
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t p = t * TILE + threadIdx.x * 4;
        uint32_t m = 0;
        for (uint32_t q = 0; q < 4; ++q) if (p + q < n && raw[p + q] == '\n') m |= 1u << q;
        const uint32_t c = (uint32_t)__popc(m);
        uint32_t incl = c;                                       // inclusive scan inside the wave
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if ((threadIdx.x & 63) >= (uint32_t)o) incl += v; }
        if ((threadIdx.x & 63) == 63) wave_base[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t base = tile_off[t];
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) base += wave_base[w];
        uint32_t k = base + incl - c;
        for (uint32_t q = 0; q < 4; ++q) if (m & (1u << q)) line_start[++k] = (uint32_t)(p + q + 1);
        __syncthreads();
    }
}

__device__ __forceinline__ bool is_space(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }
__device__ __forceinline__ uint8_t nt4_dev(uint8_t c)
{
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

// err bits: 1 header does not start with '@', 2 third line does not start with '+', 4 sequence / quality lengths differ, 8 empty read
// raw: the block; base: its offset in the buffer the records' offsets refer to; the record goes to rec[stride * i + which]
__global__ void __launch_bounds__(256) k_fq_parse(const uint8_t *__restrict__ raw, const uint32_t *__restrict__ line_start, uint32_t n_rec,
                                                   FqRec *__restrict__ rec, uint32_t *__restrict__ len, uint32_t *__restrict__ ctl /* [0] err, [1] max len, [2] first bad record */,
                                                   uint32_t base, uint32_t stride, uint32_t which)
{
    TSTRIDE(i, n_rec) {
        const uint32_t l0 = line_start[4 * i], l1 = line_start[4 * i + 1], l2 = line_start[4 * i + 2], l3 = line_start[4 * i + 3], l4 = line_start[4 * i + 4];
        uint32_t err = 0;
        if (raw[l0] != '@') err |= 1;
        uint32_t ne = l0 + 1;
        const uint32_t he = l1 - 1;                              // the header's newline
        while (ne < he && !is_space(raw[ne])) ++ne;
        uint32_t name_len = ne - (l0 + 1);
        if (name_len > 2 && raw[ne - 2] == '/' && raw[ne - 1] >= '0' && raw[ne - 1] <= '9') name_len -= 2;      // trim_readno (query.c:139-143)
        uint32_t se = l2 - 1; while (se > l1 && raw[se - 1] == '\r') --se;
        if (l2 >= l3 || raw[l2] != '+') err |= 2;
        uint32_t qe = l4 - 1; while (qe > l3 && raw[qe - 1] == '\r') --qe;
        const uint32_t L = se - l1;
        if (qe - l3 != L) err |= 4;
        if (L == 0) err |= 8;
        FqRec r; r.name_off = base + l0 + 1; r.name_len = name_len; r.seq_off = base + l1; r.len = L; r.qual_off = base + l3;
        rec[stride * i + which] = r; len[stride * i + which] = L;
        if (err) { atomicOr(ctl, err); atomicMin(ctl + 2, (uint32_t)i); }
        atomicMax(ctl + 1, L);
    }
}

// one wave per record: the aligner's byte codes
__global__ void __launch_bounds__(256) k_fq_codes(const uint8_t *__restrict__ raw, const FqRec *__restrict__ rec, const uint32_t *__restrict__ offs, uint32_t n_rec,
                                                   uint8_t *__restrict__ seqs)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t i = wave; i < n_rec; i += n_waves) {
        const FqRec r = rec[i];
        const uint32_t o = offs[i];
        for (uint32_t j = lane; j < r.len; j += 64) seqs[o + j] = nt4_dev(raw[r.seq_off + j]);
    }
}

// ---- SAM text --------------------------------------------------------------------------------------------------------------
template <bool WRITE> struct Emit {
    char *p; uint32_t n; uint32_t cap = 0xFFFFFFFFu;           // bytes past cap are counted, not written
    __device__ __forceinline__ void put(char c) { if (WRITE && n < cap) p[n] = c; ++n; }
    __device__ __forceinline__ void puts(const char *t) { while (*t) put(*t++); }
    __device__ __forceinline__ void putn(const uint8_t *t, uint32_t k) { for (uint32_t i = 0; i < k; ++i) put((char)t[i]); }
    __device__ __forceinline__ void putu(uint64_t v)
    {
        char tmp[20]; int k = 0;
        do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (k) put(tmp[--k]);
    }
};

__device__ __forceinline__ int seq_id_dev(const SamDev &d, int64_t coor)            // bns_coor_pac2real's search (bntseq.c:269-289)
{
    int left = 0, mid = 0, right = d.n_contigs;
    while (left < right) {
        mid = (left + right) >> 1;
        if (coor >= d.c_off[mid]) {
            if (mid == d.n_contigs - 1) break;
            if (coor < d.c_off[mid + 1]) break;
            left = mid + 1;
        } else right = mid;
    }
    return mid;
}
__device__ __forceinline__ uint32_t pac_at_dev(const uint32_t *text, uint32_t l) { return (text[l >> 4] >> (30 - 2 * (l & 15u))) & 3u; }
__device__ __forceinline__ uint32_t mask_at_dev(const uint32_t *ref, uint32_t l) { return (ref[l >> 3] >> (4 * (l & 7u))) & 15u; }

template <bool WRITE> __device__ __forceinline__ void put_cigar_dev(Emit<WRITE> &o, const uint16_t *ops, int n)
{
    for (int i = 0; i < n; ++i) { o.putu(ops[i] >> 4); o.put("MID?"[ops[i] & 3]); }
}

// the aligned base i of the read: codes as read, or the reverse complement for strand 1
__device__ __forceinline__ uint32_t aligned_base(const uint8_t *sq, uint32_t L, uint32_t strand, uint32_t i)
{
    if (!strand) return sq[i];
    const uint32_t c = sq[L - 1 - i];
    return c < 4 ? 3u - c : c;
}

template <bool WRITE> __device__ void sam_tags(Emit<WRITE> &o, const SamDev &d, const salt_result_t *q, const uint8_t *sq, uint32_t L, uint32_t strand, bool md);

// A record is  name | head | SEQ | '\t' | QUAL | tail | newline(s).  head (flag ... the tab in front of SEQ) and tail (the tags) are small
// and irregular: one thread formats them (sam_head / sam_tail).  name, SEQ and QUAL are most of the bytes and plain copies: the lanes of
// a group share them (k_sam_write).  what: 0 a record, 1 an empty line (a skipped read), 2 a record without tail (unmapped, sam.c:105-125).
struct SamShape { uint32_t what, strand; };
template <bool WRITE>
__device__ SamShape sam_head(Emit<WRITE> &o, const SamDev &d, uint32_t i)
{
    const salt_result_t *q = d.res + i;
    if (q->skipped) return SamShape{ 1u, 0u };                       // > 200 N: the reference prints an empty line (alnse.c:1328,1437)
    if (q->pos == 0xFFFFFFFFu) { o.puts("\t4\t*\t0\t0\t*\t*\t0\t0\t"); return SamShape{ 2u, 0u }; }      // sam.c:105-125
    const uint32_t strand = q->strand;
    const int rid = seq_id_dev(d, q->pos);
    o.put('\t'); o.putu(strand ? 16 : 0); o.put('\t');
    o.putn(reinterpret_cast<const uint8_t *>(d.c_names) + d.c_name_off[rid], d.c_name_off[rid + 1] - d.c_name_off[rid]); o.put('\t');
    o.putu((uint64_t)((int64_t)q->pos - d.c_off[rid] + 1)); o.put('\t'); o.putu(q->mapq); o.put('\t');
    put_cigar_dev(o, q->cigar, q->n_cigar);
    o.puts("\t*\t0\t0\t");
    return SamShape{ 0u, strand };
}
template <bool WRITE>
__device__ void sam_tail(Emit<WRITE> &o, const SamDev &d, uint32_t i, SamShape sh)
{
    if (sh.what) return;
    sam_tags<WRITE>(o, d, d.res + i, d.seqs + d.offs[i], d.rec[i].len, sh.strand, d.nm_md != 0);
}
// SEQ and QUAL as the record shows them: the read's bases, or their reverse complement and the reversed qualities on strand 1
__device__ __forceinline__ char seq_char(const uint8_t *sq, uint32_t L, uint32_t strand, uint32_t j)
{
    const uint32_t c = aligned_base(sq, L, strand, j);
    return (char)((0x4E54474341ull >> (8u * (c > 4 ? 4u : c))) & 0xFFu);      // "ACGTN"
}
__device__ __forceinline__ char qual_char(const uint8_t *qual, uint32_t L, uint32_t strand, uint32_t j) { return (char)qual[strand ? L - 1 - j : j]; }

template <bool WRITE> __device__ SamShape sam_head_pe(Emit<WRITE> &o, const SamDev &d, uint32_t i);
template <bool WRITE> __device__ void sam_tail_pe(Emit<WRITE> &o, const SamDev &d, uint32_t i, SamShape sh);

// the whole record by one thread (records whose head or tail outgrow their slot)
template <bool WRITE>
__device__ uint32_t sam_record(const SamDev &d, uint32_t i, char *dst)
{
    Emit<WRITE> o{ dst, 0 };
    const FqRec r = d.rec[i];
    const uint8_t *name = d.raw + r.name_off, *qual = d.raw + r.qual_off, *sq = d.seqs + d.offs[i];
    Emit<false> probe{ nullptr, 0 };                                 // (what kind of record: a skipped read has no name either)
    const SamShape sh = d.pe ? sam_head_pe<false>(probe, d, i) : sam_head<false>(probe, d, i);
    if (sh.what == 1) return 0;
    o.putn(name, r.name_len);
    if (d.pe) (void)sam_head_pe<WRITE>(o, d, i); else (void)sam_head<WRITE>(o, d, i);
    for (uint32_t j = 0; j < r.len; ++j) o.put(seq_char(sq, r.len, sh.strand, j));
    o.put('\t');
    for (uint32_t j = 0; j < r.len; ++j) o.put(qual_char(qual, r.len, sh.strand, j));
    if (d.pe) sam_tail_pe<WRITE>(o, d, i, sh); else sam_tail<WRITE>(o, d, i, sh);
    return o.n;
}

// sixteen bases from base b0 on of a 2-bit array (16 per word, the first in the top bits); nv: how many of them exist
__device__ __forceinline__ uint32_t tb_window(const uint32_t *w, uint32_t b0, uint32_t nv)
{
    const uint32_t wi = b0 >> 4, s2 = 2u * (b0 & 15u);
    uint32_t x = w[wi] << s2;
    if (s2 && nv > 16u - (b0 & 15u)) x |= w[wi + 1] >> (32u - s2);
    return x;
}
// false: not this way (a read with N: the packed words carry N as 0)
template <bool WRITE>
__device__ bool sam_md_words(Emit<WRITE> &o, const SamDev &d, const salt_result_t *q, uint32_t i, uint32_t strand)
{
    const uint32_t *rec = d.tb + (uint64_t)i * d.pg.tb_stride;
    const uint32_t *rw = rec + (strand ? d.pg.nw16 : 0u), *nw = rec + 2u * d.pg.nw16 + (strand ? d.pg.nw32 : 0u);
    uint32_t any_n = 0;
    for (uint32_t k = 0; k < d.pg.nw32; ++k) any_n |= nw[k];
    if (any_n) return false;
    const uint32_t n = q->cigar[0] >> 4, rp0 = q->pos, si0 = (uint32_t)q->seq_start;
    const char *NT = "ACGTN";
    int nm = 0, n_rs = 0, prev = -1;
    o.puts("\tMD:Z:");
    for (uint32_t c = 0; c < n; c += 16) {
        const uint32_t nv = n - c < 16u ? n - c : 16u;
        const uint32_t R = tb_window(rw, si0 + c, nv), G = tb_window(d.text, rp0 + c, nv);
        uint32_t m = ((R ^ G) | ((R ^ G) >> 1)) & 0x55555555u;              // base j of the window: bit 30 - 2j
        if (nv < 16u) m &= ~((1u << (32u - 2u * nv)) - 1u);
        while (m) {
            const uint32_t j = ((uint32_t)__clz((int)m) - 1u) >> 1, sh = 30u - 2u * j;
            const int k = (int)(c + j);
            if (k - prev - 1 > 0) o.putu((uint64_t)(k - prev - 1));
            o.put(NT[(G >> sh) & 3u]);
            if ((mask_at_dev(d.ref, rp0 + (uint32_t)k) & (1u << ((R >> sh) & 3u))) != 0 && n_rs < 64) ++n_rs;
            prev = k; ++nm;
            m &= ~(1u << sh);
        }
    }
    if ((int)n - prev - 1 > 0) o.putu((uint64_t)((int)n - prev - 1));
    o.puts("\tNM:i:"); o.putu((uint64_t)nm);
    if (n_rs > 0) {                                                   // the offsets of the mismatches that are listed alleles
        o.puts("\tXV:i:");
        int seen = 0;
        for (uint32_t c = 0; c < n && seen < n_rs; c += 16) {
            const uint32_t nv = n - c < 16u ? n - c : 16u;
            const uint32_t R = tb_window(rw, si0 + c, nv), G = tb_window(d.text, rp0 + c, nv);
            uint32_t m = ((R ^ G) | ((R ^ G) >> 1)) & 0x55555555u;
            if (nv < 16u) m &= ~((1u << (32u - 2u * nv)) - 1u);
            while (m && seen < n_rs) {
                const uint32_t j = ((uint32_t)__clz((int)m) - 1u) >> 1, sh = 30u - 2u * j;
                if ((mask_at_dev(d.ref, rp0 + c + j) & (1u << ((R >> sh) & 3u))) != 0) { if (seen) o.put(','); o.putu((uint64_t)(c + j)); ++seen; }
                m &= ~(1u << sh);
            }
        }
    }
    return true;
}

// XA and MD / NM / XV of one record, shared by both record kinds (sam_add_xa sam.c:186-240, sam_add_md_nm sam.c:246-328)
template <bool WRITE>
__device__ void sam_tags(Emit<WRITE> &o, const SamDev &d, const salt_result_t *q, const uint8_t *sq, uint32_t L, uint32_t strand, bool md)
{
    const char *NT = "ACGTN";
    {
        bool first = true; int h = 0;
        for (int s = 0; s < 2; ++s)
            for (int k = 0; k < q->n_hits[s]; ++k, ++h) {
                const salt_hit_t hit = q->hits[s][k];
                if (hit.pos == q->pos) continue;
                if (first) { o.puts("\tXA:Z:"); first = false; }
                const int r2 = seq_id_dev(d, hit.pos);
                o.putn(reinterpret_cast<const uint8_t *>(d.c_names) + d.c_name_off[r2], d.c_name_off[r2 + 1] - d.c_name_off[r2]); o.put(','); o.put("+-"[s]);
                o.putu((uint64_t)((int64_t)hit.pos - d.c_off[r2] + 1)); o.put(',');
                if (d.xa_cigar) {
                    if (hit.is_gap) put_cigar_dev(o, q->hit_cigar[h], q->hit_n_cigar[h]);
                    else { o.putu((uint64_t)L); o.put('M'); }
                    o.put(',');
                } else o.puts("*,");
                o.putu(hit.n_diff); o.put(';');
            }
    }
    // MD / NM / XV of an alignment without gaps (one M operation: nearly every record) from the packed 2-bit words of the read (k_pack's
    // tb record of the aligned strand) and of the genome, sixteen bases per XOR: every load is issued before any is needed.  (The walk
    // below takes a base of each per step and decides on it before the next load: ~100 dependent round trips per record.)
    if (md && d.tb && q->n_cigar == 1 && (q->cigar[0] & 15) == 0 && sam_md_words<WRITE>(o, d, q, (uint32_t)(q - d.res), strand)) { }
    else if (md) {
        int nm = 0, n_match = 0, n_rs = 0;
        uint32_t rp = q->pos; int si = q->seq_start;
        o.puts("\tMD:Z:");
        for (int c = 0; c < q->n_cigar; ++c) {
            const int n = q->cigar[c] >> 4, op = q->cigar[c] & 15;
            if (op == 0) {
                for (int k = 0; k < n; ++k, ++rp, ++si) {
                    const uint32_t bt = pac_at_dev(d.text, rp), b = aligned_base(sq, L, strand, (uint32_t)si);
                    if (bt == b) { ++n_match; continue; }
                    if (b < 5 && (mask_at_dev(d.ref, rp) & (1u << b)) != 0 && n_rs < 64) ++n_rs;
                    ++nm;
                    if (n_match) o.putu((uint64_t)n_match);
                    n_match = 0;
                    o.put(NT[bt]);
                }
            } else if (op == 1) { nm += n; si += n; }
            else if (op == 2) {
                if (n_match) o.putu((uint64_t)n_match);
                n_match = 0; nm += n; o.put('^');
                for (int k = 0; k < n; ++k, ++rp) o.put(NT[pac_at_dev(d.text, rp)]);
            }
        }
        if (n_match) o.putu((uint64_t)n_match);
        o.puts("\tNM:i:"); o.putu((uint64_t)nm);
        if (n_rs > 0) {                                               // the offsets of the mismatches that are listed alleles: second walk, no array
            o.puts("\tXV:i:");
            int seen = 0; rp = q->pos; si = q->seq_start;
            for (int c = 0; c < q->n_cigar && seen < n_rs; ++c) {
                const int n = q->cigar[c] >> 4, op = q->cigar[c] & 15;
                if (op == 0) {
                    for (int k = 0; k < n && seen < n_rs; ++k, ++rp, ++si) {
                        const uint32_t bt = pac_at_dev(d.text, rp), b = aligned_base(sq, L, strand, (uint32_t)si);
                        if (bt == b) continue;
                        if (b < 5 && (mask_at_dev(d.ref, rp) & (1u << b)) != 0) { if (seen) o.put(','); o.putu((uint64_t)(si - q->seq_start)); ++seen; }
                    }
                } else if (op == 1) si += n;
                else if (op == 2) rp += (uint32_t)n;
            }
        }
    }
    if (d.rg_len) { o.puts("\tRG:Z:"); o.putn(reinterpret_cast<const uint8_t *>(d.rg), (uint32_t)d.rg_len); }
}

// one record of a pair (alnpe_sam, sam.c:331-457): record i = mate i & 1 of pair i >> 1; the writer adds the reference's two newlines
template <bool WRITE>
__device__ SamShape sam_head_pe(Emit<WRITE> &o, const SamDev &d, uint32_t i)
{
    const uint32_t me = i, ot = i ^ 1u, m0 = i & ~1u;
    const salt_result_t *q = d.res + me, *qo = d.res + ot, *q0 = d.res + m0, *q1 = q0 + 1;
    const uint32_t L = d.rec[me].len;
    const bool map_me = q->pos != 0xFFFFFFFFu, map_ot = qo->pos != 0xFFFFFFFFu;
    int rid_me = -1, rid_ot = -1; uint32_t pos_me = 0, pos_ot = 0;
    if (map_me) { rid_me = seq_id_dev(d, q->pos); pos_me = q->pos - (uint32_t)d.c_off[rid_me] + 1; }
    if (map_ot) { rid_ot = seq_id_dev(d, qo->pos); pos_ot = qo->pos - (uint32_t)d.c_off[rid_ot] + 1; }
    int tlen = 0;
    if (map_me && map_ot) {                                           // computed on (mate 0, mate 1) whichever record this is
        const uint32_t p0 = (i & 1u) ? pos_ot : pos_me, p1 = (i & 1u) ? pos_me : pos_ot;
        if (rid_me != rid_ot) tlen = 0;
        else if (p0 < p1) tlen = (int)(p1 + q1->seq_end - q1->seq_start + 1 - p0);
        else tlen = (int)(p0 + q0->seq_end - q1->seq_start + 1 - p1);                        // sam.c:355-356: q[1].seq_start in both arms
        if ((uint32_t)tlen > d.max_tlen || (uint32_t)tlen < d.min_tlen) tlen = 0;
    }
    uint32_t flag = 0x1;
    if (!map_me) flag |= 0x4;
    if (!map_ot) flag |= 0x8;
    if (q->strand == 1) flag |= 0x10;
    if (qo->strand == 1) flag |= 0x20;
    if (tlen != 0) flag |= 0x2;
    flag |= (i & 1u) ? 0x80 : 0x40;
    o.put('\t'); o.putu(flag); o.put('\t');
    auto contig = [&](int rid) { o.putn(reinterpret_cast<const uint8_t *>(d.c_names) + d.c_name_off[rid], d.c_name_off[rid + 1] - d.c_name_off[rid]); };
    if (map_me) {
        contig(rid_me); o.put('\t'); o.putu(pos_me); o.put('\t'); o.putu(q->mapq); o.put('\t');
        if (q->seq_start != 0) { o.putu(q->seq_start); o.put('S'); }
        put_cigar_dev(o, q->cigar, q->n_cigar);
        if (q->seq_end != L - 1) { o.putu((uint64_t)(L - q->seq_end - 1)); o.put('S'); }
        o.put('\t');
    } else if (map_ot) { contig(rid_ot); o.put('\t'); o.putu(pos_ot); o.puts("\t255\t*\t"); }
    else o.puts("*\t0\t255\t*\t");
    if (map_ot) {
        if (rid_me == rid_ot || !map_me) o.puts("=\t"); else { contig(rid_ot); o.put('\t'); }
        o.putu(pos_ot); o.put('\t');
    } else o.puts("*\t0\t");
    if (tlen != 0) { if (q->pos >= qo->pos) o.put('-'); o.putu((uint64_t)tlen); o.put('\t'); }
    else o.puts("0\t");
    return SamShape{ 0u, q->strand == 1 ? 1u : 0u };
}
template <bool WRITE>
__device__ void sam_tail_pe(Emit<WRITE> &o, const SamDev &d, uint32_t i, SamShape sh)
{
    const salt_result_t *q = d.res + i;
    sam_tags<WRITE>(o, d, q, d.seqs + d.offs[i], d.rec[i].len, sh.strand, d.nm_md && q->pos != 0xFFFFFFFFu);
}

// k_sam_len: one thread per record formats the record's head and tail ONCE, into the record's slot (SAM_HEAD_CAP + SAM_TAIL_CAP bytes), and
// leaves their lengths: the record's length is then arithmetic (name + head + 2 L + 1 + tail + newlines) and k_sam_write copies.
// total64: the block's byte count in 64 bits next to the 32-bit offsets of the scan (a block whose SAM text passes 4 GiB is refused)
__global__ void __launch_bounds__(256) k_sam_len(SamDev d, uint32_t n, uint32_t *__restrict__ len, unsigned long long *__restrict__ total64)
{
    unsigned long long mine = 0;
    TSTRIDE(i, n) {
        char *slot = d.slot + (uint64_t)i * SAM_SLOT;
        Emit<true> h{ slot, 0, SAM_HEAD_CAP }, t{ slot + SAM_HEAD_CAP, 0, SAM_TAIL_CAP };
        SamShape sh;
        if (d.pe) { sh = sam_head_pe<true>(h, d, (uint32_t)i); sam_tail_pe<true>(t, d, (uint32_t)i, sh); }
        else { sh = sam_head<true>(h, d, (uint32_t)i); sam_tail<true>(t, d, (uint32_t)i, sh); }
        const FqRec r = d.rec[i];
        SamSeg g; g.head_len = (uint16_t)(h.n < 0xFFFFu ? h.n : 0xFFFFu); g.tail_len = (uint16_t)(t.n < 0xFFFFu ? t.n : 0xFFFFu);
        g.what = (uint8_t)sh.what; g.strand = (uint8_t)sh.strand; g.over = (uint8_t)(h.n > SAM_HEAD_CAP || t.n > SAM_TAIL_CAP); g.pad = 0;
        d.seg[i] = g;
        const uint32_t l = (sh.what == 1 ? 0u : r.name_len + h.n + 2u * r.len + 1u + t.n) + (d.pe ? 2u : 1u);      // + the newline (PE: and the driver's, alnpe.c:640-648)
        len[i] = l; mine += l;
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
    if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(total64, mine);
}
// k_sam_write: eight lanes per record copy its pieces to the record's place in the block (off[] is the scan of the lengths): the name out of
// the FASTQ text, the head out of the slot, SEQ from the aligner's codes (reverse complement on strand 1), QUAL from the FASTQ text
// (reversed on strand 1), the tail out of the slot, the newline(s).  Every load is independent of every other: no lane walks a record byte
// by byte behind its own loads, which is what a record per thread did (1.4 ms per 154 000 records, three waves per CU beside the 48 KB
// LDS image the threads formatted into).  A record whose head or tail outgrew the slot is formatted by one lane, the old way.
__global__ void __launch_bounds__(256) k_sam_write(SamDev d, uint32_t n, const uint32_t *__restrict__ off, char *__restrict__ out)
{
    const uint32_t s = threadIdx.x & 7u;
    const uint64_t grp = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3, n_grp = ((uint64_t)gridDim.x * blockDim.x) >> 3;
    for (uint64_t i = grp; i < n; i += n_grp) {
        const SamSeg g = d.seg[i];
        char *dst = out + off[i];
        const uint32_t nl = d.pe ? 2u : 1u;
        if (g.what == 1) { if (s < nl) dst[s] = '\n'; continue; }
        if (g.over) {
            if (s == 0) { const uint32_t w = sam_record<true>(d, (uint32_t)i, dst); dst[w] = '\n'; if (d.pe) dst[w + 1] = '\n'; }
            continue;
        }
        const FqRec r = d.rec[i];
        const uint8_t *name = d.raw + r.name_off, *qual = d.raw + r.qual_off, *sq = d.seqs + d.offs[i];
        const char *slot = d.slot + i * SAM_SLOT;
        const uint32_t L = r.len, strand = g.strand;
        for (uint32_t j = s; j < r.name_len; j += 8) dst[j] = (char)name[j];
        dst += r.name_len;
        for (uint32_t j = s; j < g.head_len; j += 8) dst[j] = slot[j];
        dst += g.head_len;
        for (uint32_t j = s; j < L; j += 8) { dst[j] = seq_char(sq, L, strand, j); dst[L + 1 + j] = qual_char(qual, L, strand, j); }
        if (s == 0) dst[L] = '\t';
        dst += 2u * L + 1u;
        for (uint32_t j = s; j < g.tail_len; j += 8) dst[j] = slot[SAM_HEAD_CAP + j];
        if (s < nl) dst[g.tail_len + s] = '\n';
    }
}

// ---- launchers (all on stream st; scan temporaries are the caller's) ----
hipError_t text_warm()
{
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_fq_lines));
}

size_t text_scan_bytes(uint64_t max_items)
{
    size_t b = 0;
    uint32_t *p = nullptr;
    (void)rocprim::exclusive_scan(nullptr, b, p, p, 0u, (size_t)max_items, rocprim::plus<uint32_t>(), nullptr);
    return b;
}
// newline count per tile, scanned in place: tile_cnt[t] = newlines in front of tile t, tile_cnt[n_tiles] = all of them
hipError_t launch_fq_count(const uint8_t *raw, uint64_t n, uint32_t *tile_cnt, void *tmp, size_t tmp_bytes, hipStream_t st)
{
    const uint64_t n_tiles = (n + TILE - 1) / TILE;
    hipError_t e = hipMemsetAsync(tile_cnt + n_tiles, 0, 4, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fq_count, dim3(tgrid(n_tiles * 256)), dim3(256), 0, st, raw, n, tile_cnt);
    return rocprim::exclusive_scan(tmp, tmp_bytes, tile_cnt, tile_cnt, 0u, (size_t)n_tiles + 1, rocprim::plus<uint32_t>(), st);
}
// line_start[0 .. newlines]: the caller has made sure the table holds newlines + 1 entries
hipError_t launch_fq_lines(const uint8_t *raw, uint64_t n, const uint32_t *tile_off, uint32_t *line_start, hipStream_t st)
{
    const uint64_t n_tiles = (n + TILE - 1) / TILE;
    hipError_t e = hipMemsetAsync(line_start, 0, 4, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fq_lines, dim3(tgrid(n_tiles * 256)), dim3(256), 0, st, raw, n, tile_off, line_start);
    return hipGetLastError();
}
// ctl = { 0, 0, 0xFFFFFFFF, 0 } by two memsets (no host buffer that would have to outlive the enqueue)
hipError_t launch_fq_ctl_init(uint32_t *ctl, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(ctl, 0, 16, st);
    if (e != hipSuccess) return e;
    return hipMemsetAsync(ctl + 2, 0xFF, 4, st);
}
// records -> FqRec + lengths; offs[0..n_rec] = exclusive scan of the lengths (offs[n_rec] = all bases); ctl = { error bits, longest read, first bad record }
hipError_t launch_fq_parse(const uint8_t *raw, const uint32_t *line_start, uint32_t n_rec, FqRec *rec, uint32_t *offs, uint32_t *ctl,
                           void *tmp, size_t tmp_bytes, hipStream_t st)
{
    hipError_t e = launch_fq_ctl_init(ctl, st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(offs + n_rec, 0, 4, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fq_parse, dim3(tgrid(n_rec)), dim3(256), 0, st, raw, line_start, n_rec, rec, offs, ctl, 0u, 1u, 0u);
    return rocprim::exclusive_scan(tmp, tmp_bytes, offs, offs, 0u, (size_t)n_rec + 1, rocprim::plus<uint32_t>(), st);
}
hipError_t launch_fq_parse_mate(const uint8_t *raw, uint32_t base, const uint32_t *line_start, uint32_t n_rec, uint32_t which, FqRec *rec, uint32_t *len,
                                uint32_t *ctl, hipStream_t st)
{
    hipLaunchKernelGGL(k_fq_parse, dim3(tgrid(n_rec)), dim3(256), 0, st, raw + base, line_start, n_rec, rec, len, ctl, base, 2u, which);
    return hipGetLastError();
}
hipError_t launch_text_scan(uint32_t *v, uint32_t n_plus_1, void *tmp, size_t tmp_bytes, hipStream_t st)
{
    return rocprim::exclusive_scan(tmp, tmp_bytes, v, v, 0u, (size_t)n_plus_1, rocprim::plus<uint32_t>(), st);
}
hipError_t launch_fq_codes(const uint8_t *raw, const FqRec *rec, const uint32_t *offs, uint32_t n_rec, uint8_t *seqs, hipStream_t st)
{
    hipLaunchKernelGGL(k_fq_codes, dim3(tgrid((uint64_t)n_rec * 64)), dim3(256), 0, st, raw, rec, offs, n_rec, seqs);
    return hipGetLastError();
}
hipError_t launch_sam_len(const SamDev &d, uint32_t n, uint32_t *off, unsigned long long *total64, void *tmp, size_t tmp_bytes, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(off + n, 0, 4, st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(total64, 0, 8, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_sam_len, dim3(tgrid(n)), dim3(256), 0, st, d, n, off, total64);
    return rocprim::exclusive_scan(tmp, tmp_bytes, off, off, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), st);                        // off[n] = all bytes
}
hipError_t launch_sam_write(const SamDev &d, uint32_t n, const uint32_t *off, char *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_sam_write, dim3(tgrid((uint64_t)n * 8)), dim3(256), 0, st, d, n, off, out);
    return hipGetLastError();
}

} // namespace salt
