// salt_amd/csrc/salt_text.hip -- FASTQ text in, SAM text out, on the device (gfx950).
//
// The reference parses FASTQ with kseq (Align_src/query.c:146-239, kseq.h) and formats SAM with ksprintf one character at a time
// (sam.c:87-328) on the host; at the rate the align kernels run that is two orders of magnitude too slow (SURVEY 7 "host I/O at
// target rate", row N2).  Here both ends are kernels, and the host only moves bytes:
//   k_fq_count / k_fq_lines   newline scan of the raw block -> start offset of every line
//   k_fq_parse                one thread per 4-line record: name (up to the first white space, a trailing /1 /2 trimmed: query.c:139-143),
//                             sequence and quality spans (CR trimmed), checks; read lengths for the offsets scan
//   k_fq_codes                bases -> codes A0 C1 G2 T3 other 4 (nst_nt4_table), the aligner's input layout
//   k_sam_len / k_sam_write   one thread per read: aln_samse (sam.c:87-182) with sam_add_xa (186-240) and sam_add_md_nm (246-328):
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
    char *p; uint32_t n;
    __device__ __forceinline__ void put(char c) { if (WRITE) p[n] = c; ++n; }
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

template <bool WRITE>
__device__ uint32_t sam_record(const SamDev &d, uint32_t i, char *dst)
{
    Emit<WRITE> o{ dst, 0 };
    const salt_result_t *q = d.res + i;
    if (q->skipped) return 0;                                         // > 200 N: the reference prints an empty line (alnse.c:1328,1437)
    const FqRec r = d.rec[i];
    const uint8_t *name = d.raw + r.name_off, *qual = d.raw + r.qual_off, *sq = d.seqs + d.offs[i];
    const uint32_t L = r.len;
    const char *NT = "ACGTN";
    if (q->pos == 0xFFFFFFFFu) {                                      // sam.c:105-125
        o.putn(name, r.name_len); o.puts("\t4\t*\t0\t0\t*\t*\t0\t0\t");
        for (uint32_t j = 0; j < L; ++j) o.put(NT[sq[j] > 4 ? 4 : sq[j]]);
        o.put('\t'); o.putn(qual, L);
        return o.n;
    }
    const uint32_t strand = q->strand;
    const int rid = seq_id_dev(d, q->pos);
    o.putn(name, r.name_len); o.put('\t'); o.putu(strand ? 16 : 0); o.put('\t');
    o.putn(reinterpret_cast<const uint8_t *>(d.c_names) + d.c_name_off[rid], d.c_name_off[rid + 1] - d.c_name_off[rid]); o.put('\t');
    o.putu((uint64_t)((int64_t)q->pos - d.c_off[rid] + 1)); o.put('\t'); o.putu(q->mapq); o.put('\t');
    put_cigar_dev(o, q->cigar, q->n_cigar);
    o.puts("\t*\t0\t0\t");
    for (uint32_t j = 0; j < L; ++j) { const uint32_t c = aligned_base(sq, L, strand, j); o.put(NT[c > 4 ? 4 : c]); }
    o.put('\t');
    if (strand) for (uint32_t j = L; j > 0; --j) o.put((char)qual[j - 1]);
    else o.putn(qual, L);
    sam_tags<WRITE>(o, d, q, sq, L, strand, d.nm_md != 0);
    return o.n;
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
    if (md) {
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

// one record of a pair (alnpe_sam, sam.c:331-457): record i = mate i & 1 of pair i >> 1; the caller adds the reference's two newlines
template <bool WRITE>
__device__ uint32_t sam_record_pe(const SamDev &d, uint32_t i, char *dst)
{
    Emit<WRITE> o{ dst, 0 };
    const uint32_t me = i, ot = i ^ 1u, m0 = i & ~1u;
    const salt_result_t *q = d.res + me, *qo = d.res + ot, *q0 = d.res + m0, *q1 = q0 + 1;
    const FqRec r = d.rec[me];
    const uint8_t *name = d.raw + r.name_off, *qual = d.raw + r.qual_off, *sq = d.seqs + d.offs[me];
    const uint32_t L = r.len;
    const char *NT = "ACGTN";
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
    o.putn(name, r.name_len); o.put('\t'); o.putu(flag); o.put('\t');
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
    const uint32_t strand = q->strand == 1 ? 1u : 0u;
    for (uint32_t j = 0; j < L; ++j) { const uint32_t c = aligned_base(sq, L, strand, j); o.put(NT[c > 4 ? 4 : c]); }
    o.put('\t');
    if (strand) for (uint32_t j = L; j > 0; --j) o.put((char)qual[j - 1]);
    else o.putn(qual, L);
    sam_tags<WRITE>(o, d, q, sq, L, strand, d.nm_md && map_me);
    return o.n;
}

// total64: the block's byte count in 64 bits next to the 32-bit offsets of the scan (a block whose SAM text passes 4 GiB is refused)
__global__ void __launch_bounds__(256) k_sam_len(SamDev d, uint32_t n, uint32_t *__restrict__ len, unsigned long long *__restrict__ total64)
{
    unsigned long long mine = 0;
    if (d.pe) { TSTRIDE(i, n) { const uint32_t l = sam_record_pe<false>(d, (uint32_t)i, nullptr) + 2u; len[i] = l; mine += l; } }   // + the record's newline and the driver's (alnpe.c:640-648)
    else { TSTRIDE(i, n) { const uint32_t l = sam_record<false>(d, (uint32_t)i, nullptr) + 1u; len[i] = l; mine += l; } }          // + the newline
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
    if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(total64, mine);
}
// One wave per 64 consecutive records.  A thread that formats its record straight into the output writes single bytes at its own
// address: 64 different cache lines per store instruction.  The records of a wave are contiguous in the output (off[] is a scan), so
// the threads format into an LDS image of that span and the wave then copies the span out with whole-line stores (PE, per 10^6
// mates: 13 ms -> see DESIGN 4.3).  A span that does not fit the image (very long reads) is written the direct way.
static constexpr uint32_t SAM_STAGE = 48u << 10;
__global__ void __launch_bounds__(64) k_sam_write(SamDev d, uint32_t n, const uint32_t *__restrict__ off, char *__restrict__ out)
{
    __shared__ char stage[SAM_STAGE];
    for (uint64_t base = (uint64_t)blockIdx.x * 64; base < n; base += (uint64_t)gridDim.x * 64) {
        const uint32_t i = (uint32_t)base + threadIdx.x, last = (uint32_t)(base + 64 < n ? base + 64 : n);
        const uint32_t o0 = off[base], span = off[last] - o0;
        const bool staged = span <= SAM_STAGE;
        if (i < n) {
            char *dst = staged ? stage + (off[i] - o0) : out + off[i];
            if (d.pe) { const uint32_t w = sam_record_pe<true>(d, i, dst); dst[w] = '\n'; dst[w + 1] = '\n'; }
            else { const uint32_t w = sam_record<true>(d, i, dst); dst[w] = '\n'; }
        }
        __syncthreads();
        if (staged) for (uint32_t k = threadIdx.x; k < span; k += 64) out[o0 + k] = stage[k];
        __syncthreads();
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
    const uint64_t waves = ((uint64_t)n + 63) / 64;
    hipLaunchKernelGGL(k_sam_write, dim3((uint32_t)(waves < 8192 ? waves : 8192)), dim3(64), 0, st, d, n, off, out);
    return hipGetLastError();
}

} // namespace salt
