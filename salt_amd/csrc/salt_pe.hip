// salt_amd/csrc/salt_pe.hip -- paired-end stage kernels (gfx950): pairing decisions and Smith-Waterman mate rescue.
//
//   k_pair   one lane per pair: pairing2 / pairing_singleton up to the point where they call Smith-Waterman
//            (Align_src/alnpe.c:94-258, 395-480): accept the primaries if properly oriented and spaced, else the best
//            pair among the alternative hits, else queue up to two rescue windows (tried in the reference's order).
//   k_sw     eight lanes per rescue request = the eight 16-bit lanes of the reference's SSE2 word kernel
//            (sw_sse2_word, ssw.c:371-547), emulated operation by operation because its "lazy F" pass is not the
//            textbook recurrence (a deletion may not follow an insertion across a stripe boundary); forward pass,
//            reverse pass from the end point (ssw_align, ssw.c:771-856), second-best score outside the mask, then the
//            banded traceback (banded_sw, ssw.c:549-727) by the group's first lane.
// The results are applied to the mates by k_pe_final (salt_align.hip), which also writes the CIGARs.
#include "salt_device.h"
#include <stdlib.h>
#include "salt_kernels.h"

namespace salt {

// score of a reference symbol against a read base code (0..3, 4 = N) as ssw_init builds its profile from
// score_mat2 indexed [ref*16 + (1<<code)] (aware, alnpe.c:58-73,283) or score_mat [ref*5 + code] (alnpe.c:52-56)
__device__ __forceinline__ int sw_score(bool aware, uint32_t ref, uint32_t code)
{
    if (aware) {
        if (code > 3) return -3;                              // 1<<4 = 16 indexes column 0 of the next row: always -3
        const uint32_t bit = 1u << code;                      // only rows 1,2,4,8 of the (transposed) matrix are non-trivial
        return ((ref == 1 || ref == 2 || ref == 4 || ref == 8) && (ref & bit)) ? 1 : -3;
    }
    if (ref > 3 || code > 3) return -1;
    return ref == code ? 1 : -3;
}

__device__ __forceinline__ int sat16(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }
__device__ __forceinline__ int subu16(int a, int b) { unsigned x = (unsigned)a & 0xFFFFu; return (int)(short)(x > (unsigned)b ? x - (unsigned)b : 0u); }

struct SwLds {                       // per 8-lane group: views into the block's dynamic LDS, seg = ceil(max read length / 8)
    short *H[2], *E, *Hmax;          // [seg][8]
    uint8_t *read;                   // [8 * seg]
};
__device__ __forceinline__ uint32_t sw_group_bytes(uint32_t seg) { return 4u * seg * 16u + ((8u * seg + 15u) & ~15u); }


__device__ __forceinline__ uint32_t ref_symbol(const IndexView &ix, const uint8_t *pac, bool aware, uint32_t p)
{
    if (aware) return (ix.ref[p >> 3] >> (4 * (p & 7u))) & 15u;
    return (pac[p >> 2] >> ((~p & 3u) << 1)) & 3u;
}

// one striped pass; all 8 lanes of the group call it together.  rd(q): read code at position q of this pass.
template <class ReadAt>
__device__ void sw_word_pass(const IndexView &ix, const uint8_t *pac, bool aware, SwLds &s, uint32_t ref0, int ref_dir, int refLen,
                             int readLen, ReadAt rd, int terminate, uint16_t *maxColumn, int &out_max, int &out_end_ref, int &out_end_read)
{
    const int lane = (int)(threadIdx.x & 7u);
    const uint64_t gmask = 0xFFull << (threadIdx.x & 56u);
    const int segLen = (readLen + 7) / 8, go = 3, ge = 1;      // aln.h:137-138
    int cur = 0, max = 0, end_ref = 0;
    for (int j = 0; j < segLen; ++j) { s.H[0][j * 8 + lane] = 0; s.H[1][j * 8 + lane] = 0; s.E[j * 8 + lane] = 0; s.Hmax[j * 8 + lane] = 0; }
    int vMaxScore = 0, vMaxMark = 0;
    const int begin = ref_dir ? refLen - 1 : 0, end = ref_dir ? -1 : refLen, step = ref_dir ? -1 : 1;
    for (int i = begin; i != end; i += step) {
        const uint32_t sym = ref_symbol(ix, pac, aware, ref0 + (uint32_t)i);
        int vF = 0, vMaxColumn = 0;
        int vH = __shfl_up((int)s.H[cur][(segLen - 1) * 8 + lane], 1, 8);
        if (lane == 0) vH = 0;
        const int ld = cur, st = cur ^ 1;                        // pvHLoad = old store, pvHStore = the other buffer
        cur = st;
        for (int j = 0; j < segLen; ++j) {
            const int q = j + lane * segLen;
            const int prof = q >= readLen ? 0 : sw_score(aware, sym, rd(q));
            int h = sat16(vH + prof);
            int e = s.E[j * 8 + lane];
            h = h > e ? h : e; h = h > vF ? h : vF;
            vMaxColumn = vMaxColumn > h ? vMaxColumn : h;
            s.H[st][j * 8 + lane] = (short)h;
            h = subu16(h, go);
            e = subu16(e, ge); e = e > h ? e : h; s.E[j * 8 + lane] = (short)e;
            vF = subu16(vF, ge); vF = vF > h ? vF : h;
            vH = s.H[ld][j * 8 + lane];
        }
        bool lazy_done = false;
        for (int k = 0; k < 8 && !lazy_done; ++k) {              // lazy F (ssw.c:487-497)
            vF = __shfl_up(vF, 1, 8);
            if (lane == 0) vF = 0;
            for (int j = 0; j < segLen; ++j) {
                int h = s.H[st][j * 8 + lane];
                h = h > vF ? h : vF;
                s.H[st][j * 8 + lane] = (short)h;
                h = subu16(h, go);
                vF = subu16(vF, ge);
                if ((__ballot(vF > h) & gmask) == 0) { lazy_done = true; break; }
            }
        }
        vMaxScore = vMaxScore > vMaxColumn ? vMaxScore : vMaxColumn;
        if (__ballot(vMaxMark != vMaxScore) & gmask) {
            vMaxMark = vMaxScore;
            int temp = vMaxScore;
            for (int o = 1; o < 8; o <<= 1) { int t = __shfl_xor(temp, o, 8); temp = temp > t ? temp : t; }
            if (temp > max) { max = temp; end_ref = i; for (int j = 0; j < segLen; ++j) s.Hmax[j * 8 + lane] = s.H[st][j * 8 + lane]; }
        }
        int mc = vMaxColumn;
        for (int o = 1; o < 8; o <<= 1) { int t = __shfl_xor(mc, o, 8); mc = mc > t ? mc : t; }
        if (lane == 0 && maxColumn) maxColumn[i] = (uint16_t)mc;
        if (mc == terminate) break;
    }
    // smallest read position holding the maximum in the best column (ssw.c:504-512)
    int end_read = readLen - 1;
    for (int j = 0; j < segLen; ++j) if ((int)s.Hmax[j * 8 + lane] == max) { int t = j + lane * segLen; if (t < end_read) end_read = t; }
    for (int o = 1; o < 8; o <<= 1) { int t = __shfl_xor(end_read, o, 8); end_read = end_read < t ? end_read : t; }
    out_max = max; out_end_ref = end_ref; out_end_read = end_read;
}

// banded_sw (ssw.c:549-727) by one lane; h_b/e_b/h_c and the direction bytes live in this group's global scratch.
// Returns the number of ops written to cig (len<<4|op), 0 on a traceback error (as the reference: no CIGAR), -1 when the
// band, the direction bytes or the CIGAR (SALT_MAX_CIGAR_OPS) would not fit (the caller counts that as an overflow).
__device__ int sw_banded(const IndexView &ix, const uint8_t *pac, bool aware, uint32_t ref0, const uint8_t *read, int refLen, int readLen,
                         int score, int band_width, int32_t *hb, int32_t *eb, int32_t *hc, int8_t *direction, uint32_t dir_cap,
                         uint16_t *cig, int cig_cap)
{
#define SET_U(u, w, i, j) { int x_ = (i) - (w); x_ = x_ > 0 ? x_ : 0; (u) = (j) - x_ + 1; }
#define SET_D(u, w, i, j, p) { int x_ = (i) - (w); x_ = x_ > 0 ? x_ : 0; x_ = (j) - x_; (u) = x_ * 3 + p; }
    const int go = 3, ge = 1;
    int i, j, e = 0, f, temp1, temp2, l, max = 0, width, width_d;
    int8_t *direction_line = direction;
    do {
        width = band_width * 2 + 3; width_d = band_width * 2 + 1;
        if (width > (int)SW_BAND_W || (uint64_t)width_d * (uint64_t)readLen * 3u + 8u > dir_cap) return -1;
        for (j = 1; j < width - 1; ++j) hb[j] = 0;
        for (i = 0; i < readLen; ++i) {
            int beg = 0, end = refLen - 1, u = 0, edge;
            j = i - band_width; beg = beg > j ? beg : j;
            j = i + band_width; end = end < j ? end : j;
            edge = end + 1 < width - 1 ? end + 1 : width - 1;
            f = hb[0] = eb[0] = hb[edge] = eb[edge] = hc[0] = 0;
            direction_line = direction + width_d * i * 3;
            const uint32_t rc = read[i];
            for (j = beg; j <= end; ++j) {
                int b, e1, f1, d, de, df, dh;
                SET_U(u, band_width, i, j); SET_U(e, band_width, i - 1, j);
                SET_U(b, band_width, i, j - 1); SET_U(d, band_width, i - 1, j - 1);
                SET_D(de, band_width, i, j, 0); SET_D(df, band_width, i, j, 1); SET_D(dh, band_width, i, j, 2);
                temp1 = i == 0 ? -go : hb[e] - go;
                temp2 = i == 0 ? -ge : eb[e] - ge;
                eb[u] = temp1 > temp2 ? temp1 : temp2;
                direction_line[de] = temp1 > temp2 ? 3 : 2;
                temp1 = hc[b] - go; temp2 = f - ge;
                f = temp1 > temp2 ? temp1 : temp2;
                direction_line[df] = temp1 > temp2 ? 5 : 4;
                e1 = eb[u] > 0 ? eb[u] : 0; f1 = f > 0 ? f : 0;
                temp1 = e1 > f1 ? e1 : f1;
                temp2 = hb[d] + sw_score(aware, ref_symbol(ix, pac, aware, ref0 + (uint32_t)j), rc);
                hc[u] = temp1 > temp2 ? temp1 : temp2;
                if (hc[u] > max) max = hc[u];
                if (temp1 <= temp2) direction_line[dh] = 1;
                else direction_line[dh] = e1 > f1 ? direction_line[de] : direction_line[df];
            }
            for (j = 1; j <= u; ++j) hb[j] = hc[j];
        }
        band_width *= 2;
    } while (max < score);
    band_width /= 2;
    i = readLen - 1; j = refLen - 1; e = 0; l = 0; f = max = 0; temp2 = 2;
    uint16_t c[SALT_MAX_CIGAR_OPS + 4];
    while (i > 0) {
        SET_D(temp1, band_width, i, j, temp2);
        switch (direction_line[temp1]) {
        case 1: --i; --j; temp2 = 2; direction_line -= width_d * 3; f = 0; break;
        case 2: --i; temp2 = 0; direction_line -= width_d * 3; f = 1; break;
        case 3: --i; temp2 = 2; direction_line -= width_d * 3; f = 1; break;
        case 4: --j; temp2 = 1; f = 2; break;
        case 5: --j; temp2 = 2; f = 2; break;
        default: return 0;
        }
        if (f == max) ++e;
        else { ++l; if (l > SALT_MAX_CIGAR_OPS) return -1; c[l - 1] = (uint16_t)(e << 4 | max); max = f; e = 1; }
    }
    if (f == 0) { ++l; if (l > SALT_MAX_CIGAR_OPS) return -1; c[l - 1] = (uint16_t)((e + 1) << 4); }
    else { l += 2; if (l > SALT_MAX_CIGAR_OPS) return -1; c[l - 2] = (uint16_t)(e << 4 | f); c[l - 1] = 16; }
    if (l > cig_cap) return -1;
    for (i = 0; i < l; ++i) cig[i] = c[l - 1 - i];
    return l;
#undef SET_U
#undef SET_D
}

// ---------------------------------------------------------------------------------------------
// k_sw: persistent groups of 8 lanes pull rescue requests
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_sw(IndexView ix, const uint8_t *__restrict__ pac, const uint8_t *__restrict__ seqs, const uint32_t *__restrict__ offs,
     const PeSwReq *__restrict__ req, const uint32_t *__restrict__ pctl, PeSwRes *__restrict__ res, uint32_t *__restrict__ head,
     uint32_t *__restrict__ overflow, uint8_t *__restrict__ scratch, uint32_t maxcol_bytes, uint32_t group_bytes, uint32_t seg, int dbg_skip_tb)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t sw_lds[];
    const uint32_t grp = threadIdx.x >> 3, lane = threadIdx.x & 7u;
    SwLds s;
    {
        uint8_t *base = sw_lds + (size_t)grp * sw_group_bytes(seg);
        s.H[0] = reinterpret_cast<short *>(base); s.H[1] = s.H[0] + seg * 8; s.E = s.H[1] + seg * 8; s.Hmax = s.E + seg * 8;
        s.read = reinterpret_cast<uint8_t *>(s.Hmax + seg * 8);
    }
    const uint32_t n_req = pctl[0];
    uint8_t *my = scratch + ((size_t)blockIdx.x * 8 + grp) * group_bytes;
    uint16_t *maxColumn = reinterpret_cast<uint16_t *>(my);
    int32_t *hb = reinterpret_cast<int32_t *>(my + maxcol_bytes), *eb = hb + SW_BAND_W, *hc = eb + SW_BAND_W;
    int8_t *direction = reinterpret_cast<int8_t *>(hc + SW_BAND_W);
    const uint32_t dir_cap = group_bytes - maxcol_bytes - 3u * SW_BAND_W * 4u;
    for (;;) {
        uint32_t it = 0;
        if (lane == 0) it = atomicAdd(head, 1u);
        it = (uint32_t)__shfl((int)it, 0, 8);
        if (it >= n_req) break;
        const PeSwReq rq = req[it];
        PeSwRes out; out.score1 = 0; out.score2 = 0; out.ref_begin = -1; out.ref_end = 0; out.read_begin = -1; out.read_end = 0; out.n_cigar = 0; out.ok = 0; out.start = rq.start; out.strand = rq.strand;
        const uint32_t off = offs[rq.mate], L = offs[rq.mate + 1] - off;
        const int refLen = (int)(rq.end - rq.start + 1);
        const bool aware = rq.aware != 0;
        const bool sane = rq.start < ix.ref_len && refLen > 0;
        const bool fits = sane && (uint64_t)refLen * 2u <= maxcol_bytes && L <= seg * 8u;
        if (sane && !fits && lane == 0) atomicAdd(overflow, 1u);
        if (fits) {
            // the mate's bases on the requested strand
            for (uint32_t i = lane; i < L; i += 8) {
                uint32_t c = rq.strand ? seqs[off + (L - 1 - i)] : seqs[off + i];
                if (rq.strand && c < 4) c = 3 - c;
                s.read[i] = (uint8_t)(c > 4 ? 4 : c);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            int max1, end_ref1, end_read1;
            auto fwd = [&](int q) -> uint32_t { return s.read[q]; };
            sw_word_pass(ix, pac, aware, s, rq.start, 0, refLen, (int)L, fwd, 0xFFFF, maxColumn, max1, end_ref1, end_read1);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            // second best outside +-maskLen around the end column (ssw.c:529-542), maskLen = L/2 >= 15 or none
            int score2 = 0;
            const int maskLen = (int)L / 2;
            if (maskLen >= 15) {
                int edge = end_ref1 - maskLen > 0 ? end_ref1 - maskLen : 0;
                for (int i = (int)lane; i < edge; i += 8) { int v = maxColumn[i]; score2 = score2 > v ? score2 : v; }
                edge = end_ref1 + maskLen > refLen ? refLen : end_ref1 + maskLen;
                for (int i = edge + (int)lane; i < refLen; i += 8) { int v = maxColumn[i]; score2 = score2 > v ? score2 : v; }
                for (int o = 1; o < 8; o <<= 1) { int t = __shfl_xor(score2, o, 8); score2 = score2 > t ? score2 : t; }
            }
            // reverse pass from the end point to find the beginning (ssw.c:817-830)
            int max2, beg_ref, beg_read_rev;
            auto rev = [&](int q) -> uint32_t { return s.read[end_read1 - q]; };
            sw_word_pass(ix, pac, aware, s, rq.start, 1, end_ref1 + 1, end_read1 + 1, rev, max1, (uint16_t *)nullptr, max2, beg_ref, beg_read_rev);
            const int read_begin = end_read1 - beg_read_rev;
            out.score1 = max1; out.score2 = score2; out.ref_begin = beg_ref; out.ref_end = end_ref1; out.read_begin = read_begin; out.read_end = end_read1;
            // banded traceback for the CIGAR (ssw.c:837-848)
            int n_cig = 0;
            if (lane == 0) {
                const int rfl = end_ref1 - beg_ref + 1, rdl = end_read1 - read_begin + 1;
                int bw = rfl - rdl; bw = (bw < 0 ? -bw : bw) + 1;
                if (dbg_skip_tb) { out.cigar[0] = (uint16_t)(rdl << 4); n_cig = 1; }
                else if (rfl > 0 && rdl > 0)
                    n_cig = sw_banded(ix, pac, aware, rq.start + (uint32_t)beg_ref, s.read + read_begin, rfl, rdl, max1, bw, hb, eb, hc, direction,
                                      dir_cap, out.cigar, SALT_MAX_CIGAR_OPS);
            }
            if (lane == 0 && n_cig < 0) { atomicAdd(overflow, 1u); n_cig = 0; }
            n_cig = __shfl(n_cig, 0, 8);
            out.n_cigar = (uint16_t)n_cig;
            out.ok = (uint16_t)((n_cig > 0 && end_read1 - read_begin + 1 >= 20) ? 1 : 0);   // alnpe.c:297 (filters = 0, filterd = 20)
        }
        if (lane == 0) res[it] = out;
    }
}

// ---------------------------------------------------------------------------------------------
// k_pair: pairing2 / pairing_singleton up to the Smith-Waterman calls
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int in_range(uint32_t a, uint32_t b, uint32_t small, uint32_t large)      // CHECK_IN_RANGE (alnpe.c:76-81)
{
    const uint32_t r = a < b ? b - a : a - b;
    if (a > b || r < small) return -1;
    return r > large ? 1 : 0;
}

__global__ void __launch_bounds__(256)
k_pair(uint32_t n_pairs, uint32_t min_tlen, uint32_t max_tlen, uint32_t l_pac, const uint32_t *__restrict__ offs,
       salt_result_t *__restrict__ res, PePair *__restrict__ pairs, PeSwReq *__restrict__ req, uint32_t *__restrict__ pctl)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    salt_result_t *q0 = res + 2 * p, *q1 = q0 + 1;
    const uint32_t l0 = offs[2 * p + 1] - offs[2 * p], l1 = offs[2 * p + 2] - offs[2 * p + 1], l2 = l0 + l1;
    const uint32_t min_isize = min_tlen > l2 ? min_tlen - l2 : 0, max_isize = max_tlen > l2 ? max_tlen - l2 : 0;
    PePair pr; pr.n_req = 0; pr.req0 = 0; pr.rescued[0] = pr.rescued[1] = 0xFF;
    const bool m0 = q0->pos != 0xFFFFFFFFu, m1 = q1->pos != 0xFFFFFFFFu;
    uint32_t st[2], en[2]; uint8_t who[2], str[2], aw[2]; int nr = 0;
    auto add = [&](uint32_t s_, uint32_t e_, int mate, int strand, int aware) { st[nr] = s_; en[nr] = e_; who[nr] = (uint8_t)mate; str[nr] = (uint8_t)strand; aw[nr] = (uint8_t)aware; ++nr; };
    if (m0 && m1) {                                                          // pairing2 (alnpe.c:94-258)
        bool done = false;
        if (q0->strand == 0 && q1->strand == 1 && q0->pos < q1->pos) done = in_range(q0->pos + l0, q1->pos, min_isize, max_isize) == 0;
        else if (q1->strand == 0 && q0->strand == 1 && q1->pos < q0->pos) done = in_range(q1->pos + l1, q0->pos, min_isize, max_isize) == 0;
        if (!done) {
            uint32_t min_err = 0xFFFFFFFFu; salt_hit_t b0 = q0->hits[0][0], b1 = q1->hits[0][0];
            for (int pass = 0; pass < 2; ++pass) {                           // alternative hits only (alnpe.c:128-200)
                const salt_result_t *qf = pass == 0 ? q0 : q1, *qb = pass == 0 ? q1 : q0;
                const uint32_t nf = qf->n_hits[0], nb = qb->n_hits[1], lf = pass == 0 ? l0 : l1;
                if (!(nf > 0 && nb > 0)) continue;
                for (uint32_t i = 0; i < nf; ++i)
                    for (uint32_t jj = 0; jj < nb; ++jj) {
                        const int rg = in_range(qf->hits[0][i].pos + lf, qb->hits[1][jj].pos, min_isize, max_isize);
                        if (rg == 0) {
                            const uint32_t e = (uint32_t)qf->hits[0][i].n_diff + qb->hits[1][jj].n_diff;
                            if (e < min_err) { min_err = e; if (pass == 0) { b0 = qf->hits[0][i]; b1 = qb->hits[1][jj]; } else { b1 = qf->hits[0][i]; b0 = qb->hits[1][jj]; } }
                        } else if (rg == 1) break;
                    }
            }
            if (min_err != 0xFFFFFFFFu) {
                q0->pos = b0.pos; q0->strand = (uint8_t)b0.strand; q0->n_diff = b0.n_diff; q0->is_gap = b0.is_gap;
                q1->pos = b1.pos; q1->strand = (uint8_t)b1.strand; q1->n_diff = b1.n_diff; q1->is_gap = b1.is_gap;
            } else {                                                          // SNP-aware rescue, q0 then q1 as anchor (alnpe.c:213-252)
                uint32_t s_, e_;
                if (q0->strand == 0) { s_ = q0->pos + min_isize + l0; e_ = q0->pos + max_isize + l0 + l1; e_ = e_ >= l_pac ? l_pac : e_; add(s_, e_, 1, 1, 1); }
                else { s_ = q0->pos > max_isize + l1 ? q0->pos - max_isize - l1 : 0; e_ = q0->pos > min_isize ? q0->pos - min_isize : 0; e_ = e_ >= l_pac ? l_pac : e_; add(s_, e_, 1, 0, 1); }
                if (q1->strand == 0) { s_ = q1->pos + min_isize + l1; e_ = q1->pos + max_isize + l1 + l0; e_ = e_ >= l_pac ? l_pac : e_; add(s_, e_, 0, 1, 1); }
                else { s_ = q1->pos > max_isize + l0 ? q1->pos - max_isize - l0 : 0; e_ = q1->pos > min_isize ? q1->pos - min_isize : 0; e_ = e_ >= l_pac ? l_pac : e_; add(s_, e_, 0, 0, 1); }
            }
        }
    } else if (m0 || m1) {                                                   // pairing_singleton (alnpe.c:395-480): plain 2-bit SW
        const uint32_t lim = l_pac - 1; uint32_t s_, e_;
        if (m0) {
            if (q0->strand == 0) { s_ = q0->pos + min_isize + l0; s_ = s_ < lim ? s_ : lim; e_ = q0->pos + max_isize + l0 + l1; e_ = e_ < lim ? e_ : lim; add(s_, e_, 1, 1, 0); }
            else { s_ = q0->pos > max_isize + l1 ? q0->pos - max_isize - l1 : 0; s_ = s_ < lim ? s_ : lim; e_ = q0->pos > min_isize ? q0->pos - min_isize : 0; e_ = e_ < lim ? e_ : lim; add(s_, e_, 1, 0, 0); }
        }
        if (m1) {
            if (q1->strand == 0) { s_ = q1->pos + min_isize + l1; s_ = s_ < lim ? s_ : lim; e_ = q1->pos + max_isize + l1 + l0; e_ = e_ < lim ? e_ : lim; add(s_, e_, 0, 1, 0); }
            else { s_ = q1->pos > max_isize + l0 ? q1->pos - max_isize - l0 : 0; s_ = s_ < lim ? s_ : lim; e_ = q1->pos > min_isize ? q1->pos - min_isize : 0; e_ = e_ < lim ? e_ : lim; add(s_, e_, 0, 0, 0); }
        }
    }
    if (nr) {
        const uint32_t base = atomicAdd(&pctl[0], (uint32_t)nr);
        pr.n_req = (uint8_t)nr; pr.req0 = base;
        for (int k = 0; k < nr; ++k) { PeSwReq r; r.start = st[k]; r.end = en[k]; r.mate = 2 * p + who[k]; r.strand = str[k]; r.aware = aw[k]; r.pad = 0; req[base + k] = r; pr.rescued[k] = who[k]; }
    }
    pairs[p] = pr;
}

void launch_pair(uint32_t n_pairs, uint32_t min_tlen, uint32_t max_tlen, uint32_t l_pac, const uint32_t *offs, salt_result_t *res,
                 PePair *pairs, PeSwReq *req, uint32_t *pctl, hipStream_t st)
{
    if (n_pairs) hipLaunchKernelGGL(k_pair, dim3((n_pairs + 255) / 256), dim3(256), 0, st, n_pairs, min_tlen, max_tlen, l_pac, offs, res, pairs, req, pctl);
}

// Blocks per CU are bounded by the dynamic LDS (8 groups x sw_group_bytes), i.e. by the longest read of the batch:
// 150-bp mates fit 8 blocks per CU where 512-bp reads fit 4.
uint32_t sw_lds_bytes(uint32_t max_len) { const uint32_t seg = (max_len + 7) / 8; return 8u * (4u * seg * 16u + ((8u * seg + 15u) & ~15u)); }
uint32_t sw_blocks_per_cu(uint32_t max_len)
{
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_sw, 64, sw_lds_bytes(max_len)) != hipSuccess || n < 1) n = 1;
    return (uint32_t)(n > (int)SW_MAX_BLOCKS_PER_CU ? (int)SW_MAX_BLOCKS_PER_CU : n);
}
// Scratch geometry of one k_sw launch: the per-column maxima cover the longest window (max_window columns), the direction
// bytes the widest band the row buffers hold; the grid is cut back before the groups' scratch passes SW_SCRATCH_TOTAL.
SwGeom sw_geom(uint32_t max_len, uint64_t max_window, uint32_t cus)
{
    SwGeom g;
    const uint64_t mc = (2 * max_window + 255) & ~255ull;
    const uint64_t dir = (3ull * max_len * (SW_BAND_W - 3) + 8 + 255) & ~255ull;
    const uint64_t grp = mc + 3ull * SW_BAND_W * 4 + dir;
    g.maxcol_bytes = (uint32_t)mc; g.group_bytes = (uint32_t)grp;
    uint64_t blocks = (uint64_t)cus * sw_blocks_per_cu(max_len);
    const uint64_t cap = SW_SCRATCH_TOTAL / (8 * grp);
    if (blocks > cap) blocks = cap;
    g.n_blocks = (uint32_t)(blocks ? blocks : 1);
    return g;
}

void launch_sw(const IndexView &ix, const uint8_t *pac, const uint8_t *seqs, const uint32_t *offs, const PeSwReq *req, const uint32_t *pctl,
               PeSwRes *res, uint32_t *head, uint32_t *overflow, uint8_t *scratch, SwGeom g, uint32_t max_len, hipStream_t st)
{
    const uint32_t seg = (max_len + 7) / 8;
    hipLaunchKernelGGL(k_sw, dim3(g.n_blocks), dim3(64), sw_lds_bytes(max_len), st, ix, pac, seqs, offs, req, pctl, res, head, overflow, scratch,
                       g.maxcol_bytes, g.group_bytes, seg, (getenv("SALT_GPU_SW_SKIP_TB") && atoi(getenv("SALT_GPU_SW_SKIP_TB"))) ? 1 : 0);
}

} // namespace salt
