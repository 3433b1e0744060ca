// salt_amd/csrc/salt_pe.hip -- paired-end stage kernels (gfx950): pairing decisions and Smith-Waterman mate rescue.
//
//   k_pair   one lane per pair: pairing2 / pairing_singleton up to the point where they call Smith-Waterman
//            (Align_src/alnpe.c:94-258, 395-480): accept the primaries if properly oriented and spaced, else the best
//            pair among the alternative hits, else queue up to two rescue windows (tried in the reference's order).
//   k_swf    eight lanes per PAIR of rescue requests = the eight 16-bit lanes of the reference's SSE2 word kernel
//            (sw_sse2_word, ssw.c:371-547), two requests in the halves of every register, emulated operation by operation because
//            its "lazy F" pass is not the textbook recurrence (a deletion may not follow an insertion across a stripe
//            boundary): the forward pass and the second-best score outside the mask (ssw_align, ssw.c:771-816).  k_swf1: the
//            same for requests that cannot be paired, one per group.
//   k_swr    the reverse pass from the end point (ssw.c:817-830), eight lanes per request.
//   k_swtb   the banded traceback (banded_sw, ssw.c:549-727), eight lanes per alignment.
// The results are applied to the mates by k_pe_final (salt_align.hip), which also writes the CIGARs.
#include "salt_device.h"
#include <stdlib.h>
#include "salt_kernels.h"

namespace salt {

// score of a reference symbol against a read base code (0..3, 4 = N) as ssw_init builds its profile from
// score_mat2 indexed [ref*16 + (1<<code)] (aware, alnpe.c:58-73,283) or score_mat [ref*5 + code] (alnpe.c:52-56)
// aware = 2: polish's own matrix (+2 / -2, N scores 0; Polish_src/polish.c:48-52) over the 2-bit genome
__device__ __forceinline__ int sw_score(int aware, uint32_t ref, uint32_t code)
{
    if (aware == 2) return (ref > 3 || code > 3) ? 0 : (ref == code ? 2 : -2);
    if (aware) {
        if (code > 3) return -3;                              // 1<<4 = 16 indexes column 0 of the next row: always -3
        const uint32_t bit = 1u << code;                      // only rows 1,2,4,8 of the (transposed) matrix are non-trivial
        return ((ref == 1 || ref == 2 || ref == 4 || ref == 8) && (ref & bit)) ? 1 : -3;
    }
    if (ref > 3 || code > 3) return -1;
    return ref == code ? 1 : -3;
}

// the same value without a branch (the traceback calls it once per cell of a row, eight alignments side by side in one wave: the
// short-circuit form above compiles to a dozen divergent branches per row)
__device__ __forceinline__ int sw_score_flat(int aware, uint32_t ref, uint32_t code)
{
    const bool eq = ref == code, bases = (ref | code) < 4u;
    const int s0 = bases ? (eq ? 1 : -3) : -1, s2 = bases ? (eq ? 2 : -2) : 0;
    const bool one_allele = __popc(ref) == 1, listed = ((ref >> (code & 3u)) & 1u) != 0;
    const int s1 = ((code < 4u) & one_allele & listed) ? 1 : -3;
    return aware == 2 ? s2 : aware ? s1 : s0;
}

__device__ __forceinline__ int sat16(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }
__device__ __forceinline__ int subu16(int a, int b) { unsigned x = (unsigned)a & 0xFFFFu; return (int)(short)(x > (unsigned)b ? x - (unsigned)b : 0u); }

struct SwLds {                       // per 8-lane group: views into the block's dynamic LDS, seg = ceil(max read length / 8)
    short *H[2], *E, *Hmax;          // [seg][8]
    uint8_t *read;                   // [8 * seg]
};
__device__ __forceinline__ uint32_t sw_group_bytes(uint32_t seg) { return 4u * seg * 16u + ((8u * seg + 15u) & ~15u); }
// where the groups' window words begin in k_swf's / k_swr's LDS: behind the rows (LDS variant, variant 0) or the reads and H-at-best rows
__host__ __device__ __forceinline__ uint32_t sw_lds_words_at(uint32_t seg, uint32_t variant, bool fwd)
{
    const uint32_t rd = (8u * seg + 15u) & ~15u;
    return variant ? (fwd ? 2u : 1u) * (8u * rd + 8u * variant * 8u * 2u) : 8u * (4u * seg * 16u + rd);
}


__device__ __forceinline__ uint32_t ref_symbol(const IndexView &ix, const uint8_t *pac, int aware, uint32_t p)
{
    if (aware == 1) return (ix.ref[p >> 3] >> (4 * (p & 7u))) & 15u;
    return (pac[p >> 2] >> ((~p & 3u) << 1)) & 3u;
}

// one striped pass; all 8 lanes of the group call it together.  rd(q): read code at position q of this pass.
template <class ReadAt>
__device__ void sw_word_pass(const IndexView &ix, const uint8_t *pac, int aware, SwLds &s, uint32_t ref0, int ref_dir, int refLen,
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

// cross-lane moves inside an 8-lane group as DPP modifiers (no LDS crossbar round trip): shift towards higher lanes within the
// 16-lane row (the callers zero the lanes that would read across the group's boundary), and an 8-lane max by two quad
// permutes and the half-row mirror
template <int N> __device__ __forceinline__ int dpp_row_shr(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x110 + N, 0xF, 0xF, true); }
__device__ __forceinline__ int dpp_max8(int x)
{
    int t = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true); x = x > t ? x : t;        // quad_perm [1,0,3,2]
    t = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true); x = x > t ? x : t;            // quad_perm [2,3,0,1]
    t = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true); return x > t ? x : t;        // row_half_mirror
}

// The same pass with the stripe rows in REGISTERS (SEG = compile-time bound on segLen, every loop fully unrolled): the LDS
// version spends a column in ~5 dependent LDS round trips per stripe with two waves per SIMD to hide them; here a column is
// ~12 VALU instructions per stripe and nothing else.  One buffer suffices for H: the old H[j] is read (it feeds stripe j+1)
// right before the new one overwrites it.  Same operations in the same order as above, so the same lazy-F behaviour.
//
// The window's symbols reach the lanes without a memory wait per column: the eight lanes of a group hold the eight 32-bit words of the
// window's current block (8 allele masks of the mixRef or 16 bases of the 2-bit genome per word; a block = 7 words' worth of columns, so
// that an unaligned block still touches at most 8 words) and drop them into the group's LDS, from where a column's word is read a column
// ahead of its use: one exposed load per block of 56 or 112 columns and no memory wait inside a column (a register that a load MAY have
// written this iteration costs a wait for every store in flight at its first use, each column).  (Loading "the next word" into a register
// rotation, as this pass first did, compiled into a memory wait at the top of every column: the copies of the rotation need the load.)
// The column maxima leave the same way: lane c % 8 keeps column c's and the group stores eight at a time.
__device__ __forceinline__ uint32_t sat_add_u32(uint32_t a, uint32_t b) { return __builtin_elementwise_add_sat(a, b); }
__device__ __forceinline__ uint32_t sat_sub_u32(uint32_t a, uint32_t b) { return __builtin_elementwise_sub_sat(a, b); }
struct RefStream {
    uint32_t *lw;                      // the current block's eight words (the group's LDS)
    uint32_t wb;                       // word index of lw[0]
    int cnext;                         // pass column at which the next block begins
};
// the block that begins at pass column c0 (forward: lw[0] is the lowest word, reverse: the highest); a block beyond the window (the
// shorter window of a pair) loads some valid word and is never used
__device__ __forceinline__ void ref_stream_turn(RefStream &r, const uint32_t *words, uint32_t ref_len, uint32_t ws, uint32_t ref0, int n, int dir, int c0, int lane)
{
    const uint32_t last = ref_len - 1u;
    uint32_t w;
    if (dir) { r.wb = sat_sub_u32(ref0 + (uint32_t)(n - 1), (uint32_t)c0) >> ws; w = sat_sub_u32(r.wb, (uint32_t)lane); }
    else { const uint32_t p = sat_add_u32(ref0, (uint32_t)c0); r.wb = (p < last ? p : last) >> ws; w = r.wb + (uint32_t)lane; w = w < (last >> ws) ? w : (last >> ws); }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();      // the word asked for a column ahead has been read
    r.lw[lane] = words[w];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    r.cnext = c0 + (int)(7u << ws);
}
// the word holding the symbol of pass column c (window position ref0 + c forward, ref0 + n - 1 - c reverse), out of the current block;
// asked for one column ahead of its use
__device__ __forceinline__ uint32_t ref_stream_word(const RefStream &r, uint32_t ws, uint32_t ref0, int n, int dir, int c)
{
    const uint32_t p = dir ? ref0 + (uint32_t)(n - 1 - c) : ref0 + (uint32_t)c;
    const uint32_t k = dir ? r.wb - (p >> ws) : (p >> ws) - r.wb;
    return r.lw[k & 7u];
}
__device__ __forceinline__ uint32_t ref_word_symbol(uint32_t w, bool masks, uint32_t ref0, int n, int dir, int c)
{
    const uint32_t p = dir ? ref0 + (uint32_t)(n - 1 - c) : ref0 + (uint32_t)c;
    // masks: 8 per word, LSB first (metaref.c:54-56); 2-bit genome: 4 bases per byte, the first in the byte's top bits (bntseq.c:88-139)
    const uint32_t sh = masks ? 4u * (p & 7u) : 8u * ((p >> 2) & 3u) + ((~p & 3u) << 1);
    return (w >> sh) & (masks ? 15u : 3u);
}
// the six 4-bit profile fields of a column: sw_score for read codes 0..3, N and "past the read", + bias (3, polish's matrix 2)
__device__ __forceinline__ uint32_t sw_prof_fields(bool masks, bool polish, uint32_t sym)
{
    const bool single = sym != 0u && (sym & (sym - 1u)) == 0u;                    // SNP-aware: exactly one allele (alnpe.c:58-73 as SSW indexes it)
    const uint32_t idx = masks ? (uint32_t)__ffs((int)sym) - 1u : sym;
    const uint32_t hit = (masks ? single : true) ? 4u << ((4u * idx) & 31u) : 0u;
    return ((polish ? 2u : 3u) << 20) | (masks ? 0u : 2u << 16) | hit;
}

// The pass is cut into begin / column / end so that the groups of a wave can be at different columns of different requests (k_swr).
template <int SEG> struct Pass1 {                                // one request's pass between two columns
    int H[SEG], E[SEG]; uint32_t shp[(SEG + 5) / 6];             // shp: the 5-bit profile shifts of six stripes per register
    RefStream rs; uint32_t w_ahead;                              // the next column's word
    int last, max, end_ref, vMaxScore, vMaxMark;
};
// Every H, E and F here lies in [0, 32767] (scores are at most the read length), where the SSE2 operations reduce to plain
// integer ones: adds_epi16(vH, profile) = vH + profile, subs_epu16(x, g) = max(x - g, 0).  The profile of a column is six
// 4-bit fields (value + 3 per read code 0..3, N, "past the read"), so a cell's lookup is one bit-field extract.
// H at the best column so far (what the end point is read from, ssw.c:504-512) is written a few dozen times per pass and read once:
// it lives in the group's LDS (hm[j * 8 + lane]).
template <int SEG, class ReadAt>
__device__ __forceinline__ void pass1_begin(Pass1<SEG> &f, int readLen, ReadAt rd, short *hm, uint32_t *lw, const IndexView &ix, const uint8_t *pac, int aware,
                                            uint32_t ref0, int n, int dir, int lane)
{
    const int segLen = (readLen + 7) / 8;
#pragma unroll
    for (int k = 0; k < (SEG + 5) / 6; ++k) f.shp[k] = 0;
#pragma unroll
    for (int j = 0; j < SEG; ++j) {
        f.H[j] = 0; f.E[j] = 0; hm[j * 8 + lane] = 0;
        const int q = j + lane * segLen;
        const uint32_t code = (j < segLen && q < readLen) ? rd(q) : 5u;      // 5: past the read (profile 0)
        f.shp[j / 6] |= (4u * (code > 5u ? 4u : code)) << (5 * (j % 6));
    }
    const bool masks = aware == 1;                               // 4-bit allele masks (mixRef) or the 2-bit genome
    const uint32_t ws = masks ? 3u : 4u;
    f.rs.lw = lw;
    ref_stream_turn(f.rs, masks ? ix.ref : reinterpret_cast<const uint32_t *>(pac), ix.ref_len, ws, ref0, n, dir, 0, lane);
    f.w_ahead = ref_stream_word(f.rs, ws, ref0, n, dir, 0);
    f.last = 0; f.max = 0; f.end_ref = 0; f.vMaxScore = 0; f.vMaxMark = 0;
}
// pass column c (window position i = c forward, n - 1 - c reverse); returns the column's maximum
template <int SEG>
__device__ __forceinline__ int pass1_column(Pass1<SEG> &f, int c, int readLen, short *hm, const IndexView &ix, const uint8_t *pac, int aware,
                                            uint32_t ref0, int n, int dir, int lane)
{
    const uint64_t gmask = 0xFFull << (threadIdx.x & 56u);
    const int segLen = (readLen + 7) / 8, go = 3, ge = 1;      // aln.h:137-138
    const bool masks = aware == 1, polish = aware == 2;
    const uint32_t ws = masks ? 3u : 4u;
    const int bias = polish ? 2 : 3;                             // the smallest score of the matrix in use, negated
    const int i = dir ? n - 1 - c : c;
    const uint32_t prof4 = sw_prof_fields(masks, polish, ref_word_symbol(f.w_ahead, masks, ref0, n, dir, c));
    if (c + 1 == f.rs.cnext) ref_stream_turn(f.rs, masks ? ix.ref : reinterpret_cast<const uint32_t *>(pac), ix.ref_len, ws, ref0, n, dir, c + 1, lane);
    f.w_ahead = ref_stream_word(f.rs, ws, ref0, n, dir, c + 1);
    int vF = 0, vMaxColumn = 0;
    int vH = dpp_row_shr<1>(f.last);                             // H of the last stripe, as the column before left it
    if (lane == 0) vH = 0;
    // (keeps the stripes' shifts packed: left alone, the compiler extracts all of them ahead of the loop into registers it does
    // not have and reloads them from scratch memory stripe by stripe)
#pragma unroll
    for (int k = 0; k < (SEG + 5) / 6; ++k) asm volatile("" : "+v"(f.shp[k]));
#pragma unroll
    for (int j = 0; j < SEG; ++j) {
        if (j < segLen) {
            int h = vH + (int)((prof4 >> ((f.shp[j / 6] >> (5 * (j % 6))) & 31u)) & 15u) - bias;
            int e = f.E[j];
            h = h > e ? h : e; h = h > vF ? h : vF;
            vMaxColumn = vMaxColumn > h ? vMaxColumn : h;
            vH = f.H[j];                                         // the previous column's value: input of stripe j+1
            f.H[j] = h;
            h -= go; h = h > 0 ? h : 0;
            e -= ge; e = e > h ? e : h; f.E[j] = e;
            vF -= ge; vF = vF > h ? vF : h;
        }
    }
    // lazy F (ssw.c:487-497) in closed form.  The reference shifts vF one lane up and sweeps the stripes (H = max(H, vF); vF -= ge),
    // up to 8 times, leaving early once no lane has vF - ge > H - go.  The sweeps only EXTEND gaps (no new F is opened), so what
    // stripe j of lane l can receive is max over k of (vF at the end of lane l-1-k's column) - k * segLen - j, floored at 0 by the
    // saturating subtraction; and after the reference's early exit no H changes any more (from there on the lane's own F chain of
    // the main loop, >= H[j] - go - ..., dominates the carried one), so taking the full maximum gives the same H.  A max-plus scan
    // over the 8 lanes and one pass over the stripes replace the sweeps (which the 8 requests of a wave would each stretch to
    // the longest of them).  E and the column maximum are not touched by the lazy pass there either.
    {
        int cf = dpp_row_shr<1>(vF);
        if (lane == 0) cf = 0;
        const int seg_ge = segLen * ge;
        { int t = dpp_row_shr<1>(cf); t = lane >= 1 ? t - seg_ge : 0; t = t > 0 ? t : 0; cf = cf > t ? cf : t; }
        { int t = dpp_row_shr<2>(cf); t = lane >= 2 ? t - 2 * seg_ge : 0; t = t > 0 ? t : 0; cf = cf > t ? cf : t; }
        { int t = dpp_row_shr<4>(cf); t = lane >= 4 ? t - 4 * seg_ge : 0; t = t > 0 ? t : 0; cf = cf > t ? cf : t; }
#pragma unroll
        for (int j = 0; j < SEG; ++j) {
            if (j < segLen) { int fj = cf - j * ge; fj = fj > 0 ? fj : 0; f.H[j] = f.H[j] > fj ? f.H[j] : fj; f.last = f.H[j]; }
        }
    }
    f.vMaxScore = f.vMaxScore > vMaxColumn ? f.vMaxScore : vMaxColumn;
    if (__ballot(f.vMaxMark != f.vMaxScore) & gmask) {
        f.vMaxMark = f.vMaxScore;
        const int temp = dpp_max8(f.vMaxScore);
        if (temp > f.max) {
            f.max = temp; f.end_ref = i;
#pragma unroll
            for (int j = 0; j < SEG; ++j) hm[j * 8 + lane] = (short)f.H[j];
        }
    }
    return dpp_max8(vMaxColumn);
}
// the smallest read position holding the maximum in the best column (ssw.c:504-512)
template <int SEG>
__device__ __forceinline__ int pass1_end(const Pass1<SEG> &f, int readLen, const short *hm, int lane)
{
    const int segLen = (readLen + 7) / 8;
    int end_read = readLen - 1;
#pragma unroll
    for (int j = 0; j < SEG; ++j) if (j < segLen && (int)hm[j * 8 + lane] == f.max) { int t = j + lane * segLen; if (t < end_read) end_read = t; }
    for (int o = 1; o < 8; o <<= 1) { int t = __shfl_xor(end_read, o, 8); end_read = end_read < t ? end_read : t; }
    return end_read;
}
// a whole pass of one request, the groups of the wave in step
template <int SEG, class ReadAt>
__device__ __forceinline__ void sw_word_pass_reg(const IndexView &ix, const uint8_t *pac, int aware, uint32_t ref0, int ref_dir, int refLen,
                                                 int readLen, ReadAt rd, int terminate, uint16_t *maxColumn, short *hm, uint32_t *lw,
                                                 int &out_max, int &out_end_ref, int &out_end_read, uint32_t *dbg_cols = nullptr)
{
    const int lane = (int)(threadIdx.x & 7u);
    Pass1<SEG> f;
    pass1_begin<SEG>(f, readLen, rd, hm, lw, ix, pac, aware, ref0, refLen, ref_dir, lane);
    int mc_keep = 0, c = 0;                                      // lane c % 8 keeps column c's maximum until the group stores eight
    for (; c < refLen; ++c) {
        const int mc = pass1_column<SEG>(f, c, readLen, hm, ix, pac, aware, ref0, refLen, ref_dir, lane);
        if (maxColumn) {                                         // (forward passes only: column c is window position c)
            if (lane == (c & 7)) mc_keep = mc;
            if ((c & 7) == 7) maxColumn[c - 7 + lane] = (uint16_t)mc_keep;
        }
        if (mc == terminate) { ++c; break; }
    }
    if (maxColumn && (c & 7) && lane < (c & 7)) maxColumn[(c & ~7) + lane] = (uint16_t)mc_keep;
    out_max = f.max; out_end_ref = f.end_ref; out_end_read = pass1_end<SEG>(f, readLen, hm, lane);
    if (dbg_cols && lane == 0) { atomicAdd(dbg_cols, (uint32_t)c); atomicAdd(dbg_cols + 1, (uint32_t)c * (uint32_t)((readLen + 7) / 8)); }
}

// The forward pass of TWO requests in one group: each lane's H, E, F hold request A in the low and request B in the high 16 bits of a
// register and the column's operations are the packed 16-bit ones (v_pk_add_u16, v_pk_max_i16, v_pk_sub_u16 clamp = subs_epu16), so a
// column of both costs what a column of one did, apart from the two profile look-ups.  The requests of the whole LAUNCH that take this
// path have the same read length and scoring (the stripes then line up and their count is a scalar; paired-end mates do); their windows
// may differ in length -- the shorter one's half keeps running with its maximum, end point and column maxima frozen.  Same operations per
// half in the same order as sw_word_pass_reg, so the same values.  The pass is cut into begin / column / end so that the groups of a wave
// can be at different columns of different pairs (k_swf).
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b)); }
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b))); }
__device__ __forceinline__ uint32_t pk_subs(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b))); }
__device__ __forceinline__ uint32_t dpp_pk_max8(uint32_t x)
{
    uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true); x = pk_max(x, t);
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true); x = pk_max(x, t);
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, true); return pk_max(x, t);
}
// The queues of the Smith-Waterman kernels.  All pulls of a launch on one counter are served one after the other (~14 ns each): when
// every group of the device asks for its first item at the same moment the last one has waited a third of a millisecond, and again when
// they all come back to find the queue empty.  So the FIRST item of a puller is its own index (no atomic), the counter hands out what
// lies behind those, and a puller looks at the counter before it adds to it (the look is a plain load: no queue).
// n_first: items the static first round covers (pullers x step).  Returns the item index, >= n_items when there is none.
__device__ __forceinline__ uint32_t sw_pull(uint32_t *head, const uint32_t step, const uint32_t n_first, const uint32_t n_items)
{
    if (n_first >= n_items || n_first + __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_items) return 0xFFFFFFFFu;
    return n_first + atomicAdd(head, step);
}
template <int SEG> struct Fwd2 {                                 // a pair's forward pass between two columns
    uint32_t H[SEG], E[SEG], shp[(SEG + 2) / 3];                 // shp: per stripe the 5-bit profile shifts of A and of B, three stripes per register
    RefStream rsA, rsB;                                          // (both streams turn their blocks at the same columns)
    uint32_t wA, wB;                                             // the next column's words
    uint32_t last, mc_keep, vMaxScore, vMaxMark;
    int maxA, maxB, erA, erB;
};
// segLen: the same for every group of the wave (a scalar); awA / awB: the scoring of the two requests (0 plain, 1 SNP-aware, 2 polish matrix)
template <int SEG>
__device__ __forceinline__ void fwd2_begin(Fwd2<SEG> &f, int segLen, int readLen, const uint8_t *rdA, const uint8_t *rdB, short *hmA, short *hmB, uint32_t *lw,
                                           const IndexView &ix, const uint8_t *pac, int awA, int awB, uint32_t ref0A, uint32_t ref0B, int nCols, int lane)
{
    const uint32_t *wordsA = awA == 1 ? ix.ref : reinterpret_cast<const uint32_t *>(pac), *wordsB = awB == 1 ? ix.ref : reinterpret_cast<const uint32_t *>(pac);
    const uint32_t wsA = awA == 1 ? 3u : 4u, wsB = awB == 1 ? 3u : 4u, ref_len = ix.ref_len;
#pragma unroll
    for (int k = 0; k < (SEG + 2) / 3; ++k) f.shp[k] = 0;
#pragma unroll
    for (int j = 0; j < SEG; ++j) {
        f.H[j] = 0; f.E[j] = 0; hmA[j * 8 + lane] = 0; hmB[j * 8 + lane] = 0;
        const int q = j + lane * segLen;
        const bool in = j < segLen && q < readLen;
        const uint32_t ca = in ? rdA[q] : 5u, cb = in ? rdB[q] : 5u;
        f.shp[j / 3] |= ((4u * (ca > 5u ? 4u : ca)) | (4u * (cb > 5u ? 4u : cb)) << 5) << (10 * (j % 3));
    }
    f.rsA.lw = lw; f.rsB.lw = lw + 8;
    ref_stream_turn(f.rsA, wordsA, ref_len, wsA, ref0A, nCols, 0, 0, lane); ref_stream_turn(f.rsB, wordsB, ref_len, wsB, ref0B, nCols, 0, 0, lane);
    f.wA = ref_stream_word(f.rsA, wsA, ref0A, nCols, 0, 0); f.wB = ref_stream_word(f.rsB, wsB, ref0B, nCols, 0, 0);
    f.last = 0; f.mc_keep = 0; f.vMaxScore = 0; f.vMaxMark = 0; f.maxA = 0; f.maxB = 0; f.erA = 0; f.erB = 0;
}
template <int SEG>
__device__ __forceinline__ void fwd2_column(Fwd2<SEG> &f, int i, int segLen, const IndexView &ix, const uint8_t *pac, int awA, int awB,
                                            uint32_t ref0A, uint32_t ref0B, int refLenA, int refLenB, int nCols, uint16_t *mcA, uint16_t *mcB,
                                            short *hmA, short *hmB, int lane)
{
    const uint64_t gmask = 0xFFull << (threadIdx.x & 56u);
    const uint32_t go2 = 0x00030003u, ge2 = 0x00010001u;         // aln.h:137-138, both halves
    const bool masksA = awA == 1, masksB = awB == 1, polishA = awA == 2, polishB = awB == 2;
    const uint32_t wsA = masksA ? 3u : 4u, wsB = masksB ? 3u : 4u;
    const uint32_t negb2 = (uint32_t)(-(polishA ? 2 : 3) & 0xFFFF) | (uint32_t)(-(polishB ? 2 : 3) & 0xFFFF) << 16;
    const uint32_t seg_ge = (uint32_t)segLen * 0x00010001u;
    // (past the end of the shorter window its half sees whatever follows in the genome: nothing of it is kept)
    const uint32_t profA = sw_prof_fields(masksA, polishA, ref_word_symbol(f.wA, masksA, ref0A, nCols, 0, i));
    const uint32_t profB = sw_prof_fields(masksB, polishB, ref_word_symbol(f.wB, masksB, ref0B, nCols, 0, i));
    if (i + 1 == f.rsA.cnext || i + 1 == f.rsB.cnext) {           // (a block of the 2-bit genome is twice as many columns: turning it early is harmless)
        ref_stream_turn(f.rsA, masksA ? ix.ref : reinterpret_cast<const uint32_t *>(pac), ix.ref_len, wsA, ref0A, nCols, 0, i + 1, lane);
        ref_stream_turn(f.rsB, masksB ? ix.ref : reinterpret_cast<const uint32_t *>(pac), ix.ref_len, wsB, ref0B, nCols, 0, i + 1, lane);
    }
    f.wA = ref_stream_word(f.rsA, wsA, ref0A, nCols, 0, i + 1); f.wB = ref_stream_word(f.rsB, wsB, ref0B, nCols, 0, i + 1);
    uint32_t vF = 0, vMaxColumn = 0;
    uint32_t vH = (uint32_t)dpp_row_shr<1>((int)f.last);
    if (lane == 0) vH = 0;
#pragma unroll
    for (int k = 0; k < (SEG + 2) / 3; ++k) asm volatile("" : "+v"(f.shp[k]));          // (as in sw_word_pass_reg)
#pragma unroll
    for (int j = 0; j < SEG; ++j) {
        if (j < segLen) {
            const uint32_t t = f.shp[j / 3] >> (10 * (j % 3));
            const uint32_t p = __builtin_amdgcn_ubfe(profA, t, 4u) | (__builtin_amdgcn_ubfe(profB, t >> 5, 4u) << 16);
            uint32_t h = pk_add(pk_add(vH, p), negb2);
            uint32_t e = f.E[j];
            h = pk_max(h, e); h = pk_max(h, vF);
            vMaxColumn = pk_max(vMaxColumn, h);
            vH = f.H[j];
            f.H[j] = h;
            h = pk_subs(h, go2);
            e = pk_subs(e, ge2); e = pk_max(e, h); f.E[j] = e;
            vF = pk_subs(vF, ge2); vF = pk_max(vF, h);
        }
    }
    {                                                            // lazy F in closed form (see sw_word_pass_reg), both halves
        uint32_t cf = (uint32_t)dpp_row_shr<1>((int)vF);
        if (lane == 0) cf = 0;
        { uint32_t t = (uint32_t)dpp_row_shr<1>((int)cf); t = lane >= 1 ? pk_subs(t, seg_ge) : 0u; cf = pk_max(cf, t); }
        { uint32_t t = (uint32_t)dpp_row_shr<2>((int)cf); t = lane >= 2 ? pk_subs(t, 2u * seg_ge) : 0u; cf = pk_max(cf, t); }
        { uint32_t t = (uint32_t)dpp_row_shr<4>((int)cf); t = lane >= 4 ? pk_subs(t, 4u * seg_ge) : 0u; cf = pk_max(cf, t); }
#pragma unroll
        for (int j = 0; j < SEG; ++j) {
            if (j < segLen) { f.H[j] = pk_max(f.H[j], pk_subs(cf, (uint32_t)j * 0x00010001u)); f.last = f.H[j]; }
        }
    }
    f.vMaxScore = pk_max(f.vMaxScore, vMaxColumn);
    if (__ballot(f.vMaxMark != f.vMaxScore) & gmask) {
        f.vMaxMark = f.vMaxScore;
        const uint32_t temp = dpp_pk_max8(f.vMaxScore);
        const int tA = (int)(temp & 0xFFFFu), tB = (int)(temp >> 16);
        if (tA > f.maxA && i < refLenA) {
            f.maxA = tA; f.erA = i;
#pragma unroll
            for (int j = 0; j < SEG; ++j) hmA[j * 8 + lane] = (short)(f.H[j] & 0xFFFFu);
        }
        if (tB > f.maxB && i < refLenB) {
            f.maxB = tB; f.erB = i;
#pragma unroll
            for (int j = 0; j < SEG; ++j) hmB[j * 8 + lane] = (short)(f.H[j] >> 16);
        }
    }
    const uint32_t mc = dpp_pk_max8(vMaxColumn);
    if (lane == (i & 7)) f.mc_keep = mc;
    if ((i & 7) == 7) {
        const int t = i - 7 + lane;
        if (t < refLenA) mcA[t] = (uint16_t)(f.mc_keep & 0xFFFFu);
        if (t < refLenB) mcB[t] = (uint16_t)(f.mc_keep >> 16);
    }
}
// after the last column: the column maxima still in the lanes, and the smallest read position holding each maximum (ssw.c:504-512)
template <int SEG>
__device__ __forceinline__ void fwd2_end(const Fwd2<SEG> &f, int segLen, int readLen, int refLenA, int refLenB, int nCols, uint16_t *mcA, uint16_t *mcB,
                                         const short *hmA, const short *hmB, int lane, int &endReadA, int &endReadB)
{
    if ((nCols & 7) && lane < (nCols & 7)) {
        const int t = (nCols & ~7) + lane;
        if (t < refLenA) mcA[t] = (uint16_t)(f.mc_keep & 0xFFFFu);
        if (t < refLenB) mcB[t] = (uint16_t)(f.mc_keep >> 16);
    }
    int erdA = readLen - 1, erdB = readLen - 1;
#pragma unroll
    for (int j = 0; j < SEG; ++j) if (j < segLen) {
        const int t = j + lane * segLen;
        if ((int)hmA[j * 8 + lane] == f.maxA && t < erdA) erdA = t;
        if ((int)hmB[j * 8 + lane] == f.maxB && t < erdB) erdB = t;
    }
    for (int o = 1; o < 8; o <<= 1) {
        int t = __shfl_xor(erdA, o, 8); erdA = erdA < t ? erdA : t;
        t = __shfl_xor(erdB, o, 8); erdB = erdB < t ? erdB : t;
    }
    endReadA = erdA; endReadB = erdB;
}

// ---------------------------------------------------------------------------------------------
// The banded traceback of ssw_align (banded_sw, ssw.c:549-727), re-designed for eight lanes per alignment (k_swtb below).
//
// What the reference computes: a banded DP over the slice [ref_begin, ref_end] x [read_begin, read_end] the two striped passes found,
// row by row (i = read position), the band [i - w, i + w] of every row kept in arrays indexed from the band's left end, three
// direction codes per cell; w doubles until the best cell reaches the alignment's score; then a walk back from the last cell.
// Here a row is ONE step of the group: lane t takes cell beg + t (rows wider than eight cells go in chunks with a carry), E comes from
// the row above element-wise, and the in-row recurrence F[j] = max(H[j-1] - go, F[j-1] - ge) is a max-plus prefix scan over the lanes
// (DPP row shifts): opening a gap from an H that is itself an F never beats extending that F (go > ge), so
//     F[j] = max( -(j - beg + 1) ge ,  max_{j' < j} ( A[j'] - go - (j - 1 - j') ge ) ),   A = max(E+, diagonal)  (H without its F term)
// gives the same F -- and the same "opened or extended" direction bit, taken from the finished neighbours (H[j-1] - go > F[j-1] - ge).
// The three direction codes of a cell (ssw.c: 1 diagonal, 2 / 3 from E extended / opened, 4 / 5 from F extended / opened) are one
// nibble: which of {diagonal, E, F} made H, E's bit, F's bit -- a byte per cell in the cell order of the reference's direction
// array, in LDS for the bands mate rescue produces (a few cells wide), in global scratch beyond.  Kept exactly: the band arrays with
// their indices (rows shifted by one once i > w), the cells the reference zeroes at the band's right edge (also when that edge is
// the slice's end and the cell above is real), band doubling with the running maximum, the walk and its run-length bookkeeping.
// ---------------------------------------------------------------------------------------------
struct TbGeom { uint32_t read_b, ref_b, row_w, dir_b, group_b; };       // per-group LDS: read bases | reference symbols | 3 rows x row_w ints | direction bytes
static constexpr int TB_NEG = -(1 << 28);

// one pass at band half-width bw; returns the largest H of the pass (all 8 lanes).  RL / DL: rows / direction bytes in LDS (else global)
template <bool RL, bool DL>
__device__ __forceinline__ int tb_band_pass(int32_t *__restrict__ hb, int32_t *__restrict__ eb, int32_t *__restrict__ hc, uint8_t *__restrict__ dir,
                                            const uint8_t *rd, const uint8_t *rf, const IndexView &ix, const uint8_t *pac, const int aware,
                                            const uint32_t ref0, const int refLen, const int readLen, const int bw)
{
    const int lane = (int)(threadIdx.x & 7u);
    const int go = 3, ge = 1;                                                   // aln.h:137-138
    const int width = 2 * bw + 3, width_d = 2 * bw + 1;
    auto gsync = [&]() {                                                        // the group's lanes hand rows to each other
        if (RL && DL) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }       // global rows: stores drained before the loads
    };
    for (int t = 1 + lane; t < width - 1; t += 8) hb[t] = 0;
    int mx = 0;
    for (int i = 0; i < readLen; ++i) {
        const int x = i - bw > 0 ? i - bw : 0, xp = i - 1 - bw > 0 ? i - 1 - bw : 0;        // left ends of this row's and the upper row's band
        const int beg = x, end = refLen - 1 < i + bw ? refLen - 1 : i + bw;
        const int edge = end + 1 < width - 1 ? end + 1 : width - 1;
        gsync();
        if (lane == 0) { hb[0] = 0; eb[0] = 0; hb[edge] = 0; eb[edge] = 0; hc[0] = 0; }
        gsync();
        const uint32_t rc = rd[i];
        uint8_t *drow = dir + (size_t)i * (size_t)width_d;
        const bool one = end - beg < 8;                                          // the whole row in one step: no copy through hc
        int carry = TB_NEG, prevH = 0, prevF = 0;                                // cell beg - 1: hc[0] = 0, f = 0
        for (int c0 = beg; c0 <= end; c0 += 8) {
            const int j = c0 + lane;
            const bool act = j <= end;
            const int u = j - x + 1, e = j - xp + 1;
            int hb_e = 0, eb_e = 0, hb_d = 0; uint32_t sym = 0;
            if (act) { hb_e = hb[e]; eb_e = eb[e]; hb_d = hb[e - 1]; sym = rf ? (uint32_t)rf[j] : ref_symbol(ix, pac, aware, ref0 + (uint32_t)j); }
            const int t1 = i == 0 ? -go : hb_e - go, t2 = i == 0 ? -ge : eb_e - ge;
            const int E = t1 > t2 ? t1 : t2, e_open = t1 > t2 ? 1 : 0;
            const int e1 = E > 0 ? E : 0;
            const int diag = hb_d + sw_score_flat(aware, sym, rc);
            const int A = e1 > diag ? e1 : diag;
            // max-plus scan: inclusive prefix maximum of g = A - go + j ge over the group's lanes, then one lane down, then the chunks before
            int inc = act ? A - go + j * ge : TB_NEG;
            { const int t = dpp_row_shr<1>(inc); if (lane >= 1) inc = inc > t ? inc : t; }
            { const int t = dpp_row_shr<2>(inc); if (lane >= 2) inc = inc > t ? inc : t; }
            { const int t = dpp_row_shr<4>(inc); if (lane >= 4) inc = inc > t ? inc : t; }
            int exc = dpp_row_shr<1>(inc);
            if (lane == 0) exc = TB_NEG;
            exc = exc > carry ? exc : carry;
            const int f_init = -(j - beg + 1) * ge, f_open = exc - (j - 1) * ge;
            const int f = f_init > f_open ? f_init : f_open;
            const int f1 = f > 0 ? f : 0;
            const int tt = e1 > f1 ? e1 : f1;
            const int H = tt > diag ? tt : diag;
            int Hl = dpp_row_shr<1>(H), fl = dpp_row_shr<1>(f);
            if (lane == 0) { Hl = prevH; fl = prevF; }
            const int f_opened = (Hl - go > fl - ge) ? 1 : 0;
            const int hsel = tt <= diag ? 0 : (e1 > f1 ? 1 : 2);
            if (act) {
                eb[u] = E;
                if (one) hb[u] = H; else hc[u] = H;                              // (every read of hb in this step is already issued)
                drow[j - x] = (uint8_t)(hsel | (e_open << 2) | (f_opened << 3));
                mx = mx > H ? mx : H;
            }
            if (!one) {
                const int last = (int)((threadIdx.x & 56u) + 7u);
                const int c7 = __shfl(inc, last); carry = carry > c7 ? carry : c7;
                prevH = __shfl(H, last); prevF = __shfl(f, last);
            }
        }
        if (!one) {
            gsync();
            const int u_last = end - x + 1;
            for (int t = 1 + lane; t <= u_last; t += 8) hb[t] = hc[t];
        }
    }
    gsync();
    return dpp_max8(mx);
}

// The same pass for bands of at most eight cells (half-width <= 3: an alignment whose ends differ by up to two bases, before any doubling
// -- nearly every mate rescue), with the band arrays in REGISTERS: lane t IS array index t + 1 of h_b / e_b (index 0 is the constant 0), the
// upper row's cells come over DPP row shifts (rows shift by one index once i > w), and a row is ~35 VALU instructions with no LDS round
// trip in its dependency chain (the LDS version waits ~6 of them per row).  The arrays' contents are tracked exactly, including what the
// reference leaves behind in indices a row does not write and the index it zeroes at the band's edge.
template <int N> __device__ __forceinline__ int dpp_row_shl(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x100 + N, 0xF, 0xF, true); }
__device__ __forceinline__ int tb_band_pass_small(uint8_t *__restrict__ dir, const uint8_t *rd, const uint8_t *rf, const int aware,
                                                  const int refLen, const int readLen, const int bw)
{
    const int lane = (int)(threadIdx.x & 7u);
    const int go = 3, ge = 1;
    const int width = 2 * bw + 3, width_d = 2 * bw + 1;                          // width - 1 <= 8: indices 1 .. 8 live in lanes 0 .. 7
    int Hp = 0, Ep = 0, mx = 0;                                                  // h_b[lane + 1], e_b[lane + 1]
    // the read base and the lane's reference symbol of a row are fetched from LDS one row ahead (nothing else in a row touches memory)
    uint32_t rc_n = rd[0], sym_n = lane < refLen && lane <= bw ? (uint32_t)rf[lane] : 0u;
    for (int i = 0; i < readLen; ++i) {
        const int x = i - bw > 0 ? i - bw : 0, xp = i - 1 - bw > 0 ? i - 1 - bw : 0, d = x - xp;
        const int end = refLen - 1 < i + bw ? refLen - 1 : i + bw, n = end - x + 1;
        const int edge = end + 1 < width - 1 ? end + 1 : width - 1;
        if (lane + 1 == edge) { Hp = 0; Ep = 0; }                                // h_b[edge] = e_b[edge] = 0 (ssw.c:593)
        const bool act = lane < n;
        const int j = x + lane;
        const uint32_t rc = rc_n, sym = sym_n;
        if (i + 1 < readLen) {
            const int xn = i + 1 - bw > 0 ? i + 1 - bw : 0, endn = refLen - 1 < i + 1 + bw ? refLen - 1 : i + 1 + bw;
            rc_n = rd[i + 1];
            sym_n = lane < endn - xn + 1 ? (uint32_t)rf[xn + lane] : 0u;
        }
        // h_b[e], e_b[e], h_b[e - 1] with e = lane + d + 1: this lane's or a neighbour's register
        const int Hn = dpp_row_shl<1>(Hp), En = dpp_row_shl<1>(Ep);
        int Hl = dpp_row_shr<1>(Hp);
        if (lane == 0) Hl = 0;                                                   // h_b[0]
        const int hb_e = d ? (lane == 7 ? 0 : Hn) : Hp, eb_e = d ? (lane == 7 ? 0 : En) : Ep, hb_d = d ? Hp : Hl;
        const int t1 = i == 0 ? -go : hb_e - go, t2 = i == 0 ? -ge : eb_e - ge;
        const int E = t1 > t2 ? t1 : t2, e_open = t1 > t2 ? 1 : 0;
        const int e1 = E > 0 ? E : 0;
        const int diag = hb_d + sw_score_flat(aware, sym, rc);
        const int A = e1 > diag ? e1 : diag;
        int inc = act ? A - go + j * ge : TB_NEG;
        { const int t = dpp_row_shr<1>(inc); if (lane >= 1) inc = inc > t ? inc : t; }
        { const int t = dpp_row_shr<2>(inc); if (lane >= 2) inc = inc > t ? inc : t; }
        { const int t = dpp_row_shr<4>(inc); if (lane >= 4) inc = inc > t ? inc : t; }
        int exc = dpp_row_shr<1>(inc);
        if (lane == 0) exc = TB_NEG;
        const int f_init = -(lane + 1) * ge, f_open = exc - (j - 1) * ge;
        const int f = f_init > f_open ? f_init : f_open;
        const int f1 = f > 0 ? f : 0;
        const int tt = e1 > f1 ? e1 : f1;
        const int H = tt > diag ? tt : diag;
        int Hleft = dpp_row_shr<1>(H), fleft = dpp_row_shr<1>(f);
        if (lane == 0) { Hleft = 0; fleft = 0; }
        const int f_opened = (Hleft - go > fleft - ge) ? 1 : 0;
        const int hsel = tt <= diag ? 0 : (e1 > f1 ? 1 : 2);
        if (act) {
            Ep = E; Hp = H;
            dir[(size_t)i * (size_t)width_d + (size_t)lane] = (uint8_t)(hsel | (e_open << 2) | (f_opened << 3));
            mx = mx > H ? mx : H;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    return dpp_max8(mx);
}

// the walk back (ssw.c:650-713) by the group's first lane: (len << 4 | op) runs into cig[], first operation first.
// Returns their number, 0 on a walk that leaves the direction bytes (the reference's "Trace back error": no CIGAR), -1 when more than cap.
__device__ __forceinline__ int tb_walk(const uint8_t *dir, const int64_t n_dir, const int refLen, const int readLen, const int bw, uint16_t *stack, uint16_t *cig, const int cap)
{
    const int width_d = 2 * bw + 1;
    int i = readLen - 1, j = refLen - 1, run = 0, n = 0, op = 0, cur = 0, from = 2;        // from: the entry of the cell to read (0 E, 1 F, 2 H)
    while (i > 0) {
        const int x = i - bw > 0 ? i - bw : 0;
        const int64_t at = (int64_t)i * width_d + (j - x);
        if (at < 0 || at >= n_dir) return 0;
        const uint32_t cell = dir[at];
        const uint32_t hsel = cell & 3u, e_open = (cell >> 2) & 1u, f_opened = (cell >> 3) & 1u;
        const int code = from == 0 ? 2 + (int)e_open : from == 1 ? 4 + (int)f_opened : hsel == 0 ? 1 : hsel == 1 ? 2 + (int)e_open : 4 + (int)f_opened;
        if (code == 1) { --i; --j; from = 2; op = 0; }                          // diagonal: M
        else if (code == 2) { --i; from = 0; op = 1; }                          // E extended: I, stay in E
        else if (code == 3) { --i; from = 2; op = 1; }                          // E opened: I, back to H
        else if (code == 4) { --j; from = 1; op = 2; }                          // F extended: D, stay in F
        else { --j; from = 2; op = 2; }                                         // F opened: D
        if (op == cur) ++run;
        else { if (n >= cap + 3) return -1; stack[n++] = (uint16_t)(run << 4 | cur); cur = op; run = 1; }
    }
    if (n + 2 > cap + 3) return -1;
    if (op == 0) stack[n++] = (uint16_t)((run + 1) << 4);                       // row 0's cell joins a run of M ...
    else { stack[n++] = (uint16_t)(run << 4 | op); stack[n++] = 16; }           // ... or is "1M" of its own
    if (n > cap) return -1;
    for (int t = 0; t < n; ++t) cig[t] = stack[n - 1 - t];
    return n;
}

// ---------------------------------------------------------------------------------------------
// k_swf / k_swr: persistent groups of 8 lanes pull rescue requests -- the two striped passes of ssw_align.  k_swf runs the forward pass
// (best score, its end point, second best: ssw.c:771-816) and leaves the requests that need their begin point with ok = 1; k_swr runs
// the reverse pass from the end point for those (ssw.c:817-830) and leaves them with ok = 2 for k_swtb, the banded traceback.  Two
// kernels rather than one: each column loop then has the registers to itself (as one kernel the loops reloaded spilled registers
// at the top of every column, a memory round trip per column), and k_swf packs two requests per group (Fwd2).
// ---------------------------------------------------------------------------------------------
// The register variants for reads up to 152 bases are capped at 128 VGPRs (four waves per SIMD)
#define SALT_SW_WAVES(SEG) __attribute__((amdgpu_waves_per_eu(((SEG) == 13 || (SEG) == 19) ? 4 : 1, ((SEG) == 13 || (SEG) == 19) ? 4 : 8)))

// the mate's bases on the requested strand into the group's LDS
__device__ __forceinline__ void sw_load_read(const uint8_t *__restrict__ seqs, uint32_t off, uint32_t L, uint32_t strand, uint8_t *dst, uint32_t lane)
{
    for (uint32_t i = lane; i < L; i += 8) {
        uint32_t c = strand ? seqs[off + (L - 1 - i)] : seqs[off + i];
        if (strand && c < 4) c = 3 - c;
        dst[i] = (uint8_t)(c > 4 ? 4 : c);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
}
// what follows the forward pass: the second best score outside +-maskLen around the end column (ssw.c:529-542; maskLen = L/2 >= 15 or
// none) and the result row, without begin point and CIGAR
__device__ __forceinline__ void sw_fwd_row(PeSwRes *o, const PeSwReq &rq, bool fits, int refLen, uint32_t L, const uint16_t *maxColumn,
                                           int max1, int end_ref1, int end_read1, uint32_t lane)
{
    int score2 = 0;
    const int maskLen = (int)L / 2;
    if (fits && maskLen >= 15) {
        int edge = end_ref1 - maskLen > 0 ? end_ref1 - maskLen : 0;
        for (int i = (int)lane; i < edge; i += 8) { int v = maxColumn[i]; score2 = score2 > v ? score2 : v; }
        edge = end_ref1 + maskLen > refLen ? refLen : end_ref1 + maskLen;
        for (int i = edge + (int)lane; i < refLen; i += 8) { int v = maxColumn[i]; score2 = score2 > v ? score2 : v; }
        score2 = dpp_max8(score2);
    }
    if (lane == 0) {
        o->score1 = fits ? max1 : 0; o->score2 = score2; o->ref_begin = -1; o->ref_end = fits ? end_ref1 : 0;
        o->read_begin = -1; o->read_end = fits ? end_read1 : 0; o->start = rq.start; o->strand = rq.strand; o->n_cigar = 0;
        o->ok = (uint16_t)(fits && !(rq.pad & 1u) ? 1 : 0);          // (pad bit 0: score only -- polish's first pass over every hit: ssw_align flag 0)
    }
}

// k_swf: pairs of neighbouring requests through the packed forward pass.  Every 8-lane group runs its own pair and pulls the next one the
// moment its window ends, inside the wave's one column loop (the groups of a wave sit at different columns of different pairs): no group
// waits for the longest window of its wave, and the last pairs of the queue spread over all SIMDs.  (Wave-sized pulls of 16 requests
// left 2.9 wave-tasks per SIMD at 46 783 requests: SIMDs with four of them ran twice as long as those with two.)  The two requests of a
// pair may differ in scoring (a singleton's plain rescue next to a SNP-aware one).  A pair that cannot take the packed path -- another
// read length than request 0's, a window beyond the scratch -- leaves with ok = 3 for k_swf1.  The loop ends for the whole wave at once, when no group has work left (a uniform exit: see k_swr).
#ifndef SALT_SWF_WAVES
#define SALT_SWF_WAVES 3          // 168 registers: the pair's rows, shifts and streams without spills (at 128 the packed shifts went to scratch and back every column)
#endif
template <int SEG>                     // stripe rows in registers, segLen <= SEG
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((SEG == 13 || SEG == 19) ? SALT_SWF_WAVES : 1, (SEG == 13 || SEG == 19) ? SALT_SWF_WAVES : 8)))
k_swf(IndexView ix, const uint8_t *__restrict__ pac, const uint8_t *__restrict__ seqs, const uint32_t *__restrict__ offs,
      const PeSwReq *__restrict__ req, const uint32_t *__restrict__ pctl, PeSwRes *__restrict__ res, uint32_t *__restrict__ head,
      uint32_t *__restrict__ overflow, uint8_t *__restrict__ scratch, uint32_t maxcol_bytes, uint32_t seg, int dbg_arg)
{
    const int dbg = SALT_DIAG_VAL(dbg_arg);                      // diagnostics build only: 2 = phase clocks, 4 = column counts, 32 / 128 = everything to k_swf1
    extern __shared__ __attribute__((aligned(16))) uint8_t sw_lds[];
    const uint32_t grp = threadIdx.x >> 3, lane = threadIdx.x & 7u;
    const uint32_t rd = (8u * seg + 15u) & ~15u;
    uint8_t *readA = sw_lds + (size_t)grp * 2u * rd, *readB = readA + rd;
    short *hmA = reinterpret_cast<short *>(sw_lds + 16u * rd) + (size_t)grp * 2u * SEG * 8, *hmB = hmA + SEG * 8;      // 2 x [SEG][8] per group, behind the reads
    uint32_t *lw = reinterpret_cast<uint32_t *>(sw_lds + sw_lds_words_at(seg, SEG, true)) + grp * 16u;                   // 2 x 8 window words per group
    const uint32_t n_req = pctl[0];
    if (n_req == 0) return;
    uint16_t *mcA = reinterpret_cast<uint16_t *>(scratch + ((size_t)blockIdx.x * 8 + grp) * 2u * maxcol_bytes);
    uint16_t *mcB = reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(mcA) + maxcol_bytes);
    // the launch's packed shape: request 0's read length
    const uint32_t mate0 = req[0].mate;
    const uint32_t Lref = (uint32_t)__builtin_amdgcn_readfirstlane((int)(offs[mate0 + 1] - offs[mate0]));
    const int segLen = (int)(Lref + 7u) / 8;
    const bool shape_ok = Lref >= 1u && Lref <= seg * 8u && segLen <= SEG && !(dbg & (32 | 128));
    Fwd2<SEG> f;
    uint32_t it0 = 0, ref0A = 0, ref0B = 0;
    int i = 0, nCols = 0, refLenA = 0, refLenB = 0, awA = 0, awB = 0;
    unsigned long long t0 = 0;
    bool idle = true, done = false, first = true;
    for (;;) {
        if (idle && !done) {                                        // the group's next pair (sw_pull: its own index first)
            uint32_t t = 2u * (blockIdx.x * 8u + grp);
            if (!first && lane == 0) t = sw_pull(head, 2u, 2u * 8u * gridDim.x, n_req);
            first = false;
            it0 = (uint32_t)__shfl((int)t, 0, 8);
            if (it0 >= n_req) done = true;
            else {
                const bool two = it0 + 1 < n_req;
                const PeSwReq rq0 = req[it0], rq1 = req[two ? it0 + 1 : it0];
                const uint32_t off0 = offs[rq0.mate], L0 = offs[rq0.mate + 1] - off0, off1 = offs[rq1.mate], L1 = offs[rq1.mate + 1] - off1;
                refLenA = (int)(rq0.end - rq0.start + 1); refLenB = (int)(rq1.end - rq1.start + 1);
                const bool fits0 = rq0.start < ix.ref_len && refLenA > 0 && (uint64_t)refLenA * 2u <= maxcol_bytes;
                const bool fits1 = rq1.start < ix.ref_len && refLenB > 0 && (uint64_t)refLenB * 2u <= maxcol_bytes;
                // (the odd last request runs as a pair with itself)
                if (!(shape_ok && fits0 && fits1 && L0 == Lref && L1 == Lref)) {
                    if (lane == 0) { res[it0].ok = 3; if (two) res[it0 + 1].ok = 3; atomicAdd(head + 4, two ? 2u : 1u); }      // forward pass pending (head[4]: how many)
                } else {
                    if (dbg & 2) t0 = __builtin_amdgcn_s_memtime();
                    sw_load_read(seqs, off0, Lref, rq0.strand, readA, lane); sw_load_read(seqs, off1, Lref, rq1.strand, readB, lane);
                    ref0A = rq0.start; ref0B = rq1.start; nCols = refLenA > refLenB ? refLenA : refLenB; i = 0; awA = rq0.aware; awB = rq1.aware;
                    fwd2_begin<SEG>(f, segLen, (int)Lref, readA, readB, hmA, hmB, lw, ix, pac, awA, awB, ref0A, ref0B, nCols, (int)lane);
                    idle = false;
                }
            }
        }
        if (__all(done)) break;
        if (!idle) {
            asm volatile("; k_swf column begin");
            fwd2_column<SEG>(f, i, segLen, ix, pac, awA, awB, ref0A, ref0B, refLenA, refLenB, nCols, mcA, mcB, hmA, hmB, (int)lane);
            ++i;
            asm volatile("; k_swf column end");
            if (i == nCols) {                                       // the pair's rows; the group is free again
                int edA, edB;
                fwd2_end<SEG>(f, segLen, (int)Lref, refLenA, refLenB, nCols, mcA, mcB, hmA, hmB, (int)lane, edA, edB);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                const bool two = it0 + 1 < n_req;
                const PeSwReq rq0 = req[it0], rq1 = req[two ? it0 + 1 : it0];
                sw_fwd_row(res + it0, rq0, true, refLenA, Lref, mcA, f.maxA, f.erA, edA, lane);
                if (two) sw_fwd_row(res + it0 + 1, rq1, true, refLenB, Lref, mcB, f.maxB, f.erB, edB, lane);
                if (lane == 0) {
                    if (dbg & 2) atomicAdd(overflow + 1, (uint32_t)(__builtin_amdgcn_s_memtime() - t0));       // phase clock
                    if (dbg & 4) atomicAdd(overflow + 1, (uint32_t)nCols);
                }
                idle = true;
            }
        }
    }
}

// k_swf1: the forward pass of the requests k_swf left pending (ok = 3), or of all of them (all = 1: reads beyond the register variants),
// one request at a time per group
template <int SEG>                     // 0: stripe rows in LDS (any read length); > 0: in registers, segLen <= SEG
__global__ void __launch_bounds__(64) SALT_SW_WAVES(SEG)
k_swf1(IndexView ix, const uint8_t *__restrict__ pac, const uint8_t *__restrict__ seqs, const uint32_t *__restrict__ offs,
       const PeSwReq *__restrict__ req, const uint32_t *__restrict__ pctl, PeSwRes *__restrict__ res, uint32_t *__restrict__ head,
       uint32_t *__restrict__ overflow, uint8_t *__restrict__ scratch, uint32_t maxcol_bytes, uint32_t seg, int all, int dbg_arg)
{
    const int dbg = SALT_DIAG_VAL(dbg_arg);                      // diagnostics build only: 4 = column counts, 32 = no pass
    extern __shared__ __attribute__((aligned(16))) uint8_t sw_lds[];
    const uint32_t grp = threadIdx.x >> 3, lane = threadIdx.x & 7u;
    SwLds s;
    if (SEG == 0) {
        uint8_t *base = sw_lds + (size_t)grp * sw_group_bytes(seg);
        s.H[0] = reinterpret_cast<short *>(base); s.H[1] = s.H[0] + seg * 8; s.E = s.H[1] + seg * 8; s.Hmax = s.E + seg * 8;
        s.read = reinterpret_cast<uint8_t *>(s.Hmax + seg * 8);
    } else {
        const uint32_t rd = (8u * seg + 15u) & ~15u;
        s.H[0] = s.H[1] = s.E = nullptr;
        s.read = sw_lds + (size_t)grp * rd;
        s.Hmax = reinterpret_cast<short *>(sw_lds + 8u * rd) + (size_t)grp * (SEG ? SEG : 1) * 8;
    }
    uint32_t *lw = reinterpret_cast<uint32_t *>(sw_lds + sw_lds_words_at(seg, SEG, false)) + grp * 8u;
    const uint32_t n_req = pctl[0];
    if (!all && head[3] == 0) return;                              // (head[3]: the requests k_swf left pending, behind the four queue heads)
    uint16_t *maxColumn = reinterpret_cast<uint16_t *>(scratch + ((size_t)blockIdx.x * 8 + grp) * 2u * maxcol_bytes);
    bool first = true;
    for (;;) {                                                       // eight requests per pull, a uniform exit (see k_swr)
        uint32_t base = 8u * blockIdx.x;
        if (!first && threadIdx.x == 0) base = sw_pull(head, 8u, 8u * gridDim.x, n_req);
        first = false;
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= n_req) break;
        const uint32_t it = base + grp;
        if (it >= n_req) continue;
        if (!all && res[it].ok != 3) continue;
        const PeSwReq rq = req[it];
        const uint32_t off = offs[rq.mate], L = offs[rq.mate + 1] - off;
        const int refLen = (int)(rq.end - rq.start + 1);
        const bool sane = rq.start < ix.ref_len && refLen > 0;
        const bool fits = sane && (uint64_t)refLen * 2u <= maxcol_bytes && L <= seg * 8u && !(dbg & 32);
        if (lane == 0 && sane && !fits && !(dbg & 32)) atomicAdd(overflow, 1u);
        int m = 0, er = 0, ed = 0;
        if (fits) {
            sw_load_read(seqs, off, L, rq.strand, s.read, lane);
            const uint8_t *rdp = s.read;
            auto fwd = [&](int q) -> uint32_t { return rdp[q]; };
            if (SEG == 0) sw_word_pass(ix, pac, (int)rq.aware, s, rq.start, 0, refLen, (int)L, fwd, 0xFFFF, maxColumn, m, er, ed);
            else sw_word_pass_reg<(SEG ? SEG : 1)>(ix, pac, (int)rq.aware, rq.start, 0, refLen, (int)L, fwd, 0xFFFF, maxColumn, s.Hmax, lw, m, er, ed,
                                                   (dbg & 4) ? overflow + 1 : nullptr);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        sw_fwd_row(res + it, rq, fits, refLen, L, maxColumn, m, er, ed, lane);
    }
}

// k_swr: the reverse pass from the end point of every request k_swf / k_swf1 left with ok = 1, eight requests per wave in step.  (One
// request per group pulled inside the column loop, as k_swf does for its pairs, took 0.77 ms against 0.58 ms here: a reverse pass is
// ~120 columns, and every group's start -- request, row, offsets, bases: four dependent loads -- holds up its whole wave.)
// The wave pulls eight requests at a time, one per group, and leaves as a whole (a uniform exit): groups that drew a number past the end
// sit the round out.  (Groups pulling one request each and leaving one by one is the shape this loop first had; with hipcc 7.2 the kernel
// then never finished once the traceback had moved out of it -- tools/dbg/sw_hang.hip reproduces that -- so the exit is kept uniform.)
template <int SEG>
__global__ void __launch_bounds__(64) SALT_SW_WAVES(SEG)
k_swr(IndexView ix, const uint8_t *__restrict__ pac, const uint8_t *__restrict__ seqs, const uint32_t *__restrict__ offs,
      const PeSwReq *__restrict__ req, const uint32_t *__restrict__ pctl, PeSwRes *__restrict__ res, uint32_t *__restrict__ head,
      uint32_t *__restrict__ overflow, uint32_t seg, int dbg_arg)
{
    const int dbg = SALT_DIAG_VAL(dbg_arg);                      // diagnostics build only: 2 = phase clocks, 8 = column counts, 16 = no pass
    extern __shared__ __attribute__((aligned(16))) uint8_t sw_lds[];
    const uint32_t grp = threadIdx.x >> 3, lane = threadIdx.x & 7u;
    SwLds s;
    if (SEG == 0) {
        uint8_t *base = sw_lds + (size_t)grp * sw_group_bytes(seg);
        s.H[0] = reinterpret_cast<short *>(base); s.H[1] = s.H[0] + seg * 8; s.E = s.H[1] + seg * 8; s.Hmax = s.E + seg * 8;
        s.read = reinterpret_cast<uint8_t *>(s.Hmax + seg * 8);
    } else {
        const uint32_t rd = (8u * seg + 15u) & ~15u;
        s.H[0] = s.H[1] = s.E = nullptr;
        s.read = sw_lds + (size_t)grp * rd;
        s.Hmax = reinterpret_cast<short *>(sw_lds + 8u * rd) + (size_t)grp * (SEG ? SEG : 1) * 8;
    }
    uint32_t *lw = reinterpret_cast<uint32_t *>(sw_lds + sw_lds_words_at(seg, SEG, false)) + grp * 8u;
    const uint32_t n_req = pctl[0];
    bool first = true;
    for (;;) {
        uint32_t base = 8u * blockIdx.x;
        if (!first && threadIdx.x == 0) base = sw_pull(head, 8u, 8u * gridDim.x, n_req);
        first = false;
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= n_req) break;
        const uint32_t it = base + grp;
        if (it >= n_req) continue;
        PeSwRes *o = res + it;
        if (o->ok != 1 || (dbg & 16)) continue;
        const PeSwReq rq = req[it];
        const uint32_t off = offs[rq.mate], L = offs[rq.mate + 1] - off;
        const int max1 = o->score1, end_ref1 = o->ref_end, end_read1 = o->read_end;
        const unsigned long long t0 = dbg & 2 ? __builtin_amdgcn_s_memtime() : 0ull;
        sw_load_read(seqs, off, L, rq.strand, s.read, lane);
        const uint8_t *rdp = s.read;
        int max2, beg_ref, beg_read_rev;
        auto rev = [&](int q) -> uint32_t { return rdp[end_read1 - q]; };
        if (SEG == 0) sw_word_pass(ix, pac, (int)rq.aware, s, rq.start, 1, end_ref1 + 1, end_read1 + 1, rev, max1, (uint16_t *)nullptr, max2, beg_ref, beg_read_rev);
        else sw_word_pass_reg<(SEG ? SEG : 1)>(ix, pac, (int)rq.aware, rq.start, 1, end_ref1 + 1, end_read1 + 1, rev, max1, (uint16_t *)nullptr, s.Hmax, lw,
                                               max2, beg_ref, beg_read_rev, (dbg & 8) ? overflow + 1 : nullptr);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            o->ref_begin = beg_ref; o->read_begin = end_read1 - beg_read_rev; o->ok = 2;      // the banded traceback is k_swtb's
            if (dbg & 2) atomicAdd(overflow + 2, (uint32_t)(__builtin_amdgcn_s_memtime() - t0));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_swtb: the banded traceback of the requests k_swr left with ok = 2 (ssw.c:837-848), eight lanes per alignment (tb_band_pass)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_swtb(IndexView ix, const uint8_t *__restrict__ pac, const uint8_t *__restrict__ seqs, const uint32_t *__restrict__ offs,
       const PeSwReq *__restrict__ req, const uint32_t *__restrict__ pctl, PeSwRes *__restrict__ res, uint32_t *__restrict__ head,
       uint32_t *__restrict__ overflow, uint8_t *__restrict__ scratch, uint32_t group_bytes, TbGeom tg, int dbg_arg)
{
    const int dbg_max_bw = SALT_DIAG_VAL(dbg_arg);               // diagnostics build only: bands wider than (low 16 bits) count as overflow; bit 16: phase clocks
    extern __shared__ __attribute__((aligned(16))) uint8_t tb_lds[];
    const uint32_t grp = threadIdx.x >> 3, lane = threadIdx.x & 7u;
    uint8_t *const rd = tb_lds + (size_t)grp * tg.group_b, *const rf = rd + tg.read_b;
    int32_t *const lrows = reinterpret_cast<int32_t *>(rf + tg.ref_b);
    uint8_t *const ldir = reinterpret_cast<uint8_t *>(lrows + 3 * tg.row_w);
    uint8_t *const my = scratch + ((size_t)blockIdx.x * 8 + grp) * group_bytes;          // global: three rows of SW_BAND_W ints, then direction bytes
    int32_t *const grows = reinterpret_cast<int32_t *>(my);
    uint8_t *const gdir = my + 3u * SW_BAND_W * 4u;
    const uint64_t gdir_cap = group_bytes > 3u * SW_BAND_W * 4u ? group_bytes - 3u * SW_BAND_W * 4u : 0u;
    const uint32_t n_req = pctl[0];
    bool first = true;
    for (;;) {                                                       // eight requests per pull, a uniform exit (see k_swr)
        uint32_t base = 8u * blockIdx.x;
        if (!first && threadIdx.x == 0) base = sw_pull(head, 8u, 8u * gridDim.x, n_req);
        first = false;
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= n_req) break;
        const uint32_t it = base + grp;
        if (it >= n_req) continue;
        PeSwRes *o = res + it;
        if (o->ok != 2) continue;
        const PeSwReq rq = req[it];
        const int ref_begin = o->ref_begin, ref_end = o->ref_end, read_begin = o->read_begin, read_end = o->read_end, score = o->score1;
        const int rfl = ref_end - ref_begin + 1, rdl = read_end - read_begin + 1, aware = rq.aware;
        const uint32_t off = offs[rq.mate], L = offs[rq.mate + 1] - off;
        int n_cig = 0;
        const bool clk = (dbg_max_bw & 0x10000) != 0;                 // diagnostics build: s_memtime ticks per phase into overflow[4..7]
        const unsigned long long c0 = clk ? __builtin_amdgcn_s_memtime() : 0ull;
        unsigned long long c1 = c0, c2 = c0;
        if ((rq.pad & 2u) && rdl < 20) { if (lane == 0) { o->n_cigar = 0; o->ok = 0; } continue; }    // a mate rescue (pad bit 1): alnpe.c:297 turns it down whatever its CIGAR
        if (rfl > 0 && rdl > 0 && (uint32_t)rdl <= tg.read_b) {
            const uint32_t ref0 = rq.start + (uint32_t)ref_begin;
            // the aligned part of the mate on the requested strand and the alignment's reference symbols: 64 per trip and group (eight loads
            // of a lane in flight together; one after the other they were most of this kernel's time)
            for (int t0 = 0; t0 < rdl; t0 += 64) {
                uint32_t v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int t = t0 + 8 * k + (int)lane;
                    const uint32_t q = (uint32_t)(read_begin + (t < rdl ? t : 0));
                    v[k] = rq.strand ? seqs[off + (L - 1 - q)] : seqs[off + q];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int t = t0 + 8 * k + (int)lane;
                    uint32_t c = v[k];
                    if (rq.strand && c < 4) c = 3 - c;
                    if (t < rdl) rd[t] = (uint8_t)(c > 4 ? 4 : c);
                }
            }
            const bool ref_in_lds = (uint32_t)rfl <= tg.ref_b;
            if (ref_in_lds)
                for (int t0 = 0; t0 < rfl; t0 += 64) {
                    uint32_t v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) { const int t = t0 + 8 * k + (int)lane; v[k] = ref_symbol(ix, pac, aware, ref0 + (uint32_t)(t < rfl ? t : 0)); }
#pragma unroll
                    for (int k = 0; k < 8; ++k) { const int t = t0 + 8 * k + (int)lane; if (t < rfl) rf[t] = (uint8_t)v[k]; }
                }
            int bw = rfl - rdl; bw = (bw < 0 ? -bw : bw) + 1;
            int mx = 0;
            bool fits = true, dir_lds = true;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            if (clk) c1 = __builtin_amdgcn_s_memtime();
            for (;;) {                                               // band doubling (ssw.c:570-631): until the band holds the alignment's score
                const int width = 2 * bw + 3, width_d = 2 * bw + 1;
                const bool rows_lds = (uint32_t)width <= tg.row_w;
                dir_lds = (uint64_t)width_d * (uint64_t)rdl <= tg.dir_b;
                if (width > (int)SW_BAND_W || (!dir_lds && (uint64_t)width_d * (uint64_t)rdl > gdir_cap) || ((dbg_max_bw & 0xFFFF) && bw > (dbg_max_bw & 0xFFFF))) { fits = false; break; }
                int m;
                const uint8_t *rfp = ref_in_lds ? rf : nullptr;
                if (width - 1 <= 8 && dir_lds && ref_in_lds) m = tb_band_pass_small(ldir, rd, rf, aware, rfl, rdl, bw);
                else if (rows_lds && dir_lds) m = tb_band_pass<true, true>(lrows, lrows + tg.row_w, lrows + 2 * tg.row_w, ldir, rd, rfp, ix, pac, aware, ref0, rfl, rdl, bw);
                else if (rows_lds) m = tb_band_pass<true, false>(lrows, lrows + tg.row_w, lrows + 2 * tg.row_w, gdir, rd, rfp, ix, pac, aware, ref0, rfl, rdl, bw);
                else if (dir_lds) m = tb_band_pass<false, true>(grows, grows + SW_BAND_W, grows + 2 * SW_BAND_W, ldir, rd, rfp, ix, pac, aware, ref0, rfl, rdl, bw);
                else m = tb_band_pass<false, false>(grows, grows + SW_BAND_W, grows + 2 * SW_BAND_W, gdir, rd, rfp, ix, pac, aware, ref0, rfl, rdl, bw);
                mx = mx > m ? mx : m;                                // the reference's maximum runs over all passes
                if (mx >= score) break;
                bw *= 2;
            }
            if (clk) c2 = __builtin_amdgcn_s_memtime();
            if (!fits) n_cig = -1;
            else if (lane == 0) n_cig = tb_walk(dir_lds ? ldir : gdir, (int64_t)(2 * bw + 1) * rdl, rfl, rdl, bw, reinterpret_cast<uint16_t *>(lrows), o->cigar, SALT_MAX_CIGAR_OPS);
            n_cig = __shfl(n_cig, 0, 8);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        } else if (rfl > 0 && rdl > 0) n_cig = -1;
        if (lane == 0) {
            if (n_cig < 0) { atomicAdd(overflow, 1u); n_cig = 0; }
            o->n_cigar = (uint16_t)n_cig;
            o->ok = (uint16_t)((n_cig > 0 && rdl >= 20) ? 1 : 0);      // alnpe.c:297 (filters = 0, filterd = 20)
            if (clk) {
                const unsigned long long c3 = __builtin_amdgcn_s_memtime();
                atomicAdd(overflow + 4, (uint32_t)(c1 - c0)); atomicAdd(overflow + 5, (uint32_t)(c2 - c1)); atomicAdd(overflow + 6, (uint32_t)(c3 - c2)); atomicAdd(overflow + 7, 1u);
            }
        }
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
        for (int k = 0; k < nr; ++k) { PeSwReq r; r.start = st[k]; r.end = en[k]; r.mate = 2 * p + who[k]; r.strand = str[k]; r.aware = aw[k]; r.pad = 2; req[base + k] = r; pr.rescued[k] = who[k]; }
    }
    pairs[p] = pr;
}

void launch_pair(uint32_t n_pairs, uint32_t min_tlen, uint32_t max_tlen, uint32_t l_pac, const uint32_t *offs, salt_result_t *res,
                 PePair *pairs, PeSwReq *req, uint32_t *pctl, hipStream_t st)
{
    if (n_pairs) hipLaunchKernelGGL(k_pair, dim3((n_pairs + 255) / 256), dim3(256), 0, st, n_pairs, min_tlen, max_tlen, l_pac, offs, res, pairs, req, pctl);
}

// stripe rows in registers for reads up to 256 bases (13 / 19 / 32 stripes of 8), in LDS beyond
static int sw_seg_variant(uint32_t max_len)
{
    static const bool lds_only = getenv("SALT_GPU_SW_LDS") && atoi(getenv("SALT_GPU_SW_LDS"));
    const uint32_t seg = (max_len + 7) / 8;
    return lds_only ? 0 : seg <= 13 ? 13 : seg <= 19 ? 19 : seg <= 32 ? 32 : 0;
}
uint32_t sw_lds_bytes(uint32_t max_len, bool pairs)
{
    const uint32_t seg = (max_len + 7) / 8;
    const uint32_t v = (uint32_t)sw_seg_variant(max_len);
    // register variants: per group the read + H-at-best [SEG][8] shorts of each request in flight (k_swf: a pair, k_swf1 / k_swr: one)
    return sw_lds_words_at(seg, v, pairs && v) + 8u * (pairs && v ? 64u : 32u);        // + the groups' window words (RefStream)
}
// one-wave blocks per CU: what registers (the register variants) or LDS (the LDS variant) admit
uint32_t sw_blocks_per_cu(uint32_t max_len)
{
    int n = 0;
    const int v = sw_seg_variant(max_len);
    const void *f = v == 13 ? (const void *)k_swf<13> : v == 19 ? (const void *)k_swf<19> : v == 32 ? (const void *)k_swf<32> : (const void *)k_swf1<0>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, 64, sw_lds_bytes(max_len, true)) != hipSuccess || n < 1) n = 1;
    return (uint32_t)(n > (int)SW_MAX_BLOCKS_PER_CU ? (int)SW_MAX_BLOCKS_PER_CU : n);
}
// k_swtb's LDS per group: the aligned part of the mate, the reference symbols of alignments up to twice the read length, three band rows
// of 32 cells (band half-width <= 14) and direction bytes for 13 cells per row (half-width <= 6): what mate rescue produces
static TbGeom tb_geom(uint32_t max_len)
{
    TbGeom t;
    t.read_b = (max_len + 15u) & ~15u;
    t.ref_b = (2u * max_len + 15u) & ~15u; if (t.ref_b > 1024u) t.ref_b = 1024u;
    t.row_w = 32;
    t.dir_b = (13u * max_len + 15u) & ~15u; if (t.dir_b > 4096u) t.dir_b = 4096u;
    t.group_b = t.read_b + t.ref_b + 3u * t.row_w * 4u + t.dir_b;
    return t;
}
// Scratch geometry of one launch: k_swf's groups keep the per-column maxima of the longest window (max_window columns) in global memory;
// k_swtb's groups three band rows and the direction bytes of the widest band the rows hold (one byte per cell) for alignments whose band
// outgrows the LDS -- its grid is cut back before that passes SW_SCRATCH_TOTAL.
SwGeom sw_geom(uint32_t max_len, uint64_t max_window, uint32_t cus)
{
    SwGeom g;
    const uint64_t mc = (2 * max_window + 255) & ~255ull;
    g.maxcol_bytes = (uint32_t)mc;
    g.n_blocks = cus * sw_blocks_per_cu(max_len);
    if (!g.n_blocks) g.n_blocks = 1;
    const uint64_t dir = ((uint64_t)max_len * (SW_BAND_W - 3) + 255) & ~255ull;
    g.tb_group_bytes = (uint32_t)(3ull * SW_BAND_W * 4 + dir);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_swtb, 64, 8u * tb_geom(max_len).group_b) != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_cu > (int)SW_MAX_BLOCKS_PER_CU) per_cu = (int)SW_MAX_BLOCKS_PER_CU;
    uint64_t tb = (uint64_t)cus * (uint32_t)per_cu;
    const uint64_t cap = SW_SCRATCH_TOTAL / (8ull * g.tb_group_bytes);
    if (tb > cap) tb = cap;
    g.tb_blocks = (uint32_t)(tb ? tb : 1);
    return g;
}
void sw_geom_limit(SwGeom &g, uint32_t blocks) { if (blocks < 1) blocks = 1; if (g.n_blocks > blocks) g.n_blocks = blocks; if (g.tb_blocks > blocks) g.tb_blocks = blocks; }
uint64_t sw_scratch_bytes(const SwGeom &g) { return (uint64_t)g.n_blocks * 16 * g.maxcol_bytes + (uint64_t)g.tb_blocks * 8 * g.tb_group_bytes; }

void launch_sw(const IndexView &ix, const uint8_t *pac, const uint8_t *seqs, const uint32_t *offs, const PeSwReq *req, const uint32_t *pctl,
               PeSwRes *res, uint32_t *heads, uint32_t *overflow, uint8_t *scratch, SwGeom g, uint32_t max_len, hipStream_t st)
{
    const uint32_t seg = (max_len + 7) / 8;
#ifdef SALT_DIAG
    const int dbg = getenv("SALT_GPU_SW_SKIP_TB") ? atoi(getenv("SALT_GPU_SW_SKIP_TB")) : 0;     // diagnostics: 2 = phase clocks, 4 / 8 = column counts
#else
    const int dbg = 0;
#endif
    // heads[0..3]: the queue heads of k_swf, k_swf1, k_swr, k_swtb; heads[4]: the requests k_swf left to k_swf1
#define SALT_LAUNCH_SW(V, ALL) do { \
        hipLaunchKernelGGL(k_swf1<V>, dim3(g.n_blocks), dim3(64), sw_lds_bytes(max_len, false), st, ix, pac, seqs, offs, req, pctl, res, heads + 1, \
                           overflow, scratch, g.maxcol_bytes, seg, ALL, dbg); \
        hipLaunchKernelGGL(k_swr<V>, dim3(g.n_blocks), dim3(64), sw_lds_bytes(max_len, false), st, ix, pac, seqs, offs, req, pctl, res, heads + 2, \
                           overflow, seg, dbg); } while (0)
#define SALT_LAUNCH_SWF(V) hipLaunchKernelGGL(k_swf<V>, dim3(g.n_blocks), dim3(64), sw_lds_bytes(max_len, true), st, ix, pac, seqs, offs, req, pctl, res, heads, \
                                              overflow, scratch, g.maxcol_bytes, seg, dbg)
    switch (sw_seg_variant(max_len)) {
    case 13: SALT_LAUNCH_SWF(13); SALT_LAUNCH_SW(13, 0); break;
    case 19: SALT_LAUNCH_SWF(19); SALT_LAUNCH_SW(19, 0); break;
    case 32: SALT_LAUNCH_SWF(32); SALT_LAUNCH_SW(32, 0); break;
    default: SALT_LAUNCH_SW(0, 1); break;
    }
#undef SALT_LAUNCH_SW
#undef SALT_LAUNCH_SWF
    const TbGeom tg = tb_geom(max_len);
    int tb_dbg = 0;
#ifdef SALT_DIAG
    if (getenv("SALT_GPU_NO_TB")) return;                                                       // diagnostics: no traceback (no CIGARs)
    if (getenv("SALT_GPU_TB_MAXBW")) tb_dbg = atoi(getenv("SALT_GPU_TB_MAXBW"));
    if (getenv("SALT_GPU_TB_CLOCKS")) tb_dbg |= 0x10000;
#endif
    hipLaunchKernelGGL(k_swtb, dim3(g.tb_blocks), dim3(64), 8u * tg.group_b, st, ix, pac, seqs, offs, req, pctl, res, heads + 3, overflow,
                       scratch + (uint64_t)g.n_blocks * 16 * g.maxcol_bytes, g.tb_group_bytes, tg, tb_dbg);
}

} // namespace salt
