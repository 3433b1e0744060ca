// salt_amd/csrc/salt_device.h -- device-side index layout and the rank/LF primitives (gfx950).
//
// The *file* formats are salt's; the layout in HBM is ours.  Every structure below
// is a pure function of the index files, so results stay bit-identical to the reference:
//
//   COcc   32 B per 64 BWT symbols of the plain-genome ("C") FM-index: 4 running counts + 2 bit
//          planes.  Occ(k,c) = one 32-byte read + one 64-bit popcount.   (replaces bwt.c:113-175 on
//          the 48 B / 128-symbol BWA blocks of bwt.h:57-64)
//   ROcc   64 B per 128 symbols of the SNP local-pattern ("R") FM-index, alphabet {A,C,G,T,#}:
//          4 running counts + 3 bit planes ('#' count = position - sum).  (replaces rbwt.c:40-191:
//          major/minor checkpoints + <=128-symbol scan through a 64 K-entry table)
//   c_sa   the FULL suffix array of the C index (4 B x (G+1)), expanded once at attach time from the
//          1-in-8 samples by the reference's own LF walk (bwt.c:89-102) -- locate is then one load.
//   r_pos  for every R suffix-array row the value Rbwt_back_bwt_sa() (rbwt.c:316-333) returns,
//          again expanded once at attach time.
//   text   the genome the C index was built over, 2 bits per base, 16 per word with the first base in the high bits
//          (rebuilt at attach time from the expanded suffix array and the BWT: text[SA[row] - 1] = BWT[row]).  Once a seed's C
//          interval is down to one row, the rest of its backward search is a comparison against this text.
//   wlkt   W-mer table, 16 B per W-mer (W = 12..16, default 14, never above the seed length): for every W-mer
//          the SA intervals both searches hold after consuming it (.x/.y = C, .z/.w = R; one gather serves both) -- C: LKT_lookup_sa on its last 12 bases
//          (lookup.h:39-53, with that table's A-padded tail quirk) followed by W-12 steps of
//          bwt_match_exact_alt (bwt.c:281-309); R: the first W iterations of Rbwt_exact_match_backward
//          from (0, textLength) (rbwt.c:619-648).  Tabulated once at attach time; (1,0) = dead.
//   c_ctx  (optional, 16 B per C suffix-array row) .x = the row's suffix-array value again, .y/.z/.w = the 2-bit genome around that
//          suffix: CTX_N bases behind the first ctx_k (= seed length) bases of the suffix and CTX_N bases in front of it, as two bit
//          planes, plus per side the number of positions whose allele mask is not exactly the genome base (SNP sites, N runs).
//          A locate that reads its rows from here can bound the mismatches of the candidate window from below WITHOUT touching
//          the window: rows come 4 to a 64-byte line and in order, windows are one random DRAM row each.  A candidate whose
//          bound exceeds 3 would fail alnse_check_nogap's ed_mismatch (alnse.c:734-782) and is dropped before it; nothing else
//          changes (see ctx_reject below).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace salt {

struct COcc { uint32_t cnt[4]; uint64_t lo, hi; };                 // 32 B
struct ROcc { uint32_t cnt[4]; uint64_t b0[2], b1[2], b2[2]; };    // 64 B
static_assert(sizeof(COcc) == 32, "COcc");
static_assert(sizeof(ROcc) == 64, "ROcc");

// Header at the start of the device image; offsets are relative to the image base so that the
// image can be broadcast to other GPUs byte-for-byte.
struct ImageHeader {
    uint64_t magic, bytes;
    uint32_t c_primary, c_L2[5], c_seq_len, c_sa_intv;
    uint32_t lkt_len, lkt_n;
    uint32_t r_text_len, r_inv_sa0, r_cum[6];
    uint32_t ref_len, r_lkt_len;
    uint64_t off_c_occ, off_c_sa, off_lkt, off_r_occ, off_r_pos, off_wlkt, off_ref, off_text;
    uint64_t n_c_blocks, n_r_blocks;
    uint64_t off_ctx;                                               // 0: no context table; else behind the W-mer table (not part of the compact image)
    uint32_t ctx_k, pad0;
    uint64_t reserved[5];
};
static const uint64_t IMAGE_MAGIC = 0x53414c5447465841ull;          // "SALTGFXA"

// What kernels receive (by value): resolved pointers + scalars.
struct IndexView {
    const COcc *c_occ; const uint32_t *c_sa; const uint32_t *lkt;
    const ROcc *r_occ; const uint32_t *r_pos; const uint4 *wlkt; const uint32_t *ref; const uint32_t *text;
    const uint4 *c_ctx; uint32_t ctx_k;                            // nullptr: absent
    const uint4 *r_ctx;                                            // the same records for the R rows (.x = r_pos of the row); outside the image, built with the paired-end genome (set_pac); nullptr: absent
    uint32_t c_primary, c_L2[5], c_seq_len;
    uint32_t r_text_len, r_inv_sa0, r_cum[6];
    uint32_t ref_len, lkt_len, r_lkt_len;
};

// ---- c_ctx records -------------------------------------------------------------------------------------------------
// Bit t of a plane, t < CTX_N: genome base s + ctx_k + t (side A, behind the seed); bit CTX_N + u: genome base s - 1 - u (side B, in
// front of the suffix).  96 bits behind .x: low plane (46), high plane (46), special-site count of A (2), of B (2); a count of 3
// means "3 or more, or the side runs off the genome": that side bounds nothing.
static const uint32_t CTX_N = 23;
static const uint64_t CTX_MASK_A = (1ull << CTX_N) - 1ull, CTX_MASK_B = CTX_MASK_A << CTX_N;
__device__ __forceinline__ uint4 ctx_pack(uint32_t sa, uint64_t lo, uint64_t hi, uint32_t ns_a, uint32_t ns_b)
{
    return make_uint4(sa, (uint32_t)lo, (uint32_t)(lo >> 32) | ((uint32_t)hi << 14), (uint32_t)(hi >> 18) | (ns_a << 28) | (ns_b << 30));
}
// The read's side of the comparison, the same for every row of one seed interval: planes of the read bases facing the record's
// positions, and `use` = the positions that face a base of the read which is not N.
struct CtxRead { uint64_t lo, hi, use; };
// true: the window of this row has more than `bound` mismatches for certain (so ed_mismatch(..., bound) would return -1).
// Mismatches between the 2-bit genome and the read are counted over the usable positions; every special site among them may be a
// match after all (the mask holds more than the genome base), so each side's count is lowered by its special-site count.
__device__ __forceinline__ bool ctx_reject(const uint4 rec, const CtxRead rd, uint32_t bound)
{
    const uint64_t lo = (uint64_t)rec.y | ((uint64_t)(rec.z & 0x3FFFu) << 32), hi = (uint64_t)(rec.z >> 14) | ((uint64_t)(rec.w & 0x0FFFFFFFu) << 18);
    const uint64_t m = ((lo ^ rd.lo) | (hi ^ rd.hi)) & rd.use;
    const uint32_t ns_a = (rec.w >> 28) & 3u, ns_b = rec.w >> 30;
    const uint32_t ca = (uint32_t)__popcll(m & CTX_MASK_A), cb = (uint32_t)__popcll(m & CTX_MASK_B);
    const uint32_t la = (ns_a == 3u || ca < ns_a) ? 0u : ca - ns_a, lb = (ns_b == 3u || cb < ns_b) ? 0u : cb - ns_b;
    return la + lb > bound;
}

__device__ __forceinline__ uint32_t sel4(uint4 v, uint32_t c)
{
    return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w;
}

// a[c] for a per-lane c without indexing the kernel argument (a dynamic index would move the array to LDS or scratch)
__device__ __forceinline__ uint32_t pick4(const uint32_t *a, uint32_t c) { return c == 0 ? a[0] : c == 1 ? a[1] : c == 2 ? a[2] : a[3]; }
__device__ __forceinline__ uint32_t pick5(const uint32_t *a, uint32_t c) { return c == 0 ? a[0] : c == 1 ? a[1] : c == 2 ? a[2] : c == 3 ? a[3] : a[4]; }

// Occ(k, c) of the C index: occurrences of c in BWT rows [0, k]  (bwt_occ, bwt.c:113-138)
__device__ __forceinline__ uint32_t c_occ(const IndexView &ix, uint32_t k, uint32_t c)
{
    if (k == ix.c_seq_len) return ix.c_L2[c + 1] - ix.c_L2[c];
    if (k == 0xFFFFFFFFu) return 0;
    k -= (k >= ix.c_primary);                       // '$' is not stored (bwt.c:120)
    const uint4 *p = reinterpret_cast<const uint4 *>(ix.c_occ + (k >> 6));
    uint4 cnt = p[0];
    uint4 pl = p[1];
    uint64_t lo = (uint64_t)pl.x | ((uint64_t)pl.y << 32), hi = (uint64_t)pl.z | ((uint64_t)pl.w << 32);
    uint32_t m = (k & 63u) + 1u;
    uint64_t mask = m == 64 ? ~0ull : ((1ull << m) - 1ull);
    uint64_t eq = ((c & 1) ? lo : ~lo) & ((c & 2) ? hi : ~hi) & mask;
    return sel4(cnt, c) + (uint32_t)__popcll(eq);
}

// Occ(k, c) and Occ(l, c) of one backward-search step (bwt_2occ, bwt.c:140-175).  Once an interval is narrow both
// rows fall into the same checkpoint block, which is then fetched once: half the gather work of two c_occ calls.
__device__ __forceinline__ uint32_t c_occ_eval(const uint4 cnt, const uint4 pl, const uint32_t k, const uint32_t c)
{
    const uint64_t lo = (uint64_t)pl.x | ((uint64_t)pl.y << 32), hi = (uint64_t)pl.z | ((uint64_t)pl.w << 32);
    const uint32_t m = (k & 63u) + 1u;
    const uint64_t mask = m == 64 ? ~0ull : ((1ull << m) - 1ull);
    const uint64_t eq = ((c & 1) ? lo : ~lo) & ((c & 2) ? hi : ~hi) & mask;
    return sel4(cnt, c) + (uint32_t)__popcll(eq);
}
// returns the number of 32-byte blocks fetched (0..2)
__device__ __forceinline__ uint32_t c_occ2(const IndexView &ix, const uint32_t k, const uint32_t l, const uint32_t c, uint32_t &ok, uint32_t &ol)
{
    const uint32_t full = pick4(ix.c_L2 + 1, c) - pick4(ix.c_L2, c);
    const bool ks = k == ix.c_seq_len || k == 0xFFFFFFFFu, ls = l == ix.c_seq_len || l == 0xFFFFFFFFu;
    const uint32_t kk = k - (k >= ix.c_primary), ll = l - (l >= ix.c_primary);
    uint4 cnt = make_uint4(0, 0, 0, 0), pl = make_uint4(0, 0, 0, 0);
    if (!ks) { const uint4 *p = reinterpret_cast<const uint4 *>(ix.c_occ + (kk >> 6)); cnt = p[0]; pl = p[1]; }
    ok = ks ? (k == ix.c_seq_len ? full : 0u) : c_occ_eval(cnt, pl, kk, c);
    if (ls) { ol = l == ix.c_seq_len ? full : 0u; return ks ? 0u : 1u; }
    const bool second = ks || (ll >> 6) != (kk >> 6);
    if (second) { const uint4 *p = reinterpret_cast<const uint4 *>(ix.c_occ + (ll >> 6)); cnt = p[0]; pl = p[1]; }
    ol = c_occ_eval(cnt, pl, ll, c);
    return (ks ? 0u : 1u) + (second ? 1u : 0u);
}

// symbol k of the $-removed C BWT (bwt_B0, bwt.h:64) -- used only by the attach-time SA expansion
__device__ __forceinline__ uint32_t c_sym(const IndexView &ix, uint32_t k)
{
    const COcc *r = ix.c_occ + (k >> 6);
    uint32_t b = k & 63u;
    return (uint32_t)((r->lo >> b) & 1ull) | ((uint32_t)((r->hi >> b) & 1ull) << 1);
}

// Occ(index, c) of the R index: occurrences of c among the first `index` stored symbols
// (Rbwt_BWTOccValue, rbwt.c:159-191).  c in 0..4 ('#' = 4).
__device__ __forceinline__ uint32_t r_occ(const IndexView &ix, uint32_t index, uint32_t c)
{
    index -= (index > ix.r_inv_sa0);                // '$' is not stored (rbwt.c:165)
    uint32_t blk = index >> 7, m = index & 127u;
    const uint4 *p = reinterpret_cast<const uint4 *>(ix.r_occ + blk);
    uint4 cnt = p[0], q0 = p[1], q1 = p[2], q2 = p[3];
    uint64_t b0l = (uint64_t)q0.x | ((uint64_t)q0.y << 32), b0h = (uint64_t)q0.z | ((uint64_t)q0.w << 32);
    uint64_t b1l = (uint64_t)q1.x | ((uint64_t)q1.y << 32), b1h = (uint64_t)q1.z | ((uint64_t)q1.w << 32);
    uint64_t b2l = (uint64_t)q2.x | ((uint64_t)q2.y << 32), b2h = (uint64_t)q2.z | ((uint64_t)q2.w << 32);
    uint64_t ml = m >= 64 ? ~0ull : ((1ull << m) - 1ull);
    uint64_t mh = m > 64 ? ((1ull << (m - 64)) - 1ull) : 0ull;
    uint64_t el = ((c & 1) ? b0l : ~b0l) & ((c & 2) ? b1l : ~b1l) & ((c & 4) ? b2l : ~b2l) & ml;
    uint64_t eh = ((c & 1) ? b0h : ~b0h) & ((c & 2) ? b1h : ~b1h) & ((c & 4) ? b2h : ~b2h) & mh;
    uint32_t base = c < 4 ? sel4(cnt, c) : (blk << 7) - (cnt.x + cnt.y + cnt.z + cnt.w);
    return base + (uint32_t)__popcll(el) + (uint32_t)__popcll(eh);
}

// Occ(a, c) and Occ(b, c) of one R backward-search step; one block fetch when both indices share it
struct ROccBlk { uint4 cnt, q0, q1, q2; };
__device__ __forceinline__ ROccBlk r_occ_load(const IndexView &ix, const uint32_t blk)
{
    const uint4 *p = reinterpret_cast<const uint4 *>(ix.r_occ + blk);
    return ROccBlk{ p[0], p[1], p[2], p[3] };
}
__device__ __forceinline__ uint32_t r_occ_eval(const ROccBlk &r, const uint32_t index, const uint32_t c)
{
    const uint32_t blk = index >> 7, m = index & 127u;
    const uint64_t b0l = (uint64_t)r.q0.x | ((uint64_t)r.q0.y << 32), b0h = (uint64_t)r.q0.z | ((uint64_t)r.q0.w << 32);
    const uint64_t b1l = (uint64_t)r.q1.x | ((uint64_t)r.q1.y << 32), b1h = (uint64_t)r.q1.z | ((uint64_t)r.q1.w << 32);
    const uint64_t b2l = (uint64_t)r.q2.x | ((uint64_t)r.q2.y << 32), b2h = (uint64_t)r.q2.z | ((uint64_t)r.q2.w << 32);
    const uint64_t ml = m >= 64 ? ~0ull : ((1ull << m) - 1ull);
    const uint64_t mh = m > 64 ? ((1ull << (m - 64)) - 1ull) : 0ull;
    const uint64_t el = ((c & 1) ? b0l : ~b0l) & ((c & 2) ? b1l : ~b1l) & ((c & 4) ? b2l : ~b2l) & ml;
    const uint64_t eh = ((c & 1) ? b0h : ~b0h) & ((c & 2) ? b1h : ~b1h) & ((c & 4) ? b2h : ~b2h) & mh;
    const uint32_t base = c < 4 ? sel4(r.cnt, c) : (blk << 7) - (r.cnt.x + r.cnt.y + r.cnt.z + r.cnt.w);
    return base + (uint32_t)__popcll(el) + (uint32_t)__popcll(eh);
}
// returns the number of 64-byte blocks fetched (1 or 2)
__device__ __forceinline__ uint32_t r_occ2(const IndexView &ix, uint32_t a, uint32_t b, const uint32_t c, uint32_t &oa, uint32_t &ob)
{
    a -= (a > ix.r_inv_sa0); b -= (b > ix.r_inv_sa0);              // '$' is not stored (rbwt.c:165)
    ROccBlk r = r_occ_load(ix, a >> 7);
    oa = r_occ_eval(r, a, c);
    const bool second = (b >> 7) != (a >> 7);
    if (second) r = r_occ_load(ix, b >> 7);
    ob = r_occ_eval(r, b, c);
    return second ? 2u : 1u;
}

// Rbwt_bwt2nt (rbwt.h:103-122): BWT symbol of row pos; the '$' row reads as '#'
__device__ __forceinline__ uint32_t r_bwt2nt(const IndexView &ix, uint32_t pos)
{
    if (pos == ix.r_inv_sa0) return 4;
    pos -= (pos > ix.r_inv_sa0);
    const ROcc *r = ix.r_occ + (pos >> 7);
    uint32_t h = (pos >> 6) & 1u, b = pos & 63u;
    return (uint32_t)((r->b0[h] >> b) & 1ull) | ((uint32_t)((r->b1[h] >> b) & 1ull) << 1) |
           ((uint32_t)((r->b2[h] >> b) & 1ull) << 2);
}

} // namespace salt
