// salt_amd/csrc/salt_index.hip -- attach-time expansion kernels (run once per index, gfx950).
//
// They replay the reference's own locate walks for EVERY suffix-array row so that the per-read
// kernels replace a data-dependent walk by one load:
//   k_build_c_sa   bwt_sa / bwt_invPsi                 (Align_src/bwt.c:89-102, bwt.h:67-71)
//   k_build_r_pos  Rbwt_back_bwt_sa                    (Align_src/rbwt.c:316-333)
//   k_build_text   the indexed genome, from SA and BWT (text[SA[row] - 1] = BWT[row])
//   k_build_wlkt   both searches' interval after the last W bases of a seed
//                                                      (Align_src/rbwt.c:619-648, alnse.c:273-275)
#include "salt_device.h"
#include "salt_kernels.h"

namespace salt {

// grid of a grid-stride kernel over n items: never more than 2^20 blocks (a HIP launch is limited to 2^32 threads, and a
// GRCh38-sized index has more rows than that allows with one thread per row and a margin)
static inline uint32_t stride_grid(uint64_t n) { uint64_t b = (n + 255) / 256; if (b > (1u << 20)) b = 1u << 20; return b ? (uint32_t)b : 1u; }

// ---- re-packing of the file-format arrays into the device layout (salt_device.h) ----
// COcc block b = BWT symbols [64 b, 64 b + 64) of the $-removed C BWT: running counts in front of it + two bit planes.  The file
// (bwt.h:57-64) has 4 counts every 128 symbols followed by 8 words of 16 symbols, first symbol in the top bits.
__global__ void __launch_bounds__(256)
k_pack_c_occ(const uint32_t *__restrict__ bwt, uint64_t bwt_words, uint32_t seq_len, uint64_t n_blocks, COcc *__restrict__ out, uint32_t *__restrict__ err)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += stride) {
        const uint64_t k0 = b * 64, base = k0 / 128 * 12;
        COcc rec; rec.cnt[0] = rec.cnt[1] = rec.cnt[2] = rec.cnt[3] = 0; rec.lo = rec.hi = 0;
        if (base + 4 <= bwt_words) for (int c = 0; c < 4; ++c) rec.cnt[c] = bwt[base + c];
        else if (k0 < seq_len) atomicOr(err, 1u);
        const uint32_t half = (uint32_t)(k0 % 128) / 64;                   // second half of a file block: add the first half's symbols
        for (uint32_t h = 0; h <= half; ++h) {
            for (uint32_t w = 0; w < 4; ++w) {
                const uint64_t kw = k0 / 128 * 128 + h * 64 + w * 16;
                if (kw >= seq_len) break;
                const uint64_t widx = base + 4 + h * 4 + w;
                if (widx >= bwt_words) { atomicOr(err, 1u); break; }
                const uint32_t word = bwt[widx];
                for (uint32_t q = 0; q < 16; ++q) {
                    if (kw + q >= seq_len) break;
                    const uint32_t sy = (word >> ((15u - q) << 1)) & 3u;
                    if (h < half) ++rec.cnt[sy];
                    else { const uint32_t i = w * 16 + q; rec.lo |= (uint64_t)(sy & 1u) << i; rec.hi |= (uint64_t)(sy >> 1) << i; }
                }
            }
        }
        out[b] = rec;
    }
}
// ROcc block b = symbols [128 b, 128 b + 128) of the stored R BWT (8 nibbles per word, first in the top nibble, rbwt.h:115-119):
// counts of A, C, G, T in front of it + three bit planes.  The counts come from the file's own explicit Occ values -- a 32-bit
// major value every 65536 symbols plus a 16-bit minor value every 256 (two per word, the even one in the high half,
// rbwt.c:40-80) -- and, for odd blocks, the 128 symbols of the block before.
__global__ void __launch_bounds__(256)
k_pack_r_occ(const uint32_t *__restrict__ code, uint64_t code_words, const uint32_t *__restrict__ minor, uint64_t minor_words, const uint32_t *__restrict__ major,
             uint64_t major_words, uint32_t text_len, uint64_t n_blocks, ROcc *__restrict__ out, uint32_t *__restrict__ err)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += stride) {
        ROcc rec; rec.cnt[0] = rec.cnt[1] = rec.cnt[2] = rec.cnt[3] = 0;
        rec.b0[0] = rec.b0[1] = rec.b1[0] = rec.b1[1] = rec.b2[0] = rec.b2[1] = 0;
        const uint64_t e = b / 2;                                           // explicit value at symbol 256 e
        const uint64_t mi = e / 2 * 5, ma = e / 256 * 5;
        if (mi + 4 < minor_words && ma + 4 < major_words) {
            for (int c = 0; c < 4; ++c) { const uint32_t mv = minor[mi + c]; rec.cnt[c] = major[ma + c] + ((e & 1) ? (mv & 0xFFFFu) : (mv >> 16)); }
        } else if (b * 128 <= text_len) atomicOr(err, 2u);
        for (uint32_t h = (uint32_t)(b & 1) ^ 1u; h < 2; ++h) {                  // odd block: count the even block before it, then pack its own
            const bool own = h == 1 || (b & 1) == 0;
            const uint64_t k0 = own ? b * 128 : (b - 1) * 128;
            for (uint32_t w = 0; w < 16; ++w) {
                const uint64_t kw = k0 + w * 8;
                if (kw >= text_len) break;
                if (kw / 8 >= code_words) { atomicOr(err, 2u); break; }
                const uint32_t word = code[kw / 8];
                for (uint32_t q = 0; q < 8; ++q) {
                    if (kw + q >= text_len) break;
                    const uint32_t sy = (word >> ((7u - q) * 4u)) & 15u;
                    if (sy > 4) { atomicOr(err, 4u); continue; }
                    if (!own) { if (sy < 4) ++rec.cnt[sy]; }
                    else {
                        const uint32_t i = w * 8 + q, hh = i >> 6, bb = i & 63u;
                        rec.b0[hh] |= (uint64_t)(sy & 1u) << bb; rec.b1[hh] |= (uint64_t)((sy >> 1) & 1u) << bb; rec.b2[hh] |= (uint64_t)(sy >> 2) << bb;
                    }
                }
            }
            if (own) break;
        }
        out[b] = rec;
    }
}
void launch_pack_c_occ(const uint32_t *bwt, uint64_t bwt_words, uint32_t seq_len, uint64_t n_blocks, COcc *out, uint32_t *err, hipStream_t st)
{
    hipLaunchKernelGGL(k_pack_c_occ, dim3(stride_grid(n_blocks)), dim3(256), 0, st, bwt, bwt_words, seq_len, n_blocks, out, err);
}
void launch_pack_r_occ(const uint32_t *code, uint64_t code_words, const uint32_t *minor, uint64_t minor_words, const uint32_t *major, uint64_t major_words,
                       uint32_t text_len, uint64_t n_blocks, ROcc *out, uint32_t *err, hipStream_t st)
{
    hipLaunchKernelGGL(k_pack_r_occ, dim3(stride_grid(n_blocks)), dim3(256), 0, st, code, code_words, minor, minor_words, major, major_words, text_len, n_blocks, out, err);
}

__global__ void __launch_bounds__(256)
k_build_c_sa(IndexView ix, const uint32_t *__restrict__ sa_sampled, uint32_t intv, uint32_t *__restrict__ out)
{
    const uint64_t n = (uint64_t)ix.c_seq_len + 1, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        uint32_t k = (uint32_t)j, steps = 0;
        while (k % intv != 0) {
            ++steps;
            if (k == ix.c_primary) k = 0;
            else {
                uint32_t c = c_sym(ix, k < ix.c_primary ? k : k - 1);
                k = ix.c_L2[c] + c_occ(ix, k, c);
            }
        }
        out[j] = steps + sa_sampled[k / intv];      // sa_sampled[0] = 0xFFFFFFFF: wraps exactly as in C
    }
}

// text[SA[row] - 1] = BWT[row]: the genome the C index was built over, 16 bases per word, first base in the high bits
__global__ void __launch_bounds__(256)
k_build_text(IndexView ix, uint32_t *__restrict__ out)
{
    const uint64_t n = (uint64_t)ix.c_seq_len + 1, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        if (j == ix.c_primary) continue;                           // the primary row holds '$'
        uint32_t p = ix.c_sa[j];
        if (p == 0xFFFFFFFFu) p = ix.c_seq_len;                    // row 0: the empty suffix (bwt_sa's sa[0] = -1)
        if (p == 0 || p > ix.c_seq_len) continue;
        const uint32_t c = c_sym(ix, (uint32_t)(j < ix.c_primary ? j : j - 1)), i = p - 1;
        atomicOr(out + (i >> 4), c << (30 - 2 * (i & 15u)));
    }
}

// c_ctx (salt_device.h): one record per suffix-array row, from the expanded suffix array, the 2-bit text and the allele masks
__global__ void __launch_bounds__(256)
k_build_c_ctx(IndexView ix, uint32_t ctx_k, uint4 *__restrict__ out)
{
    const uint64_t n = (uint64_t)ix.c_seq_len + 1, stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t len = ix.c_seq_len < ix.ref_len ? ix.c_seq_len : ix.ref_len;      // positions both the text and the masks hold
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const uint32_t sa = ix.c_sa[j];
        const uint64_t s = sa == 0xFFFFFFFFu ? (uint64_t)ix.c_seq_len : (uint64_t)sa;
        uint64_t lo = 0, hi = 0;
        uint32_t ns[2] = { 0, 0 };
        bool whole[2] = { s + ctx_k + CTX_N <= len, s >= CTX_N && s <= len };
        for (int side = 0; side < 2; ++side) {
            if (!whole[side]) { ns[side] = 3; continue; }
            for (uint32_t t = 0; t < CTX_N; ++t) {
                const uint64_t p = side == 0 ? s + ctx_k + t : s - 1 - t;
                const uint32_t b = (ix.text[p >> 4] >> (30 - 2 * (uint32_t)(p & 15u))) & 3u;
                const uint32_t mask = (ix.ref[p >> 3] >> (4 * (uint32_t)(p & 7u))) & 15u;
                const uint32_t bit = side * CTX_N + t;
                lo |= (uint64_t)(b & 1u) << bit; hi |= (uint64_t)(b >> 1) << bit;
                ns[side] += mask != (1u << b);
            }
            if (ns[side] > 3) ns[side] = 3;
        }
        out[j] = ctx_pack(sa, lo, hi, ns[0], ns[1]);
    }
}

// r_ctx: the same record for every row of the R index -- the genome around the position r_pos gives the row (the seed's first base): a
// paired-end mate of a repeat enumerates up to max_locate rows of every R interval too (alnse.c:538-595), and two thirds of the candidate
// windows k_heavy_pe looked at came from there
__global__ void __launch_bounds__(256)
k_build_r_ctx(IndexView ix, uint32_t ctx_k, uint4 *__restrict__ out)
{
    const uint64_t n = (uint64_t)ix.r_text_len + 1, stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t len = ix.c_seq_len < ix.ref_len ? ix.c_seq_len : ix.ref_len;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const uint32_t rp = ix.r_pos[j];
        const uint64_t s = rp;
        uint64_t lo = 0, hi = 0;
        uint32_t ns[2] = { 0, 0 };
        bool whole[2] = { s + ctx_k + CTX_N <= len, s >= CTX_N && s <= len };
        for (int side = 0; side < 2; ++side) {
            if (!whole[side]) { ns[side] = 3; continue; }
            for (uint32_t t = 0; t < CTX_N; ++t) {
                const uint64_t p = side == 0 ? s + ctx_k + t : s - 1 - t;
                const uint32_t b = (ix.text[p >> 4] >> (30 - 2 * (uint32_t)(p & 15u))) & 3u;
                const uint32_t mask = (ix.ref[p >> 3] >> (4 * (uint32_t)(p & 7u))) & 15u;
                const uint32_t bit = side * CTX_N + t;
                lo |= (uint64_t)(b & 1u) << bit; hi |= (uint64_t)(b >> 1) << bit;
                ns[side] += mask != (1u << b);
            }
            if (ns[side] > 3) ns[side] = 3;
        }
        out[j] = ctx_pack(rp, lo, hi, ns[0], ns[1]);
    }
}
void launch_build_r_ctx(const IndexView &ix, uint32_t ctx_k, uint4 *out, hipStream_t st)
{
    const uint64_t n = (uint64_t)ix.r_text_len + 1;
    hipLaunchKernelGGL(k_build_r_ctx, dim3(stride_grid(n)), dim3(256), 0, st, ix, ctx_k, out);
}

void launch_build_c_ctx(const IndexView &ix, uint32_t ctx_k, uint4 *out, hipStream_t st)
{
    const uint64_t n = (uint64_t)ix.c_seq_len + 1;
    hipLaunchKernelGGL(k_build_c_ctx, dim3(stride_grid(n)), dim3(256), 0, st, ix, ctx_k, out);
}

void launch_build_text(const IndexView &ix, uint32_t *out, hipStream_t st)
{
    const uint64_t n = (uint64_t)ix.c_seq_len + 1;
    hipLaunchKernelGGL(k_build_text, dim3(stride_grid(n)), dim3(256), 0, st, ix, out);
}

__global__ void __launch_bounds__(256)
k_build_r_pos(IndexView ix, const uint32_t *__restrict__ r_sa, uint32_t *__restrict__ out)
{
    const uint64_t n = (uint64_t)ix.r_text_len + 1, stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t n_acgt = ix.r_cum[4];
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        uint32_t sa_index = (uint32_t)j, step = 0;
        // every walk ends at a '#' (the text starts with one); the bound only guards a corrupt index
        while (sa_index <= n_acgt && step < (1u << 20)) {
            uint32_t c = r_bwt2nt(ix, sa_index);
            sa_index = ix.r_cum[c] + r_occ(ix, sa_index, c) + 1;
            ++step;
        }
        out[j] = sa_index > n_acgt ? r_sa[sa_index - n_acgt - 1] + step - 1 : 0xFFFFFFFFu;
    }
}

// One 32-byte entry (two uint4) per W-mer x (first base in the high bits).  First half: the C interval after LKT_lookup_sa on the last
// lkt_len bases + bwt_match_exact_alt on the W - lkt_len bases before them (lookup.h:39-53, bwt.c:281-309) in .x/.y, and the R interval
// after the first W steps of Rbwt_exact_match_backward from (0, textLength) (rbwt.c:619-648) in .z/.w; (1, 0) = empty.  Second half,
// for a C interval of ONE row (the usual case at W = 16): .x = the genome position of that suffix and .y = the 16 bases in front of
// it (2 bits each, the base right in front of the suffix in the lowest bits) -- so k_seed finishes such a seed (the remaining k - W
// bases must equal the text in front of the suffix, the rest of bwt_match_exact_alt cannot branch any more) from this one gather,
// without touching the suffix array or the text.
__global__ void __launch_bounds__(256)
k_build_wlkt(IndexView ix, uint32_t len, uint4 *__restrict__ out)
{
    const uint64_t n = 1ull << (2 * len), stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t lk = ix.lkt_len;
    for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n; x += stride) {
        const uint32_t tail = (uint32_t)(x & ((1ull << (2 * lk)) - 1ull));
        uint32_t k = ix.lkt[tail], l = ix.lkt[tail + 1] - 1;
        for (uint32_t t = 0; t < len - lk && k <= l; ++t) {
            uint32_t c = (uint32_t)(x >> (2 * (lk + t))) & 3u;
            uint32_t ok = c_occ(ix, k - 1, c), ol = c_occ(ix, l, c);
            k = ix.c_L2[c] + ok + 1; l = ix.c_L2[c] + ol;
        }
        uint32_t k0 = 0, l0 = ix.r_text_len;
        for (uint32_t step = 0; step < len && k0 <= l0; ++step) {
            uint32_t c = (uint32_t)(x >> (2 * step)) & 3u;     // last base of the W-mer first
            k0 = ix.r_cum[c] + r_occ(ix, k0, c) + 1;
            l0 = ix.r_cum[c] + r_occ(ix, l0 + 1, c);
        }
        uint4 e = make_uint4(1u, 0u, 1u, 0u), f = make_uint4(0u, 0u, 0u, 0u);
        if (k <= l) { e.x = k; e.y = l; }
        if (k0 <= l0) { e.z = k0; e.w = l0; }
        // a C interval of one or two rows: the position of each suffix and the 16 bases in front of it (.x/.y row k, .z/.w row k + 1)
        for (uint32_t q = 0; q < 2 && k <= l && l - k <= 1 && q <= l - k; ++q) {
            uint32_t p0 = ix.c_sa[k + q];
            if (p0 == 0xFFFFFFFFu) p0 = ix.c_seq_len;          // row 0: the empty suffix (bwt_sa's sa[0] = -1)
            uint32_t prev = 0;
            if (p0 >= 16) {
                const uint32_t t0 = p0 - 16, tj = t0 >> 4, tr = t0 & 15u;
                const uint64_t vt = ((uint64_t)ix.text[tj] << 32) | ix.text[tj + 1];
                prev = (uint32_t)(vt >> (32 - 2 * tr));
            } else if (p0 > 0) prev = ix.text[0] >> (32 - 2 * p0);
            if (q == 0) { f.x = p0; f.y = prev; } else { f.z = p0; f.w = prev; }
        }
        out[2 * x] = e; out[2 * x + 1] = f;
    }
}

void launch_build_wlkt(const IndexView &ix, uint32_t len, uint4 *out, hipStream_t st)
{
    uint64_t n = 1ull << (2 * len), blocks = (n + 255) / 256;
    if (blocks > (1u << 20)) blocks = 1u << 20;
    hipLaunchKernelGGL(k_build_wlkt, dim3((uint32_t)blocks), dim3(256), 0, st, ix, len, out);
}

void launch_build_c_sa(const IndexView &ix, const uint32_t *sa_sampled, uint32_t sa_intv, uint32_t *out, hipStream_t st)
{
    uint64_t n = (uint64_t)ix.c_seq_len + 1;
    hipLaunchKernelGGL(k_build_c_sa, dim3(stride_grid(n)), dim3(256), 0, st, ix, sa_sampled, sa_intv, out);
}

void launch_build_r_pos(const IndexView &ix, const uint32_t *r_sa, uint32_t *out, hipStream_t st)
{
    uint64_t n = (uint64_t)ix.r_text_len + 1;
    hipLaunchKernelGGL(k_build_r_pos, dim3(stride_grid(n)), dim3(256), 0, st, ix, r_sa, out);
}

} // namespace salt
