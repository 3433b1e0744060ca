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

__global__ void __launch_bounds__(256)
k_build_c_sa(IndexView ix, const uint32_t *__restrict__ sa_sampled, uint32_t intv, uint32_t *__restrict__ out)
{
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > ix.c_seq_len) return;
    uint32_t k = (uint32_t)j, steps = 0;
    while (k % intv != 0) {
        ++steps;
        if (k == ix.c_primary) k = 0;
        else {
            uint32_t c = c_sym(ix, k < ix.c_primary ? k : k - 1);
            k = ix.c_L2[c] + c_occ(ix, k, c);
        }
    }
    out[j] = steps + sa_sampled[k / intv];          // sa_sampled[0] = 0xFFFFFFFF: wraps exactly as in C
}

// text[SA[row] - 1] = BWT[row]: the genome the C index was built over, 16 bases per word, first base in the high bits
__global__ void __launch_bounds__(256)
k_build_text(IndexView ix, uint32_t *__restrict__ out)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > ix.c_seq_len || j == ix.c_primary) return;             // the primary row holds '$'
    uint32_t p = ix.c_sa[j];
    if (p == 0xFFFFFFFFu) p = ix.c_seq_len;                        // row 0: the empty suffix (bwt_sa's sa[0] = -1)
    if (p == 0 || p > ix.c_seq_len) return;
    const uint32_t c = c_sym(ix, (uint32_t)(j < ix.c_primary ? j : j - 1)), i = p - 1;
    atomicOr(out + (i >> 4), c << (30 - 2 * (i & 15u)));
}

void launch_build_text(const IndexView &ix, uint32_t *out, hipStream_t st)
{
    const uint64_t n = (uint64_t)ix.c_seq_len + 1;
    hipLaunchKernelGGL(k_build_text, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, ix, out);
}

__global__ void __launch_bounds__(256)
k_build_r_pos(IndexView ix, const uint32_t *__restrict__ r_sa, uint32_t *__restrict__ out)
{
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > ix.r_text_len) return;
    uint32_t sa_index = (uint32_t)j, step = 0;
    const uint32_t n_acgt = ix.r_cum[4];
    // every walk ends at a '#' (the text starts with one); the bound only guards a corrupt index
    while (sa_index <= n_acgt && step < (1u << 20)) {
        uint32_t c = r_bwt2nt(ix, sa_index);
        sa_index = ix.r_cum[c] + r_occ(ix, sa_index, c) + 1;
        ++step;
    }
    out[j] = sa_index > n_acgt ? r_sa[sa_index - n_acgt - 1] + step - 1 : 0xFFFFFFFFu;
}

// One entry per W-mer x (first base in the high bits): the C interval after LKT_lookup_sa on the last lkt_len bases
// + bwt_match_exact_alt on the W - lkt_len bases before them (lookup.h:39-53, bwt.c:281-309) in .x/.y, and the R
// interval after the first W steps of Rbwt_exact_match_backward from (0, textLength) (rbwt.c:619-648) in .z/.w;
// (1, 0) = empty.  One 16-byte gather serves both searches of a seed.
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
        uint4 e = make_uint4(1u, 0u, 1u, 0u);
        if (k <= l) { e.x = k; e.y = l; }
        if (k0 <= l0) { e.z = k0; e.w = l0; }
        out[x] = e;
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
    hipLaunchKernelGGL(k_build_c_sa, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, ix, sa_sampled, sa_intv, out);
}

void launch_build_r_pos(const IndexView &ix, const uint32_t *r_sa, uint32_t *out, hipStream_t st)
{
    uint64_t n = (uint64_t)ix.r_text_len + 1;
    hipLaunchKernelGGL(k_build_r_pos, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, ix, r_sa, out);
}

} // namespace salt
