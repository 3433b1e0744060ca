// salt_amd/csrc/salt_align.hip -- per-batch kernels of the single-end alignment path (gfx950 only).
//
//   k_seed   one lane per (read, strand, seed slot): 12-mer tables + backward search on the C and R
//            FM-indexes + interval-shrinking left extension            (alnse_seed_overlap, alnse.c:199-312)
//   k_align  one 64-lane wave per read: order seeds (klib introsort), locate under the max_locate
//            cap, sort + dedup loci, masked-Hamming verify of all candidates in parallel with the
//            reference's sequential best/first-hit rule replayed by ballots, Landau-Vishkin on the
//            4-bit masks when nothing matched gap-free, hit selection, MAPQ, CIGAR
//            (alnse_locate_alt alnse.c:633-731, alnse_check_nogap :734-782, alnse_check_withgap
//             :871-901, alnse_overlap_alt :1045-1104, query_set_hits/gen_mapq/query_gen_cigar
//             query.c:270-333, LandauVishkin.c:19-122,176-470, editdistance.c:88-284)
//
// All arithmetic is integer; outputs are bit-exact with the reference CPU path.
#include "salt_device.h"
#include "salt_kernels.h"

namespace salt {

// Every block is exactly one wave, and the LDS unit executes one wave's DS instructions in issue order, so
// lanes only need the COMPILER not to move LDS accesses across the hand-off point: a wavefront-scope
// fence + wave_barrier costs no instruction, where __syncthreads() costs s_waitcnt vmcnt(0) + s_barrier.
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                     __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); } while (0)

static constexpr int MAXL = SALT_MAX_READ_LEN;
static constexpr int SLOTS = SALT_MAX_SEED_SLOTS;
static constexpr int MAXLOC = SALT_MAX_LOCATE;
static constexpr int LVK = 31;                  // MAX_K (LandauVishkin.c:13)
// k_light writes the first 24 bytes of a result row as six dwords
static_assert(offsetof(salt_result_t, strand) == 4 && offsetof(salt_result_t, mapq) == 7 && offsetof(salt_result_t, b0) == 8 &&
              offsetof(salt_result_t, b1) == 12 && offsetof(salt_result_t, seq_start) == 16 && offsetof(salt_result_t, seq_end) == 18 &&
              offsetof(salt_result_t, n_hits) == 20 && offsetof(salt_result_t, n_cigar) == 22 && offsetof(salt_result_t, skipped) == 23 &&
              offsetof(salt_result_t, hits) == 24 && sizeof(salt_hit_t) == 8, "salt_result_t header layout");
static constexpr int NHIT = 6;                  // first hits kept per strand (5 + the primary)
// gen_mapq's quotient 255 * x / b0 (query.c:270-281; x = |b0 - b1| <= 100000, so the product fits 32 bits).  b0 <= 3 whenever the
// best hit is gap-free: constant divisors there instead of a runtime division (a 64-bit one costs ~100 instructions per read).
__device__ __forceinline__ uint32_t mapq_quot(uint32_t x, uint32_t b0)
{
    const uint32_t v = 255u * x;
    return b0 == 1u ? v : b0 == 2u ? v / 2u : b0 == 3u ? v / 3u : v / b0;
}
static constexpr uint32_t INF = 255;

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Pointers that reach a non-inlined device function lose their address space (flat_load + coupled
// vmcnt/lgkmcnt waits); these casts state that they point to global memory.
typedef const uint32_t __attribute__((address_space(1))) *gp_u32;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef const u32x4_t __attribute__((address_space(1))) *gp_u32x4;
__device__ __forceinline__ gp_u32 as_global(const uint32_t *p) { return (gp_u32)p; }
__device__ __forceinline__ gp_u32x4 as_global(const uint4 *p) { return (gp_u32x4)(const void *)p; }

// diagnostic phase clock: adds the cycles since the previous stamp to ctr[slot] (lane 0, only when counting)
struct PhaseClock {                    // acc: a per-wave LDS array (flushed once per wave), or nullptr
    unsigned long long *acc; uint64_t t;
    __device__ PhaseClock(unsigned long long *a) : acc(a), t(a ? __builtin_amdgcn_s_memtime() : 0) {}
    __device__ __forceinline__ void stamp(int slot)
    {
        if (!acc) return;
        uint64_t n = __builtin_amdgcn_s_memtime();
        if (lane_id() == 0) acc[slot] += (unsigned long long)(n - t);
        t = n;
    }
    __device__ __forceinline__ void add(int slot, unsigned long long v) { if (acc && lane_id() == 0) acc[slot] += v; }
};

__device__ __forceinline__ uint32_t read_base(const uint8_t *seq, uint32_t L, int strand, uint32_t i)
{
    if (strand == 0) return seq[i];
    uint32_t c = seq[L - 1 - i];
    return c < 4 ? 3 - c : c;                    // query_seq_reverse (query.c:46-71)
}

// ---------------------------------------------------------------------------------------------
// k_pack: the batch's reads in the two packed, fixed-stride forms the other kernels gather from (PackGeom,
// salt_kernels.h).  A block of 256 threads takes 256 / (2*nw32) consecutive reads: their bytes are staged in LDS
// as both strands (coalesced loads; query_seq_reverse / nst_nt4_table semantics, codes > 4 read as N,
// query.c:46-71,177-183), then one thread per (read, strand, 32 bases) packs with shift-and-mask steps only.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pk_nibbles(uint32_t x)        // 4 bytes (low nibbles) -> 16 bits, byte 0 lowest
{
    uint32_t t = x & 0x0F0F0F0Fu;
    t = (t | (t >> 4)) & 0x00FF00FFu;
    return (t | (t >> 8)) & 0xFFFFu;
}

// Phase 1: one thread per (read, 32 forward bases): the bytes come straight from global memory as aligned dwords (never
// past the ends of seqs[]), are clamped (codes > 4 read as N) and packed with shift-and-mask steps; the forward words go to
// global memory and to LDS.  Phase 2: one thread per reverse-strand word, derived from the forward WORDS: reversing a
// read reverses its bit string, complementing a base reverses the bits of its one-hot nibble / inverts its 2-bit code.
struct PackWords { uint32_t pm[64], tb[32], nm[16]; };               // forward words of one read (up to 512 bases)

__global__ void __launch_bounds__(256)
k_pack(PackGeom pg, uint32_t n_reads, const uint8_t *__restrict__ seqs, const uint32_t *__restrict__ offs,
       uint32_t *__restrict__ pm, uint32_t *__restrict__ tb)
{
    extern __shared__ uint32_t pk_lds[];                                    // [rpb][nw8 + nw16 + nw32] forward words
    __shared__ uint32_t so[65];
    const uint32_t tpr = pg.nw32, rpb = 256 / tpr < 64 ? 256 / tpr : 64;   // threads per read in phase 1, reads per block
    const uint32_t wpr = pg.nw8 + pg.nw16 + pg.nw32;                       // forward words per read
    const uint32_t r0 = blockIdx.x * rpb, nr = n_reads - r0 < rpb ? n_reads - r0 : rpb;
    if (threadIdx.x <= nr) so[threadIdx.x] = offs[r0 + threadIdx.x];
    __syncthreads();
    const uint32_t cap = 8 * pg.nw8;                                        // bases the records hold per strand
    const uint64_t total_bases = offs[n_reads];                             // end of the valid bytes of seqs[]
    {
        const uint32_t rr = threadIdx.x / tpr, j = threadIdx.x - rr * tpr;
        if (rr < nr) {
            uint32_t L = so[rr + 1] - so[rr];
            if (L > cap) L = cap;                                           // longer than max_read_len says: truncated, never out of bounds
            // 32 bytes from seqs + so[rr] + 32 j as 9 aligned dwords; dwords outside [seqs, seqs + total_bases) read as 0
            const uintptr_t a0 = (uintptr_t)seqs + so[rr] + 32u * j, lo = (uintptr_t)seqs & ~(uintptr_t)3, hi = (uintptr_t)seqs + total_bases;
            const uintptr_t ab = a0 & ~(uintptr_t)3;
            const uint32_t sh = (uint32_t)(a0 & 3u) * 8u;
            uint32_t wv[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) { const uintptr_t ad = ab + 4u * t; wv[t] = (ad >= lo && ad < hi) ? *reinterpret_cast<const uint32_t *>(ad) : 0u; }
            uint32_t d[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                uint32_t x = __funnelshift_r(wv[t], wv[t + 1], sh);
                const uint32_t big = ((((x & 0x7F7F7F7Fu) + 0x7B7B7B7Bu) | x) & 0x80808080u) >> 7;   // 1 in every byte >= 5
                const uint32_t mb = big * 0xFFu;
                x = (x & ~mb) | (0x04040404u & mb);                                            // codes > 4 read as N
                const int nv = (int)L - (int)(32u * j + 4u * t);                               // bases of the read in this dword
                const uint32_t vm = nv >= 4 ? 0xFFFFFFFFu : nv <= 0 ? 0u : (0xFFFFFFFFu >> (8 * (4 - nv)));
                d[t] = (x & vm) | (0x08080808u & ~vm);                                         // 8 = no base
            }
            uint32_t *pmr = pm + (uint64_t)(r0 + rr) * pg.pm_stride, *tbr = tb + (uint64_t)(r0 + rr) * pg.tb_stride;
            uint32_t *fw = pk_lds + (size_t)rr * wpr;
            uint32_t p16[4], nb = 0;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const uint32_t n8 = pk_nibbles(d[2 * h]) | (pk_nibbles(d[2 * h + 1]) << 16);     // base q of the chunk in nibble q
                const uint32_t one = 0x11111111u;
                const uint32_t aa = n8 & one, bb = (n8 >> 1) & one, isn = (n8 >> 2) & one, valid = ~(n8 >> 3) & one;
                const uint32_t na = aa ^ one, nb_ = bb ^ one;
                uint32_t oh = (na & nb_) | ((aa & nb_) << 1) | ((na & bb) << 2) | ((aa & bb) << 3);  // 1 << code
                oh = (oh | isn * 15u) & (valid * 15u);                                          // N -> 15, no base -> 0 (nt2bit, editdistance.c:40)
                if (4 * j + h < pg.nw8) { pmr[4 * j + h] = oh; fw[4 * j + h] = oh; }
                uint32_t t = n8 & 0x33333333u;                                                  // 2-bit codes (N and 'no base' read 0)
                t = (t | (t >> 2)) & 0x0F0F0F0Fu; t = (t | (t >> 4)) & 0x00FF00FFu; p16[h] = (t | (t >> 8)) & 0xFFFFu;
                uint32_t u = isn;
                u = (u | (u >> 3)) & 0x03030303u; u = (u | (u >> 6)) & 0x000F000Fu; u = (u | (u >> 12)) & 0xFFu;
                nb |= u << (8 * h);
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const uint32_t rv = __brev(p16[2 * g] | (p16[2 * g + 1] << 16));               // first base highest; then put each pair back in order
                const uint32_t wd = ((rv & 0x55555555u) << 1) | ((rv >> 1) & 0x55555555u);
                if (2 * j + g < pg.nw16) { tbr[2 * j + g] = wd; fw[pg.nw8 + 2 * j + g] = wd; }
            }
            const uint32_t nmw = __brev(nb);
            tbr[2 * pg.nw16 + j] = nmw; fw[pg.nw8 + pg.nw16 + j] = nmw;
            if (j == 0) { pmr[2 * pg.nw8] = L; tbr[2 * pg.nw16 + 2 * pg.nw32] = L; }
        }
    }
    __syncthreads();
    // ---- phase 2: the reverse-complement strand, word by word, from the forward words ----
    for (uint32_t t = threadIdx.x; t < nr * wpr; t += 256) {
        const uint32_t rr = t / wpr, wi = t - rr * wpr;
        uint32_t L = so[rr + 1] - so[rr];
        if (L > cap) L = cap;
        const uint32_t *fw = pk_lds + (size_t)rr * wpr;
        uint32_t *pmr = pm + (uint64_t)(r0 + rr) * pg.pm_stride, *tbr = tb + (uint64_t)(r0 + rr) * pg.tb_stride;
        if (wi < pg.nw8) {
            // nibbles 8k .. 8k+7 of the reverse strand = forward nibbles L-1-8k down to L-8-8k, each bit-reversed:
            // take forward nibbles a .. a+7 (a = L-8-8k; nibbles before the read are zero) and reverse all 32 bits
            const uint32_t k = wi;
            const int a = (int)L - 8 - 8 * (int)k;
            uint32_t x = 0;
            if (a > -8) {
                if (a >= 0) { const uint32_t j0 = (uint32_t)a >> 3; x = __funnelshift_r(fw[j0], j0 + 1 < pg.nw8 ? fw[j0 + 1] : 0u, 4u * ((uint32_t)a & 7u)); }
                else x = fw[0] << (4 * -a);
            }
            pmr[pg.nw8 + k] = __brev(x);
        } else if (wi < pg.nw8 + pg.nw16) {
            // 16 bases of the reverse strand (first base highest) = forward bases a+15 down to a (a = L-16-16k), complemented; N -> 0
            const uint32_t k = wi - pg.nw8;
            const int a = (int)L - 16 - 16 * (int)k;
            const uint32_t *f2 = fw + pg.nw8, *fn = fw + pg.nw8 + pg.nw16;
            uint32_t y = 0, z = 0;                                          // y: forward bases a..a+15 (a at the top), z: their N flags (a at bit 15)
            if (a > -16) {
                if (a >= 0) {
                    const uint32_t j0 = (uint32_t)a >> 4, s2 = 2u * ((uint32_t)a & 15u);
                    const uint32_t w0 = f2[j0], w1 = j0 + 1 < pg.nw16 ? f2[j0 + 1] : 0u;
                    y = s2 ? (w0 << s2) | (w1 >> (32 - s2)) : w0;
                    const uint32_t n0 = (uint32_t)a >> 5, s1 = (uint32_t)a & 31u;
                    const uint32_t m0 = fn[n0], m1 = n0 + 1 < pg.nw32 ? fn[n0 + 1] : 0u;
                    z = (s1 ? (m0 << s1) | (m1 >> (32 - s1)) : m0) >> 16;
                } else { y = f2[0] >> (2 * -a); z = (fn[0] >> 16) >> (-a); }
            }
            const uint32_t ry = __brev(y);                                  // base a+15 first, bits inside every pair swapped
            uint32_t rev = ~(((ry & 0x55555555u) << 1) | ((ry >> 1) & 0x55555555u));            // pairs back in order, complemented
            uint32_t zz = __brev(z) >> 16;                                  // N flag of base a+15 at bit 15 ... base a at bit 0
            zz = (zz | (zz << 8)) & 0x00FF00FFu; zz = (zz | (zz << 4)) & 0x0F0F0F0Fu;           // spread every flag to a pair of bits
            zz = (zz | (zz << 2)) & 0x33333333u; zz = (zz | (zz << 1)) & 0x55555555u; zz |= zz << 1;
            const int nv = (int)L - 16 * (int)k;                            // bases of the reverse strand in this word
            const uint32_t vm = nv >= 16 ? 0xFFFFFFFFu : nv <= 0 ? 0u : ~(0xFFFFFFFFu >> (2 * nv));
            tbr[pg.nw16 + k] = rev & ~zz & vm;
        } else {
            // 32 N flags of the reverse strand (first base highest) = forward flags a+31 down to a (a = L-32-32k)
            const uint32_t k = wi - pg.nw8 - pg.nw16;
            const int a = (int)L - 32 - 32 * (int)k;
            const uint32_t *fn = fw + pg.nw8 + pg.nw16;
            uint32_t y = 0;
            if (a > -32) {
                if (a >= 0) { const uint32_t n0 = (uint32_t)a >> 5, s1 = (uint32_t)a & 31u; const uint32_t m0 = fn[n0], m1 = n0 + 1 < pg.nw32 ? fn[n0 + 1] : 0u; y = s1 ? (m0 << s1) | (m1 >> (32 - s1)) : m0; }
                else y = fn[0] >> (-a);
            }
            tbr[2 * pg.nw16 + pg.nw32 + k] = __brev(y);
        }
    }
}

void launch_pack(const PackGeom &pg, uint32_t n_reads, const uint8_t *seqs, const uint32_t *offs, uint32_t *pm, uint32_t *tb, hipStream_t st)
{
    if (!n_reads) return;
    const uint32_t rpb = 256 / pg.nw32 < 64 ? 256 / pg.nw32 : 64;
    const uint32_t lds = rpb * (pg.nw8 + pg.nw16 + pg.nw32) * 4u;
    hipLaunchKernelGGL(k_pack, dim3((n_reads + rpb - 1) / rpb), dim3(256), lds, st, pg, n_reads, seqs, offs, pm, tb);
}

// ---------------------------------------------------------------------------------------------
// k_seed
// ---------------------------------------------------------------------------------------------
// One seed = (read, strand, slot).  Every seed needs the W-mer table gather; after it most C intervals are ONE row and are
// finished on the spot against the genome text, while the seeds that still have to walk -- C intervals with several rows
// (repeats) and seeds alive in the R index (k-mers around SNPs, ~1 in 6) -- are minorities.  A wave pays for the longest of
// its 64 lanes, so those walks are not done where they arise: the block collects them in LDS and its first lanes adopt them,
// densely packed (typically one wave of the four keeps walking and the other three retire).
struct SeedCtx {                       // passed BY VALUE everywhere: anything reached through `this` or a reference stays in scratch memory
    const uint32_t *t2, *tn;           // 2-bit codes / N flags of this strand in the tb record
    uint32_t L, k, W, s, wb, nb, w0, w1, w2, n0, n1;
    bool inreg;                        // seeds of up to 33 bases sit in 3 + 2 registers (bases wb*16 .. wb*16+47)
};
__device__ __forceinline__ bool seed_valid(const SeedCtx c) { return c.L >= c.k && c.s + c.k <= c.L; }
__device__ __forceinline__ SeedCtx seed_ctx(const SeedParams &sp, const uint32_t *tb, uint32_t item, uint32_t lkt_len)
{
    SeedCtx c;
    const uint32_t slot = item % sp.spr, rs = item / sp.spr, strand = rs & 1u, r = rs >> 1;
    // the read as k_pack left it: 2-bit codes (first base in the high bits) and 'is N' bits of this strand
    const uint32_t *rec = tb + (uint64_t)r * sp.pg.tb_stride;
    c.t2 = rec + strand * sp.pg.nw16; c.tn = rec + 2 * sp.pg.nw16 + strand * sp.pg.nw32;
    c.L = rec[2 * sp.pg.nw16 + 2 * sp.pg.nw32];
    c.k = (uint32_t)sp.l_seed; c.W = lkt_len; c.s = slot * (uint32_t)sp.l_overlap; c.inreg = c.k <= 33;
    c.wb = c.s >> 4; c.nb = c.s >> 5;
    c.w0 = c.w1 = c.w2 = c.n0 = c.n1 = 0;
    if (seed_valid(c)) { c.w0 = c.t2[c.wb]; c.w1 = c.t2[c.wb + 1]; c.w2 = c.t2[c.wb + 2]; c.n0 = c.tn[c.nb]; c.n1 = c.tn[c.nb + 1]; }     // stays inside the record (PackGeom)
    return c;
}
__device__ __forceinline__ uint32_t seed_base2(const SeedCtx c, uint32_t i)
{
    if (c.inreg && i >= c.s) { const uint32_t rel = i - (c.wb << 4); const uint32_t ws = rel < 16 ? c.w0 : rel < 32 ? c.w1 : c.w2; return (ws >> (30 - 2 * (rel & 15u))) & 3u; }
    return (c.t2[i >> 4] >> (30 - 2 * (i & 15u))) & 3u;
}
__device__ __forceinline__ bool seed_is_n(const SeedCtx c, uint32_t i)
{
    if (c.inreg && i >= c.s) { const uint32_t rel = i - (c.nb << 5); const uint32_t ns = rel < 32 ? c.n0 : c.n1; return (ns >> (31 - (rel & 31u))) & 1u; }
    return (c.tn[i >> 5] >> (31 - (i & 31u))) & 1u;
}

// One-row C interval `row` with head bases s .. s+it still to consume (newest first): the located position, or 0xFFFFFFFF when the
// read and the text in front of that suffix differ.  Everything by value: what a lambda captures by reference ends up in scratch.
__device__ __forceinline__ uint32_t seed_resolve_unique(const uint32_t *__restrict__ c_sa, const uint32_t *__restrict__ text, uint32_t c_seq_len,
                                                        uint32_t s, uint32_t wb, uint32_t nb, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t n0, uint32_t n1,
                                                        int it, uint32_t row, uint32_t &n_aux, const uint4 *__restrict__ c_ctx = nullptr, uint32_t ctx_len = 0)
{
        const uint32_t m = (uint32_t)it + 1u;
        const uint32_t reln = s - (nb << 5);
        const uint64_t vn = ((uint64_t)n0 << 32) | n1;
        if (c_ctx && m <= CTX_N) {
            // the context table's record of the row holds the suffix's position AND the 23 bases in front of it: one load instead of
            // the suffix-array load followed by the text load (record layout: salt_device.h; side B is complete iff CTX_N <= p0 <= ctx_len)
            const uint4 rec = c_ctx[row];
            const uint32_t p0c = rec.x == 0xFFFFFFFFu ? c_seq_len : rec.x;
            if (p0c >= CTX_N && p0c <= ctx_len) {
                n_aux += 1u << 10;
                if (((vn >> (64 - reln - m)) & ((1ull << m) - 1ull)) != 0) return 0xFFFFFFFFu;
                const uint64_t lo = ((uint64_t)rec.y | ((uint64_t)(rec.z & 0x3FFFu) << 32)) >> CTX_N, hi = ((uint64_t)(rec.z >> 14) | ((uint64_t)(rec.w & 0x0FFFFFFFu) << 18)) >> CTX_N;
                uint32_t rlo = 0, rhi = 0;
                for (uint32_t u = 0; u < m; ++u) {                            // read base s + it - u faces genome base p0 - 1 - u
                    const uint32_t rel = s + (uint32_t)it - u - (wb << 4);
                    const uint32_t ws = rel < 16 ? w0 : rel < 32 ? w1 : w2;
                    const uint32_t code = (ws >> (30 - 2 * (rel & 15u))) & 3u;
                    rlo |= (code & 1u) << u; rhi |= (code >> 1) << u;
                }
                const uint32_t mk = (1u << m) - 1u;
                return ((((uint32_t)lo ^ rlo) | ((uint32_t)hi ^ rhi)) & mk) == 0 ? p0c - m : 0xFFFFFFFFu;
            }
        }
        uint32_t p0 = c_sa[row];
        if (p0 == 0xFFFFFFFFu) p0 = c_seq_len;            // row 0: the empty suffix
        bool ok = ((vn >> (64 - reln - m)) & ((1ull << m) - 1ull)) == 0 && p0 >= m;
        uint32_t steps = m;
        for (uint32_t done = 0; done < m && ok; ) {          // at most two pieces of up to 16 bases, the later bases (consumed first) first
            const uint32_t cnt = (m - done) > 16u ? 16u : (m - done);
            const uint32_t r0 = s + (m - done - cnt), t0 = p0 - done - cnt;       // read bases r0 .. r0+cnt-1 against text t0 ..
            const uint32_t rel = r0 - (wb << 4), rr = rel & 15u;
            const uint64_t vr = rel < 16 ? (((uint64_t)w0 << 32) | w1) : (((uint64_t)w1 << 32) | w2);
            const uint32_t xr = (uint32_t)((vr >> (64 - 2 * rr - 2 * cnt)) & ((1ull << (2 * cnt)) - 1ull));
            const uint32_t tj = t0 >> 4, tr = t0 & 15u;
            const uint64_t vt = ((uint64_t)text[tj] << 32) | text[tj + 1];
            n_aux += 1u << 21;                               // one 8-byte text load
            const uint32_t xt = (uint32_t)((vt >> (64 - 2 * tr - 2 * cnt)) & ((1ull << (2 * cnt)) - 1ull));
            const uint32_t diff = xr ^ xt;
            if (diff) { ok = false; steps = done + ((uint32_t)__ffs((int)diff) - 1u) / 2u + 1u; }
            done += cnt;
        }
        (void)steps;
        n_aux += 1u << 10;                                   // one suffix-array load
        return ok ? p0 - m : 0xFFFFFFFFu;
    }

// The rest of a C search (bwt.c:281-309) from interval [kc, lc] with head bases s .. s+i_top still to consume, newest first,
// then the interval-shrinking extension (alnse.c:246-258).  A C interval of ONE row cannot branch any more: the search
// succeeds iff the read's remaining bases equal the text in front of that suffix, so one suffix-array load and one text load
// replace the remaining Occ steps and the seed leaves already located (.w = 2: .x = .y = the genome position).
__device__ __forceinline__ uint4 seed_c_rest(const IndexView &ix, const SeedParams &sp, const SeedCtx c, uint32_t kc, uint32_t lc, int i_top,
                                             uint32_t &n_occ_c, uint32_t &n_aux)
{
    const uint32_t s = c.s;
    const bool uniq = c.inreg && sp.resolve_unique;
    bool alive = true, located = false;
    if (uniq && kc == lc && i_top >= 0) { const uint32_t p = seed_resolve_unique(ix.c_sa, ix.text, ix.c_seq_len, s, c.wb, c.nb, c.w0, c.w1, c.w2, c.n0, c.n1, i_top, kc, n_aux, ix.c_ctx, ix.c_seq_len < ix.ref_len ? ix.c_seq_len : ix.ref_len); alive = p != 0xFFFFFFFFu; located = alive; if (alive) kc = lc = p; }
    for (int i = i_top; i >= 0 && alive && !located; --i) {
        if (seed_is_n(c, s + (uint32_t)i)) { alive = false; break; }
        const uint32_t b = seed_base2(c, s + (uint32_t)i);
        uint32_t ok, ol; n_occ_c += c_occ2(ix, kc - 1, lc, b, ok, ol);
        { const uint32_t l2 = pick4(ix.c_L2, b); kc = l2 + ok + 1; lc = l2 + ol; } alive = kc <= lc;
        if (uniq && alive && kc == lc && i > 0) { const uint32_t p = seed_resolve_unique(ix.c_sa, ix.text, ix.c_seq_len, s, c.wb, c.nb, c.w0, c.w1, c.w2, c.n0, c.n1, i - 1, kc, n_aux, ix.c_ctx, ix.c_seq_len < ix.ref_len ? ix.c_seq_len : ix.ref_len); alive = p != 0xFFFFFFFFu; located = alive; if (alive) kc = lc = p; }
    }
    if (!alive) return make_uint4(1, 0, 0, 0);
    if (located) return make_uint4(kc, kc, s, 2);
    uint32_t ext = 0;                                         // shrink big intervals leftwards (alnse.c:246-258)
    while (lc - kc > sp.max_seed && ext < s) {
        if (seed_is_n(c, s - ext - 1)) break;
        const uint32_t b = seed_base2(c, s - ext - 1);
        uint32_t ok, ol; n_occ_c += c_occ2(ix, kc - 1, lc, b, ok, ol);
        if (ok + 1 > ol) break;
        { const uint32_t l2 = pick4(ix.c_L2, b); kc = l2 + ok + 1; lc = l2 + ol; } ++ext;
        if (lc - kc <= sp.max_seed) break;
    }
    return make_uint4(kc, lc, s - ext, 1);
}

// The rest of an R search (rbwt.c:619-648) and its extension, which has no N guard (alnse.c:279-291)
__device__ __forceinline__ uint4 seed_r_rest(const IndexView &ix, const SeedParams &sp, const SeedCtx c, uint32_t kr, uint32_t lr, int i_top,
                                             uint32_t &n_occ_r)
{
    const uint32_t s = c.s;
    bool alive = true;
    for (int i = i_top; i >= 0 && alive; --i) {
        if (seed_is_n(c, s + (uint32_t)i)) { alive = false; break; }
        const uint32_t b = seed_base2(c, s + (uint32_t)i);
        uint32_t ok, ol; n_occ_r += r_occ2(ix, kr, lr + 1, b, ok, ol);
        { const uint32_t cm = pick5(ix.r_cum, b); kr = cm + ok + 1; lr = cm + ol; } alive = kr <= lr;
    }
    if (!alive) return make_uint4(1, 0, 0, 0);
    uint32_t ext = 0;
    while (lr - kr > sp.max_seed && ext < s) {
        const uint32_t b = seed_is_n(c, s - ext - 1) ? 4u : seed_base2(c, s - ext - 1);               // an N walks the '#' column
        uint32_t ok, ol; n_occ_r += r_occ2(ix, kr, lr + 1, b, ok, ol);
        if (ok + 1 > ol) break;
        { const uint32_t cm = pick5(ix.r_cum, b); kr = cm + ok + 1; lr = cm + ol; } ++ext;
        if (lr - kr <= sp.max_seed) break;
    }
    return make_uint4(kr, lr, s - ext, 1);
}

static constexpr uint32_t WQ_SEG = 64, WQ_STRIDE = 64;            // segments per walk queue; words between their counters
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_seed(IndexView ix, SeedParams sp, const uint32_t *__restrict__ tb,
       uint4 *__restrict__ sai_c, uint4 *__restrict__ sai_r, uint4 *__restrict__ wq, uint32_t *__restrict__ wq_cnt, uint32_t wq_seg_cap,
       unsigned long long *__restrict__ ctr)
{
    // walks still to do go to the queues k_seed_walk drains: [0] R searches, [1] C searches.  The block counts its walks in LDS, reserves
    // their slots in ONE of WQ_SEG segments per list with one atomic (an atomic on one address is served every ~11 ns: 31 000 blocks on one
    // counter would be a third of a millisecond) and every lane stores its own record.
    __shared__ uint32_t q_n[2], q_base[2];
    const uint64_t item = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, n_items = (uint64_t)sp.n_reads * 2u * sp.spr;
    const uint64_t lt = (1ull << lane_id()) - 1ull;
    uint32_t n_lkt = 0, n_occ_c = 0, n_occ_r = 0;
    if (threadIdx.x < 2) { q_n[threadIdx.x] = 0; q_base[threadIdx.x] = 0; }
    __syncthreads();
    bool pend_c = false, pend_r = false;
    uint32_t pk_c = 0, pl_c = 0, pk_r = 0, pl_r = 0, at_r = 0, at_c = 0;
    if (item < n_items) {
        const SeedCtx c = seed_ctx(sp, tb, (uint32_t)item, ix.r_lkt_len);
        uint4 oc = make_uint4(1, 0, 0, 0), orr = make_uint4(1, 0, 0, 0);
        if (seed_valid(c)) {
            const uint32_t e = c.s + c.k - 1, W = c.W;
            // W-mer at the seed tail: both searches start from their tabulated interval
            // (LKT_seq2LktItem / LKT_lookup_sa lookup.c:163-177 + the first steps of bwt.c:281-309, rbwt.c:619-648)
            uint32_t x = 0; bool has_n = false;
            const uint32_t a0 = e - W + 1;
            if (c.inreg) {
                const uint32_t rel = a0 - (c.wb << 4), rr = rel & 15u;
                const uint64_t v = rel < 16 ? (((uint64_t)c.w0 << 32) | c.w1) : rel < 32 ? (((uint64_t)c.w1 << 32) | c.w2) : ((uint64_t)c.w2 << 32);
                x = (uint32_t)((v >> (64 - 2 * rr - 2 * W)) & ((1ull << (2 * W)) - 1ull));
                const uint32_t reln = a0 - (c.nb << 5);
                const uint64_t vn = ((uint64_t)c.n0 << 32) | c.n1;
                has_n = ((vn >> (64 - reln - W)) & ((1ull << W) - 1ull)) != 0;
            } else {
                for (uint32_t t = 0; t < W; ++t) { has_n |= seed_is_n(c, a0 + t); x = (x << 2) | seed_base2(c, a0 + t); }
            }
            if (!has_n) {
                const uint4 v = ix.wlkt[2 * (uint64_t)x];    // one 32-byte gather (one sector): C interval in .x/.y, R interval in .z/.w,
                const uint4 u = ix.wlkt[2 * (uint64_t)x + 1];//   and for a one-row C interval its genome position + the 16 bases in front of it
                ++n_lkt;                                     // device counters: bits 0..9 W-mer gathers, 10..20 SA loads, 21..31 text loads
                const int i_top = (int)(c.k - W) - 1;        // head bases s .. s+i_top are still to consume
                if (v.x <= v.y) {
                    const uint32_t m = c.k - W;
                    if (v.x == v.y && c.inreg && sp.resolve_unique && m >= 1 && m <= 16) {
                        // one row: the search succeeds iff the m bases in front of the W-mer equal the text in front of that suffix
                        // (what the remaining steps of bwt_match_exact_alt, bwt.c:281-309, decide) -- both are in registers
                        const uint32_t rel = c.s - (c.wb << 4), rr = rel & 15u;
                        const uint64_t vr = rel < 16 ? (((uint64_t)c.w0 << 32) | c.w1) : (((uint64_t)c.w1 << 32) | c.w2);
                        const uint32_t mk = m == 16 ? 0xFFFFFFFFu : ((1u << (2 * m)) - 1u);
                        const uint32_t xr = (uint32_t)(vr >> (64 - 2 * rr - 2 * m)) & mk;
                        const uint32_t reln2 = c.s - (c.nb << 5);
                        const uint64_t vn2 = ((uint64_t)c.n0 << 32) | c.n1;
                        const bool head_n = ((vn2 >> (64 - reln2 - m)) & ((1ull << m) - 1ull)) != 0;
                        if (!head_n && u.x >= m && xr == (u.y & mk)) oc = make_uint4(u.x - m, u.x - m, c.s, 2);
                    }
                    else if (v.y == v.x + 1u && c.inreg && sp.resolve_unique && m >= 1 && m <= 16) {
                        // two rows (a tenth of the seeds at W = 16 on 3.1e9 bases: chance repeats of the 16-mer): the entry holds both
                        // suffixes' positions and the bases in front of them.  The m steps still to do keep exactly the rows whose text
                        // continues like the read: none -> dead, one -> that row, located; both -> the walk decides (a real repeat)
                        const uint32_t rel = c.s - (c.wb << 4), rr = rel & 15u;
                        const uint64_t vr = rel < 16 ? (((uint64_t)c.w0 << 32) | c.w1) : (((uint64_t)c.w1 << 32) | c.w2);
                        const uint32_t mk = m == 16 ? 0xFFFFFFFFu : ((1u << (2 * m)) - 1u);
                        const uint32_t xr = (uint32_t)(vr >> (64 - 2 * rr - 2 * m)) & mk;
                        const uint32_t reln2 = c.s - (c.nb << 5);
                        const uint64_t vn2 = ((uint64_t)c.n0 << 32) | c.n1;
                        const bool head_n = ((vn2 >> (64 - reln2 - m)) & ((1ull << m) - 1ull)) != 0;
                        const bool ok0 = !head_n && u.x >= m && xr == (u.y & mk), ok1 = !head_n && u.z >= m && xr == (u.w & mk);
                        if (ok0 && ok1) { pend_c = true; pk_c = v.x; pl_c = v.y; }
                        else if (ok0) oc = make_uint4(u.x - m, u.x - m, c.s, 2);
                        else if (ok1) oc = make_uint4(u.z - m, u.z - m, c.s, 2);
                    }
                    else if ((v.x == v.y && c.inreg && sp.resolve_unique) || i_top < 0) oc = seed_c_rest(ix, sp, c, v.x, v.y, i_top, n_occ_c, n_lkt);   // no walk left (or only the extension)
                    else { pend_c = true; pk_c = v.x; pl_c = v.y; }
                }
                if (!sp.seed_only_ref) {
                    if (v.z <= v.w) { pend_r = true; pk_r = v.z; pl_r = v.w; }
                }
            }
        }
        if (!pend_c) sai_c[item] = oc;
        if (!pend_r) sai_r[item] = orr;
    }
    // collect: one LDS atomic per wave and list
    {
        const uint64_t mr = __ballot(pend_r), mc = __ballot(pend_c);
        uint32_t br = 0, bc = 0;
        if (lane_id() == 0) { if (mr) br = atomicAdd(&q_n[0], (uint32_t)__popcll(mr)); if (mc) bc = atomicAdd(&q_n[1], (uint32_t)__popcll(mc)); }
        br = (uint32_t)__shfl((int)br, 0); bc = (uint32_t)__shfl((int)bc, 0);
        at_r = br + (uint32_t)__popcll(mr & lt); at_c = bc + (uint32_t)__popcll(mc & lt);
    }
    __syncthreads();
    const uint32_t seg = blockIdx.x & (WQ_SEG - 1u);
    if (threadIdx.x < 2 && q_n[threadIdx.x]) q_base[threadIdx.x] = atomicAdd(&wq_cnt[(threadIdx.x * WQ_SEG + seg) * WQ_STRIDE], q_n[threadIdx.x]);
    __syncthreads();
    if (pend_r) wq[(size_t)seg * wq_seg_cap + q_base[0] + at_r] = make_uint4((uint32_t)item, pk_r, pl_r, 0u);
    if (pend_c) wq[((size_t)WQ_SEG + seg) * wq_seg_cap + q_base[1] + at_c] = make_uint4((uint32_t)item, pk_c, pl_c, 0u);
    if (ctr) {                                                // one atomic per wave and counter
        for (int o = 32; o > 0; o >>= 1) {
            n_lkt += __shfl_down(n_lkt, o); n_occ_c += __shfl_down(n_occ_c, o); n_occ_r += __shfl_down(n_occ_r, o);
        }
        if (lane_id() == 0) {
            atomicAdd(ctr + SALT_CTR_LKT, n_lkt & 1023u); atomicAdd(ctr + SALT_CTR_OCC_C, n_occ_c); atomicAdd(ctr + SALT_CTR_OCC_R, n_occ_r);
            atomicAdd(ctr + SALT_CTR_D_WLKT, n_lkt & 1023u); atomicAdd(ctr + SALT_CTR_D_COCC_SEED, n_occ_c); atomicAdd(ctr + SALT_CTR_D_ROCC_SEED, n_occ_r);
            atomicAdd(ctr + SALT_CTR_D_SA_SEED, (n_lkt >> 10) & 2047u); atomicAdd(ctr + SALT_CTR_D_TEXT_SEED, n_lkt >> 21);
        }
    }
}

// k_seed_walk: the walks k_seed queued, one per lane and lanes densely packed: R searches first, then C searches (similar lengths side by
// side: a wave pays for the longest of its 64 walks).  Inside k_seed a block of 256 seeds kept ~70 lanes busy with walks while it held its
// wave slots; here every lane of every resident wave has one, which is what raises the number of requests in flight.  Block b serves
// segment b % WQ_SEG with the other blocks of that segment, in strides.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_seed_walk(IndexView ix, SeedParams sp, const uint32_t *__restrict__ tb, uint4 *__restrict__ sai_c, uint4 *__restrict__ sai_r,
            const uint4 *__restrict__ wq, const uint32_t *__restrict__ wq_cnt, uint32_t wq_seg_cap, unsigned long long *__restrict__ ctr)
{
    const uint32_t seg = blockIdx.x & (WQ_SEG - 1u), bb = blockIdx.x / WQ_SEG, nbb = gridDim.x / WQ_SEG;
    uint32_t n_aux = 0, n_occ_c = 0, n_occ_r = 0;
    for (uint32_t which = 0; which < 2; ++which) {
        const uint32_t n = wq_cnt[(which * WQ_SEG + seg) * WQ_STRIDE];
        const uint4 *q = wq + ((size_t)which * WQ_SEG + seg) * wq_seg_cap;
        for (uint32_t t = bb * 256u + threadIdx.x; t < n; t += nbb * 256u) {
            const uint4 e = q[t];
            const SeedCtx c = seed_ctx(sp, tb, e.x, ix.r_lkt_len);
            const int i_top = (int)(c.k - c.W) - 1;
            if (which == 0) sai_r[e.x] = seed_r_rest(ix, sp, c, e.y, e.z, i_top, n_occ_r);
            else sai_c[e.x] = seed_c_rest(ix, sp, c, e.y, e.z, i_top, n_occ_c, n_aux);
        }
    }
    if (ctr) {
        for (int o = 32; o > 0; o >>= 1) { n_aux += __shfl_down(n_aux, o); n_occ_c += __shfl_down(n_occ_c, o); n_occ_r += __shfl_down(n_occ_r, o); }
        if (lane_id() == 0) {
            atomicAdd(ctr + SALT_CTR_OCC_C, n_occ_c); atomicAdd(ctr + SALT_CTR_OCC_R, n_occ_r);
            atomicAdd(ctr + SALT_CTR_D_COCC_SEED, n_occ_c); atomicAdd(ctr + SALT_CTR_D_ROCC_SEED, n_occ_r);
            atomicAdd(ctr + SALT_CTR_D_SA_SEED, (n_aux >> 10) & 2047u); atomicAdd(ctr + SALT_CTR_D_TEXT_SEED, n_aux >> 21);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_align: LDS of one wave
// ---------------------------------------------------------------------------------------------
template <int NS> struct SaiListsT { uint32_t sp[2][NS], ep[2][NS], off[2][NS]; };     // [0]=C, [1]=R
struct SaiLists { uint32_t *sp[2], *ep[2], *off[2]; };                                // the same lists by reference (the sort replica below)
template <int NS> __device__ __forceinline__ SaiLists sai_ref(SaiListsT<NS> &a) { return SaiLists{ { a.sp[0], a.sp[1] }, { a.ep[0], a.ep[1] }, { a.off[0], a.off[1] } }; }
struct LvTables { short L[LVK][64]; char A[LVK][64]; };
// lane-parallel LV (one candidate per lane): nibble-packed text window per lane + two DP rows per lane
static constexpr int LLV_K = 12;                 // handles k <= 12 (reads up to 129 bp at k = L/10)
static constexpr int LLV_TW = 22;                // words per lane: up to 168 text nibbles (+1 pad word): reads up to 164 bases
static constexpr int LLV_W = 2 * LLV_K + 3;      // diagonals -k-1 .. k+1
static constexpr int LLV_N = 64;                 // candidates per round: one per lane
struct LaneLv { uint32_t T[LLV_N * LLV_TW]; uint8_t rows[2][LLV_W][LLV_N]; };
struct LvBytes { uint8_t T[MAXL + 4 + 64]; uint8_t P[MAXL + 64]; };
// NS: seed slots per list the block can hold; LLV: with the lane-LV scratch (9 KB) for gapped passes finished inside the block.
// k_heavy's usual shape is <32, false> = 7.5 KB: its LDS is what decides how much of a CU it leaves to the kernels of the other
// batches in flight (18.6 KB per block with 512 slots and the lane-LV scratch: 303-345 Mreads/s on the 4-stream step; 7.5 KB: 383-388)
template <int NS, bool LLV>
struct WaveLdsT {
    static constexpr bool HAS_LLV = LLV;
    static constexpr int N_SLOTS = NS;
    uint32_t pm[2][MAXL / 8];                               // one-hot nibble words of the read, both strands (k_pack)
    union U { SaiListsT<NS> sai; typename std::conditional<LLV, LaneLv, uint32_t>::type llv; LvBytes lvb; } u;     // seeds | lane-LV scratch | wave-LV byte strings
    uint8_t  cand_e[MAXLOC];
    uint32_t loci[MAXLOC];
    uint32_t hit_pos[2][NHIT];
    uint8_t  hit_nd[2][NHIT], hit_gap[2][NHIT];
    uint16_t cig[SALT_MAX_CIGAR_OPS];
    int      n_cig;
    alignas(8) uint8_t cargs[128];                          // build_candidates' arguments (CandArgs): see there
};
typedef WaveLdsT<SLOTS, true> WaveLds;                      // any read the ABI admits, everything finished inside the block
typedef WaveLdsT<32, false> WaveLdsSmall;                   // <= 32 seed slots per strand; a gapped read without a k_gap slot goes to the overflow queue
typedef WaveLdsT<1, false> WaveLdsCigar;                    // k_cigar: the read, the byte strings of one traceback, the CIGAR

// ---- klib introsort replica on the (sp,ep,off) triple arrays (ksort.h:159-228) ------------------
struct Sai { uint32_t sp, ep, off; };
__device__ __forceinline__ Sai sai_get(const SaiLists &s, int w, int i) { return Sai{ s.sp[w][i], s.ep[w][i], s.off[w][i] }; }
__device__ __forceinline__ void sai_set(SaiLists &s, int w, int i, Sai v) { s.sp[w][i] = v.sp; s.ep[w][i] = v.ep; s.off[w][i] = v.off; }
__device__ __forceinline__ bool sai_ltv(Sai a, Sai b) { return a.ep - a.sp < b.ep - b.sp; }       // alnse.c:35
__device__ __forceinline__ bool sai_lt(const SaiLists &s, int w, int i, int j) { return sai_ltv(sai_get(s, w, i), sai_get(s, w, j)); }
__device__ __forceinline__ void sai_swap(SaiLists &s, int w, int i, int j) { Sai a = sai_get(s, w, i), b = sai_get(s, w, j); sai_set(s, w, i, b); sai_set(s, w, j, a); }

__device__ void sai_insertsort(SaiLists &s, int w, int lo, int hi)
{
    for (int i = lo + 1; i < hi; ++i)
        for (int j = i; j > lo && sai_lt(s, w, j, j - 1); --j) sai_swap(s, w, j, j - 1);
}

__device__ void sai_combsort(SaiLists &s, int w, int base, int n)
{
    const double shrink = 1.2473309501039786540366528676643;
    int gap = n; bool do_swap;
    do {
        if (gap > 2) { gap = (int)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
        do_swap = false;
        for (int i = 0; i < n - gap; ++i)
            if (sai_lt(s, w, base + i + gap, base + i)) { sai_swap(s, w, base + i, base + i + gap); do_swap = true; }
    } while (do_swap || gap > 2);
    if (gap != 1) sai_insertsort(s, w, base, base + n);
}

__device__ __attribute__((noinline)) void sai_introsort(SaiLists &s, int w, int n)
{
    if (n < 1) return;
    if (n == 2) { if (sai_lt(s, w, 1, 0)) sai_swap(s, w, 0, 1); return; }
    int d; for (d = 2; (1 << d) < n; ++d) { }
    d <<= 1;
    int st_l[24], st_r[24], st_d[24], top = 0;
    int lo = 0, hi = n - 1;
    for (;;) {
        if (lo < hi) {
            if (--d == 0) { sai_combsort(s, w, lo, hi - lo + 1); hi = lo; continue; }
            int i = lo, j = hi, k = i + ((j - i) >> 1) + 1;
            if (sai_lt(s, w, k, i)) { if (sai_lt(s, w, k, j)) k = j; }
            else k = sai_lt(s, w, j, i) ? i : j;
            Sai rp = sai_get(s, w, k);
            if (k != hi) sai_swap(s, w, k, hi);
            for (;;) {
                do ++i; while (sai_ltv(sai_get(s, w, i), rp));
                do --j; while (i <= j && sai_ltv(rp, sai_get(s, w, j)));
                if (j <= i) break;
                sai_swap(s, w, i, j);
            }
            sai_swap(s, w, i, hi);
            if (i - lo > hi - i) {
                if (i - lo > 16 && top < 24) { st_l[top] = lo; st_r[top] = i - 1; st_d[top] = d; ++top; }
                lo = hi - i > 16 ? i + 1 : hi;
            } else {
                if (hi - i > 16 && top < 24) { st_l[top] = i + 1; st_r[top] = hi; st_d[top] = d; ++top; }
                hi = i - lo > 16 ? i - 1 : lo;
            }
        } else {
            if (top == 0) { sai_insertsort(s, w, 0, n); return; }
            --top; lo = st_l[top]; hi = st_r[top]; d = st_d[top];
        }
    }
}

// ---- wave-wide bitonic sort of loci[0..n) in LDS (ks_introsort(uint32_t): any total sort) --------
__device__ void sort_loci(uint32_t *a, uint32_t n)
{
    if (n < 2) return;
    uint32_t P = 2; while (P < n) P <<= 1;
    for (uint32_t i = n + lane_id(); i < P; i += 64) a[i] = 0xFFFFFFFFu;
    WSYNC();
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = lane_id(); t < (P >> 1); t += 64) {
                uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));       // element with bit j clear
                uint32_t p = i | j;
                uint32_t x = a[i], y = a[p];
                bool up = (i & k) == 0;
                if ((x > y) == up) { a[i] = y; a[p] = x; }
            }
            WSYNC();
        }
    }
}

// drop duplicates and out-of-range loci of a sorted list, keeping order (alnse.c:758-762 / 890-894)
__device__ __forceinline__ uint32_t dedup_loci(uint32_t *loci, uint32_t n, bool gap_mode, uint32_t L, uint32_t ref_len)
{
    const uint32_t lane = lane_id();
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t n_out = 0;
    for (uint32_t b = 0; b < n; b += 64) {
        uint32_t i = b + lane;
        bool keep = false; uint32_t pos = 0;
        if (i < n) {
            pos = loci[i];
            bool dup = i > 0 && loci[i - 1] == pos;
            bool out = gap_mode ? (pos + L + 4 >= ref_len) : (pos >= ref_len);
            keep = !dup && !out;
        }
        uint64_t m = __ballot(keep);
        WSYNC();                                           // all reads of this chunk are done
        if (keep) loci[n_out + (uint32_t)__popcll(m & lt)] = pos;
        n_out += (uint32_t)__popcll(m);
        WSYNC();
    }
    return n_out;
}

// ---- candidates of one strand: gather seeds, order them, locate, sort, dedup ----------------------
// Leaves the candidate positions in w.loci[0..return).  gap_mode selects the range filter of
// alnse_check_withgap (alnse.c:894) instead of alnse_check_nogap's (alnse.c:762).
struct CandStats { uint32_t n_cand, n_sa_c, n_sa_r, n_loci; uint32_t in_lds; uint32_t n_ctx_rej, n_ctx_rows; };     // in_lds: the loci went to w.loci (PE lists that fit)
struct CandArgs {                      // everything by value: a by-reference IndexView would live in scratch memory
    const uint32_t *c_sa, *r_pos; const uint4 *sai_c, *sai_r;
    uint32_t ref_len, spr, max_locate, r, L; int strand; bool gap_mode; unsigned long long *phase;
    uint32_t *loci; uint32_t loci_cap; int pe;
    bool finish;                   // false: stop after locate (unsorted, duplicates and out-of-range loci still in)
    const uint4 *r_ctx;                    // non-null (with c_ctx): the R rows' records, used the same way
    const uint4 *c_ctx; uint32_t ctx_k;    // non-null: rows come from the context table and a row whose window has more than 3 mismatches for
                                           // certain (ctx_reject, salt_device.h) is counted against the caps but not stored.  Gap-free pass only.
};

// The read's side of a context comparison for a seed that starts at read offset `off` (pm: the strand's one-hot words in LDS).
// Lane t < CTX_N faces genome base s + ctx_k + t, i.e. read base off + ctx_k + t; lane CTX_N + u faces s - 1 - u, read base off - 1 - u.
__device__ __forceinline__ CtxRead ctx_read(const uint32_t *pm, uint32_t L, uint32_t off, uint32_t ctx_k)
{
    const uint32_t t = lane_id();
    const int p = t < CTX_N ? (int)(off + ctx_k + t) : (int)off - 1 - (int)(t - CTX_N);
    const bool v = t < 2u * CTX_N && p >= 0 && p < (int)L;
    const uint32_t nib = v ? (pm[(uint32_t)p >> 3] >> (4u * ((uint32_t)p & 7u))) & 15u : 0u;
    CtxRead rd;
    rd.lo = __ballot(v && (nib & 0xAu) != 0);                // one-hot 1 << code: code bit 0 set for 2 and 8, bit 1 for 4 and 8
    rd.hi = __ballot(v && (nib & 0xCu) != 0);
    rd.use = __ballot(v && (nib == 1u || nib == 2u || nib == 4u || nib == 8u));     // N (15) matches everything: not counted
    return rd;
}
// Not inlined (three call sites, ~1 400 instructions).  Its arguments are the same for all 64 lanes: passed by value they would be
// written to and read back from scratch memory once per lane and call (~6 KB per call, measured as 430 MB of writes per launch), so the
// caller leaves ONE copy in the wave's LDS.
template <bool PE, bool GL, class W>       // GL: the block has a global list for located rows beyond the LDS list (paired end; single end with -m above it).
__device__ __attribute__((noinline)) CandStats build_candidates(W &w)      // A template parameter, not a run-time flag: a list that may live in either memory is reached through flat loads
{
    const CandArgs a = *reinterpret_cast<const CandArgs *>(w.cargs);
    const uint32_t lane = lane_id();
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint32_t L = a.L, r = a.r; const int strand = a.strand; const bool gap_mode = a.gap_mode;
    const gp_u32x4 sai_c = as_global(a.sai_c), sai_r = as_global(a.sai_r);
    struct { gp_u32 c_sa, r_pos; uint32_t ref_len; } ix = { as_global(a.c_sa), as_global(a.r_pos), a.ref_len };
    struct { uint32_t spr, max_locate; } ap = { a.spr, a.max_locate };
    uint32_t n_sa_c = 0, n_sa_r = 0, n_loci_out = 0;
    constexpr bool glob = GL;                                 // a global list is there: used when the rows outgrow the LDS
    uint32_t *loci = glob ? a.loci : w.loci;                 // decided below, once the number of rows is known
    PhaseClock pc(a.phase);
    const uint64_t base_item = ((uint64_t)r * 2u + (uint32_t)strand) * ap.spr;
    uint32_t n_list[2] = { 0, 0 };
    const uint32_t cap_total = PE ? a.loci_cap : ap.max_locate;
    uint32_t rows = 0;                                        // suffix-array rows the locate loops would enumerate (saturating)
    // gather valid seeds in seed order (n_C / n_back_R grow in seed_start order, alnse.c:265-300)
    for (uint32_t b = 0; b < ap.spr; b += 64) {
        const uint32_t slot = b + lane;
        uint4 v2[2] = { make_uint4(1, 0, 0, 0), make_uint4(1, 0, 0, 0) };
        if (slot < ap.spr) {                                  // both lists' loads are in flight together
            const u32x4_t tc = sai_c[base_item + slot], tr = sai_r[base_item + slot];
            v2[0] = make_uint4(tc.x, tc.y, tc.z, tc.w); v2[1] = make_uint4(tr.x, tr.y, tr.z, tr.w);
        }
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const uint4 v = v2[which];
            const uint32_t n = n_list[which];
            uint64_t m = __ballot(v.w != 0);
            uint32_t sz = 0;
            if (v.w) {
                uint32_t at = n + (uint32_t)__popcll(m & lt); w.u.sai.sp[which][at] = v.x; w.u.sai.ep[which][at] = v.y; w.u.sai.off[which][at] = v.z | (v.w == 2 ? 0x80000000u : 0u);
                sz = v.y - v.x + 1u;
                if (which == 1 && !PE) { uint32_t skip = sz / 0x40000u; if (skip > 1) sz = (sz + skip - 1) / skip; }
                if (PE && which == 0 && sz > ap.max_locate + 1u) sz = ap.max_locate + 1u;              // rows the per-interval cap lets through (alnse.c:523)
                if (PE && which == 1 && v.y - v.x > ap.max_locate) sz = (v.y - v.x) / ((v.y - v.x) / ap.max_locate) + 1u;   // rows the subsampling visits
                if (sz > cap_total) sz = cap_total + 1u;
            }
            for (int o = 32; o > 0; o >>= 1) sz += (uint32_t)__shfl_xor((int)sz, o);
            rows = rows + sz > cap_total ? cap_total + 1u : rows + sz;
            n_list[which] = n + (uint32_t)__popcll(m);
        }
    }
    WSYNC();
    // The reference orders both lists by interval size (alnse.c:307-308, klib introsort).  The order only decides WHICH
    // rows are located before a cap stops the loops; when every row is located the loci are the same set, and they
    // are sorted right after -- so the (serial) replica of the sort runs only when the cap can bite.
    if (rows > cap_total && lane < 2) { SaiLists sl = sai_ref(w.u.sai); sai_introsort(sl, (int)lane, (int)n_list[lane]); }
    // a PE mate may enumerate 0x40000 loci, which need the global scratch; the usual few hundred stay in LDS like an SE read's (the
    // verify and rule passes then pay one memory round trip per trip instead of two or three)
    const bool pe_in_lds = !glob || rows <= (uint32_t)MAXLOC;
    if (pe_in_lds) loci = w.loci;
    WSYNC();
    pc.stamp(SALT_CTR_T_GATHER);
    // locate: SE under the global max_locate cap (alnse_locate_alt, alnse.c:633-731); PE with the per-interval cap
    // and the 0x40000 global cap of alnse_locate (alnse.c:501-629; here bounded by the scratch capacity)
    uint32_t n = 0, ns = 0, n_ctx_rows = 0;                   // rows that count against the cap; rows stored (ns < n only with the context table)
    bool full = false;
    const gp_u32x4 c_ctx = as_global(a.c_ctx);
    const bool use_ctx = a.c_ctx != nullptr && !gap_mode;             // paired end too: a rejected row counts against both caps like any row (alnse.c:523-533)
    for (uint32_t i = 0; i < n_list[0] && !full; ++i) {
        const uint32_t sp = w.u.sai.sp[0][i], off = w.u.sai.off[0][i] & 0x7FFFFFFFu;
        const bool located = (w.u.sai.off[0][i] >> 31) != 0;                  // k_seed resolved this one-row interval to its position
        uint32_t ep = w.u.sai.ep[0][i];
        if (PE && ep - sp > ap.max_locate) ep = sp + ap.max_locate;             // j - sp <= max_locate (alnse.c:523)
        const bool ctx_here = use_ctx && !located;
        CtxRead rd = { 0, 0, 0 };
        if (ctx_here) rd = ctx_read(w.pm[strand], L, off, a.ctx_k);
        for (uint64_t j0 = sp; j0 <= ep && !full; j0 += 64) {
            pc.add(SALT_CTR_X0, 1);
            uint64_t j = j0 + lane;
            bool in = j <= ep, keep = false, rej = false;
            uint32_t pos = 0;
            if (in) {
                uint32_t sa;
                if (ctx_here) { const u32x4_t rc = c_ctx[j]; sa = rc.x; rej = ctx_reject(make_uint4(rc.x, rc.y, rc.z, rc.w), rd, 3u); }
                else sa = located ? (uint32_t)j : ix.c_sa[j];
                pos = sa - off; keep = !(pos + L > ix.ref_len);                 // u32 wrap as in alnse.c:672-673
            }
            uint64_t m = __ballot(keep);
            uint32_t rank = (uint32_t)__popcll(m & lt);
            const bool st = keep && !rej && n + rank < cap_total;              // inside the cap and not ruled out by its context
            const uint64_t ms = __ballot(st);
            if (st) loci[ns + (uint32_t)__popcll(ms & lt)] = pos;
            ns += (uint32_t)__popcll(ms);
            uint32_t tot = (uint32_t)__popcll(m);
            if (n + tot >= cap_total) {
                // lookups the sequential loop would have made before stopping
                uint64_t last = m;                       // position of the cap_total-th kept lane
                uint32_t need = cap_total - n;           // >= 1
                for (uint32_t q = 1; q < need; ++q) last &= last - 1;
                uint32_t stop_lane = (uint32_t)__ffsll((long long)last) - 1;
                n_sa_c += stop_lane + 1; if (ctx_here) n_ctx_rows += stop_lane + 1;
                n = cap_total; full = true;
            } else { n += tot; const uint32_t looked = (uint32_t)__popcll(__ballot(in)); n_sa_c += looked; if (ctx_here) n_ctx_rows += looked; }
        }
    }
    pc.stamp(SALT_CTR_X2);
    const gp_u32x4 r_ctx = as_global(a.r_ctx);
    const bool use_rctx = use_ctx && a.r_ctx != nullptr;
    for (uint32_t i = 0; i < n_list[1] && !full; ++i) {
        const uint32_t sp = w.u.sai.sp[1][i], ep = w.u.sai.ep[1][i], off = w.u.sai.off[1][i];
        uint32_t skip = (ep + 1 - sp) / 0x40000u;                                       // alnse.c:707-708
        if ((int)skip <= 0) skip = 1;
        // PE: an interval wider than max_locate is subsampled with rand() in the reference (alnse.c:587-595), i.e. its
        // result is not defined; the stand-in keeps the first row of every block of (ep-sp)/max_locate rows
        if (PE) skip = ep - sp > ap.max_locate ? (ep - sp) / ap.max_locate : 1;
        CtxRead rd = { 0, 0, 0 };
        if (use_rctx) rd = ctx_read(w.pm[strand], L, off, a.ctx_k);
        for (uint64_t j0 = sp; j0 <= ep && !full; j0 += 64ull * skip) {
            pc.add(SALT_CTR_X1, 1);
            uint64_t j = j0 + (uint64_t)lane * skip;
            bool in = j <= ep, keep = false, rej = false;
            uint32_t pos = 0;
            if (in) {
                uint32_t rp;
                if (use_rctx) { const u32x4_t rc = r_ctx[j]; rp = rc.x; rej = ctx_reject(make_uint4(rc.x, rc.y, rc.z, rc.w), rd, 3u); }
                else rp = ix.r_pos[j];
                pos = rp - off; keep = !(pos > ix.ref_len || pos + L > ix.ref_len);      // alnse.c:715-717
            }
            uint64_t m = __ballot(keep);
            uint32_t rank = (uint32_t)__popcll(m & lt);
            const bool st = keep && !rej && n + rank < cap_total;                       // inside the cap and not ruled out by its context
            const uint64_t ms = __ballot(st);
            if (st) loci[ns + (uint32_t)__popcll(ms & lt)] = pos;
            ns += (uint32_t)__popcll(ms);
            uint32_t tot = (uint32_t)__popcll(m);
            if (n + tot >= cap_total) {
                uint64_t last = m; uint32_t need = cap_total - n;
                for (uint32_t q = 1; q < need; ++q) last &= last - 1;
                const uint32_t looked = (uint32_t)__ffsll((long long)last);
                n_sa_r += looked; if (use_rctx) n_ctx_rows += looked;
                n = cap_total; full = true;
            } else { n += tot; const uint32_t looked = (uint32_t)__popcll(__ballot(in)); n_sa_r += looked; if (use_rctx) n_ctx_rows += looked; }
        }
    }
    const uint32_t n_ctx_rej = n - ns;
    n_loci_out += n;
    n = ns;                                                   // what the passes below see
    WSYNC();
    bool in_lds = pe_in_lds;
    if (glob && !in_lds && n <= (uint32_t)MAXLOC) {
        // the rows that were ENUMERATED outgrew the LDS list, the rows the context table let through do not: bring them in (one trip now
        // instead of two or three per verify / rule pass)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        for (uint32_t i = lane; i < n; i += 64) w.loci[i] = loci[i];
        loci = w.loci; in_lds = true;
        WSYNC();
    }
    pc.stamp(SALT_CTR_T_LOCATE);
    if (!a.finish) return CandStats{ n, n_sa_c, n_sa_r, n_loci_out, in_lds ? 1u : 0u, n_ctx_rej, n_ctx_rows };
    sort_loci(loci, n);
    WSYNC();
    pc.stamp(SALT_CTR_T_SORT);
    const uint32_t n_out = dedup_loci(loci, n, gap_mode, L, ix.ref_len);
    pc.stamp(SALT_CTR_T_DEDUP);
    return CandStats{ n_out, n_sa_c, n_sa_r, n_loci_out, in_lds ? 1u : 0u, n_ctx_rej, n_ctx_rows };
}

template <bool PE, bool GL, class W>
__device__ __forceinline__ CandStats build_candidates_call(const CandArgs a, W &w)
{
    static_assert(sizeof(CandArgs) <= sizeof(w.cargs), "CandArgs outgrew its LDS slot");
    WSYNC();
    if (lane_id() == 0) *reinterpret_cast<CandArgs *>(w.cargs) = a;
    WSYNC();
    return build_candidates<PE, GL, W>(w);
}

// ---- masked Hamming distance, capped: returns 0..3 or INF (ed_mismatch, editdistance.c:88-163) ----
__device__ __forceinline__ uint32_t mismatch_capped(const IndexView &ix, const uint32_t *pm, uint32_t L, uint32_t pos)
{
    // a locate whose `pos - offset` wrapped below 0 survives the range check the way it does in the reference (alnse.c:672-673) and is
    // only dropped by the candidate rule (pos >= mixRef.l, alnse.c:762): it must not be used as an address
    if (pos >= ix.ref_len) return INF;
    const uint32_t nw = (L + 7) >> 3, w0 = pos >> 3, sh = (pos & 7u) * 4u;
    const uint32_t *ref = ix.ref + w0;
    uint32_t lo = ref[0], mism = 0;
    for (uint32_t j = 0; j < nw; ++j) {
        uint32_t hi = ref[j + 1];
        uint32_t rw = sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
        uint32_t x = rw & pm[j];
        uint32_t nz = (x | (x >> 1) | (x >> 2) | (x >> 3)) & 0x11111111u;
        uint32_t rem = L - j * 8;
        uint32_t vm = rem >= 8 ? 0x11111111u : (0x11111111u >> (4 * (8 - rem)));
        mism += (uint32_t)__popc(vm) - (uint32_t)__popc(nz & vm);
        lo = hi;
    }
    return mism > 3 ? INF : mism;
}

// ---- masked Hamming distance, four lanes per candidate -------------------------------------------
// Each lane of a quad loads 16 contiguous bytes of the candidate's window (one dwordx4), so a wave-wide
// load touches 16 candidates x 64 B instead of 64 lanes x 14 scattered dwords: ~8x fewer cache-line
// lookups per candidate in the texture-address path, which is what bounds this stage.
// Needs (L+7)/8 + 1 <= 16 words, i.e. L <= 120.  Writes min(count, INF) for candidates [0, n) to out[].
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

// mismatches of one candidate, computed by its 4 lanes (sub = 0..3 holds words 4*sub..4*sub+3 of the window).
// pmw: the read's one-hot words of this lane (zero past the read's end), nvalid: bases of the read inside them.
// A base matches when (reference mask & one-hot) != 0 (ed_mismatch, editdistance.c:88-163); matches are counted
// per nibble with one add (bit 3 of (n & 7) + 7 | n is set iff the nibble n is non-zero) and subtracted from nvalid.
// LN = 4 lanes per candidate cover 16 window words (reads up to 120 bases), LN = 8 cover 32 (up to 248 bases: 150-bp mates).
template <int LN = 4>
__device__ __forceinline__ uint32_t quad_mismatch(const u32x4_a4 x, const uint32_t pos, const uint32_t (&pmw)[4], const uint32_t nvalid)
{
    const uint32_t sh = (pos & 7u) * 4u;
    const uint32_t nxt = (uint32_t)__shfl_down((int)x.x, 1);             // first word of the next lane of the quad
    const uint32_t xs[5] = { x.x, x.y, x.z, x.w, nxt };
    uint32_t match = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint32_t w = __funnelshift_r(xs[t], xs[t + 1], sh);        // window word: reference nibbles pos+8j .. pos+8j+7
        const uint32_t y = w & pmw[t];
        match += (uint32_t)__popc((((y & 0x77777777u) + 0x77777777u) | y) & 0x88888888u);
    }
    uint32_t mism = nvalid - match;
    mism += (uint32_t)__shfl_xor((int)mism, 1);
    mism += (uint32_t)__shfl_xor((int)mism, 2);
    if (LN == 8) mism += (uint32_t)__shfl_xor((int)mism, 4);
    return mism;
}

template <int G, int LN = 4>                                             // groups of 64 / LN candidates whose loads are in flight together
__device__ __forceinline__ void verify_quads(const uint32_t *__restrict__ ref, uint32_t ref_len, const uint32_t *pm, uint32_t L,
                                             const uint32_t *cand, uint32_t n, uint8_t *out)
{
    constexpr uint32_t CPW = 64u / LN;                                   // candidates per wave-wide load
    const uint32_t lane = lane_id(), sub = lane & (uint32_t)(LN - 1), q = lane / (uint32_t)LN;
    const uint32_t nw = (L + 7) >> 3;
    const uint32_t nvalid = L > 32u * sub ? (L - 32u * sub < 32u ? L - 32u * sub : 32u) : 0u;
    uint32_t pmw[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) pmw[t] = (4 * sub + t) < nw ? pm[4 * sub + t] : 0u;
    for (uint32_t c0 = 0; c0 < n; c0 += CPW * G) {                      // up to G groups of CPW candidates per trip
        uint32_t pos[G]; u32x4_a4 x[G]; bool act[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (c0 + CPW * g >= n) break;                                // (uniform) nothing left for this group
            const uint32_t c = c0 + CPW * g + q;
            act[g] = c < n;
            pos[g] = act[g] ? cand[c] : 0u;
            if (pos[g] >= ref_len) pos[g] = 0xFFFFFFFFu;                 // wrapped below 0 (see mismatch_capped): no load, INF
            x[g] = *reinterpret_cast<const u32x4_a4 *>(ref + ((pos[g] == 0xFFFFFFFFu ? 0u : pos[g]) >> 3) + 4 * sub);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (c0 + CPW * g >= n) break;
            const uint32_t mism = quad_mismatch<LN>(x[g], pos[g], pmw, nvalid);
            if (act[g] && sub == 0) out[c0 + CPW * g + q] = (uint8_t)((mism > 3 || pos[g] == 0xFFFFFFFFu) ? INF : mism);
        }
    }
}

// Reads of 121 .. 248 bases (150-base mates), two phases.  A window is 76+ bytes: eight lanes per candidate move 128 (two or three
// 64-byte sectors), and nearly every located row of a repeat read is a false one.  Phase 1 looks at the first 120 bases only -- four
// lanes, one 64-byte stretch per candidate, 128 candidates in flight: a candidate with more than 3 mismatches there cannot pass
// ed_mismatch(..., 3) (editdistance.c:88-163 counts over the whole read).  Phase 2 counts the few survivors over the whole read with
// eight lanes each.  Same out[] as verify_quads<G, 8>.
__device__ __forceinline__ void verify_two_phase(const uint32_t *__restrict__ ref, uint32_t ref_len, const uint32_t *pm, uint32_t L,
                                                 const uint32_t *cand, uint32_t n, uint8_t *out)
{
    verify_quads<8, 4>(ref, ref_len, pm, 120u, cand, n, out);                  // partial counts (<= 3) or INF
    WSYNC();
    const uint32_t lane = lane_id(), sub = lane & 7u, q = lane >> 3;
    const uint32_t nw = (L + 7) >> 3;
    const uint32_t nvalid = L > 32u * sub ? (L - 32u * sub < 32u ? L - 32u * sub : 32u) : 0u;
    uint32_t pmw[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) pmw[t] = (4 * sub + t) < nw ? pm[4 * sub + t] : 0u;
    const bool need = 4u * sub < nw + 1u;                                    // this lane's words overlap the window (the rest is never fetched)
    for (uint32_t b = 0; b < n; b += 64) {
        const uint32_t i = b + lane;
        uint64_t m = __ballot(i < n && out[i] != INF);
        while (m) {                                                          // (uniform) eight survivors per trip, one per lane group
            uint64_t mm = m; uint32_t c = 0xFFFFFFFFu;
            for (uint32_t k = 0; k < 8; ++k) {                               // group q takes the q-th survivor
                if (!mm) break;
                const uint32_t bit = (uint32_t)__ffsll((long long)mm) - 1u;
                if (k == q) c = b + bit;
                mm &= mm - 1;
            }
            m = mm;
            const bool act = c != 0xFFFFFFFFu;
            uint32_t pos = act ? cand[c] : 0u;
            if (pos >= ref_len) pos = 0xFFFFFFFFu;                           // (cannot survive phase 1; kept for symmetry with verify_quads)
            u32x4_a4 x = { 0u, 0u, 0u, 0u };
            if (act && need && pos != 0xFFFFFFFFu) x = *reinterpret_cast<const u32x4_a4 *>(ref + (pos >> 3) + 4 * sub);
            const uint32_t mism = quad_mismatch<8>(x, pos, pmw, nvalid);
            if (act && sub == 0) out[c] = (uint8_t)((mism > 3 || pos == 0xFFFFFFFFu) ? INF : mism);
        }
    }
}

// Both strands of one read in the same trips (k_light): candidates c0[0..n0) use pm0, c1[0..n1) use pm1.
template <int LN = 4>
__device__ __forceinline__ void verify_quads_2(const uint32_t *__restrict__ ref, uint32_t ref_len, const uint32_t *pm0, const uint32_t *pm1, uint32_t L,
                                               const uint32_t *c0, uint32_t n0, const uint32_t *c1, uint32_t n1, uint8_t *o0, uint8_t *o1)
{
    constexpr uint32_t CPW = 64u / LN;
    const uint32_t lane = lane_id(), sub = lane & (uint32_t)(LN - 1), q = lane / (uint32_t)LN;
    const uint32_t nw = (L + 7) >> 3, n = n0 + n1;
    const uint32_t nvalid = L > 32u * sub ? (L - 32u * sub < 32u ? L - 32u * sub : 32u) : 0u;
    uint32_t pa[4], pb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { const bool in = (4 * sub + t) < nw; pa[t] = in ? pm0[4 * sub + t] : 0u; pb[t] = in ? pm1[4 * sub + t] : 0u; }
    for (uint32_t b = 0; b < n; b += 4u * CPW) {
        uint32_t pos[4]; u32x4_a4 x[4]; bool act[4], rev[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (b + CPW * g >= n) break;
            const uint32_t c = b + CPW * g + q;
            act[g] = c < n; rev[g] = c >= n0;
            pos[g] = act[g] ? (rev[g] ? c1[c - n0] : c0[c]) : 0u;
            if (pos[g] >= ref_len) pos[g] = 0xFFFFFFFFu;
            x[g] = *reinterpret_cast<const u32x4_a4 *>(ref + ((pos[g] == 0xFFFFFFFFu ? 0u : pos[g]) >> 3) + 4 * sub);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (b + CPW * g >= n) break;
            const uint32_t pw[4] = { rev[g] ? pb[0] : pa[0], rev[g] ? pb[1] : pa[1], rev[g] ? pb[2] : pa[2], rev[g] ? pb[3] : pa[3] };
            const uint32_t mism = quad_mismatch<LN>(x[g], pos[g], pw, nvalid);
            if (act[g] && sub == 0) {
                const uint32_t c = b + CPW * g + q;
                const uint8_t v = (uint8_t)((mism > 3 || pos[g] == 0xFFFFFFFFu) ? INF : mism);
                if (rev[g]) o1[c - n0] = v; else o0[c] = v;
            }
        }
    }
}

// ---- Landau-Vishkin on byte masks, one diagonal per lane ---------------------------------------
// T: text masks (tlen bytes, zero padded), P: one-hot pattern (plen bytes, zero padded).
// Returns e (<= k), or -1.  When tab != nullptr also fills the L / action tables and returns the
// finishing diagonal in d_fin (order 0,-1,1,... LandauVishkin.c:248); otherwise the order is
// irrelevant for the distance (LandauVishkin.c:67).
__device__ __attribute__((noinline)) int lv_wave(const uint8_t *T, int tlen, const uint8_t *P, int plen, int k, LvTables *tab, int &d_fin)
{
    const int lane = (int)lane_id();
    const int d = lane - 31;
    const int ad = d < 0 ? -d : d;
    int Lprev = -2;
    const int end0 = plen < tlen ? plen : tlen;
    if (d == 0) { int i = 0; while (i < end0 && (P[i] & T[i]) != 0) ++i; Lprev = i; }
    const int r0 = __shfl(Lprev, 31);
    if (tab) tab->L[0][lane] = (short)Lprev;
    d_fin = 0;
    if (r0 == end0) return plen > end0 ? plen - end0 : 0;
    if (k > LVK - 1) k = LVK - 1;
    for (int e = 1; e <= k; ++e) {
        int left = __shfl_up(Lprev, 1), right = __shfl_down(Lprev, 1);
        if (lane == 0) left = -2;
        if (lane == 63) right = -2;
        const bool active = ad <= e && ad <= LVK - 1;
        int best = Lprev + 1; char act = 'X';
        if (left > best) { best = left; act = 'D'; }
        if (right + 1 > best) { best = right + 1; act = 'I'; }
        int cur = -2;
        if (active) {
            if (P[best] == T[d + best]) {                     // equality gate (LandauVishkin.c:79,264)
                int end = plen < tlen - d ? plen : tlen - d;
                if (best >= end) best = end;
                else { int i = best; while (i < end && (P[i] & T[d + i]) != 0) ++i; best = i; }
            }
            cur = best;
        }
        if (tab) { tab->L[e][lane] = (short)cur; tab->A[e][lane] = act; }
        uint64_t reach = __ballot(active && cur == plen);
        if (reach) {
            if (tab) {
                // first finishing diagonal in the order 0,-1,1,-2,2,...
                int rank = d == 0 ? 0 : (d < 0 ? 2 * ad - 1 : 2 * ad);
                int my = (active && cur == plen) ? rank : 1000;
                for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(my, o); my = t < my ? t : my; }
                d_fin = my == 0 ? 0 : ((my & 1) ? -((my + 1) >> 1) : (my >> 1));
            }
            return e;
        }
        Lprev = cur;
    }
    return -1;
}

// ---- Landau-Vishkin distance, one candidate per lane (computeEditDistance, LandauVishkin.c:19-122) ----
// Text: the lane's (L+4)-base window of the mixRef, nibble-packed in LDS (zero beyond tlen, like the
// reference's zero-padded byte buffer); pattern: the read's one-hot nibbles (N = 15), shared.
// Returns e in 0..k, or 255 for "more than k" / inactive lanes.  The distance does not depend on the
// diagonal order, so all lanes walk the (e, d) cells in the same order.
__device__ __attribute__((noinline)) uint32_t lv_lanes(LaneLv &s, const uint32_t *pm, int plen, int tlen, int k, bool active)
{
    const int lane = (int)lane_id() & (LLV_N - 1);      // callers pass active = false for lanes >= LLV_N
    const uint32_t *T = s.T + lane * LLV_TW;
    const int nwp = (plen + 7) >> 3;
    // 8 nibbles starting at base i: pattern (zeros past plen) / text (zeros outside [0, tlen); i may be negative)
    auto wP = [&](int i) -> uint32_t {
        const int j = i >> 3;
        const uint32_t lo = j < nwp ? pm[j] : 0u, hi = j + 1 < nwp ? pm[j + 1] : 0u;
        return __funnelshift_r(lo, hi, 4u * (uint32_t)(i & 7));
    };
    auto wT = [&](int i) -> uint32_t {
        if (i < 0) return i <= -8 ? 0u : T[0] << (4 * -i);
        const int j = i >> 3;
        const uint32_t lo = j < LLV_TW ? T[j] : 0u, hi = j + 1 < LLV_TW ? T[j + 1] : 0u;
        return __funnelshift_r(lo, hi, 4u * (uint32_t)(i & 7));
    };
    // bases matching from i on diagonal d, 8 per step: a base matches when mask & one-hot != 0 (LandauVishkin.c:42-53,85-96)
    auto extend = [&](int d, int i, int end) -> int {
        while (i < end) {
            const uint32_t y = wP(i) & wT(d + i);
            const uint32_t z = ~(((y & 0x77777777u) + 0x77777777u) | y) & 0x88888888u;     // bit 3 of a nibble: no match there
            const int run = z ? (__ffs((int)z) - 1) >> 2 : 8;
            i += run;
            if (run < 8) break;
        }
        return i < end ? i : end;
    };
    auto RD = [&](int row, int d) -> int { return (int)s.rows[row][d + LLV_K + 1][lane] - 2; };
    auto WR = [&](int row, int d, int v) { s.rows[row][d + LLV_K + 1][lane] = (uint8_t)(v + 2); };
    for (int row = 0; row < 2; ++row) for (int d = 0; d < LLV_W; ++d) s.rows[row][d][lane] = 0;      // -2 everywhere
    const int end0 = plen < tlen ? plen : tlen;
    uint32_t result = 255;
    bool done = !active;
    if (active) {
        const int i = extend(0, 0, end0);
        WR(0, 0, i);
        if (i == end0) { result = (uint32_t)(plen > end0 ? plen - end0 : 0); done = true; }
    }
    for (int e = 1; e <= k; ++e) {
        if (__ballot(!done) == 0) break;
        const int cur = e & 1, prev = cur ^ 1;
        for (int d = -e; d <= e; ++d) {
            if (done) continue;
            int best = RD(prev, d) + 1;
            const int left = RD(prev, d - 1), right = RD(prev, d + 1) + 1;
            if (left > best) best = left;
            if (right > best) best = right;
            if ((wP(best) & 15u) == (wT(d + best) & 15u)) {               // equality gate (LandauVishkin.c:79)
                const int end = plen < tlen - d ? plen : tlen - d;
                best = best >= end ? end : extend(d, best, end);
            }
            if (best == plen) { result = (uint32_t)e; done = true; }
            else WR(cur, d, best);
        }
    }
    return result;
}

// ---------------------------------------------------------------------------------------------
// The reference's candidate loop (code_kmismatch alnse.c:348-369 / code_kdiff alnse.c:371-393) walks the SORTED,
// duplicate-free list: a candidate with distance v counts iff v <= the running bound, the bound drops to the smallest
// distance seen, the first such candidate and every strictly better one become the best, and the first NHIT that
// count are kept as hits.  Equivalently, with P = {candidates with v <= bound_in, inside the reference}:
//   a candidate counts        iff  no member of P at a smaller position has a smaller distance,
//   best                      =    the smallest position among the members of P with the minimal distance,
//   hits                      =    the NHIT smallest distinct positions that count, a0 = the distance of the first.
// None of this needs the list sorted or free of duplicates (equal positions have equal distances), so the kernels run it
// on the located rows as they come: per distance t the smallest position minpos[t] over P (one wave reduction each), then
// "counts" is  pos < min(minpos[0..v-1]),  and the hits are extracted by repeated minimum.
// pos / val: n entries in LDS or global memory; val > VMAX = no candidate.  GAPF: alnse_check_withgap's range filter.
// ---------------------------------------------------------------------------------------------
template <int VMAX, bool GAPF>
__device__ __forceinline__ void rule_unsorted(const uint32_t *pos, const uint8_t *val, const uint32_t n, const uint32_t L, const uint32_t ref_len,
                                              uint32_t &bound, bool &any, uint32_t &best_pos, uint32_t &best_v,
                                              uint32_t &n_hits, uint32_t &a0, uint32_t *hit_pos, uint8_t *hit_nd)
{
    const uint32_t lane = lane_id();
    const uint32_t NONE = 0xFFFFFFFFu;
    auto in_range = [&](uint32_t p) -> bool { return GAPF ? !(p + L + 4 >= ref_len) : p < ref_len; };
    uint32_t mp[VMAX + 1];
#pragma unroll
    for (int t = 0; t <= VMAX; ++t) mp[t] = NONE;
    // The members of P are few (a repeat read's list: ~900 located rows, a handful inside the bound), and the hits below are six more
    // passes over whatever they are taken from: while P has at most 64 members, member k is copied to lane k (scalar steps per member,
    // none for a trip without one) and the hit passes run on that register copy instead of the list.
    uint32_t my_p = 0, my_v = NONE, n_p = 0;                      // n_p > 64: P outgrew the lanes, the passes walk the list
    for (uint32_t b = 0; b < n; b += 64) {
        const uint32_t i = b + lane;
        uint32_t p = 0, v = NONE;
        if (i < n) { p = pos[i]; v = val[i]; }
        const bool in = i < n && v <= bound && in_range(p);
        if (in) {
#pragma unroll
            for (int t = 0; t <= VMAX; ++t) if (v == (uint32_t)t && p < mp[t]) mp[t] = p;
        }
        uint64_t m = __ballot(in);
        if (m && n_p <= 64u) {
            if (n_p + (uint32_t)__popcll(m) > 64u) n_p = 65u;
            else
                while (m) {
                    const int l = __ffsll((long long)m) - 1;
                    const uint32_t pl = (uint32_t)__builtin_amdgcn_readlane((int)p, l), vl = (uint32_t)__builtin_amdgcn_readlane((int)v, l);
                    if (lane == n_p) { my_p = pl; my_v = vl; }
                    ++n_p; m &= m - 1;
                }
        }
    }
#pragma unroll
    for (int t = 0; t <= VMAX; ++t)
        for (int o = 32; o > 0; o >>= 1) { const uint32_t x = (uint32_t)__shfl_xor((int)mp[t], o); mp[t] = x < mp[t] ? x : mp[t]; }
    // before[t] = smallest position of P with a distance below t
    uint32_t before[VMAX + 1];
    before[0] = NONE;
#pragma unroll
    for (int t = 1; t <= VMAX; ++t) before[t] = mp[t - 1] < before[t - 1] ? mp[t - 1] : before[t - 1];
    any = false; n_hits = 0;
#pragma unroll
    for (int t = VMAX; t >= 0; --t) if (mp[t] != NONE) { any = true; best_v = (uint32_t)t; best_pos = mp[t]; }
    if (!any) return;
    unsigned long long last = 0;
    const bool in_lanes = n_p <= 64u;
    unsigned long long my_key = ~0ull;                            // this lane's member, if it counts (p < the smallest position of a smaller distance)
    if (in_lanes && lane < n_p) {
        uint32_t lim = NONE;
#pragma unroll
        for (int t = 1; t <= VMAX; ++t) if (my_v == (uint32_t)t) lim = before[t];
        if (my_p < lim) my_key = ((unsigned long long)my_p << 8) | my_v;
    }
    for (uint32_t h = 0; h < (uint32_t)NHIT; ++h) {
        unsigned long long cur = ~0ull;
        if (in_lanes) { if (h == 0 || my_key > last) cur = my_key; }
        else
        for (uint32_t b = 0; b < n; b += 64) {
            const uint32_t i = b + lane;
            if (i < n) {
                const uint32_t p = pos[i], v = val[i];
                if (v <= bound && in_range(p)) {
                    uint32_t lim = NONE;
#pragma unroll
                    for (int t = 1; t <= VMAX; ++t) if (v == (uint32_t)t) lim = before[t];
                    const unsigned long long key = ((unsigned long long)p << 8) | v;
                    if (p < lim && (h == 0 || key > last) && key < cur) cur = key;
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long x = ((unsigned long long)(uint32_t)__shfl_xor((int)(cur >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)cur, o);
            cur = x < cur ? x : cur;
        }
        if (cur == ~0ull) break;
        if (lane == 0) { hit_pos[h] = (uint32_t)(cur >> 8); hit_nd[h] = (uint8_t)(cur & 255u); }
        if (h == 0) a0 = (uint32_t)(cur & 255u);
        last = cur; ++n_hits;
    }
    bound = best_v < bound ? best_v : bound;
}

// rule_unsorted for short lists with few candidates inside the bound (k_light): the reductions walk the set bits of a
// ballot with v_readlane, i.e. they cost scalar instructions per candidate instead of six cross-lane steps per value.
template <int VMAX, bool GAPF>
__device__ __forceinline__ void rule_sparse(const uint32_t *pos, const uint8_t *val, const uint32_t n, const uint32_t L, const uint32_t ref_len,
                                            uint32_t &bound, bool &any, uint32_t &best_pos, uint32_t &best_v,
                                            uint32_t &n_hits, uint32_t &a0, uint32_t *hit_pos, uint8_t *hit_nd)
{
    const uint32_t lane = lane_id();
    const uint32_t NONE = 0xFFFFFFFFu;
    auto in_range = [&](uint32_t p) -> bool { return GAPF ? !(p + L + 4 >= ref_len) : p < ref_len; };
    uint32_t mp[VMAX + 1];
#pragma unroll
    for (int t = 0; t <= VMAX; ++t) mp[t] = NONE;
    any = false; n_hits = 0;
    if (n == 0) return;
    if (n <= 64) {                                            // the usual case: every candidate inside the bound is the same locus
        uint32_t p = 0, v = NONE;
        if (lane < n) { p = pos[lane]; v = val[lane]; }
        const bool ok = v <= bound && in_range(p);
        const uint64_t m = __ballot(ok);
        if (m == 0) return;
        const int l0 = __ffsll((long long)m) - 1;
        const uint32_t p0 = (uint32_t)__builtin_amdgcn_readlane((int)p, l0);
        if (__ballot(ok && p != p0) == 0) {
            const uint32_t v0 = (uint32_t)__builtin_amdgcn_readlane((int)v, l0);
            any = true; best_pos = p0; best_v = v0; n_hits = 1; a0 = v0;
            if (lane == 0) { hit_pos[0] = p0; hit_nd[0] = (uint8_t)v0; }
            bound = v0 < bound ? v0 : bound;
            return;
        }
    }
    for (uint32_t b = 0; b < n; b += 64) {
        const uint32_t i = b + lane;
        uint32_t p = 0, v = NONE;
        if (i < n) { p = pos[i]; v = val[i]; }
        uint64_t m = __ballot(v <= bound && in_range(p));
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            const uint32_t pl = (uint32_t)__builtin_amdgcn_readlane((int)p, l), vl = (uint32_t)__builtin_amdgcn_readlane((int)v, l);
#pragma unroll
            for (int t = 0; t <= VMAX; ++t) if (vl == (uint32_t)t && pl < mp[t]) mp[t] = pl;
            m &= m - 1;
        }
    }
    uint32_t before[VMAX + 1];
    before[0] = NONE;
#pragma unroll
    for (int t = 1; t <= VMAX; ++t) before[t] = mp[t - 1] < before[t - 1] ? mp[t - 1] : before[t - 1];
#pragma unroll
    for (int t = VMAX; t >= 0; --t) if (mp[t] != NONE) { any = true; best_v = (uint32_t)t; best_pos = mp[t]; }
    if (!any) return;
    uint32_t last_p = 0;
    for (uint32_t h = 0; h < (uint32_t)NHIT; ++h) {
        uint32_t cur_p = NONE, cur_v = 0;
        for (uint32_t b = 0; b < n; b += 64) {
            const uint32_t i = b + lane;
            uint32_t p = 0, v = NONE;
            if (i < n) { p = pos[i]; v = val[i]; }
            uint32_t lim = NONE;
#pragma unroll
            for (int t = 1; t <= VMAX; ++t) if (v == (uint32_t)t) lim = before[t];
            uint64_t m = __ballot(v <= bound && in_range(p) && p < lim && (h == 0 || p > last_p));
            while (m) {
                const int l = __ffsll((long long)m) - 1;
                const uint32_t pl = (uint32_t)__builtin_amdgcn_readlane((int)p, l);
                if (pl < cur_p) { cur_p = pl; cur_v = (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
                m &= m - 1;
            }
        }
        if (cur_p == NONE) break;
        if (lane == 0) { hit_pos[h] = cur_p; hit_nd[h] = (uint8_t)cur_v; }
        if (h == 0) a0 = cur_v;
        last_p = cur_p; ++n_hits;
    }
    bound = best_v < bound ? best_v : bound;
}

// unpack text masks / one-hot pattern for LV (editdistance.c:183-227)
template <class W>
__device__ void lv_unpack(const uint32_t *ref_generic, W &w, int strand, uint32_t L, uint32_t pos)
{
    const gp_u32 ref = as_global(ref_generic);
    const uint32_t tlen = L + 4;
    for (uint32_t i = lane_id(); i < tlen + 48; i += 64) {
        uint32_t p = pos + i;
        w.u.lvb.T[i] = i < tlen ? (uint8_t)((ref[p >> 3] >> (4 * (p & 7u))) & 15u) : (uint8_t)0;
    }
    for (uint32_t i = lane_id(); i < L + 48; i += 64) {
        w.u.lvb.P[i] = i < L ? (uint8_t)((w.pm[strand][i >> 3] >> (4 * (i & 7u))) & 15u) : (uint8_t)0;
    }
    WSYNC();
}

// CIGAR of a gapped hit into w.cig / w.n_cig (computeEditDistanceWithCigar, useM=1) ------------------
template <class W>
__device__ __attribute__((noinline)) void lv_cigar(const uint32_t *ref, W &w, LvTables *tabp, int strand, uint32_t L, uint32_t pos, int k)
{
    LvTables &tab = *tabp;                  // per-block table in global memory (L2-resident, rare path)
    lv_unpack(ref, w, strand, L, pos);
    int d_fin = 0;
    WSYNC();
    int e = lv_wave(w.u.lvb.T, (int)L + 4, w.u.lvb.P, (int)L, k, &tab, d_fin);
    __threadfence_block();
    WSYNC();
    if (lane_id() == 0) {
        int n = 0;
        uint16_t *cg = w.cig;
        if (e == 0) { cg[n++] = (uint16_t)((L << 4) | 0u); }
        else if (e > 0) {
            char act[LVK + 1]; int matched[LVK + 1];
            int cd = d_fin;
            for (int ce = e; ce >= 1; --ce) {
                char a = tab.A[ce][cd + 31];
                act[ce] = a;
                int cur = tab.L[ce][cd + 31];
                if (a == 'I') { matched[ce] = cur - tab.L[ce - 1][cd + 1 + 31] - 1; cd += 1; }
                else if (a == 'D') { matched[ce] = cur - tab.L[ce - 1][cd - 1 + 31]; cd -= 1; }
                else { matched[ce] = cur - tab.L[ce - 1][cd + 31] - 1; }
            }
            int acc = tab.L[0][31];
            int ce = 1;
            while (ce <= e) {
                char a = act[ce]; int cnt = 1;
                while (ce + 1 <= e && matched[ce] == 0 && act[ce + 1] == a) { ++cnt; ++ce; }
                if (a == 'X') acc += cnt;
                else {
                    if (acc != 0 && n < SALT_MAX_CIGAR_OPS) { cg[n++] = (uint16_t)((acc << 4) | 0); }
                    acc = 0;
                    if (n < SALT_MAX_CIGAR_OPS) cg[n++] = (uint16_t)((cnt << 4) | (a == 'I' ? 1 : 2));
                }
                if (matched[ce] > 0) acc += matched[ce];
                ++ce;
            }
            if (acc != 0 && n < SALT_MAX_CIGAR_OPS) cg[n++] = (uint16_t)((acc << 4) | 0);
        }
        w.n_cig = n;
    }
    WSYNC();
}

// ---------------------------------------------------------------------------------------------
// k_align
// ---------------------------------------------------------------------------------------------
// the text window of one candidate for lv_lanes: tl reference masks from pos, nibble-packed, zero past the end
__device__ __forceinline__ void lane_text(const uint32_t *__restrict__ ref, uint32_t *T, uint32_t pos, uint32_t tl)
{
    const uint32_t w0 = pos >> 3, sh = (pos & 7u) * 4u, nwt = (tl + 7) >> 3;
    // every word of the window is asked for before the first is used (one load, one wait, one store per word was 22 memory round trips
    // in a row per candidate)
    uint32_t wv[LLV_TW + 1];
#pragma unroll
    for (int j = 0; j <= LLV_TW; ++j) wv[j] = (uint32_t)j <= nwt ? ref[w0 + (uint32_t)j] : 0u;
#pragma unroll
    for (int j = 0; j < LLV_TW; ++j) {
        uint32_t word = 0;
        if ((uint32_t)j < nwt) {
            word = __funnelshift_r(wv[j], wv[j + 1], sh);
            const uint32_t rem = tl - (uint32_t)j * 8;
            if (rem < 8) word &= (1u << (4 * rem)) - 1u;
        }
        T[j] = word;
    }
}

// A read without a gap-free hit needs Landau-Vishkin over all its candidates plus LV tracebacks for its CIGARs:
// hundreds of microseconds on one wave, i.e. the tail of a persistent kernel.  k_heavy therefore only stores such a
// read's candidate lists; k_gap computes the distances with one wave per (read, strand, 32 candidates), k_gapfin
// replays the reference's sequential rule over the stored distances and writes the result, and k_cigar runs one
// traceback per wave.  (GapBufs is declared in salt_kernels.h; a read that finds no free slot is finished inline.)
__device__ __forceinline__ uint32_t store_gap_list(const uint32_t *loci, uint32_t n, uint32_t L, uint32_t ref_len, uint32_t *__restrict__ dst)
{
    const uint32_t lane = lane_id();
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t n_out = 0;
    for (uint32_t b = 0; b < n; b += 64) {                    // the gap-free list minus alnse_check_withgap's range filter (alnse.c:894)
        const uint32_t i = b + lane;
        const uint32_t pos = i < n ? loci[i] : 0u;
        const bool keep = i < n && !(pos + L + 4 >= ref_len);
        const uint64_t m = __ballot(keep);
        if (keep) dst[n_out + (uint32_t)__popcll(m & lt)] = pos;
        n_out += (uint32_t)__popcll(m);
    }
    return n_out;
}

template <bool PE, bool GL, class W>
__device__ __forceinline__ void align_general(const IndexView ix, const AlignParams ap, W &w, const uint32_t r,
                              const uint32_t *__restrict__ pm,
                              const uint4 *__restrict__ sai_c, const uint4 *__restrict__ sai_r,
                              salt_result_t *__restrict__ results, unsigned long long *__restrict__ ctr,
                              unsigned long long *phase, LvTables *lvtab, const GapBufs g, uint32_t *pe_loci, uint8_t *pe_cand)
{
    const uint32_t lane = lane_id();
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint32_t *rec = pm + (uint64_t)r * ap.pg.pm_stride;               // k_pack's record of this read
    const uint32_t L = rec[2 * ap.pg.nw8];
    salt_result_t *out = results + r;
    uint32_t c_sa_c = 0, c_sa_r = 0, c_verify = 0, c_vwords = 0, c_lv = 0, c_loci = 0, c_ctx_rej = 0, c_ctx_rows = 0;
    PhaseClock pc(phase);
    const uint64_t rt0 = phase ? __builtin_amdgcn_s_memrealtime() : 0;
    constexpr bool glob = GL;                                 // paired end (0x40000 loci per strand, alnse.c:42) or single end with -m above the LDS list
    uint32_t *loci = glob ? pe_loci : w.loci;                 // candidate loci: LDS, or the block's global scratch when a list outgrows it
    uint8_t *cand_e = glob ? pe_cand : w.cand_e;              // (set after every build_candidates call from what it reports)
    const uint32_t loci_cap = PE ? PE_LOCI_CAP : (uint32_t)MAXLOC;

    // ---- the read: one-hot masks of both strands, 8 bases per word, LSB first like the mixRef (editdistance.c:40) ----
    const uint32_t nw = (L + 7) >> 3;
    uint32_t n_amb = 0;
    for (uint32_t t = lane; t < 2 * nw; t += 64) {
        const uint32_t s = t >= nw, j = s ? t - nw : t;
        const uint32_t word = rec[s * ap.pg.nw8 + j];
        w.pm[s][j] = word;
        if (!s) n_amb += (uint32_t)__popc(word & (word >> 1) & (word >> 2) & (word >> 3) & 0x11111111u);     // N = all four bits
    }
    if (ap.max_amb < L) { for (int o = 32; o > 0; o >>= 1) n_amb += (uint32_t)__shfl_xor((int)n_amb, o); }
    else n_amb = 0;                                           // (uniform) the limit cannot be passed
    WSYNC();
    // result defaults (query_read_seq, query.c:199-206)
    uint32_t q_pos = 0xFFFFFFFFu; uint32_t q_strand = 3, q_ndiff = 255, q_gap = 255;
    const bool too_short = L < (uint32_t)ap.l_seed;
    if (n_amb > ap.max_amb) {                                 // alnse.c:1328 (200) / alnpe.c:495 (5): record left untouched
        if (lane == 0) {
            out->pos = q_pos; out->strand = 3; out->n_diff = 255; out->is_gap = 255; out->mapq = 0;
            out->b0 = -1; out->b1 = -1; out->seq_start = 0; out->seq_end = (uint16_t)(L - 1);
            out->n_hits[0] = out->n_hits[1] = 0; out->n_cigar = 0; out->skipped = 1;
        }
        return;
    }
    pc.stamp(SALT_CTR_T_LOAD);
    // ---- gap-free pass over both strands (alnse.c:1077-1084) ----
    uint32_t bound = 3;
    bool found[2] = { false, false };
    uint32_t n_hits_s[2] = { 0, 0 };            // hits recorded (<= NHIT) per strand
    uint32_t a0[2] = { 0, 0 };                  // n_diff of the first hit of each list
    uint32_t n_cand_nogap = 0, n_loc_s[2] = { 0, 0 };
    if (!too_short)
    for (int strand = 0; strand < 2; ++strand) {
        pc.stamp(SALT_CTR_T_SCAN);
        // Located rows first, unsorted: only loci that can pass (<= 3 mismatches, inside the reference) matter to the
        // sequential rule, so the sort (alnse.c:726-729), the duplicate filter (alnse.c:758-762) and the rule run on
        // those few; the result is the one the full sorted list gives.
        const CandStats cs = build_candidates_call<PE, GL>(CandArgs{ ix.c_sa, ix.r_pos, sai_c, sai_r, ix.ref_len, ap.spr, ap.max_locate, r, L, strand, false, phase, pe_loci, loci_cap, ap.pe, false, ix.r_ctx, ix.c_ctx, ix.ctx_k }, w);
        if (glob) { loci = cs.in_lds ? w.loci : pe_loci; cand_e = cs.in_lds ? w.cand_e : pe_cand; }
        pc.t = phase ? __builtin_amdgcn_s_memtime() : 0;
        const uint32_t n_loc = cs.n_cand; c_sa_c += cs.n_sa_c; c_sa_r += cs.n_sa_r; c_loci += cs.n_loci; c_ctx_rej += cs.n_ctx_rej; c_ctx_rows += cs.n_ctx_rows;
        uint32_t call_best_n = INF, call_best_pos = 0;
        if (SALT_DIAG_VAL(ap.heavy_stop) == 1) continue;
        auto verify_all = [&](uint32_t n) {                   // cand_e[i] = min(mismatches, INF) of loci[i], loads of 128 candidates in flight
            if (L <= 120) verify_quads<8>(ix.ref, ix.ref_len, w.pm[strand], L, loci, n, cand_e);
            else if (L <= 248) verify_two_phase(ix.ref, ix.ref_len, w.pm[strand], L, loci, n, cand_e);
            else for (uint32_t i = lane; i < n; i += 64) cand_e[i] = (uint8_t)mismatch_capped(ix, w.pm[strand], L, loci[i]);
            WSYNC();
        };
        verify_all(n_loc);
        pc.stamp(SALT_CTR_T_VERIFY);
        if (SALT_DIAG_VAL(ap.heavy_stop) == 2) continue;
        {
            bool any = false; uint32_t bp = 0, bv = 0, nh = 0, a0s = 0;
            rule_unsorted<3, false>(loci, cand_e, n_loc, L, ix.ref_len, bound, any, bp, bv, nh, a0s, w.hit_pos[strand], w.hit_nd[strand]);
            if (any) {
                found[strand] = true; call_best_pos = bp; call_best_n = bv; n_hits_s[strand] = nh; a0[strand] = a0s;
                if (lane < NHIT) w.hit_gap[strand][lane] = 0;
            }
        }
        const uint32_t n_cand = n_loc;
        c_verify += n_loc; n_cand_nogap += n_loc + cs.n_ctx_rej; n_loc_s[strand] = n_loc;
        for (uint32_t b = lane; b < n_cand; b += 64) c_vwords += ((loci[b] & 7u) + L + 7) >> 3;
        if (found[strand]) { q_pos = call_best_pos; q_ndiff = call_best_n; q_gap = 0; q_strand = (uint32_t)strand; }
        WSYNC();
        pc.stamp(SALT_CTR_T_SCAN);
    }

    if (SALT_DIAG_VAL(ap.heavy_stop)) return;
    // ---- gapped pass (alnse.c:1089-1096): sequential per candidate, LV across lanes ----
    if (!too_short && !found[0] && !found[1]) {
        int maxd = PE ? 3 : (int)(L / 10);                    // alnse.c:1090 (SE) / alnse.c:1016-1028 (PE keeps 3)
        const int gap_k0 = maxd;
        const bool lanes_fit = gap_k0 <= LLV_K && L + 4 <= 8u * (LLV_TW - 1);
        if (lanes_fit && g.cap && n_cand_nogap > 0) {
            // k_gap / k_gapfin / k_cigar take it from here: both strands' located rows go to the pool as they are (unsorted,
            // duplicates included -- rule_unsorted needs neither).  Strand 1's rows are still in `loci` (unless the context table
            // thinned them: the gapped pass needs every row); strand 0's are located again.  (Single end and paired end alike.)
            uint32_t slot = 0;
            if (lane == 0) slot = atomicAdd(&g.gctl[QC(2)], 1u);
            slot = (uint32_t)__shfl((int)slot, 0);
            if (slot < g.cap) {
                uint32_t ns[2], off[2];
                bool ok = true;
                for (int k = 0; k < 2 && ok; ++k) {
                    const int strand = 1 - k;
                    uint32_t n = n_loc_s[1];
                    if (strand == 0 || ix.c_ctx != nullptr) {
                        WSYNC();
                        const CandStats cs = build_candidates_call<PE, GL>(CandArgs{ ix.c_sa, ix.r_pos, sai_c, sai_r, ix.ref_len, ap.spr, ap.max_locate, r, L, strand, true, phase, pe_loci, loci_cap, ap.pe, false, nullptr, nullptr, 0 }, w);
                        if (glob) { loci = cs.in_lds ? w.loci : pe_loci; cand_e = cs.in_lds ? w.cand_e : pe_cand; }
                        c_sa_c += cs.n_sa_c; c_sa_r += cs.n_sa_r; c_loci += cs.n_loci;
                        n = cs.n_cand;
                    }
                    if (n > 1024u * LLV_N) { ok = false; break; }        // a work item names its chunk in 10 bits; PE lists can be longer
                    uint32_t o = 0;
                    if (lane == 0) o = atomicAdd(&g.gctl[QC(8)], n);
                    o = (uint32_t)__shfl((int)o, 0);
                    ok = (uint64_t)o + n <= g.pool;
                    if (ok) for (uint32_t i = lane; i < n; i += 64) g.gloci[o + i] = loci[i];
                    ns[strand] = n; off[strand] = o;
                }
                const uint32_t ch0 = ok ? (ns[0] + LLV_N - 1) / LLV_N : 0, ch1 = ok ? (ns[1] + LLV_N - 1) / LLV_N : 0;
                uint32_t base = 0;
                if (ok && lane == 0) base = atomicAdd(&g.gctl[QC(5)], ch0 + ch1);
                base = (uint32_t)__shfl((int)base, 0);
                ok = ok && (uint64_t)base + ch0 + ch1 <= g.items_cap;
                if (lane == 0) {
                    g.gq[slot] = ok ? r : 0xFFFFFFFFu;                    // an unusable slot is skipped by the later kernels
                    g.gn[2 * slot] = ok ? ns[0] : 0; g.gn[2 * slot + 1] = ok ? ns[1] : 0;
                    g.goff[2 * slot] = off[0]; g.goff[2 * slot + 1] = off[1];
                }
                if (ok) {
                    for (uint32_t c = lane; c < ch0 + ch1; c += 64) g.gitems[base + c] = (slot << 11) | (c >= ch0 ? (0x400u | (c - ch0)) : c);
                    if (ctr) {
                        for (int o2 = 32; o2 > 0; o2 >>= 1) c_vwords += __shfl_down(c_vwords, o2);
                        if (lane == 0) {
                            atomicAdd(ctr + SALT_CTR_SA_C, c_sa_c); atomicAdd(ctr + SALT_CTR_SA_R, c_sa_r);
                            atomicAdd(ctr + SALT_CTR_VERIFY, c_verify); atomicAdd(ctr + SALT_CTR_VERIFY_WORDS, c_vwords);
                            atomicAdd(ctr + SALT_CTR_LV, ns[0] + ns[1]); atomicAdd(ctr + SALT_CTR_READS, 1ull);
                            atomicAdd(ctr + SALT_CTR_BASES, L); atomicAdd(ctr + SALT_CTR_LOCI, c_loci);
                            atomicAdd(ctr + SALT_CTR_D_SA_HEAVY, c_sa_c + c_sa_r);
                            atomicAdd(ctr + SALT_CTR_D_VERIFY_HEAVY, c_verify * (L <= 248 ? 4u : (((L + 7) >> 3) + 4u) / 4u));      // 121 .. 248 bases: phase 1 of verify_two_phase (the survivors' second look is not counted)
                            atomicAdd(ctr + SALT_CTR_D_OUT_HEAVY, 5u * (ns[0] + ns[1]) + 16u);      // the located rows + distances handed to k_gap
                            if (ix.c_ctx) { atomicAdd(ctr + SALT_CTR_D_CTX_ROWS, c_ctx_rows); atomicAdd(ctr + SALT_CTR_D_CTX_REJECTED, c_ctx_rej); }
                        }
                    }
                    return;
                }
                WSYNC();                                                  // pool exhausted: finish this read here
            }
        }
        if constexpr (!W::HAS_LLV) {
            // ... which takes the lane-LV scratch this block does not carry: the read goes to the overflow queue and the pass behind
            // this kernel (the same code in a block that has it) starts it again.  Nothing of it has been written yet.
            if (lanes_fit && n_cand_nogap > 0) {
                if (lane == 0) g.ovq[atomicAdd(&g.gctl[QC(9)], 1u)] = r;
                return;
            }
        }
        for (int strand = 0; strand < 2; ++strand) {
            pc.stamp(SALT_CTR_T_GAP);
            const CandStats cs = build_candidates_call<PE, GL>(CandArgs{ ix.c_sa, ix.r_pos, sai_c, sai_r, ix.ref_len, ap.spr, ap.max_locate, r, L, strand, true, phase, pe_loci, loci_cap, ap.pe, true, nullptr, nullptr, 0 }, w);
            if (glob) { loci = cs.in_lds ? w.loci : pe_loci; cand_e = cs.in_lds ? w.cand_e : pe_cand; }
            pc.t = phase ? __builtin_amdgcn_s_memtime() : 0;
            const uint32_t n_cand = cs.n_cand; c_sa_c += cs.n_sa_c; c_sa_r += cs.n_sa_r; c_loci += cs.n_loci;
            bool any = false;
            // all candidates' distances at the call's initial bound, 64 at a time (one per lane); the
            // sequential rule below then only compares numbers
            const bool lanes_ok = lanes_fit && W::HAS_LLV;
            if constexpr (W::HAS_LLV) if (lanes_ok) {
                const int k0 = gap_k0;
                for (uint32_t b = 0; b < n_cand; b += LLV_N) {
                    WSYNC();
                    const uint32_t i = b + lane;
                    bool act = false;
                    if (lane < LLV_N && i < n_cand) {
                        const uint32_t pos = loci[i];
                        act = !(pos > ix.ref_len || pos + L + 4 > ix.ref_len);           // ed_diff guard (editdistance.c:178)
                        if (act) lane_text(ix.ref, w.u.llv.T + lane * LLV_TW, pos, L + 4);
                    }
                    const uint32_t e = lv_lanes(w.u.llv, w.pm[strand], (int)L, (int)L + 4, k0, act);
                    if (lane < LLV_N && i < n_cand) cand_e[i] = (uint8_t)e;
                }
                WSYNC();
            }
            for (uint32_t i = 0; i < n_cand; ++i) {
                uint32_t pos = loci[i];
                int e = -1;
                if (lanes_ok) { const uint32_t ev = cand_e[i]; e = (ev != 255 && (int)ev <= maxd) ? (int)ev : -1; }
                else if (!(pos > ix.ref_len || pos + L + 4 > ix.ref_len)) {   // ed_diff guard (editdistance.c:178)
                    lv_unpack(ix.ref, w, strand, L, pos);
                    int dd;
                    e = lv_wave(w.u.lvb.T, (int)L + 4, w.u.lvb.P, (int)L, maxd, nullptr, dd);
                    WSYNC();
                }
                ++c_lv;
                if (e >= 0) {
                    if (e < maxd || !any) { maxd = e; q_gap = 1; q_ndiff = (uint32_t)e; q_strand = (uint32_t)strand; q_pos = pos; }
                    if (n_hits_s[strand] < NHIT && lane == 0) {
                        uint32_t h = n_hits_s[strand];
                        w.hit_pos[strand][h] = pos; w.hit_nd[strand][h] = (uint8_t)e; w.hit_gap[strand][h] = 1;
                    }
                    if (n_hits_s[strand] == 0) a0[strand] = (uint32_t)e;
                    if (n_hits_s[strand] < NHIT) ++n_hits_s[strand];
                    any = true;
                }
            }
            found[strand] = any;
            WSYNC();
        }
    }
    WSYNC();
    pc.stamp(SALT_CTR_T_GAP);

    // ---- hits, MAPQ (query_set_hits / gen_mapq, query.c:270-333) ----
    // every lane computes the same small loop; lane 0 writes
    int b0 = (int)q_ndiff, b1 = 100000, tot = 0;
    uint32_t nh[2] = { 0, 0 };
    uint32_t sel_idx[2][SALT_MAX_HITS];
    for (int s = 0; s < 2 && tot < ap.max_hits; ++s) {
        for (uint32_t j = 0; j < n_hits_s[s]; ++j) {
            uint32_t p = w.hit_pos[s][j];
            if (p == 0xFFFFFFFFu || p == q_pos) continue;
            if (a0[s] <= q_ndiff) {
                if ((int)a0[s] <= b1) b1 = (int)a0[s];
                if (nh[s] < SALT_MAX_HITS) sel_idx[s][nh[s]] = j;
                ++nh[s]; ++tot;
            }
            if (tot == ap.max_hits) break;
        }
    }
    uint32_t mapq = 0;
    if (b0 != 0) {                                            // integer form of 255*|b0-b1|/b0, identical for all inputs
        uint32_t x = (uint32_t)(b0 > b1 ? b0 - b1 : b1 - b0);
        uint64_t q = mapq_quot(x, (uint32_t)b0);
        mapq = q < 254 ? (uint32_t)q : 254u;
    }
    if (lane == 0) {
        out->pos = q_pos; out->strand = (uint8_t)q_strand; out->n_diff = (uint8_t)q_ndiff; out->is_gap = (uint8_t)q_gap;
        out->mapq = (uint8_t)mapq; out->b0 = b0; out->b1 = b1; out->seq_start = 0; out->seq_end = (uint16_t)(L - 1);
        out->n_hits[0] = (uint8_t)nh[0]; out->n_hits[1] = (uint8_t)nh[1]; out->skipped = 0;
        for (int s = 0; s < 2; ++s)
            for (uint32_t j = 0; j < nh[s]; ++j) {
                uint32_t h = sel_idx[s][j];
                out->hits[s][j].pos = w.hit_pos[s][h]; out->hits[s][j].n_diff = w.hit_nd[s][h];
                out->hits[s][j].is_gap = w.hit_gap[s][h]; out->hits[s][j].strand = (uint16_t)s;
            }
    }
    // ---- CIGARs (query_gen_cigar query.c:282-296; XA cigars sam.c:216-225) ----
    if (q_pos != 0xFFFFFFFFu) {
        if (q_gap) {
            lv_cigar(ix.ref, w, lvtab, (int)q_strand, L, q_pos, (int)q_ndiff);
            if (lane < (uint32_t)w.n_cig) out->cigar[lane] = w.cig[lane];
            if (lane == 0) out->n_cigar = (uint8_t)w.n_cig;
        } else if (lane == 0) { out->cigar[0] = (uint16_t)((L << 4) | 0u); out->n_cigar = 1; }
    } else if (lane == 0) out->n_cigar = 0;
    uint32_t hidx = 0;
    for (int s = 0; s < 2; ++s)
        for (uint32_t j = 0; j < nh[s]; ++j, ++hidx) {
            uint32_t h = sel_idx[s][j];
            if (w.hit_gap[s][h]) {
                WSYNC();
                lv_cigar(ix.ref, w, lvtab, s, L, w.hit_pos[s][h], (int)w.hit_nd[s][h]);
                if (lane < (uint32_t)w.n_cig) out->hit_cigar[hidx][lane] = w.cig[lane];
                if (lane == 0) out->hit_n_cigar[hidx] = (uint8_t)w.n_cig;
            } else if (lane == 0) out->hit_n_cigar[hidx] = 0;
        }

    pc.stamp(SALT_CTR_T_TAIL);
    if (phase) {
        const unsigned long long dt = (unsigned long long)(__builtin_amdgcn_s_memrealtime() - rt0);
        pc.add(SALT_CTR_X3, dt);
        if (ctr && lane == 0) atomicMax(ctr + SALT_CTR_MAX_HEAVY, (dt << 32) | r);
    }
    if (ctr) {
        for (int o = 32; o > 0; o >>= 1) c_vwords += __shfl_down(c_vwords, o);
        if (lane == 0) {
            atomicAdd(ctr + SALT_CTR_SA_C, c_sa_c); atomicAdd(ctr + SALT_CTR_SA_R, c_sa_r);
            atomicAdd(ctr + SALT_CTR_VERIFY, c_verify); atomicAdd(ctr + SALT_CTR_VERIFY_WORDS, c_vwords);
            atomicAdd(ctr + SALT_CTR_LV, c_lv); atomicAdd(ctr + SALT_CTR_READS, 1ull);
            atomicAdd(ctr + SALT_CTR_BASES, L); atomicAdd(ctr + SALT_CTR_LOCI, c_loci);
            atomicAdd(ctr + SALT_CTR_D_SA_HEAVY, c_sa_c + c_sa_r);
            atomicAdd(ctr + SALT_CTR_D_VERIFY_HEAVY, c_verify * (L <= 248 ? 4u : (((L + 7) >> 3) + 4u) / 4u));      // 121 .. 248 bases: phase 1 of verify_two_phase (the survivors' second look is not counted)
            atomicAdd(ctr + SALT_CTR_D_OUT_HEAVY, 128u);
            if (ix.c_ctx) { atomicAdd(ctr + SALT_CTR_D_CTX_ROWS, c_ctx_rows); atomicAdd(ctr + SALT_CTR_D_CTX_REJECTED, c_ctx_rej); }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Persistent kernels: one-wave blocks pull work items through counters in qctl[] until the head passes the count (control word k lives
// at qctl[QC(k)], a cache line of its own; k_heavy, k_gap and k_gapfin take their items through ranged heads instead: pop_ranged)
//   qctl[0] reads queued by k_light    qctl[1] (was k_heavy's head)
//   qctl[2] gapped reads (slots)       qctl[3] k_gap head        qctl[4] k_gapfin head
//   qctl[5] k_gap items                qctl[6] CIGAR items       qctl[7] k_cigar head      qctl[8] pool entries used
// ---------------------------------------------------------------------------------------------
static constexpr uint32_t HEAVY_HEADS = 64, HEAVY_HEAD_STRIDE = 64;          // ranges of the queue with a head of their own; words between the heads
// Thread 0 of a persistent one-wave block: the next item of a queue of n_items cut into HEAVY_HEADS ranges (heads[range * HEAVY_HEAD_STRIDE],
// zero at launch); 0xFFFFFFFF once every range is empty.  seg / tried: where this block is (start at blockIdx % HEAVY_HEADS, 0).
__device__ __forceinline__ uint32_t pop_ranged(uint32_t *heads, const uint32_t n_items, uint32_t &seg, uint32_t &tried)
{
    while (tried < HEAVY_HEADS) {
        const uint32_t lo = (uint32_t)((uint64_t)n_items * seg / HEAVY_HEADS), hi = (uint32_t)((uint64_t)n_items * (seg + 1u) / HEAVY_HEADS);
        uint32_t *h = heads + seg * HEAVY_HEAD_STRIDE;
        if (lo < hi && __hip_atomic_load(h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < hi - lo) {      // (a look first: late waves do not queue up on an empty range's counter)
            const uint32_t k = atomicAdd(h, 1u);
            if (k < hi - lo) return lo + k;
        }
        seg = (seg + 1u) & (HEAVY_HEADS - 1u); ++tried;
    }
    return 0xFFFFFFFFu;
}
template <bool PE, bool GL, class W>
__device__ __forceinline__ void heavy_body(const IndexView &ix, const AlignParams &ap, const uint32_t *__restrict__ pm,
                                           const uint4 *__restrict__ sai_c, const uint4 *__restrict__ sai_r,
                                           salt_result_t *__restrict__ results, const uint32_t *__restrict__ queue,
                                           unsigned long long *__restrict__ ctr, LvTables *__restrict__ lvtab, const GapBufs g,
                                           uint8_t *__restrict__ pe_scr, const int overflow_pass)
{
    __shared__ W w;
    __shared__ uint32_t s_item;
    __shared__ unsigned long long s_phase[SALT_CTR_N];
    unsigned long long *phase = ctr ? s_phase : nullptr;
    if (ctr) { for (int i = threadIdx.x; i < SALT_CTR_N; i += 64) s_phase[i] = 0; }
    WSYNC();
    // the usual pass takes the reads k_light queued (or all of them); the overflow pass the ones a block without the lane-LV scratch left
    const uint32_t n_items = overflow_pass ? g.gctl[QC(9)] : ap.all_heavy ? ap.n_reads : g.gctl[QC(0)];
    // One read per pop.  The pops of all waves on ONE counter are served one after the other, ~14 ns each whatever the waves do in between
    // (75 700 queued reads: 1.09 of the kernel's 1.165 ms, the same at 8, 12 and 16 blocks per CU; time = 0.08 ms + 14.4 ns x reads from
    // 9 000 to 150 000 reads, profiles/r03/heavy_vs_batch.log).  So the queue is cut into HEAVY_HEADS ranges with a head each, 256 bytes
    // apart; a wave starts at range blockIdx % HEAVY_HEADS and moves on to the next when one is empty (pop_ranged).  k_gap and k_gapfin
    // take their items the same way (heads + 1, + 2); the counters k_heavy's gapped reads push through are a cache line apart each (QC()).
    uint32_t seg = blockIdx.x & (HEAVY_HEADS - 1u), tried = 0;
    for (;;) {
        if (threadIdx.x == 0) {
            uint32_t got = 0xFFFFFFFFu;
            if (overflow_pass) { got = atomicAdd(&g.gctl[QC(10)], 1u); if (got >= n_items) got = 0xFFFFFFFFu; }
            else got = pop_ranged(g.qheads, n_items, seg, tried);
            s_item = got;
        }
        WSYNC();
        const uint32_t it = s_item;
        WSYNC();
        if (it == 0xFFFFFFFFu) break;
        const uint32_t r = overflow_pass ? g.ovq[it] : ap.all_heavy ? it : queue[it];
        align_general<PE, GL>(ix, ap, w, r, pm, sai_c, sai_r, results, ctr, phase, lvtab + blockIdx.x, g,
                          pe_scr ? reinterpret_cast<uint32_t *>(pe_scr + (size_t)blockIdx.x * PE_LOCI_CAP * 5) : nullptr,
                          pe_scr ? pe_scr + (size_t)blockIdx.x * PE_LOCI_CAP * 5 + (size_t)PE_LOCI_CAP * 4 : nullptr);
        if (phase && threadIdx.x == 0) s_phase[SALT_CTR_HEAVY_READS] += 1;
        WSYNC();
    }
    if (phase) {                                                         // one flush per wave
        WSYNC();
        for (int i = SALT_CTR_T_LOAD + (int)threadIdx.x; i < SALT_CTR_N; i += 64) if (s_phase[i]) atomicAdd(ctr + i, s_phase[i]);
    }
}

// Registers of the persistent kernels capped for four waves per SIMD (128 VGPRs, ~20 spilled on rare paths): 16 one-wave blocks per CU.
// Uncapped (12 per CU) k_heavy takes 0.83 ms per 10^6 GRCh38-scale reads and k_heavy_pe 2.01 per 10^6 mates; capped 0.73 and 1.76, and the
// kernels of the other batches in flight find more room beside them (profiles/r03/ab_heavy_ranged_pops.log).  -DSALT_HEAVY_WAVES=0: no cap.
#ifndef SALT_HEAVY_WAVES
#define SALT_HEAVY_WAVES 4
#endif
#if SALT_HEAVY_WAVES > 0
#define HEAVY_OCC __attribute__((amdgpu_waves_per_eu(SALT_HEAVY_WAVES, SALT_HEAVY_WAVES)))
#else
#define HEAVY_OCC
#endif
#define HEAVY_KERNEL(NAME, PE, GL, LDS)                                                                                    \
__global__ void __launch_bounds__(64) HEAVY_OCC                                                                         \
NAME(IndexView ix, AlignParams ap, const uint32_t *__restrict__ pm,                                                     \
     const uint4 *__restrict__ sai_c, const uint4 *__restrict__ sai_r, salt_result_t *__restrict__ results,             \
     const uint32_t *__restrict__ queue, unsigned long long *__restrict__ ctr,                                          \
     LvTables *__restrict__ lvtab, GapBufs g, uint8_t *__restrict__ pe_scr, int overflow_pass)                          \
{ heavy_body<PE, GL, LDS>(ix, ap, pm, sai_c, sai_r, results, queue, ctr, lvtab, g, pe_scr, overflow_pass); }
HEAVY_KERNEL(k_heavy, false, false, WaveLdsSmall)      // the usual shape: <= 32 seed slots per strand, gapped reads handed to k_gap
HEAVY_KERNEL(k_heavy_pe, true, true, WaveLdsSmall)    // paired-end mates: PE locate rule, loci in global scratch, gap bound 3
HEAVY_KERNEL(k_heavy_big, false, false, WaveLds)      // any read the ABI admits, everything finished inside the block: batches with more seed
HEAVY_KERNEL(k_heavy_pe_big, true, true, WaveLds)
HEAVY_KERNEL(k_heavy_glob, false, true, WaveLds)    // single end with -m above the LDS list (SALT_MAX_LOCATE): located rows in the block's global list, everything in this one shape
//    // slots, runs without the k_gap buffers, and the overflow pass behind the usual shape

// k_gap: one item = 32 candidates of one strand of one queued read: their Landau-Vishkin distances at the bound L/10
// (ed_diff -> computeEditDistance, editdistance.c:174-232, LandauVishkin.c:19-122), one candidate per lane
struct GapLds { LaneLv llv; uint32_t pm[MAXL / 8]; };
__global__ void __launch_bounds__(64)
k_gap(IndexView ix, AlignParams ap, const uint32_t *__restrict__ pm, GapBufs g)
{
    __shared__ GapLds s;
    __shared__ uint32_t s_item;
    const uint32_t lane = lane_id();
    const uint32_t n_items = g.gctl[QC(5)] < g.items_cap ? g.gctl[QC(5)] : g.items_cap;
    uint32_t seg = blockIdx.x & (HEAVY_HEADS - 1u), tried = 0;
    for (;;) {
        if (threadIdx.x == 0) s_item = pop_ranged(g.qheads + 1, n_items, seg, tried);
        WSYNC();
        const uint32_t it = s_item;
        WSYNC();
        if (it == 0xFFFFFFFFu) break;
        const uint32_t item = g.gitems[it], slot = item >> 11, strand = (item >> 10) & 1u, chunk = item & 1023u;
        const uint32_t r = g.gq[slot];
        const uint32_t *rec = pm + (uint64_t)r * ap.pg.pm_stride;
        const uint32_t L = rec[2 * ap.pg.nw8], nw = (L + 7) >> 3;
        for (uint32_t t = lane; t < nw; t += 64) s.pm[t] = rec[strand * ap.pg.nw8 + t];
        const uint32_t n = g.gn[2 * slot + strand], i = chunk * LLV_N + lane;
        const size_t row = g.goff[2 * slot + strand];
        bool act = false;
        if (lane < LLV_N && i < n) {
            const uint32_t pos = g.gloci[row + i];
            act = !(pos > ix.ref_len || pos + L + 4 > ix.ref_len);               // ed_diff guard (editdistance.c:178)
            if (act) lane_text(ix.ref, s.llv.T + lane * LLV_TW, pos, L + 4);
        }
        WSYNC();
        const uint32_t e = lv_lanes(s.llv, s.pm, (int)L, (int)L + 4, ap.pe ? 3 : (int)(L / 10), act);   // alnse.c:1090 / 1016-1028
        if (lane < LLV_N && i < n) g.ge[row + i] = (uint8_t)e;
        WSYNC();
    }
}

// k_gapfin: one queued read per wave: alnse_check_withgap's sequential rule (alnse.c:871-901, code_kdiff 371-393)
// replayed by ballots over the stored distances, strand 0 then 1 with the bound carried over; then
// query_set_hits / gen_mapq (query.c:270-333) and the result row; every LV traceback becomes a k_cigar item.
struct FinLds { uint32_t hit_pos[2][NHIT]; uint8_t hit_nd[2][NHIT]; };
__global__ void __launch_bounds__(64)
k_gapfin(IndexView ix, AlignParams ap, const uint32_t *__restrict__ pm, salt_result_t *__restrict__ results, GapBufs g,
         unsigned long long *__restrict__ ctr)
{
    __shared__ FinLds w;
    __shared__ uint32_t s_item;
    const uint32_t lane = lane_id();
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint32_t n_items = g.gctl[QC(2)] < g.cap ? g.gctl[QC(2)] : g.cap;
    uint32_t seg = blockIdx.x & (HEAVY_HEADS - 1u), tried = 0;
    for (;;) {
        if (threadIdx.x == 0) s_item = pop_ranged(g.qheads + 2, n_items, seg, tried);
        WSYNC();
        const uint32_t slot = s_item;
        WSYNC();
        if (slot == 0xFFFFFFFFu) break;
        const uint64_t rt0 = ctr ? __builtin_amdgcn_s_memrealtime() : 0;
        const uint32_t r = g.gq[slot];
        if (r == 0xFFFFFFFFu) continue;                                   // the pool was full: k_heavy finished this read itself
        const uint32_t L = pm[(uint64_t)r * ap.pg.pm_stride + 2 * ap.pg.nw8];
        salt_result_t *out = results + r;
        uint32_t q_pos = 0xFFFFFFFFu, q_strand = 3, q_ndiff = 255, q_gap = 255;
        uint32_t bound = ap.pe ? 3u : L / 10;                             // alnse.c:1090 (SE) / alnse.c:1016-1028 (PE keeps 3)
        uint32_t n_hits_s[2] = { 0, 0 }, a0[2] = { 0, 0 };
        for (int strand = 0; strand < 2; ++strand) {
            const uint32_t n = g.gn[2 * slot + strand];
            const size_t row = g.goff[2 * slot + strand];
            bool any = false; uint32_t bp = 0, bv = 0;
            rule_unsorted<LLV_K, true>(g.gloci + row, g.ge + row, n, L, ix.ref_len, bound, any, bp, bv, n_hits_s[strand], a0[strand],
                                       w.hit_pos[strand], w.hit_nd[strand]);
            if (any) { q_pos = bp; q_ndiff = bv; q_gap = 1; q_strand = (uint32_t)strand; }
        }
        WSYNC();
        int b0 = (int)q_ndiff, b1 = 100000, tot = 0;
        uint32_t nh[2] = { 0, 0 };
        uint32_t sel_idx[2][SALT_MAX_HITS];
        for (int s = 0; s < 2 && tot < ap.max_hits; ++s)
            for (uint32_t j = 0; j < n_hits_s[s]; ++j) {
                const uint32_t p = w.hit_pos[s][j];
                if (p == 0xFFFFFFFFu || p == q_pos) continue;
                if (a0[s] <= q_ndiff) {
                    if ((int)a0[s] <= b1) b1 = (int)a0[s];
                    if (nh[s] < SALT_MAX_HITS) sel_idx[s][nh[s]] = j;
                    ++nh[s]; ++tot;
                }
                if (tot == ap.max_hits) break;
            }
        uint32_t mapq = 0;
        if (b0 != 0) {
            const uint32_t x = (uint32_t)(b0 > b1 ? b0 - b1 : b1 - b0);
            const uint64_t q = mapq_quot(x, (uint32_t)b0);
            mapq = q < 254 ? (uint32_t)q : 254u;
        }
        const uint32_t n_cig_items = q_pos != 0xFFFFFFFFu ? 1u + nh[0] + nh[1] : 0u;
        const uint32_t first_cig = ap.pe ? 1u : 0u;                        // paired end: the alignment's own CIGAR is made after pairing (k_pe_final)
        if (lane == 0) {
            out->pos = q_pos; out->strand = (uint8_t)q_strand; out->n_diff = (uint8_t)q_ndiff; out->is_gap = (uint8_t)q_gap;
            out->mapq = (uint8_t)mapq; out->b0 = b0; out->b1 = b1; out->seq_start = 0; out->seq_end = (uint16_t)(L - 1);
            out->n_hits[0] = (uint8_t)nh[0]; out->n_hits[1] = (uint8_t)nh[1]; out->skipped = 0; out->n_cigar = 0;
            for (int s = 0; s < 2; ++s)
                for (uint32_t j = 0; j < nh[s]; ++j) {
                    const uint32_t h = sel_idx[s][j];
                    out->hits[s][j].pos = w.hit_pos[s][h]; out->hits[s][j].n_diff = w.hit_nd[s][h];
                    out->hits[s][j].is_gap = 1; out->hits[s][j].strand = (uint16_t)s;
                }
            if (n_cig_items > first_cig) {                                // main hit (which = 0) and every alternative hit (1 + index)
                const uint32_t base = atomicAdd(&g.gctl[QC(6)], n_cig_items - first_cig);
                for (uint32_t c = first_cig; c < n_cig_items; ++c) g.cq[base + c - first_cig] = (r << 3) | c;
            }
            if (ctr) atomicMax(ctr + SALT_CTR_MAX_GAPFIN, ((unsigned long long)(__builtin_amdgcn_s_memrealtime() - rt0) << 32) | r);
        }
        WSYNC();
    }
}

// k_cigar: one LV traceback per wave (query_gen_cigar query.c:282-296; XA CIGARs sam.c:216-225).
// items[i] = (read << 3) | which, which 0 = the alignment itself, 1 + h = alternative hit h; *count items, *head = the queue head
__global__ void __launch_bounds__(64)
k_cigar(IndexView ix, PackGeom pg, const uint32_t *__restrict__ pm, salt_result_t *__restrict__ results,
        const uint32_t *__restrict__ items, const uint32_t *__restrict__ count, uint32_t *__restrict__ head, uint32_t cap_items,
        LvTables *__restrict__ lvtab)
{
    __shared__ WaveLdsCigar w;
    __shared__ uint32_t s_item;
    const uint32_t lane = lane_id();
    const uint32_t n_items = *count < cap_items ? *count : cap_items;
    struct { PackGeom pg; } ap = { pg };
    bool first = true;
    for (;;) {
        // the block's own index first, then what the counter hands out behind those -- after a look at it (a plain load): the pops of a
        // launch on one counter are served one after the other, ~14 ns each, also the ones that only find the queue empty
        if (threadIdx.x == 0)
            s_item = first ? blockIdx.x
                   : (gridDim.x >= n_items || gridDim.x + __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_items) ? 0xFFFFFFFFu : gridDim.x + atomicAdd(head, 1u);
        first = false;
        WSYNC();
        const uint32_t it = s_item;
        WSYNC();
        if (it >= n_items) break;
        const uint32_t item = items[it], which = item & 7u, r = item >> 3;
        salt_result_t *out = results + r;
        const uint32_t *rec = pm + (uint64_t)r * ap.pg.pm_stride;
        const uint32_t L = rec[2 * ap.pg.nw8];
        uint32_t strand, pos, k;
        if (which == 0) { strand = out->strand; pos = out->pos; k = out->n_diff; }
        else {
            const uint32_t h = which - 1, n0 = out->n_hits[0];
            strand = h >= n0; const uint32_t j = strand ? h - n0 : h;
            pos = out->hits[strand][j].pos; k = out->hits[strand][j].n_diff;
        }
        for (uint32_t t = lane; t < (L + 7) >> 3; t += 64) w.pm[strand][t] = rec[strand * ap.pg.nw8 + t];
        WSYNC();
        lv_cigar(ix.ref, w, lvtab + blockIdx.x, (int)strand, L, pos, (int)k);
        uint16_t *dst = which == 0 ? out->cigar : out->hit_cigar[which - 1];
        if (lane < (uint32_t)w.n_cig) dst[lane] = w.cig[lane];
        if (lane == 0) { if (which == 0) out->n_cigar = (uint8_t)w.n_cig; else out->hit_n_cigar[which - 1] = (uint8_t)w.n_cig; }
        WSYNC();
    }
}

// ---------------------------------------------------------------------------------------------
// k_light: one wave per read, the common case in three memory round trips.
//
// A read stays here when (a) it is short enough for the small LDS image (L <= 160, <= 16 seed slots),
// (b) each of its four seed lists (C/R x strand) enumerates at most 64 suffix-array rows, so that the
// max_locate cap (>= 128) can never bite and locate order is irrelevant once the loci are sorted, and
// (c) a gap-free hit exists.  Everything else is queued for k_heavy, which replays the reference's
// general control flow.  Same results either way; the split only changes who computes them.
// ---------------------------------------------------------------------------------------------
// The queue of reads for k_heavy.  One counter for all of them was what k_light2 waited for: an atomic on ONE address is served every
// ~11 ns however many waves ask (75 700 queued reads: 0.83 of its 0.87 ms).  The reads of workgroup-sized groups of four go to one
// of QSEG segments, each with a counter 256 bytes from the next and room for every read that can map to it; k_queue_pack then
// lays the segments end to end (the order of the queue does not matter) and leaves the count where k_heavy reads it.
static constexpr uint32_t QSEG = 64, QSEG_STRIDE = 64;                      // segments; words between their counters
__host__ __device__ __forceinline__ uint32_t qseg_cap(uint32_t n_reads) { return n_reads / QSEG + 8u; }
__device__ __forceinline__ void queue_push(uint32_t *__restrict__ qseg, uint32_t *__restrict__ qsub, uint32_t n_reads, uint32_t r)
{
    const uint32_t s = (r >> 2) & (QSEG - 1u);
    qseg[(size_t)s * qseg_cap(n_reads) + atomicAdd(&qsub[s * QSEG_STRIDE], 1u)] = r;
}
__global__ void __launch_bounds__(256)
k_queue_pack(const uint32_t *__restrict__ qseg, const uint32_t *__restrict__ qsub, uint32_t n_reads, uint32_t *__restrict__ queue, uint32_t *__restrict__ qctl)
{
    __shared__ uint32_t base_s, cnt_s;
    if (threadIdx.x == 0) {
        uint32_t base = 0, total = 0;
        for (uint32_t s = 0; s < QSEG; ++s) { const uint32_t c = qsub[s * QSEG_STRIDE]; if (s < blockIdx.x) base += c; total += c; }
        base_s = base; cnt_s = qsub[blockIdx.x * QSEG_STRIDE];
        if (blockIdx.x == 0) qctl[QC(0)] = total;
    }
    __syncthreads();
    const uint32_t *src = qseg + (size_t)blockIdx.x * qseg_cap(n_reads);
    for (uint32_t i = threadIdx.x; i < cnt_s; i += 256) queue[base_s + i] = src[i];
}

static constexpr int LT_SLOTS = 16;      // seed slots per list handled here
static constexpr int LT_LOCI = 128;      // loci per strand handled here
static constexpr int LT_MAXL = 160;

struct LightLds {
    uint32_t pm[2][LT_MAXL / 8];
    uint32_t sp[4][LT_SLOTS], off[4][LT_SLOTS];
    uint32_t pre[4][LT_SLOTS + 1];       // rows enumerated before slot i of list l
    uint32_t loci[2][LT_LOCI];
    uint32_t hit_pos[2][NHIT];
    uint8_t  hit_nd[2][NHIT];
    uint8_t  val[2][LT_LOCI];
};

static constexpr int LT_WAVES = 2;       // independent reads (waves) per workgroup: half the workgroups to dispatch (as k_light2)
__global__ void __launch_bounds__(64 * LT_WAVES) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_light(IndexView ix, AlignParams ap, const uint32_t *__restrict__ pm,
        const uint4 *__restrict__ sai_c, const uint4 *__restrict__ sai_r, salt_result_t *__restrict__ results,
        uint32_t *__restrict__ queue /* the segments */, uint32_t *__restrict__ qsub, unsigned long long *__restrict__ ctr)
{
    __shared__ LightLds w_all[LT_WAVES];
    LightLds &w = w_all[threadIdx.x >> 6];
    const uint32_t lane = lane_id();
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint32_t r = __builtin_amdgcn_readfirstlane(blockIdx.x * LT_WAVES + (threadIdx.x >> 6));
    if (r >= ap.n_reads) return;
    const uint32_t *rec = pm + (uint64_t)r * ap.pg.pm_stride;               // k_pack's one-hot words of this read, both strands
    const uint32_t L = rec[2 * ap.pg.nw8];
    bool heavy = L > LT_MAXL || L < (uint32_t)ap.l_seed || ap.spr > LT_SLOTS || ap.max_locate < 2 * 64;
    uint32_t c_sa_c = 0, c_sa_r = 0, c_verify = 0, c_vwords = 0, c_loci = 0;
    const bool prof = ctr && (r & 127u) == 0;                                // phase clock of every 128th read
    uint64_t tp = prof ? __builtin_amdgcn_s_memtime() : 0;
    auto stamp = [&](int slot) {
        if (!prof) return;
        const uint64_t n = __builtin_amdgcn_s_memtime();
        if (lane == 0) atomicAdd(ctr + slot, (unsigned long long)(n - tp));
        tp = n;
    };

    if (!heavy) {
        // ---- round trip 1: the read (as one-hot nibble words, both strands) and its seeds ----
        const uint32_t nw = (L + 7) >> 3;
        uint32_t n_amb = 0;
        if (lane < 2 * nw) {
            const uint32_t s = lane >= nw, j = s ? lane - nw : lane;
            const uint32_t word = rec[s * ap.pg.nw8 + j];
            w.pm[s][j] = word;
            if (!s) n_amb = (uint32_t)__popc(word & (word >> 1) & (word >> 2) & (word >> 3) & 0x11111111u);   // N = all four bits
        }
        if (ap.max_amb < L)                                                 // (uniform) otherwise the limit cannot be passed
            for (int o = 32; o > 0; o >>= 1) n_amb += (uint32_t)__shfl_xor((int)n_amb, o);
        if (n_amb > ap.max_amb) {                                           // alnse.c:1328 / alnpe.c:495: record left untouched
            if (lane == 0) {
                salt_result_t *out = results + r;
                out->pos = 0xFFFFFFFFu; out->strand = 3; out->n_diff = 255; out->is_gap = 255; out->mapq = 0;
                out->b0 = -1; out->b1 = -1; out->seq_start = 0; out->seq_end = (uint16_t)(L - 1);
                out->n_hits[0] = out->n_hits[1] = 0; out->n_cigar = 0; out->skipped = 1;
            }
            return;
        }
        uint32_t tot[4] = { 0, 0, 0, 0 };
        bool small = false;
        {
            // lane = (list, slot); list: 0 C/fwd 1 R/fwd 2 C/rev 3 R/rev.  The reference orders each list by
            // interval size (alnse.c:307-308), which only decides what is located first when the max_locate
            // cap bites; here it cannot (<= 64 rows per list), and the loci are sorted afterwards anyway.
            const uint32_t l = lane >> 4, slot = lane & 15u;
            uint4 v = make_uint4(1, 0, 0, 0);
            if (slot < ap.spr) v = ((l & 1) ? sai_r : sai_c)[((uint64_t)r * 2u + (l >> 1)) * ap.spr + slot];
            uint32_t sz = v.w ? v.y - v.x + 1u : 0u;
            if (sz > 65u) sz = 65u;                                         // keeps the sums below from wrapping
            uint32_t inc = sz;                                              // inclusive prefix sum within the 16-lane row
            for (int o = 1; o < 16; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 16); if (slot >= (uint32_t)o) inc += t; }
            for (int q = 0; q < 4; ++q) tot[q] = (uint32_t)__builtin_amdgcn_readlane((int)inc, 16 * q + 15);
            small = tot[0] <= 16 && tot[1] <= 16 && tot[2] <= 16 && tot[3] <= 16;       // (uniform) the usual case
            if (small) {
                // every seed writes the suffix-array rows of its interval where the row lanes of its list will look them up
                for (uint32_t k2 = 0; k2 < sz; ++k2) { w.sp[l][inc - sz + k2] = v.x + k2; w.off[l][inc - sz + k2] = v.z | (v.w == 2 ? 0x80000000u : 0u); }
            } else {
                w.sp[l][slot] = v.x; w.off[l][slot] = v.z | (v.w == 2 ? 0x80000000u : 0u);        // bit 31: .x is a genome position already
                w.pre[l][slot + 1] = inc;
                if (slot == 0) w.pre[l][0] = 0;
            }
        }
        WSYNC();
        stamp(SALT_CTR_LT_SEEDS);
        for (int l = 0; l < 4; ++l) heavy |= tot[l] > 64;
        if (ap.dbg_stop == 1) { if (lane == 0) results[r].pos = tot[0] + tot[1] + tot[2] + tot[3] + w.pm[0][0] + w.pm[1][1]; return; }
        if (!heavy) {
            // ---- round trip 2: every suffix-array row of the four lists at once ----
            uint32_t n_s[2];
            if (small) {
                // lane = (list, row): at most 16 rows per list, so one lane per row covers all four lists in one go
                const uint32_t l = lane >> 4, x = lane & 15u;
                const uint32_t tl = l == 0 ? tot[0] : l == 1 ? tot[1] : l == 2 ? tot[2] : tot[3];
                bool keep = false; uint32_t p = 0;
                if (x < tl) {
                    const uint32_t j = w.sp[l][x], of = w.off[l][x];
                    p = ((of >> 31) ? j : (l & 1) ? ix.r_pos[j] : ix.c_sa[j]) - (of & 0x7FFFFFFFu);
                    keep = (l & 1) ? !(p > ix.ref_len || p + L > ix.ref_len) : !(p + L > ix.ref_len);       // alnse.c:672-673,715-717
                }
                stamp(SALT_CTR_LT_LOCATE);
                if (ap.dbg_stop == 2) { if (__ballot(p == 12345u) == 1) results[r].pos = p; return; }
                const uint64_t km = __ballot(keep);
                const uint32_t s = lane >> 5, ms = (uint32_t)(km >> (32 * s));
                WSYNC();                                                    // the row table aliases nothing below, but keep LDS ordered
                if (keep) w.loci[s][(uint32_t)__popc(ms & ((1u << (lane & 31u)) - 1u))] = p;
                n_s[0] = (uint32_t)__popc((uint32_t)km); n_s[1] = (uint32_t)__popc((uint32_t)(km >> 32));
                c_sa_c += tot[0] + tot[2]; c_sa_r += tot[1] + tot[3]; c_loci += n_s[0] + n_s[1];
            } else {
            uint32_t pos4[4]; bool keep4[4];
            for (int l = 0; l < 4; ++l) {
                keep4[l] = false; pos4[l] = 0;
                if (lane < tot[l]) {
                    uint32_t i = 0;
                    while (w.pre[l][i + 1] <= lane) ++i;                   // skips empty slots (equal prefix sums)
                    const uint32_t j = w.sp[l][i] + (lane - w.pre[l][i]), of = w.off[l][i];
                    const uint32_t p = ((of >> 31) ? j : (l & 1) ? ix.r_pos[j] : ix.c_sa[j]) - (of & 0x7FFFFFFFu);
                    pos4[l] = p;
                    keep4[l] = (l & 1) ? !(p > ix.ref_len || p + L > ix.ref_len) : !(p + L > ix.ref_len);   // alnse.c:672-673,715-717
                }
            }
            stamp(SALT_CTR_LT_LOCATE);
            if (ap.dbg_stop == 2) { const uint32_t x = pos4[0] + pos4[1] + pos4[2] + pos4[3]; if (__ballot(x == 12345u) == 1) results[r].pos = x; return; }
            for (int s = 0; s < 2; ++s) {                                   // both lists of a strand, as located (no order, duplicates stay)
                const uint64_t mc = __ballot(keep4[2 * s]), mr = __ballot(keep4[2 * s + 1]);
                const uint32_t nc = (uint32_t)__popcll(mc);
                if (keep4[2 * s]) w.loci[s][(uint32_t)__popcll(mc & lt)] = pos4[2 * s];
                if (keep4[2 * s + 1]) w.loci[s][nc + (uint32_t)__popcll(mr & lt)] = pos4[2 * s + 1];
                n_s[s] = nc + (uint32_t)__popcll(mr);
                c_sa_c += tot[2 * s]; c_sa_r += tot[2 * s + 1]; c_loci += n_s[s];
            }
            }
            WSYNC();
            stamp(SALT_CTR_LT_SORT);
            if (ap.dbg_stop == 3) { if (lane == 0) results[r].pos = w.loci[0][0] + w.loci[1][0] + n_s[0] + n_s[1]; return; }
            // ---- round trip 3: masked Hamming distance of every located row of both strands ----
            uint32_t bound = 3, q_pos = 0xFFFFFFFFu, q_strand = 3, q_ndiff = 255;
            uint32_t n_hits_s[2] = { 0, 0 }, a0[2] = { 0, 0 };
            bool found[2] = { false, false };
            {
                if (L <= 120) verify_quads_2(ix.ref, ix.ref_len, w.pm[0], w.pm[1], L, w.loci[0], n_s[0], w.loci[1], n_s[1], w.val[0], w.val[1]);
                else if (L <= 248) verify_quads_2<8>(ix.ref, ix.ref_len, w.pm[0], w.pm[1], L, w.loci[0], n_s[0], w.loci[1], n_s[1], w.val[0], w.val[1]);
                else
                    for (int s = 0; s < 2; ++s)
                        for (uint32_t i = lane; i < n_s[s]; i += 64) w.val[s][i] = (uint8_t)mismatch_capped(ix, w.pm[s], L, w.loci[s][i]);
                WSYNC();
                stamp(SALT_CTR_LT_VERIFY);
                if (ap.dbg_stop == 4) { if (__ballot(w.val[0][0] + w.val[1][0] == 12345u) == 1) results[r].pos = w.val[0][0]; return; }
                c_verify += n_s[0] + n_s[1];
                // ---- the sequential best/first-hit rule (alnse.c:348-369, 1079-1083) on the unsorted rows: rule_unsorted ----
                for (int s = 0; s < 2; ++s) {
                    bool any = false; uint32_t bp = 0, bv = 0;
                    rule_sparse<3, false>(w.loci[s], w.val[s], n_s[s], L, ix.ref_len, bound, any, bp, bv, n_hits_s[s], a0[s], w.hit_pos[s], w.hit_nd[s]);
                    if (any) { found[s] = true; q_pos = bp; q_ndiff = bv; q_strand = (uint32_t)s; }
                }
            }
            if (ap.dbg_stop == 5) { if (lane == 0) results[r].pos = q_pos + n_hits_s[0] + n_hits_s[1] + a0[0] + a0[1]; return; }
            if (!heavy && !found[0] && !found[1]) heavy = true;             // needs the gapped pass
            if (heavy) { }
            else {
                WSYNC();
                // ---- query_set_hits / gen_mapq (query.c:270-333): lane t < 2*NHIT looks at recorded hit (t / NHIT, t % NHIT);
                // the sequential loop keeps the first max_hits of them that are neither the primary nor worse than it ----
                const uint32_t hs = lane >= (uint32_t)NHIT, hj = lane - hs * NHIT;
                uint32_t hp = 0xFFFFFFFFu, hn = 0;
                bool cand = false;
                if (lane < 2u * NHIT && hj < (hs ? n_hits_s[1] : n_hits_s[0])) {
                    hp = w.hit_pos[hs][hj]; hn = w.hit_nd[hs][hj];
                    cand = hp != 0xFFFFFFFFu && hp != q_pos && (hs ? a0[1] : a0[0]) <= q_ndiff;
                }
                const uint64_t cm = __ballot(cand);
                const bool sel = cand && (uint32_t)__popcll(cm & lt) < (uint32_t)ap.max_hits;
                const uint64_t sm = __ballot(sel);
                const uint32_t nh0 = (uint32_t)__popcll(sm & ((1ull << NHIT) - 1ull)), nh1 = (uint32_t)__popcll(sm >> NHIT);
                const int b0 = (int)q_ndiff;
                int b1 = 100000;
                if (nh0) b1 = (int)a0[0];
                if (nh1 && (int)a0[1] <= b1) b1 = (int)a0[1];
                uint32_t mapq = 0;
                if (b0 != 0) {
                    const uint32_t x = (uint32_t)(b0 > b1 ? b0 - b1 : b1 - b0);
                    const uint64_t q = mapq_quot(x, (uint32_t)b0);
                    mapq = q < 254 ? (uint32_t)q : 254u;
                }
                salt_result_t *out = results + r;
                if (sel) {                                                  // one 8-byte store per kept hit, all in one instruction
                    const uint32_t hidx = (uint32_t)__popcll(sm & lt);
                    salt_hit_t hv; hv.pos = hp; hv.n_diff = (uint8_t)hn; hv.is_gap = 0; hv.strand = (uint16_t)hs;
                    out->hits[hs][hs ? hidx - nh0 : hidx] = hv;
                    out->hit_n_cigar[hidx] = 0;
                }
                if (lane < 6) {                                             // the 24 header bytes as six dwords
                    const uint32_t d = lane == 0 ? q_pos
                                     : lane == 1 ? (q_strand | (q_ndiff << 8) | (mapq << 24))             // strand, n_diff, is_gap = 0, mapq
                                     : lane == 2 ? (uint32_t)b0
                                     : lane == 3 ? (uint32_t)b1
                                     : lane == 4 ? ((L - 1) << 16)                                        // seq_start = 0, seq_end
                                     : (nh0 | (nh1 << 8) | (1u << 16));                                   // n_hits[2], n_cigar = 1, skipped = 0
                    reinterpret_cast<uint32_t *>(out)[lane] = d;
                } else if (lane == 6) out->cigar[0] = (uint16_t)((L << 4) | 0u);
            }
        }
    }
    if (heavy) {
        if (lane == 0) queue_push(queue, qsub, ap.n_reads, r);
        return;                                                             // k_heavy does (and counts) all of it
    }
    stamp(SALT_CTR_LT_OUT);
    if (prof && lane == 0) atomicAdd(ctr + SALT_CTR_LT_SAMPLES, 1ull);
    if (ctr) {
        for (int o = 32; o > 0; o >>= 1) c_vwords += __shfl_down(c_vwords, o);
        if (lane == 0) {
            atomicAdd(ctr + SALT_CTR_SA_C, c_sa_c); atomicAdd(ctr + SALT_CTR_SA_R, c_sa_r);
            atomicAdd(ctr + SALT_CTR_VERIFY, c_verify); atomicAdd(ctr + SALT_CTR_VERIFY_WORDS, c_vwords);
            atomicAdd(ctr + SALT_CTR_READS, 1ull); atomicAdd(ctr + SALT_CTR_BASES, L); atomicAdd(ctr + SALT_CTR_LOCI, c_loci);
            atomicAdd(ctr + SALT_CTR_D_SA_LIGHT, c_sa_c + c_sa_r);
            atomicAdd(ctr + SALT_CTR_D_VERIFY_LIGHT, c_verify * (L <= 120 ? 4u : L <= 248 ? 8u : (((L + 7) >> 3) + 4u) / 4u));
            atomicAdd(ctr + SALT_CTR_D_OUT_LIGHT, 40u);                     // 24-byte header + CIGAR word + (typically one) 8-byte hit
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_light2: k_light with TWO reads per wave (32 lanes each) for reads with few seed slots (spr <= 8; L <= 120 with four lanes per
// located row: the 100-bp single-end case; L <= 248 with eight: 150-base reads and mates).  k_light is bound by instruction issue and by the dispatch of one workgroup per read; with
// two reads per wave both halve.  Same steps, same results; every ballot / shuffle stays inside the read's half-wave.
// ---------------------------------------------------------------------------------------------
static constexpr int L2_SLOTS = 8;       // seed slots per list
struct LightLds2 {
    uint32_t pm[2][32];                  // (reads up to 120 bases use 15 words per strand, up to 248 bases 31)
    uint32_t sp[4][L2_SLOTS], off[4][L2_SLOTS];
    uint32_t pre[4][L2_SLOTS + 1];
    uint32_t loci[2][LT_LOCI];
    uint32_t hit_pos[2][NHIT];
    uint8_t  hit_nd[2][NHIT];
    uint8_t  val[2][LT_LOCI];
};

// rule_sparse on a half-wave: n candidates, lane hl of the half holds candidates hl, hl + 32, ...
__device__ __forceinline__ void rule_sparse_h(const uint32_t *pos, const uint8_t *val, const uint32_t n, const uint32_t ref_len,
                                              uint32_t &bound, bool &any, uint32_t &best_pos, uint32_t &best_v,
                                              uint32_t &n_hits, uint32_t &a0, uint32_t *hit_pos, uint8_t *hit_nd)
{
    const uint32_t lane = lane_id(), hl = lane & 31u, hb = lane & 32u;
    const uint32_t NONE = 0xFFFFFFFFu;
    auto HB = [&](bool x) -> uint32_t { const uint64_t m = __ballot(x); return hb ? (uint32_t)(m >> 32) : (uint32_t)m; };   // this half's 32 bits (a select, not a 64-bit shift)
    auto SHF = [&](uint32_t v, int src) -> uint32_t { return (uint32_t)__shfl((int)v, (int)hb + src); };
    any = false; n_hits = 0;
    if (n == 0) return;
    if (n <= 32) {                                            // the usual case: every candidate inside the bound is the same locus
        uint32_t p = 0, v = NONE;
        if (hl < n) { p = pos[hl]; v = val[hl]; }
        const bool ok = v <= bound && p < ref_len;
        const uint32_t m = HB(ok);
        if (m == 0) return;
        const int l0 = __ffs((int)m) - 1;
        const uint32_t p0 = SHF(p, l0);
        if (HB(ok && p != p0) == 0) {
            const uint32_t v0 = SHF(v, l0);
            any = true; best_pos = p0; best_v = v0; n_hits = 1; a0 = v0;
            if (hl == 0) { hit_pos[0] = p0; hit_nd[0] = (uint8_t)v0; }
            bound = v0 < bound ? v0 : bound;
            return;
        }
    }
    uint32_t mp[4] = { NONE, NONE, NONE, NONE };
    for (uint32_t b = 0; b < n; b += 32) {
        const uint32_t i = b + hl;
        uint32_t p = 0, v = NONE;
        if (i < n) { p = pos[i]; v = val[i]; }
        uint32_t m = HB(v <= bound && p < ref_len);
        while (m) {
            const int l = __ffs((int)m) - 1;
            const uint32_t pl = SHF(p, l), vl = SHF(v, l);
#pragma unroll
            for (int t = 0; t <= 3; ++t) if (vl == (uint32_t)t && pl < mp[t]) mp[t] = pl;
            m &= m - 1;
        }
    }
    uint32_t before[4];
    before[0] = NONE;
#pragma unroll
    for (int t = 1; t <= 3; ++t) before[t] = mp[t - 1] < before[t - 1] ? mp[t - 1] : before[t - 1];
#pragma unroll
    for (int t = 3; t >= 0; --t) if (mp[t] != NONE) { any = true; best_v = (uint32_t)t; best_pos = mp[t]; }
    if (!any) return;
    uint32_t last_p = 0;
    for (uint32_t h = 0; h < (uint32_t)NHIT; ++h) {
        uint32_t cur_p = NONE, cur_v = 0;
        for (uint32_t b = 0; b < n; b += 32) {
            const uint32_t i = b + hl;
            uint32_t p = 0, v = NONE;
            if (i < n) { p = pos[i]; v = val[i]; }
            const uint32_t lim = v == 1 ? before[1] : v == 2 ? before[2] : v == 3 ? before[3] : NONE;
            uint32_t m = HB(v <= bound && p < ref_len && p < lim && (h == 0 || p > last_p));
            while (m) {
                const int l = __ffs((int)m) - 1;
                const uint32_t pl = SHF(p, l);
                if (pl < cur_p) { cur_p = pl; cur_v = SHF(v, l); }
                m &= m - 1;
            }
        }
        if (cur_p == NONE) break;
        if (hl == 0) { hit_pos[h] = cur_p; hit_nd[h] = (uint8_t)cur_v; }
        if (h == 0) a0 = cur_v;
        last_p = cur_p; ++n_hits;
    }
    bound = best_v < bound ? best_v : bound;
}

// L2_WAVES waves per workgroup; each wave still works alone on its two reads (no block-level synchronisation).  Two waves per
// workgroup halve the number of workgroups the dispatcher has to place (5 x 10^5 -> 2.5 x 10^5 per batch): the kernel alone takes
// the same 0.48 ms, but with other batches' kernels on the GPU the step rate rises ~6 %; four are slower again.
// LN: lanes per located row in the window check: 4 cover reads up to 120 bases, 8 up to 248 (150-base mates and reads; the seed slots
// bound the length before that: spr <= 8)
template <int L2_WAVES, int LN = 4>
__global__ void __launch_bounds__(64 * L2_WAVES) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_light2(IndexView ix, AlignParams ap, const uint32_t *__restrict__ pm,
         const uint4 *__restrict__ sai_c, const uint4 *__restrict__ sai_r, salt_result_t *__restrict__ results,
         uint32_t *__restrict__ queue /* the segments */, uint32_t *__restrict__ qsub)
{
    __shared__ LightLds2 w2[2 * L2_WAVES];
    const uint32_t lane = lane_id(), hl = lane & 31u, hb = lane & 32u, half = lane >> 5;
    LightLds2 &w = w2[2u * (threadIdx.x >> 6) + half];
    const uint32_t lt = (1u << hl) - 1u;
    auto HB = [&](bool x) -> uint32_t { const uint64_t m = __ballot(x); return hb ? (uint32_t)(m >> 32) : (uint32_t)m; };   // this half's 32 bits (a select, not a 64-bit shift)
    auto SHF = [&](uint32_t v, int src) -> uint32_t { return (uint32_t)__shfl((int)v, (int)hb + src); };
    const uint32_t r = 2u * (blockIdx.x * (uint32_t)L2_WAVES + (threadIdx.x >> 6)) + half;
    if (r >= ap.n_reads) return;
    const uint32_t *rec = pm + (uint64_t)r * ap.pg.pm_stride;
    const uint32_t L = rec[2 * ap.pg.nw8];
    bool heavy = L > (LN == 8 ? 248u : 120u) || L < (uint32_t)ap.l_seed || ap.spr > (uint32_t)L2_SLOTS || ap.max_locate < 2 * 64;
    if (!heavy) {
        // ---- round trip 1: the read (one-hot nibble words, both strands) and its seeds ----
        const uint32_t nw = (L + 7) >> 3;                                   // <= 15 (LN = 4) / <= 31 (LN = 8)
        uint32_t n_amb = 0;
        for (uint32_t x = hl; x < 2 * nw; x += 32) {                        // (one trip for reads up to 128 bases)
            const uint32_t s = x >= nw, j = s ? x - nw : x;
            const uint32_t word = rec[s * ap.pg.nw8 + j];
            w.pm[s][j] = word;
            if (!s) n_amb += (uint32_t)__popc(word & (word >> 1) & (word >> 2) & (word >> 3) & 0x11111111u);
        }
        if (ap.max_amb < L) {
            for (int o = 16; o > 0; o >>= 1) n_amb += (uint32_t)__shfl_xor((int)n_amb, o);
            if (n_amb > ap.max_amb) {                                       // alnse.c:1328 / alnpe.c:495: record left untouched
                if (hl == 0) {
                    salt_result_t *out = results + r;
                    out->pos = 0xFFFFFFFFu; out->strand = 3; out->n_diff = 255; out->is_gap = 255; out->mapq = 0;
                    out->b0 = -1; out->b1 = -1; out->seq_start = 0; out->seq_end = (uint16_t)(L - 1);
                    out->n_hits[0] = out->n_hits[1] = 0; out->n_cigar = 0; out->skipped = 1;
                }
                return;
            }
        }
        uint32_t tot[4];
        bool small;
        {
            // half-lane = (list, slot); list: 0 C/fwd 1 R/fwd 2 C/rev 3 R/rev (see k_light)
            const uint32_t l = hl >> 3, slot = hl & 7u;
            uint4 v = make_uint4(1, 0, 0, 0);
            if (slot < ap.spr) v = ((l & 1) ? sai_r : sai_c)[((uint64_t)r * 2u + (l >> 1)) * ap.spr + slot];
            uint32_t sz = v.w ? v.y - v.x + 1u : 0u;
            if (sz > 65u) sz = 65u;
            uint32_t inc = sz;                                              // inclusive prefix sum within the 8-lane row
            for (int o = 1; o < 8; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 8); if (slot >= (uint32_t)o) inc += t; }
            for (int q = 0; q < 4; ++q) tot[q] = SHF(inc, 8 * q + 7);
            small = tot[0] <= 8 && tot[1] <= 8 && tot[2] <= 8 && tot[3] <= 8;
            if (small) {
                for (uint32_t k2 = 0; k2 < sz; ++k2) { w.sp[l][inc - sz + k2] = v.x + k2; w.off[l][inc - sz + k2] = v.z | (v.w == 2 ? 0x80000000u : 0u); }
            } else {
                w.sp[l][slot] = v.x; w.off[l][slot] = v.z | (v.w == 2 ? 0x80000000u : 0u);        // bit 31: .x is a genome position already
                w.pre[l][slot + 1] = inc;
                if (slot == 0) w.pre[l][0] = 0;
            }
        }
        WSYNC();
        for (int l = 0; l < 4; ++l) heavy |= tot[l] > 64;
        if (!heavy) {
            // ---- round trip 2: the suffix-array rows of the four lists ----
            uint32_t n_s[2] = { 0, 0 };
            if (small) {
                const uint32_t l = hl >> 3, x = hl & 7u;
                const uint32_t tl = l == 0 ? tot[0] : l == 1 ? tot[1] : l == 2 ? tot[2] : tot[3];
                bool keep = false; uint32_t p = 0;
                if (x < tl) {
                    const uint32_t j = w.sp[l][x], of = w.off[l][x];
                    p = ((of >> 31) ? j : (l & 1) ? ix.r_pos[j] : ix.c_sa[j]) - (of & 0x7FFFFFFFu);
                    keep = (l & 1) ? !(p > ix.ref_len || p + L > ix.ref_len) : !(p + L > ix.ref_len);       // alnse.c:672-673,715-717
                }
                const uint32_t km = HB(keep);
                const uint32_t s = hl >> 4, ms = (km >> (16 * s)) & 0xFFFFu;
                if (keep) w.loci[s][(uint32_t)__popc(ms & ((1u << (hl & 15u)) - 1u))] = p;
                n_s[0] = (uint32_t)__popc(km & 0xFFFFu); n_s[1] = (uint32_t)__popc(km >> 16);
            } else {
                for (int l = 0; l < 4; ++l) {
                    const int s = l >> 1;
                    for (uint32_t x = hl; x - hl < tot[l]; x += 32) {         // (x - hl) is the same for the whole half: a uniform trip count
                        bool keep = false; uint32_t p = 0;
                        if (x < tot[l]) {
                            uint32_t i = 0;
                            while (w.pre[l][i + 1] <= x) ++i;
                            const uint32_t j = w.sp[l][i] + (x - w.pre[l][i]), of = w.off[l][i];
                            p = ((of >> 31) ? j : (l & 1) ? ix.r_pos[j] : ix.c_sa[j]) - (of & 0x7FFFFFFFu);
                            keep = (l & 1) ? !(p > ix.ref_len || p + L > ix.ref_len) : !(p + L > ix.ref_len);
                        }
                        const uint32_t km = HB(keep);
                        if (keep) w.loci[s][n_s[s] + (uint32_t)__popc(km & lt)] = p;
                        n_s[s] += (uint32_t)__popc(km);
                    }
                }
            }
            WSYNC();
            // ---- round trip 3: masked Hamming distance of every located row of both strands, LN lanes per row ----
            {
                constexpr uint32_t RPG = 32u / LN;                           // rows of one group of loads per half-wave
                const uint32_t sub = hl & (uint32_t)(LN - 1), q = hl / (uint32_t)LN, n = n_s[0] + n_s[1];
                const uint32_t nvalid = L > 32u * sub ? (L - 32u * sub < 32u ? L - 32u * sub : 32u) : 0u;
                uint32_t pa[4], pb[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) { const bool in = (4 * sub + t) < nw; pa[t] = in ? w.pm[0][4 * sub + t] : 0u; pb[t] = in ? w.pm[1][4 * sub + t] : 0u; }
                for (uint32_t b = 0; b < n; b += 4u * RPG) {                // 4 groups of RPG rows per trip
                    uint32_t pos[4]; u32x4_a4 x[4]; bool act[4], rev[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (b + RPG * g >= n) break;
                        const uint32_t c = b + RPG * g + q;
                        act[g] = c < n; rev[g] = c >= n_s[0];
                        pos[g] = act[g] ? (rev[g] ? w.loci[1][c - n_s[0]] : w.loci[0][c]) : 0u;
                        if (pos[g] >= ix.ref_len) pos[g] = 0xFFFFFFFFu;      // wrapped below 0 (see mismatch_capped): no load, INF
                        if (LN == 8 && !(4u * sub < nw + 1u)) x[g] = u32x4_a4{ 0u, 0u, 0u, 0u };       // this lane's words lie behind the window: never fetched
                        else x[g] = *reinterpret_cast<const u32x4_a4 *>(ix.ref + ((pos[g] == 0xFFFFFFFFu ? 0u : pos[g]) >> 3) + 4 * sub);
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (b + RPG * g >= n) break;
                        const uint32_t pw[4] = { rev[g] ? pb[0] : pa[0], rev[g] ? pb[1] : pa[1], rev[g] ? pb[2] : pa[2], rev[g] ? pb[3] : pa[3] };
                        const uint32_t mism = quad_mismatch<LN>(x[g], pos[g], pw, nvalid);
                        if (act[g] && sub == 0) {
                            const uint32_t c = b + RPG * g + q;
                            const uint8_t v = (uint8_t)((mism > 3 || pos[g] == 0xFFFFFFFFu) ? INF : mism);
                            if (rev[g]) w.val[1][c - n_s[0]] = v; else w.val[0][c] = v;
                        }
                    }
                }
            }
            WSYNC();
            // ---- the sequential best/first-hit rule on the unsorted rows ----
            uint32_t bound = 3, q_pos = 0xFFFFFFFFu, q_strand = 3, q_ndiff = 255;
            uint32_t n_hits_s[2] = { 0, 0 }, a0[2] = { 0, 0 };
            bool found[2] = { false, false };
            for (int s = 0; s < 2; ++s) {
                bool any = false; uint32_t bp = 0, bv = 0;
                rule_sparse_h(w.loci[s], w.val[s], n_s[s], ix.ref_len, bound, any, bp, bv, n_hits_s[s], a0[s], w.hit_pos[s], w.hit_nd[s]);
                if (any) { found[s] = true; q_pos = bp; q_ndiff = bv; q_strand = (uint32_t)s; }
            }
            if (!found[0] && !found[1]) heavy = true;                       // needs the gapped pass
            else {
                WSYNC();
                // ---- query_set_hits / gen_mapq (query.c:270-333), see k_light ----
                const uint32_t hs = hl >= (uint32_t)NHIT, hj = hl - hs * NHIT;
                uint32_t hp = 0xFFFFFFFFu, hn = 0;
                bool cand = false;
                if (hl < 2u * NHIT && hj < (hs ? n_hits_s[1] : n_hits_s[0])) {
                    hp = w.hit_pos[hs][hj]; hn = w.hit_nd[hs][hj];
                    cand = hp != 0xFFFFFFFFu && hp != q_pos && (hs ? a0[1] : a0[0]) <= q_ndiff;
                }
                const uint32_t cm = HB(cand);
                const bool sel = cand && (uint32_t)__popc(cm & lt) < (uint32_t)ap.max_hits;
                const uint32_t sm = HB(sel);
                const uint32_t nh0 = (uint32_t)__popc(sm & ((1u << NHIT) - 1u)), nh1 = (uint32_t)__popc(sm >> NHIT);
                const int b0 = (int)q_ndiff;
                int b1 = 100000;
                if (nh0) b1 = (int)a0[0];
                if (nh1 && (int)a0[1] <= b1) b1 = (int)a0[1];
                uint32_t mapq = 0;
                if (b0 != 0) {
                    const uint32_t x = (uint32_t)(b0 > b1 ? b0 - b1 : b1 - b0);
                    const uint64_t qq = mapq_quot(x, (uint32_t)b0);
                    mapq = qq < 254 ? (uint32_t)qq : 254u;
                }
                salt_result_t *out = results + r;
                if (sel) {
                    const uint32_t hidx = (uint32_t)__popc(sm & lt);
                    salt_hit_t hv; hv.pos = hp; hv.n_diff = (uint8_t)hn; hv.is_gap = 0; hv.strand = (uint16_t)hs;
                    out->hits[hs][hs ? hidx - nh0 : hidx] = hv;
                    out->hit_n_cigar[hidx] = 0;
                }
                if (hl >= 16 && hl < 22) {                                  // the 24 header bytes as six dwords
                    const uint32_t t = hl - 16;
                    const uint32_t d = t == 0 ? q_pos
                                     : t == 1 ? (q_strand | (q_ndiff << 8) | (mapq << 24))
                                     : t == 2 ? (uint32_t)b0
                                     : t == 3 ? (uint32_t)b1
                                     : t == 4 ? ((L - 1) << 16)
                                     : (nh0 | (nh1 << 8) | (1u << 16));
                    reinterpret_cast<uint32_t *>(out)[t] = d;
                } else if (hl == 22) out->cigar[0] = (uint16_t)((L << 4) | 0u);
            }
        }
    }
    if (heavy && hl == 0) queue_push(queue, qsub, ap.n_reads, r);            // k_heavy does all of it
}

// ---------------------------------------------------------------------------------------------
// k_pe_final: one thread per pair -- apply the first successful mate rescue (in the order pairing2 /
// pairing_singleton try them, alnpe.c:213-252, 420-470), then query_gen_cigar for the mates that keep their
// seed-and-verify mapping (query.c:282-296): "<L>M", or a k_cigar item when the mapping is gapped
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_pe_final(PackGeom pg, uint32_t n_pairs, const uint32_t *__restrict__ pm,
           salt_result_t *__restrict__ res, const PePair *__restrict__ pairs, const PeSwRes *__restrict__ sw,
           uint32_t *__restrict__ citems, uint32_t *__restrict__ ccount)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const PePair pr = pairs[p];
    int rescued = -1;
    for (int k = 0; k < pr.n_req && rescued < 0; ++k) {
        const PeSwRes &r = sw[pr.req0 + k];
        if (!r.ok) continue;
        rescued = pr.rescued[k];
        salt_result_t *q = res + 2 * p + rescued;
        const int b0 = r.score1, b1 = r.score2;
        uint32_t mapq = 0;
        if (b0 != 0) { const uint32_t x = (uint32_t)(b0 > b1 ? b0 - b1 : b1 - b0); const uint64_t v = mapq_quot(x, (uint32_t)b0); mapq = v < 254 ? (uint32_t)v : 254u; }
        q->b0 = b0; q->b1 = b1; q->mapq = (uint8_t)mapq;
        q->pos = (uint32_t)r.ref_begin + r.start; q->strand = (uint8_t)r.strand;
        q->seq_start = (uint16_t)r.read_begin; q->seq_end = (uint16_t)r.read_end;
        q->n_cigar = (uint8_t)r.n_cigar;
        for (uint32_t c = 0; c < r.n_cigar; ++c) q->cigar[c] = r.cigar[c];
    }
    for (int m = 0; m < 2; ++m) {
        if (m == rescued) continue;
        salt_result_t *q = res + 2 * p + m;
        const uint32_t L = pm[(uint64_t)(2 * p + m) * pg.pm_stride + 2 * pg.nw8];
        q->seq_start = 0; q->seq_end = (uint16_t)(L - 1);
        if (q->pos == 0xFFFFFFFFu) q->n_cigar = 0;
        else if (q->is_gap) { q->n_cigar = 0; citems[atomicAdd(ccount, 1u)] = ((2 * p + (uint32_t)m) << 3) | 0u; }
        else { q->cigar[0] = (uint16_t)((L << 4) | 0u); q->n_cigar = 1; }
    }
}

void launch_pe_final(const IndexView &ix, const PackGeom &pg, uint32_t n_pairs, const uint32_t *pm, salt_result_t *res, const PePair *pairs,
                     const PeSwRes *sw, void *lvtab, uint32_t *citems, uint32_t *cctl, uint32_t n_blocks, hipStream_t st)
{
    if (!n_pairs) return;
    hipLaunchKernelGGL(k_pe_final, dim3((n_pairs + 255) / 256), dim3(256), 0, st, pg, n_pairs, pm, res, pairs, sw, citems, cctl);
    hipLaunchKernelGGL(k_cigar, dim3(n_blocks), dim3(64), 0, st, ix, pg, pm, res, citems, cctl, cctl + 1, 2 * n_pairs, static_cast<LvTables *>(lvtab));
}

// ---------------------------------------------------------------------------------------------
// launch wrappers (called from salt_gpu.hip)
// ---------------------------------------------------------------------------------------------
uint32_t seed_wq_seg_cap(uint64_t items) { return (uint32_t)(((items + 255) / 256 + WQ_SEG - 1) / WQ_SEG) * 256u; }      // every seed of every block of a segment
size_t seed_wq_words(uint64_t items) { return (size_t)2 * WQ_SEG * seed_wq_seg_cap(items) * 4; }                              // uint4 records of both lists, in words
uint32_t seed_wq_cnt_words() { return 2u * WQ_SEG * WQ_STRIDE; }
void launch_seed(const IndexView &ix, const SeedParams &sp, const uint32_t *tb, uint4 *sai_c, uint4 *sai_r, uint4 *wq, uint32_t *wq_cnt,
                 uint32_t walk_blocks, unsigned long long *ctr, hipStream_t st)
{
    uint64_t items = (uint64_t)sp.n_reads * 2u * sp.spr;
    if (!items) return;
    uint32_t blocks = (uint32_t)((items + 255) / 256);
    const uint32_t cap = seed_wq_seg_cap(items);
    hipMemsetAsync(wq_cnt, 0, (size_t)seed_wq_cnt_words() * 4, st);
    hipLaunchKernelGGL(k_seed, dim3(blocks), dim3(256), 0, st, ix, sp, tb, sai_c, sai_r, wq, wq_cnt, cap, ctr);
    walk_blocks = (walk_blocks + WQ_SEG - 1) / WQ_SEG * WQ_SEG;
    hipLaunchKernelGGL(k_seed_walk, dim3(walk_blocks), dim3(256), 0, st, ix, sp, tb, sai_c, sai_r, wq, wq_cnt, cap, ctr);
}

void launch_light(const IndexView &ix, const AlignParams &ap, const uint32_t *pm, const uint8_t *, const uint32_t *, const uint4 *sai_c,
                  const uint4 *sai_r, salt_result_t *results, uint32_t *queue, uint32_t *qctl, uint32_t *qseg, uint32_t *qsub, unsigned long long *ctr, hipStream_t st)
{
    if (!ap.n_reads) return;
    hipMemsetAsync(qsub, 0, (size_t)QSEG * QSEG_STRIDE * 4, st);
    static const bool no_half = getenv("SALT_GPU_NO_LIGHT2") && atoi(getenv("SALT_GPU_NO_LIGHT2"));
    if (!no_half && !ctr && !SALT_DIAG_VAL(ap.dbg_stop) && ap.spr <= (uint32_t)L2_SLOTS && ap.pg.nw8 <= 31) {          // reads of at most 120 / 248 bases with at most 8 seed slots: two per wave
        static const int wv = getenv("SALT_GPU_L2_WAVES") ? atoi(getenv("SALT_GPU_L2_WAVES")) : 2;
        if (ap.pg.nw8 > 15) hipLaunchKernelGGL((k_light2<2, 8>), dim3((ap.n_reads + 3) / 4), dim3(128), 0, st, ix, ap, pm, sai_c, sai_r, results, qseg, qsub);
        else if (wv == 1) hipLaunchKernelGGL(k_light2<1>, dim3((ap.n_reads + 1) / 2), dim3(64), 0, st, ix, ap, pm, sai_c, sai_r, results, qseg, qsub);
        else hipLaunchKernelGGL(k_light2<2>, dim3((ap.n_reads + 3) / 4), dim3(128), 0, st, ix, ap, pm, sai_c, sai_r, results, qseg, qsub);
    } else
        hipLaunchKernelGGL(k_light, dim3((ap.n_reads + LT_WAVES - 1) / LT_WAVES), dim3(64 * LT_WAVES), 0, st, ix, ap, pm, sai_c, sai_r, results, qseg, qsub, ctr);
    hipLaunchKernelGGL(k_queue_pack, dim3(QSEG), dim3(256), 0, st, qseg, qsub, ap.n_reads, queue, qctl);
}
size_t queue_words(uint32_t max_reads) { return (size_t)max_reads * 2 + (size_t)QSEG * qseg_cap(max_reads); }     // flat queue | overflow queue | segments
uint32_t queue_sub_words() { return QSEG * QSEG_STRIDE; }
uint32_t queue_heads_offset() { static_assert(HEAVY_HEADS == QSEG && HEAVY_HEAD_STRIDE == QSEG_STRIDE, "k_heavy's heads live in the segments' counter slots"); return QSEG_STRIDE / 2; }

// ---------------------------------------------------------------------------------------------
// k_diag_rule: unit access to rule_unsorted / rule_sparse (tests only).  One wave per case; out[case] = any, best_pos,
// best_v, n_hits, a0, bound_out, then NHIT x (hit_pos, hit_nd)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_diag_rule(uint32_t n_cases, const uint32_t *__restrict__ pos, const uint8_t *__restrict__ val, const uint32_t *__restrict__ offs,
            const uint32_t *__restrict__ bound_in, uint32_t L, uint32_t ref_len, int mode, uint32_t *__restrict__ out)
{
    __shared__ uint32_t hit_pos[NHIT];
    __shared__ uint8_t hit_nd[NHIT];
    const uint32_t c = blockIdx.x, lane = lane_id();
    if (c >= n_cases) return;
    const uint32_t o = offs[c], n = offs[c + 1] - o;
    uint32_t bound = bound_in[c], bp = 0, bv = 0, nh = 0, a0 = 0;
    bool any = false;
    if (lane < NHIT) { hit_pos[lane] = 0; hit_nd[lane] = 0; }
    WSYNC();
    if (mode == 0) rule_unsorted<3, false>(pos + o, val + o, n, L, ref_len, bound, any, bp, bv, nh, a0, hit_pos, hit_nd);
    else if (mode == 1) rule_sparse<3, false>(pos + o, val + o, n, L, ref_len, bound, any, bp, bv, nh, a0, hit_pos, hit_nd);
    else rule_unsorted<LLV_K, true>(pos + o, val + o, n, L, ref_len, bound, any, bp, bv, nh, a0, hit_pos, hit_nd);
    WSYNC();
    uint32_t *w = out + (size_t)c * (6 + 2 * NHIT);
    if (lane == 0) { w[0] = any; w[1] = any ? bp : 0; w[2] = any ? bv : 0; w[3] = nh; w[4] = nh ? a0 : 0; w[5] = bound; }
    if (lane < NHIT) { w[6 + 2 * lane] = lane < nh ? hit_pos[lane] : 0; w[7 + 2 * lane] = lane < nh ? hit_nd[lane] : 0; }
}

void launch_diag_rule(uint32_t n_cases, const uint32_t *pos, const uint8_t *val, const uint32_t *offs, const uint32_t *bound_in,
                      uint32_t L, uint32_t ref_len, int mode, uint32_t *out, hipStream_t st)
{
    if (n_cases) hipLaunchKernelGGL(k_diag_rule, dim3(n_cases), dim3(64), 0, st, n_cases, pos, val, offs, bound_in, L, ref_len, mode, out);
}
uint32_t diag_rule_words() { return 6 + 2 * NHIT; }

// ---------------------------------------------------------------------------------------------
// k_diag_lv: unit access to the verify / LV device functions for the golden vectors (tests only)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_diag_lv(IndexView ix, uint32_t n_cases, const uint32_t *__restrict__ pos, const uint32_t *__restrict__ kdiff,
          const uint8_t *__restrict__ seqs, const uint32_t *__restrict__ offs, int32_t *__restrict__ out /* [n][4] */,
          uint16_t *__restrict__ cig_out /* [n][64] */, LvTables *__restrict__ lvtab)
{
    __shared__ WaveLds w;
    const uint32_t c = blockIdx.x, lane = lane_id();
    if (c >= n_cases) return;
    const uint32_t off = offs[c], L = offs[c + 1] - off, p = pos[c];
    const uint32_t nw = (L + 7) >> 3;
    for (uint32_t j = lane; j < nw; j += 64) {
        uint32_t word = 0;
        for (uint32_t q = 0; q < 8; ++q) {
            uint32_t i = j * 8 + q, cc = i < L ? seqs[off + i] : 5u;
            word |= (cc < 4 ? (1u << cc) : (cc == 4 ? 15u : 0u)) << (4 * q);
        }
        w.pm[0][j] = word;
    }
    WSYNC();
    int32_t v = -2, e_wave = -2, e_lane = -2, n_cig = -2;
    if (p + L <= ix.ref_len) v = (int32_t)mismatch_capped(ix, w.pm[0], L, p);
    const bool in_range = !(p > ix.ref_len || p + L + 4 > ix.ref_len);
    const int k = (int)kdiff[c];
    if (in_range) {
        lv_unpack(ix.ref, w, 0, L, p);
        int dd;
        e_wave = lv_wave(w.u.lvb.T, (int)L + 4, w.u.lvb.P, (int)L, k, nullptr, dd);
        WSYNC();
        if (k <= LLV_K && L + 4 <= 8u * (LLV_TW - 1)) {
            const uint32_t tl = L + 4, w0 = p >> 3, sh = (p & 7u) * 4u, nwt = (tl + 7) >> 3;
            uint32_t lo = ix.ref[w0];
            for (uint32_t j = 0; j < LLV_TW && lane < LLV_N; ++j) {    // every lane stages the same window
                uint32_t word = 0;
                if (j < nwt) {
                    const uint32_t hi = ix.ref[w0 + j + 1];
                    word = sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
                    const uint32_t rem = tl - j * 8;
                    if (rem < 8) word &= (1u << (4 * rem)) - 1u;
                    lo = hi;
                }
                w.u.llv.T[lane * LLV_TW + j] = word;
            }
            uint32_t el = lv_lanes(w.u.llv, w.pm[0], (int)L, (int)L + 4, k, lane < LLV_N);
            el = (uint32_t)__shfl((int)el, 0);
            e_lane = el == 255 ? -1 : (int32_t)el;
            WSYNC();
        }
        if (e_wave >= 0 && e_wave < LVK) {
            lv_cigar(ix.ref, w, lvtab + blockIdx.x, 0, L, p, e_wave);
            n_cig = w.n_cig;
            if (lane < (uint32_t)w.n_cig) cig_out[(size_t)c * SALT_MAX_CIGAR_OPS + lane] = w.cig[lane];
        }
    } else { e_wave = -1; e_lane = -1; }
    if (lane == 0) { out[c * 4 + 0] = v; out[c * 4 + 1] = e_wave; out[c * 4 + 2] = e_lane; out[c * 4 + 3] = n_cig; }
}

size_t lv_table_bytes() { return sizeof(LvTables); }

void launch_diag_lv(const IndexView &ix, uint32_t n, const uint32_t *pos, const uint32_t *kdiff, const uint8_t *seqs,
                    const uint32_t *offs, int32_t *out, uint16_t *cig, void *lvtab, hipStream_t st)
{
    if (n) hipLaunchKernelGGL(k_diag_lv, dim3(n), dim3(64), 0, st, ix, n, pos, kdiff, seqs, offs, out, cig, static_cast<LvTables *>(lvtab));
}

// k_diag_verify: unit access to the candidate verifiers (tests only).  One wave per case: the read seqs[offs[c]..offs[c+1]) against
// candidates cand[coffs[c]..coffs[c+1]) (at most 256) on the caller's mixRef.  mode 0: mismatch_capped per lane, 1: verify_quads<8>
// (4 lanes per candidate, L <= 120), 2: verify_quads<8, 8> (L <= 248), 3 / 4: verify_quads_2<4> / <8> with the list split between the
// "strands" (both use the same read).  out[i] = 0..3 or 255.  Candidates >= ref_len (a locate that wrapped below 0, alnse.c:672-673) must
// come back 255 WITHOUT being used as an address: the test's mixRef is a few KB, so a regression reads far outside of it only by value.
__global__ void __launch_bounds__(64)
k_diag_verify(const uint32_t *__restrict__ ref, uint32_t ref_len, uint32_t n_cases, const uint8_t *__restrict__ seqs, const uint32_t *__restrict__ offs,
              const uint32_t *__restrict__ cand, const uint32_t *__restrict__ coffs, int mode, uint8_t *__restrict__ out)
{
    __shared__ uint32_t pm[64];
    __shared__ uint32_t loci[256];
    __shared__ uint8_t val[256];
    const uint32_t c = blockIdx.x, lane = lane_id();
    if (c >= n_cases) return;
    const uint32_t off = offs[c], L = offs[c + 1] - off, c0 = coffs[c];
    uint32_t n = coffs[c + 1] - c0;
    if (n > 256) n = 256;
    const uint32_t nw = (L + 7) >> 3;
    for (uint32_t j = lane; j < 64; j += 64) {
        uint32_t word = 0;
        for (uint32_t q = 0; q < 8 && j < nw; ++q) {
            const uint32_t i = j * 8 + q, cc = i < L ? seqs[off + i] : 5u;
            word |= (cc < 4 ? (1u << cc) : (cc == 4 ? 15u : 0u)) << (4 * q);
        }
        pm[j] = word;
    }
    for (uint32_t i = lane; i < n; i += 64) { loci[i] = cand[c0 + i]; val[i] = 77; }
    WSYNC();
    IndexView ix; ix.ref = ref; ix.ref_len = ref_len;
    if (mode == 0) { for (uint32_t i = lane; i < n; i += 64) val[i] = (uint8_t)mismatch_capped(ix, pm, L, loci[i]); }
    else if (mode == 1) verify_quads<8>(ref, ref_len, pm, L, loci, n, val);
    else if (mode == 2) verify_quads<8, 8>(ref, ref_len, pm, L, loci, n, val);
    else if (mode == 3) verify_quads_2(ref, ref_len, pm, pm, L, loci, n / 2, loci + n / 2, n - n / 2, val, val + n / 2);
    else verify_quads_2<8>(ref, ref_len, pm, pm, L, loci, n / 2, loci + n / 2, n - n / 2, val, val + n / 2);
    WSYNC();
    for (uint32_t i = lane; i < n; i += 64) out[c0 + i] = val[i];
}
void launch_diag_verify(const uint32_t *ref, uint32_t ref_len, uint32_t n_cases, const uint8_t *seqs, const uint32_t *offs, const uint32_t *cand,
                        const uint32_t *coffs, int mode, uint8_t *out, hipStream_t st)
{
    if (n_cases) hipLaunchKernelGGL(k_diag_verify, dim3(n_cases), dim3(64), 0, st, ref, ref_len, n_cases, seqs, offs, cand, coffs, mode, out);
}

// ---------------------------------------------------------------------------------------------
// k_polish (row N4): the re-scoring step of the reference's SAM post-processor (Polish_src/polish.c:461-497, 190-249).
// One wave per item = (read, strand, genome offset): the plain edit distance (k <= 13) between the read and the bases at that offset
// -- stock Landau-Vishkin on EQUAL bytes (Polish_src/lv.c), which lv_wave computes when both strings are one-hot: equal <=> AND != 0
// -- and, for the items that ask for it, the CIGAR (computeEditDistanceWithCigar with useM: first "e plain mismatches on diagonal 0 ->
// <L>M", lv.c:274-296, then the traceback).  The reference's window buffer is calloc'ed, so what lies behind the text reads as 'A'
// (code 0) there; a window clipped at the genome end keeps the previous hit's bytes behind the clip (polish.c:84-92): the host hands
// those few windows over explicitly (`pool`).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_polish(const uint8_t *__restrict__ pac, const uint8_t *__restrict__ codes, const uint32_t *__restrict__ offs, const salt_polish_item_t *__restrict__ items,
         uint32_t n_items, const uint8_t *__restrict__ pool, uint32_t pool_stride, int want_cigar, int32_t *__restrict__ dist,
         uint16_t *__restrict__ cigars, uint8_t *__restrict__ n_cigar, LvTables *__restrict__ tabs)
{
    __shared__ LvBytes s;
    __shared__ int s_n;
    const uint32_t lane = lane_id();
    for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        const salt_polish_item_t x = items[it];
        const uint32_t o = offs[x.read], L = offs[x.read + 1] - o, tl = x.tlen;
        WSYNC();
        for (uint32_t i = lane; i < L + 48; i += 64) {
            uint8_t pv = 1;                                                    // behind the read: calloc'ed zeros = 'A'
            if (i < L) { const uint8_t c = x.strand ? codes[o + L - 1 - i] : codes[o + i]; pv = c < 4 ? (uint8_t)(1u << (x.strand ? 3 - c : c)) : (uint8_t)(x.strand ? 0x20 : 0x10); }
            s.P[i] = pv;
        }
        for (uint32_t i = lane; i < L + 48; i += 64) {
            uint8_t tv = 1;
            if (x.pool != 0xFFFFFFFFu) { if (i < pool_stride) tv = (uint8_t)(1u << (pool[(uint64_t)x.pool * pool_stride + i] & 3u)); }
            else if (i < tl) { const uint64_t l = (uint64_t)x.offset + i; tv = (uint8_t)(1u << ((pac[l >> 2] >> ((~l & 3u) << 1)) & 3u)); }
            s.T[i] = tv;
        }
        WSYNC();
        int d_fin = 0;
        LvTables *tab = want_cigar ? tabs + blockIdx.x : nullptr;
        const int e = lv_wave(s.T, (int)tl, s.P, (int)L, (int)x.k, tab, d_fin);
        if (lane == 0) dist[it] = e;
        if (want_cigar) {
            __threadfence_block();
            WSYNC();
            // e plain mismatches on diagonal 0?  (lv.c:274-296)
            const int end0 = (int)(L < tl ? L : tl);
            int straight = 0;
            for (int i = (int)lane; i < end0; i += 64) straight += s.P[i] != s.T[i];
            for (int w = 32; w > 0; w >>= 1) straight += __shfl_xor(straight, w);
            straight += (int)L - end0;
            if (lane == 0) {
                int n = 0;
                uint16_t *cg = cigars + (uint64_t)it * SALT_MAX_CIGAR_OPS;
                if (e == 0 || (e > 0 && straight == e)) cg[n++] = (uint16_t)((L << 4) | 0u);
                else if (e > 0) {
                    char act[LVK + 1]; int matched[LVK + 1];
                    int cd = d_fin;
                    for (int ce = e; ce >= 1; --ce) {
                        const char a = tab->A[ce][cd + 31];
                        act[ce] = a;
                        const int cur = tab->L[ce][cd + 31];
                        if (a == 'I') { matched[ce] = cur - tab->L[ce - 1][cd + 1 + 31] - 1; cd += 1; }
                        else if (a == 'D') { matched[ce] = cur - tab->L[ce - 1][cd - 1 + 31]; cd -= 1; }
                        else { matched[ce] = cur - tab->L[ce - 1][cd + 31] - 1; }
                    }
                    int acc = tab->L[0][31];
                    int ce = 1;
                    while (ce <= e) {
                        const char a = act[ce]; int cnt = 1;
                        while (ce + 1 <= e && matched[ce] == 0 && act[ce + 1] == a) { ++cnt; ++ce; }
                        if (a == 'X') acc += cnt;
                        else {
                            if (acc != 0 && n < SALT_MAX_CIGAR_OPS) cg[n++] = (uint16_t)((acc << 4) | 0);
                            acc = 0;
                            if (n < SALT_MAX_CIGAR_OPS) cg[n++] = (uint16_t)((cnt << 4) | (a == 'I' ? 1 : 2));
                        }
                        if (matched[ce] > 0) acc += matched[ce];
                        ++ce;
                    }
                    if (acc != 0 && n < SALT_MAX_CIGAR_OPS) cg[n++] = (uint16_t)((acc << 4) | 0);
                }
                n_cigar[it] = (uint8_t)n;
                s_n = n;
            }
            WSYNC();
        }
    }
}
void launch_polish(const uint8_t *pac, const uint8_t *codes, const uint32_t *offs, const salt_polish_item_t *items, uint32_t n_items, const uint8_t *pool,
                   uint32_t pool_stride, int want_cigar, int32_t *dist, uint16_t *cigars, uint8_t *n_cigar, void *tabs, uint32_t n_blocks, hipStream_t st)
{
    if (n_items) hipLaunchKernelGGL(k_polish, dim3(n_blocks), dim3(64), 0, st, pac, codes, offs, items, n_items, pool, pool_stride, want_cigar, dist, cigars, n_cigar,
                                    static_cast<LvTables *>(tabs));
}

// first 128 bytes of every result row, densely packed (what the host needs of nearly every row)
__global__ void __launch_bounds__(256)
k_heads(const salt_result_t *__restrict__ res, uint32_t n, uint4 *__restrict__ heads)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)n * 8u) return;
    const uint32_t i = (uint32_t)(t >> 3), c = (uint32_t)t & 7u;
    heads[t] = reinterpret_cast<const uint4 *>(res + i)[c];
}
void launch_heads(const salt_result_t *res, uint32_t n, uint8_t *heads, hipStream_t st)
{
    if (n) hipLaunchKernelGGL(k_heads, dim3((uint32_t)(((uint64_t)n * 8u + 255) / 256)), dim3(256), 0, st, res, n, reinterpret_cast<uint4 *>(heads));
}

uint32_t heavy_blocks_per_cu()
{
    static int cached = 0;
    if (!cached) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_heavy, 64, 0) != hipSuccess || n < 1) n = 8;
        cached = n;
    }
    return (uint32_t)cached;
}

void launch_heavy(const IndexView &ix, const AlignParams &ap, const uint32_t *pm, const uint4 *sai_c,
                  const uint4 *sai_r, salt_result_t *results, const uint32_t *queue, unsigned long long *ctr,
                  uint32_t n_blocks, uint32_t gap_blocks, void *lvtab, const GapBufs &g_in, uint32_t *ovq, uint32_t *qheads, uint8_t *pe_scr, hipEvent_t *ev3, hipStream_t st)
{
    if (!ap.n_reads) { if (ev3) for (int i = 0; i < 3; ++i) hipEventRecord(ev3[i], st); return; }
    uint32_t blocks = n_blocks < ap.n_reads ? n_blocks : ap.n_reads;
    LvTables *tab = static_cast<LvTables *>(lvtab);
    GapBufs g = g_in; g.ovq = ovq; g.qheads = qheads;
    // The usual shape (7.5 KB of LDS per block) when the batch fits it: at most 32 seed slots per strand and k_gap's buffers to hand gapped
    // reads to; the few reads it cannot finish (no k_gap slot left) wait in the overflow queue for the pass right behind it, which runs
    // the all-in-one shape (18.6 KB) and leaves at once when the queue is empty.  SALT_GPU_HEAVY_BIG=1: the all-in-one shape for everything.
    static const bool force_big = getenv("SALT_GPU_HEAVY_BIG") && atoi(getenv("SALT_GPU_HEAVY_BIG"));
    const bool se_glob = !ap.pe && pe_scr != nullptr;                  // single end, -m above the LDS list
    if (se_glob) {
        hipLaunchKernelGGL(k_heavy_glob, dim3(blocks), dim3(64), 0, st, ix, ap, pm, sai_c, sai_r, results, queue, ctr, tab, g, pe_scr, 0);
    }
    const bool small = !se_glob && !force_big && g.cap > 0 && ovq != nullptr && ap.spr <= (uint32_t)WaveLdsSmall::N_SLOTS;
    if (small) {
        if (ap.pe) hipLaunchKernelGGL(k_heavy_pe, dim3(blocks), dim3(64), 0, st, ix, ap, pm, sai_c, sai_r, results, queue, ctr, tab, g, pe_scr, 0);
        else     hipLaunchKernelGGL(k_heavy, dim3(blocks), dim3(64), 0, st, ix, ap, pm, sai_c, sai_r, results, queue, ctr, tab, g, pe_scr, 0);
    }
    // (the overflow pass on a small grid: it is empty nearly always, and 1 536 blocks of 18.6 KB each that only look at a counter cost 70 us)
    const uint32_t big_blocks = small && blocks > 256u ? 256u : blocks;
    if (se_glob) { }                                                    // (done above)
    else if (ap.pe) hipLaunchKernelGGL(k_heavy_pe_big, dim3(big_blocks), dim3(64), 0, st, ix, ap, pm, sai_c, sai_r, results, queue, ctr, tab, g, pe_scr, small ? 1 : 0);
    else     hipLaunchKernelGGL(k_heavy_big, dim3(big_blocks), dim3(64), 0, st, ix, ap, pm, sai_c, sai_r, results, queue, ctr, tab, g, pe_scr, small ? 1 : 0);
    if (ev3) hipEventRecord(ev3[0], st);
    if (!g.cap) { if (ev3) { hipEventRecord(ev3[1], st); hipEventRecord(ev3[2], st); } return; }
    // the deferred gapped passes: distances by (read, strand, 32 candidates), one finishing wave per read, one traceback per wave
    // grids: k_gap has many ~70 us items and wants every wave; the other two have few, short items and start faster on fewer blocks
    hipLaunchKernelGGL(k_gap, dim3(gap_blocks), dim3(64), 0, st, ix, ap, pm, g);
    if (ev3) hipEventRecord(ev3[1], st);
    hipLaunchKernelGGL(k_gapfin, dim3((n_blocks + 3) / 4), dim3(64), 0, st, ix, ap, pm, results, g, ctr);
    if (ev3) hipEventRecord(ev3[2], st);
    hipLaunchKernelGGL(k_cigar, dim3((n_blocks + 1) / 2), dim3(64), 0, st, ix, ap.pg, pm, results, g.cq, g.gctl + QC(6), g.gctl + QC(7), g.cap * (1u + SALT_MAX_HITS), tab);
}

GapBufs gap_bufs_layout(uint8_t *base, uint32_t cap, uint32_t *gctl, size_t *bytes)
{
    GapBufs g; size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off = (off + n + 255) & ~(size_t)255; return base ? base + o : nullptr; };
    // an SE read needs at most 2 x 1000 entries; PE mates share the pool.  Beyond 32 768 slots the pool stops growing with them
    // (64 M rows = 335 MB): a batch in which EVERY read needs the gapped pass still finds a slot per read, and rows for all of them
    // unless they are repeat reads throughout (then the rest takes the overflow pass)
    { const uint64_t want = (uint64_t)cap * 2u * (uint64_t)MAXLOC; g.pool = (uint32_t)(want < (1ull << 26) ? want : (1ull << 26)); }
    g.items_cap = g.pool / LLV_N + 2u * cap;
    g.gq = reinterpret_cast<uint32_t *>(take((size_t)cap * 4));
    g.gn = reinterpret_cast<uint32_t *>(take((size_t)cap * 2 * 4));
    g.goff = reinterpret_cast<uint32_t *>(take((size_t)cap * 2 * 4));
    g.gloci = reinterpret_cast<uint32_t *>(take((size_t)g.pool * 4));
    g.ge = take((size_t)g.pool);
    g.gitems = reinterpret_cast<uint32_t *>(take((size_t)g.items_cap * 4));
    g.cq = reinterpret_cast<uint32_t *>(take((size_t)cap * (1 + SALT_MAX_HITS) * 4));
    g.gctl = gctl; g.cap = cap; g.ovq = nullptr;
    if (bytes) *bytes = off;
    return g;
}


} // namespace salt
