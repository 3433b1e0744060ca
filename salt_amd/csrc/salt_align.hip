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

#define WSYNC() __syncthreads()     /* blocks are exactly one wave: orders LDS traffic between lanes */

static constexpr int MAXL = SALT_MAX_READ_LEN;
static constexpr int SLOTS = SALT_MAX_SEED_SLOTS;
static constexpr int MAXLOC = SALT_MAX_LOCATE;
static constexpr int LVK = 31;                  // MAX_K (LandauVishkin.c:13)
static constexpr int NHIT = 6;                  // first hits kept per strand (5 + the primary)
static constexpr uint32_t INF = 255;

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ uint32_t read_base(const uint8_t *seq, uint32_t L, int strand, uint32_t i)
{
    if (strand == 0) return seq[i];
    uint32_t c = seq[L - 1 - i];
    return c < 4 ? 3 - c : c;                    // query_seq_reverse (query.c:46-71)
}

// ---------------------------------------------------------------------------------------------
// k_seed
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_seed(IndexView ix, SeedParams sp, const uint8_t *__restrict__ seqs, const uint32_t *__restrict__ offs,
       uint4 *__restrict__ sai_c, uint4 *__restrict__ sai_r, unsigned long long *__restrict__ ctr)
{
    const uint64_t item = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n_lkt = 0, n_occ_c = 0, n_occ_r = 0;
    if (item < (uint64_t)sp.n_reads * 2u * sp.spr) {
        const uint32_t slot = (uint32_t)(item % sp.spr);
        const uint32_t rs = (uint32_t)(item / sp.spr);
        const int strand = (int)(rs & 1u);
        const uint32_t r = rs >> 1;
        const uint32_t off = offs[r], L = offs[r + 1] - off;
        const uint8_t *seq = seqs + off;
        const uint32_t k = (uint32_t)sp.l_seed, s = slot * (uint32_t)sp.l_overlap;
        uint4 oc = make_uint4(1, 0, 0, 0), orr = make_uint4(1, 0, 0, 0);
        if (L >= k && s + k <= L) {
            const uint32_t e = s + k - 1, lk = ix.lkt_len;
            // 12-mer at the seed tail (LKT_seq2LktItem, lookup.c:163-177)
            uint32_t x = 0; bool has_n = false;
            for (uint32_t t = 0; t < lk; ++t) {
                uint32_t c = read_base(seq, L, strand, e - lk + 1 + t);
                has_n |= c > 3; x = (x << 2) | (c & 3u);
            }
            uint32_t kc = 1, lc = 0, kr = 1, lr = 0;
            bool alive_c = !has_n, alive_r = !has_n && !sp.seed_only_ref;
            if (alive_c) { kc = ix.lkt[x]; lc = ix.lkt[x + 1] - 1; alive_c = kc <= lc; ++n_lkt; }
            int i_r_start;                                   // first head index the R search still has to consume
            if (ix.r_lkt_len == lk) {
                if (alive_r) { uint2 v = ix.r_lkt[x]; kr = v.x; lr = v.y; alive_r = kr <= lr; n_occ_r += 2 * lk; }
                i_r_start = (int)(k - lk) - 1;
            } else {                                          // no R table: start from the full range
                kr = 0; lr = ix.r_text_len; i_r_start = (int)k - 1;
                alive_r = !sp.seed_only_ref;
            }
            // joint backward search over the seed head, newest base last (bwt.c:281-309, rbwt.c:619-648)
            for (int i = i_r_start; i >= 0 && (alive_c || alive_r); --i) {
                uint32_t c = read_base(seq, L, strand, s + (uint32_t)i);
                if (c > 3) { alive_c = false; alive_r = false; break; }
                if (alive_c && i < (int)(k - lk)) {
                    uint32_t ok = c_occ(ix, kc - 1, c), ol = c_occ(ix, lc, c);
                    kc = ix.c_L2[c] + ok + 1; lc = ix.c_L2[c] + ol; alive_c = kc <= lc; n_occ_c += 2;
                }
                if (alive_r) {
                    uint32_t ok = r_occ(ix, kr, c), ol = r_occ(ix, lr + 1, c);
                    kr = ix.r_cum[c] + ok + 1; lr = ix.r_cum[c] + ol; alive_r = kr <= lr; n_occ_r += 2;
                }
            }
            if (alive_c) {                                    // shrink big intervals leftwards (alnse.c:246-258)
                uint32_t ext = 0;
                while (lc - kc > sp.max_seed && ext < s) {
                    uint32_t c = read_base(seq, L, strand, s - ext - 1);
                    if (c > 3) break;
                    uint32_t ok = c_occ(ix, kc - 1, c), ol = c_occ(ix, lc, c);
                    n_occ_c += 2;
                    if (ok + 1 > ol) break;
                    kc = ix.c_L2[c] + ok + 1; lc = ix.c_L2[c] + ol; ++ext;
                    if (lc - kc <= sp.max_seed) break;
                }
                oc = make_uint4(kc, lc, s - ext, 1);
            }
            if (alive_r) {                                    // same, without the N guard (alnse.c:279-291)
                uint32_t ext = 0;
                while (lr - kr > sp.max_seed && ext < s) {
                    uint32_t c = read_base(seq, L, strand, s - ext - 1);
                    uint32_t ok = r_occ(ix, kr, c), ol = r_occ(ix, lr + 1, c);
                    n_occ_r += 2;
                    if (ok + 1 > ol) break;
                    kr = ix.r_cum[c] + ok + 1; lr = ix.r_cum[c] + ol; ++ext;
                    if (lr - kr <= sp.max_seed) break;
                }
                orr = make_uint4(kr, lr, s - ext, 1);
            }
        }
        sai_c[item] = oc;
        sai_r[item] = orr;
    }
    if (ctr) {                                                // one atomic per wave and counter
        for (int o = 32; o > 0; o >>= 1) {
            n_lkt += __shfl_down(n_lkt, o); n_occ_c += __shfl_down(n_occ_c, o); n_occ_r += __shfl_down(n_occ_r, o);
        }
        if (lane_id() == 0) {
            atomicAdd(ctr + SALT_CTR_LKT, n_lkt); atomicAdd(ctr + SALT_CTR_OCC_C, n_occ_c);
            atomicAdd(ctr + SALT_CTR_OCC_R, n_occ_r);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_align: LDS of one wave
// ---------------------------------------------------------------------------------------------
struct SaiLists { uint32_t sp[2][SLOTS], ep[2][SLOTS], off[2][SLOTS]; };     // [0]=C, [1]=R
struct LvTables { short L[LVK][64]; char A[LVK][64]; };
struct WaveLds {
    uint8_t  seq[2][MAXL];
    uint32_t pm[2][MAXL / 8];
    union { SaiLists sai; LvTables lv; } u;
    uint32_t loci[MAXLOC];
    uint8_t  lvT[MAXL + 4 + 64];
    uint8_t  lvP[MAXL + 64];
    uint32_t hit_pos[2][NHIT];
    uint8_t  hit_nd[2][NHIT], hit_gap[2][NHIT];
    uint16_t cig[SALT_MAX_CIGAR_OPS];
    int      n_cig;
};

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

__device__ void sai_introsort(SaiLists &s, int w, int n)
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

// ---- candidates of one strand: gather seeds, order them, locate, sort, dedup ----------------------
// Leaves the candidate positions in w.loci[0..return).  gap_mode selects the range filter of
// alnse_check_withgap (alnse.c:894) instead of alnse_check_nogap's (alnse.c:762).
__device__ uint32_t build_candidates(const IndexView &ix, const AlignParams &ap, WaveLds &w, uint32_t r, int strand,
                                     uint32_t L, const uint4 *sai_c, const uint4 *sai_r, bool gap_mode,
                                     uint32_t &n_sa_c, uint32_t &n_sa_r, uint32_t &n_loci_out)
{
    const uint32_t lane = lane_id();
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint64_t base_item = ((uint64_t)r * 2u + (uint32_t)strand) * ap.spr;
    uint32_t n_list[2] = { 0, 0 };
    // gather valid seeds in seed order (n_C / n_back_R grow in seed_start order, alnse.c:265-300)
    for (int which = 0; which < 2; ++which) {
        const uint4 *src = which == 0 ? sai_c : sai_r;
        uint32_t n = 0;
        for (uint32_t b = 0; b < ap.spr; b += 64) {
            uint32_t slot = b + lane;
            uint4 v = make_uint4(1, 0, 0, 0);
            if (slot < ap.spr) v = src[base_item + slot];
            uint64_t m = __ballot(v.w != 0);
            if (v.w) { uint32_t at = n + (uint32_t)__popcll(m & lt); w.u.sai.sp[which][at] = v.x; w.u.sai.ep[which][at] = v.y; w.u.sai.off[which][at] = v.z; }
            n += (uint32_t)__popcll(m);
        }
        n_list[which] = n;
    }
    WSYNC();
    if (lane < 2) sai_introsort(w.u.sai, (int)lane, (int)n_list[lane]);      // alnse.c:307-308
    WSYNC();
    // locate under the global cap (alnse_locate_alt, alnse.c:633-731)
    uint32_t n = 0;
    bool full = false;
    for (uint32_t i = 0; i < n_list[0] && !full; ++i) {
        const uint32_t sp = w.u.sai.sp[0][i], ep = w.u.sai.ep[0][i], off = w.u.sai.off[0][i];
        for (uint64_t j0 = sp; j0 <= ep && !full; j0 += 64) {
            uint64_t j = j0 + lane;
            bool in = j <= ep, keep = false;
            uint32_t pos = 0;
            if (in) { pos = ix.c_sa[j] - off; keep = !(pos + L > ix.ref_len); }        // u32 wrap as in alnse.c:672-673
            uint64_t m = __ballot(keep);
            uint32_t rank = (uint32_t)__popcll(m & lt);
            if (keep && n + rank < ap.max_locate) w.loci[n + rank] = pos;
            uint32_t tot = (uint32_t)__popcll(m);
            if (n + tot >= ap.max_locate) {
                // lookups the sequential loop would have made before stopping
                uint64_t last = m;                       // position of the max_locate-th kept lane
                uint32_t need = ap.max_locate - n;       // >= 1
                for (uint32_t q = 1; q < need; ++q) last &= last - 1;
                uint32_t stop_lane = (uint32_t)__ffsll((long long)last) - 1;
                n_sa_c += stop_lane + 1;
                n = ap.max_locate; full = true;
            } else { n += tot; n_sa_c += (uint32_t)__popcll(__ballot(in)); }
        }
    }
    for (uint32_t i = 0; i < n_list[1] && !full; ++i) {
        const uint32_t sp = w.u.sai.sp[1][i], ep = w.u.sai.ep[1][i], off = w.u.sai.off[1][i];
        uint32_t skip = (ep + 1 - sp) / 0x40000u;                                       // alnse.c:707-708
        if ((int)skip <= 0) skip = 1;
        for (uint64_t j0 = sp; j0 <= ep && !full; j0 += 64ull * skip) {
            uint64_t j = j0 + (uint64_t)lane * skip;
            bool in = j <= ep, keep = false;
            uint32_t pos = 0;
            if (in) { pos = ix.r_pos[j] - off; keep = !(pos > ix.ref_len || pos + L > ix.ref_len); }   // alnse.c:715-717
            uint64_t m = __ballot(keep);
            uint32_t rank = (uint32_t)__popcll(m & lt);
            if (keep && n + rank < ap.max_locate) w.loci[n + rank] = pos;
            uint32_t tot = (uint32_t)__popcll(m);
            if (n + tot >= ap.max_locate) {
                uint64_t last = m; uint32_t need = ap.max_locate - n;
                for (uint32_t q = 1; q < need; ++q) last &= last - 1;
                n_sa_r += (uint32_t)__ffsll((long long)last);
                n = ap.max_locate; full = true;
            } else { n += tot; n_sa_r += (uint32_t)__popcll(__ballot(in)); }
        }
    }
    n_loci_out += n;
    WSYNC();
    sort_loci(w.loci, n);
    WSYNC();
    // drop duplicates and out-of-range loci, keeping order (alnse.c:758-762 / 890-894)
    uint32_t n_out = 0;
    for (uint32_t b = 0; b < n; b += 64) {
        uint32_t i = b + lane;
        bool keep = false; uint32_t pos = 0;
        if (i < n) {
            pos = w.loci[i];
            bool dup = i > 0 && w.loci[i - 1] == pos;
            bool out = gap_mode ? (pos + L + 4 >= ix.ref_len) : (pos >= ix.ref_len);
            keep = !dup && !out;
        }
        uint64_t m = __ballot(keep);
        WSYNC();                                           // all reads of this chunk are done
        if (keep) w.loci[n_out + (uint32_t)__popcll(m & lt)] = pos;
        n_out += (uint32_t)__popcll(m);
        WSYNC();
    }
    return n_out;
}

// ---- masked Hamming distance, capped: returns 0..3 or INF (ed_mismatch, editdistance.c:88-163) ----
__device__ __forceinline__ uint32_t mismatch_capped(const IndexView &ix, const uint32_t *pm, uint32_t L, uint32_t pos)
{
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

// ---- Landau-Vishkin on byte masks, one diagonal per lane ---------------------------------------
// T: text masks (tlen bytes, zero padded), P: one-hot pattern (plen bytes, zero padded).
// Returns e (<= k), or -1.  When tab != nullptr also fills the L / action tables and returns the
// finishing diagonal in d_fin (order 0,-1,1,... LandauVishkin.c:248); otherwise the order is
// irrelevant for the distance (LandauVishkin.c:67).
__device__ int lv_wave(const uint8_t *T, int tlen, const uint8_t *P, int plen, int k, LvTables *tab, int &d_fin)
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

// unpack text masks / one-hot pattern for LV (editdistance.c:183-227)
__device__ void lv_unpack(const IndexView &ix, WaveLds &w, int strand, uint32_t L, uint32_t pos)
{
    const uint32_t tlen = L + 4;
    for (uint32_t i = lane_id(); i < tlen + 48; i += 64) {
        uint32_t p = pos + i;
        w.lvT[i] = i < tlen ? (uint8_t)((ix.ref[p >> 3] >> (4 * (p & 7u))) & 15u) : (uint8_t)0;
    }
    for (uint32_t i = lane_id(); i < L + 48; i += 64) {
        uint8_t c = i < L ? w.seq[strand][i] : (uint8_t)5;
        w.lvP[i] = c > 4 ? (uint8_t)0 : (c > 3 ? (uint8_t)15 : (uint8_t)(1u << c));
    }
    WSYNC();
}

// CIGAR of a gapped hit into w.cig / w.n_cig (computeEditDistanceWithCigar, useM=1) ------------------
__device__ void lv_cigar(const IndexView &ix, WaveLds &w, int strand, uint32_t L, uint32_t pos, int k)
{
    lv_unpack(ix, w, strand, L, pos);
    int d_fin = 0;
    WSYNC();
    int e = lv_wave(w.lvT, (int)L + 4, w.lvP, (int)L, k, &w.u.lv, d_fin);
    WSYNC();
    if (lane_id() == 0) {
        int n = 0;
        uint16_t *cg = w.cig;
        if (e == 0) { cg[n++] = (uint16_t)((L << 4) | 0u); }
        else if (e > 0) {
            char act[LVK + 1]; int matched[LVK + 1];
            int cd = d_fin;
            for (int ce = e; ce >= 1; --ce) {
                char a = w.u.lv.A[ce][cd + 31];
                act[ce] = a;
                int cur = w.u.lv.L[ce][cd + 31];
                if (a == 'I') { matched[ce] = cur - w.u.lv.L[ce - 1][cd + 1 + 31] - 1; cd += 1; }
                else if (a == 'D') { matched[ce] = cur - w.u.lv.L[ce - 1][cd - 1 + 31]; cd -= 1; }
                else { matched[ce] = cur - w.u.lv.L[ce - 1][cd + 31] - 1; }
            }
            int acc = w.u.lv.L[0][31];
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
__global__ void __launch_bounds__(64)
k_align(IndexView ix, AlignParams ap, const uint8_t *__restrict__ seqs, const uint32_t *__restrict__ offs,
        const uint4 *__restrict__ sai_c, const uint4 *__restrict__ sai_r, salt_result_t *__restrict__ results,
        unsigned long long *__restrict__ ctr)
{
    __shared__ WaveLds w;
    const uint32_t lane = lane_id();
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint32_t r = blockIdx.x;
    const uint32_t off = offs[r], L = offs[r + 1] - off;
    salt_result_t *out = results + r;
    uint32_t c_sa_c = 0, c_sa_r = 0, c_verify = 0, c_vwords = 0, c_lv = 0, c_loci = 0;

    // ---- load the read, both strands (query.c:177-183, 46-71) ----
    uint32_t n_amb = 0;
    for (uint32_t b = 0; b < L; b += 64) {
        uint32_t i = b + lane; bool isn = false;
        if (i < L) {
            uint8_t c = seqs[off + i];
            if (c > 4) c = 4;
            w.seq[0][i] = c; w.seq[1][L - 1 - i] = c < 4 ? (uint8_t)(3 - c) : c;
            isn = c > 3;
        }
        n_amb += (uint32_t)__popcll(__ballot(isn));
    }
    WSYNC();
    // result defaults (query_read_seq, query.c:199-206)
    uint32_t q_pos = 0xFFFFFFFFu; uint32_t q_strand = 3, q_ndiff = 255, q_gap = 255;
    const bool too_short = L < (uint32_t)ap.l_seed;
    if (n_amb > 200) {                                        // alnse.c:1328: record left untouched
        if (lane == 0) {
            out->pos = q_pos; out->strand = 3; out->n_diff = 255; out->is_gap = 255; out->mapq = 0;
            out->b0 = -1; out->b1 = -1; out->seq_start = 0; out->seq_end = (uint16_t)(L - 1);
            out->n_hits[0] = out->n_hits[1] = 0; out->n_cigar = 0; out->skipped = 1;
        }
        return;
    }
    // one-hot masks, 8 bases per word, LSB first like the mixRef (editdistance.c:40)
    const uint32_t nw = (L + 7) >> 3;
    for (uint32_t t = lane; t < 2 * nw; t += 64) {
        uint32_t s = t >= nw, j = s ? t - nw : t, word = 0;
        for (uint32_t q = 0; q < 8; ++q) {
            uint32_t i = j * 8 + q;
            uint32_t c = i < L ? w.seq[s][i] : 5u;
            uint32_t msk = c < 4 ? (1u << c) : (c == 4 ? 15u : 0u);
            word |= msk << (4 * q);
        }
        w.pm[s][j] = word;
    }
    WSYNC();

    // ---- gap-free pass over both strands (alnse.c:1077-1084) ----
    uint32_t bound = 3;
    bool found[2] = { false, false };
    uint32_t n_hits_s[2] = { 0, 0 };            // hits recorded (<= NHIT) per strand
    uint32_t a0[2] = { 0, 0 };                  // n_diff of the first hit of each list
    if (!too_short)
    for (int strand = 0; strand < 2; ++strand) {
        uint32_t n_cand = build_candidates(ix, ap, w, r, strand, L, sai_c, sai_r, false, c_sa_c, c_sa_r, c_loci);
        uint32_t call_best_n = INF, call_best_pos = 0;
        for (uint32_t b = 0; b < n_cand; b += 64) {
            uint32_t i = b + lane, v = INF, pos = 0;
            if (i < n_cand) { pos = w.loci[i]; v = mismatch_capped(ix, w.pm[strand], L, pos); }
            // ballots by value: m[t] = lanes with v <= t
            uint64_t m0 = __ballot(v <= 0), m1 = __ballot(v <= 1), m2 = __ballot(v <= 2), m3 = __ballot(v <= 3);
            // a candidate passes iff v <= bound and no earlier candidate of this chunk is smaller
            uint64_t smaller = v == 0 ? 0ull : (v == 1 ? m0 : (v == 2 ? m1 : m2));
            bool pass = v <= bound && (smaller & lt) == 0;
            uint64_t pm = __ballot(pass);
            if (pm) {
                uint32_t vmin = m0 & pm ? 0u : (m1 & pm ? 1u : (m2 & pm ? 2u : 3u));   // smallest passing value
                // record the first hits in order
                uint32_t rank = n_hits_s[strand] + (uint32_t)__popcll(pm & lt);
                if (pass && rank < NHIT) { w.hit_pos[strand][rank] = pos; w.hit_nd[strand][rank] = (uint8_t)v; w.hit_gap[strand][rank] = 0; }
                if (n_hits_s[strand] == 0) a0[strand] = (uint32_t)__shfl((int)v, __ffsll((long long)pm) - 1);
                uint32_t add = (uint32_t)__popcll(pm);
                n_hits_s[strand] = n_hits_s[strand] + add > NHIT ? NHIT : n_hits_s[strand] + add;
                if (vmin < call_best_n) {
                    uint64_t at = __ballot(pass && v == vmin);
                    call_best_n = vmin;
                    call_best_pos = (uint32_t)__shfl((int)pos, __ffsll((long long)at) - 1);
                }
                found[strand] = true;
                bound = vmin < bound ? vmin : bound;
            }
            (void)m3;
        }
        c_verify += n_cand;
        for (uint32_t b = lane; b < n_cand; b += 64) c_vwords += ((w.loci[b] & 7u) + L + 7) >> 3;
        if (found[strand]) { q_pos = call_best_pos; q_ndiff = call_best_n; q_gap = 0; q_strand = (uint32_t)strand; }
        WSYNC();
    }

    // ---- gapped pass (alnse.c:1089-1096): sequential per candidate, LV across lanes ----
    if (!too_short && !found[0] && !found[1]) {
        int maxd = (int)(L / 10);
        for (int strand = 0; strand < 2; ++strand) {
            uint32_t n_cand = build_candidates(ix, ap, w, r, strand, L, sai_c, sai_r, true, c_sa_c, c_sa_r, c_loci);
            bool any = false;
            for (uint32_t i = 0; i < n_cand; ++i) {
                uint32_t pos = w.loci[i];
                int e = -1;
                if (!(pos > ix.ref_len || pos + L + 4 > ix.ref_len)) {        // ed_diff guard (editdistance.c:178)
                    lv_unpack(ix, w, strand, L, pos);
                    int dd;
                    e = lv_wave(w.lvT, (int)L + 4, w.lvP, (int)L, maxd, nullptr, dd);
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
        uint64_t q = (uint64_t)255 * x / (uint32_t)b0;
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
            lv_cigar(ix, w, (int)q_strand, L, q_pos, (int)q_ndiff);
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
                lv_cigar(ix, w, s, L, w.hit_pos[s][h], (int)w.hit_nd[s][h]);
                if (lane < (uint32_t)w.n_cig) out->hit_cigar[hidx][lane] = w.cig[lane];
                if (lane == 0) out->hit_n_cigar[hidx] = (uint8_t)w.n_cig;
            } else if (lane == 0) out->hit_n_cigar[hidx] = 0;
        }

    if (ctr) {
        for (int o = 32; o > 0; o >>= 1) c_vwords += __shfl_down(c_vwords, o);
        if (lane == 0) {
            atomicAdd(ctr + SALT_CTR_SA_C, c_sa_c); atomicAdd(ctr + SALT_CTR_SA_R, c_sa_r);
            atomicAdd(ctr + SALT_CTR_VERIFY, c_verify); atomicAdd(ctr + SALT_CTR_VERIFY_WORDS, c_vwords);
            atomicAdd(ctr + SALT_CTR_LV, c_lv); atomicAdd(ctr + SALT_CTR_READS, 1ull);
            atomicAdd(ctr + SALT_CTR_BASES, L); atomicAdd(ctr + SALT_CTR_LOCI, c_loci);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// launch wrappers (called from salt_gpu.hip)
// ---------------------------------------------------------------------------------------------
void launch_seed(const IndexView &ix, const SeedParams &sp, const uint8_t *seqs, const uint32_t *offs, uint4 *sai_c,
                 uint4 *sai_r, unsigned long long *ctr, hipStream_t st)
{
    uint64_t items = (uint64_t)sp.n_reads * 2u * sp.spr;
    if (!items) return;
    uint32_t blocks = (uint32_t)((items + 255) / 256);
    hipLaunchKernelGGL(k_seed, dim3(blocks), dim3(256), 0, st, ix, sp, seqs, offs, sai_c, sai_r, ctr);
}

void launch_align(const IndexView &ix, const AlignParams &ap, const uint8_t *seqs, const uint32_t *offs, const uint4 *sai_c,
                  const uint4 *sai_r, salt_result_t *results, unsigned long long *ctr, hipStream_t st)
{
    if (!ap.n_reads) return;
    hipLaunchKernelGGL(k_align, dim3(ap.n_reads), dim3(64), 0, st, ix, ap, seqs, offs, sai_c, sai_r, results, ctr);
}

} // namespace salt
