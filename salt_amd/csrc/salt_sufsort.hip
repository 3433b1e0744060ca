// salt_amd/csrc/salt_sufsort.hip -- index construction on the device (gfx950): a suffix sorter for texts of up to
// 2^32 - 2 symbols and the arrays salt-idx derives from a suffix array.
//
// What it replaces in the reference (paths under Index_src/): the incremental BWT builders bwt_bwtgen (bwt_gen.c) and
// Rbwt_bwt_bwtgen (4bit_bwt_gen.c:1044-1130), bwt_bwtupdate_core (bwtmisc.c:121-143), bwt_cal_sa (bwt.c:48-68),
// LKT_build_lookuptable (LookUpTable.c:70-150) and Rbwt_gen_sa (rbwt.c:424-475).  A BWT is canonical, so any correct suffix
// order gives the reference's bytes (tests/test_index_builder.py, tests/test_gpu_index_build.py).
//
// Suffix sorter = prefix doubling over an LSD radix sort (rocPRIM's device radix sort is the one library primitive used;
// everything around it is written here):
//   round 0   key(p) = the first K symbols of suffix p packed into 64 bits (K = 32 for the 2-bit genome alphabet, 21 for
//             the 3-bit alphabet {A,C,G,T,#} of the local-pattern text), zero padded past the end.  Pairs (key, p) enter the
//             stable sort with p DESCENDING, so among equal keys the shorter suffix comes first, which is the order a
//             smallest terminator gives; a suffix shorter than K is unique by its length and is made a group of its own.
//   round r   only suffixes still tied with a neighbour take part: key = (rank of the group, rank of the suffix h symbols
//             further on) -- Larsson-Sadakane doubling -- sorted, re-ranked, and the groups that became singletons drop out.
//             h doubles every round; a tied suffix always has p + h <= n (a suffix that reaches the end is unique).
// Memory: about 33 bytes per symbol at the peak (round 0: two key and two value buffers), 102 GB for 3.1e9 bases: sized
// for the 288 GB of an MI355X, which is what makes one flat sort possible instead of a bucketed external one.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <chrono>
#include "../../include/salt_gpu.h"

namespace {

thread_local std::string s_err;
int sfail(int code, const std::string &m) { s_err = m; return code; }
#define SCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return sfail(SALT_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

constexpr uint32_t TPB = 256;
inline uint32_t grid_for(uint64_t n, uint32_t per_block = TPB)
{
    uint64_t b = (n + per_block - 1) / per_block;
    if (b > (1u << 20)) b = 1u << 20;                 // grid-stride loops: never more than 2^28 threads whatever n is
    return b ? (uint32_t)b : 1u;
}
#define GSTRIDE(i, n) for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, s_ = (uint64_t)gridDim.x * blockDim.x; i < (n); i += s_)

// ---- round 0: packed K-symbol keys, positions descending ---------------------------------------------------------------------
template <int BITS>
__global__ void __launch_bounds__(TPB) k_keys0(const uint8_t *__restrict__ text, uint64_t n, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    constexpr int K = 64 / BITS;
    __shared__ uint8_t tile[TPB + K];
    const uint64_t n_tiles = (n + TPB - 1) / TPB;
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t p0 = t * TPB;
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < TPB + K; j += TPB) tile[j] = p0 + j < n ? text[p0 + j] : 0;
        __syncthreads();
        const uint64_t p = p0 + threadIdx.x;
        if (p < n) {
            uint64_t key = 0;
#pragma unroll
            for (int j = 0; j < K; ++j) key = (key << BITS) | tile[threadIdx.x + j];
            const uint64_t i = n - 1 - p;                                  // element i holds position n-1-i
            keys[i] = key; vals[i] = (uint32_t)p;
        }
    }
}

// group starts after round 0: a new key, or a suffix shorter than K on either side
__global__ void __launch_bounds__(TPB) k_flags0(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ sa, uint64_t n, uint32_t K,
                                                 uint8_t *__restrict__ flag, uint32_t *__restrict__ start)
{
    GSTRIDE(i, n) {
        bool f = i == 0;
        if (!f) {
            const uint64_t a = sa[i - 1], b = sa[i];
            f = keys[i] != keys[i - 1] || a + K > n || b + K > n;
        }
        flag[i] = f;
        start[i] = f ? (uint32_t)i : 0u;
    }
}

// rank of suffix p = 1 + slot of the first member of its group (0 is kept for the empty suffix at p = n)
__global__ void __launch_bounds__(TPB) k_rank0(const uint32_t *__restrict__ sa, const uint32_t *__restrict__ start, uint64_t n, uint32_t *__restrict__ rank)
{
    GSTRIDE(i, n) rank[sa[i]] = start[i] + 1u;
}

// tied = member of a group of two or more
__global__ void __launch_bounds__(TPB) k_tied(const uint8_t *__restrict__ flag, uint64_t m, uint32_t *__restrict__ tied)
{
    GSTRIDE(j, m) tied[j] = !(flag[j] && (j + 1 == m || flag[j + 1])) ? 1u : 0u;
}

// compacts the tied elements: slot (identity when slot_in == nullptr) and position
__global__ void __launch_bounds__(TPB) k_compact(const uint8_t *__restrict__ flag, const uint32_t *__restrict__ dst, uint64_t m,
                                                  const uint32_t *__restrict__ slot_in, const uint32_t *__restrict__ p_in,
                                                  uint32_t *__restrict__ slot_out, uint32_t *__restrict__ p_out)
{
    GSTRIDE(j, m) {
        const bool tied = !(flag[j] && (j + 1 == m || flag[j + 1]));
        if (tied) { const uint32_t d = dst[j]; slot_out[d] = slot_in ? slot_in[j] : (uint32_t)j; p_out[d] = p_in[j]; }
    }
}

__global__ void __launch_bounds__(TPB) k_keys2(const uint32_t *__restrict__ p, const uint32_t *__restrict__ rank, uint64_t m, uint64_t n, uint64_t h,
                                                uint32_t shift, uint64_t *__restrict__ keys)
{
    GSTRIDE(j, m) {
        const uint64_t q = (uint64_t)p[j] + h;
        const uint32_t r2 = q >= n ? 0u : rank[q];                      // q == n: the empty suffix; q > n cannot be tied (see the header)
        keys[j] = ((uint64_t)rank[p[j]] << shift) | r2;
    }
}

__global__ void __launch_bounds__(TPB) k_flags2(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ slot, uint64_t m,
                                                 uint8_t *__restrict__ flag, uint32_t *__restrict__ start)
{
    GSTRIDE(j, m) {
        const bool f = j == 0 || keys[j] != keys[j - 1];
        flag[j] = f;
        start[j] = f ? slot[j] : 0u;
    }
}

__global__ void __launch_bounds__(TPB) k_apply2(const uint32_t *__restrict__ slot, const uint32_t *__restrict__ p, const uint32_t *__restrict__ start, uint64_t m,
                                                 uint32_t *__restrict__ sa, uint32_t *__restrict__ rank)
{
    GSTRIDE(j, m) { sa[slot[j]] = p[j]; rank[p[j]] = start[j] + 1u; }
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    hipError_t alloc(uint64_t bytes) { if (p) { hipFree(p); p = nullptr; } return hipMalloc(&p, bytes ? bytes : 1); }
    void release() { if (p) { hipFree(p); p = nullptr; } }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Suffix array of d_text[0..n) (one symbol per byte, values < 2^bits, bits = 2 or 3) WITHOUT the empty suffix: d_sa[n] positions in
// suffix order (the smallest terminator convention), d_rank[n+1] with d_rank[p] = 1 + row of suffix p in d_sa and d_rank[n] = 0 -- i.e.
// d_rank[p] is the row of p in the full suffix array that starts with the empty suffix.  Both arrays are caller-allocated.
int suffix_sort(const uint8_t *d_text, uint64_t n, int bits, uint32_t *d_sa, uint32_t *d_rank, int verbose)
{
    if (n == 0) return SALT_OK;
    if (n >= 0xFFFFFFF0ull) return sfail(SALT_E_INVAL, "text too long for 32-bit suffix positions");
    if (bits != 2 && bits != 3) return sfail(SALT_E_INVAL, "suffix_sort: 2 or 3 bits per symbol");
    const uint32_t K = bits == 2 ? 32u : 21u;
    const double t0 = now_s();
    DevBuf keys_a, keys_b, vals_a, flag, start, tmp;
    size_t tmp_bytes = 0;
    SCHK(keys_a.alloc(n * 8)); SCHK(keys_b.alloc(n * 8)); SCHK(vals_a.alloc(n * 4));
    SCHK(flag.alloc(n)); SCHK(start.alloc(n * 4));
    if (bits == 2) hipLaunchKernelGGL(k_keys0<2>, dim3(grid_for(n)), dim3(TPB), 0, nullptr, d_text, n, keys_a.as<uint64_t>(), vals_a.as<uint32_t>());
    else hipLaunchKernelGGL(k_keys0<3>, dim3(grid_for(n)), dim3(TPB), 0, nullptr, d_text, n, keys_a.as<uint64_t>(), vals_a.as<uint32_t>());
    SCHK(hipGetLastError());
    const unsigned key_bits = bits == 2 ? 64u : 63u;                    // 32 x 2 or 21 x 3 bits, in the low bits of the key
    SCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), vals_a.as<uint32_t>(), d_sa, (size_t)n, 0u, key_bits, nullptr));
    SCHK(tmp.alloc(tmp_bytes));
    SCHK(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), vals_a.as<uint32_t>(), d_sa, (size_t)n, 0u, key_bits, nullptr));
    hipLaunchKernelGGL(k_flags0, dim3(grid_for(n)), dim3(TPB), 0, nullptr, keys_b.as<uint64_t>(), d_sa, n, K, flag.as<uint8_t>(), start.as<uint32_t>());
    SCHK(hipGetLastError());
    keys_a.release(); keys_b.release(); vals_a.release();
    // group starts: running maximum of (flag ? slot : 0)
    size_t scan_bytes = 0;
    SCHK(rocprim::inclusive_scan(nullptr, scan_bytes, start.as<uint32_t>(), start.as<uint32_t>(), (size_t)n, rocprim::maximum<uint32_t>(), nullptr));
    if (scan_bytes > tmp_bytes) { SCHK(tmp.alloc(scan_bytes)); tmp_bytes = scan_bytes; }
    SCHK(rocprim::inclusive_scan(tmp.p, scan_bytes, start.as<uint32_t>(), start.as<uint32_t>(), (size_t)n, rocprim::maximum<uint32_t>(), nullptr));
    SCHK(hipMemsetAsync(d_rank + n, 0, 4, nullptr));
    hipLaunchKernelGGL(k_rank0, dim3(grid_for(n)), dim3(TPB), 0, nullptr, d_sa, start.as<uint32_t>(), n, d_rank);
    SCHK(hipGetLastError());
    // ---- tied elements of round 0 ----
    auto count_tied = [&](const uint8_t *f, uint64_t m, uint32_t *dst, uint64_t *out) -> int {
        hipLaunchKernelGGL(k_tied, dim3(grid_for(m)), dim3(TPB), 0, nullptr, f, m, dst);
        SCHK(hipGetLastError());
        uint32_t last_flag = 0, last_sum = 0;
        SCHK(hipMemcpy(&last_flag, dst + (m - 1), 4, hipMemcpyDeviceToHost));
        size_t b = 0;
        SCHK(rocprim::exclusive_scan(nullptr, b, dst, dst, 0u, (size_t)m, rocprim::plus<uint32_t>(), nullptr));
        if (b > tmp_bytes) { SCHK(tmp.alloc(b)); tmp_bytes = b; }
        SCHK(rocprim::exclusive_scan(tmp.p, b, dst, dst, 0u, (size_t)m, rocprim::plus<uint32_t>(), nullptr));
        SCHK(hipMemcpy(&last_sum, dst + (m - 1), 4, hipMemcpyDeviceToHost));
        *out = (uint64_t)last_sum + last_flag;
        return SALT_OK;
    };
    uint64_t m = 0;
    int rc = count_tied(flag.as<uint8_t>(), n, start.as<uint32_t>(), &m);
    if (rc) return rc;
    if (verbose) fprintf(stderr, "[sufsort] n = %llu, %d bits: round 0 (%u symbols) %.2f s, %llu suffixes still tied\n", (unsigned long long)n, bits, K, now_s() - t0, (unsigned long long)m);
    if (m == 0) return SALT_OK;
    DevBuf slot_a, slot_b, p_a, p_b, k2_a, k2_b;
    SCHK(slot_a.alloc(m * 4)); SCHK(p_a.alloc(m * 4));
    hipLaunchKernelGGL(k_compact, dim3(grid_for(n)), dim3(TPB), 0, nullptr, flag.as<uint8_t>(), start.as<uint32_t>(), n, (const uint32_t *)nullptr, d_sa,
                       slot_a.as<uint32_t>(), p_a.as<uint32_t>());
    SCHK(hipGetLastError());
    SCHK(hipDeviceSynchronize());
    flag.release(); start.release();
    SCHK(slot_b.alloc(m * 4)); SCHK(p_b.alloc(m * 4)); SCHK(k2_a.alloc(m * 8)); SCHK(k2_b.alloc(m * 8));
    SCHK(flag.alloc(m)); SCHK(start.alloc(m * 4));
    uint32_t shift = 1;
    while ((1ull << shift) <= n + 1) ++shift;                           // bits of a rank (<= n)
    uint64_t h = K;
    for (int round = 1; m > 0; ++round, h *= 2) {
        const double t1 = now_s();
        hipLaunchKernelGGL(k_keys2, dim3(grid_for(m)), dim3(TPB), 0, nullptr, p_a.as<uint32_t>(), d_rank, m, n, h, shift, k2_a.as<uint64_t>());
        SCHK(hipGetLastError());
        size_t b = 0;
        SCHK(rocprim::radix_sort_pairs(nullptr, b, k2_a.as<uint64_t>(), k2_b.as<uint64_t>(), p_a.as<uint32_t>(), p_b.as<uint32_t>(), (size_t)m, 0u, 2 * shift, nullptr));
        if (b > tmp_bytes) { SCHK(tmp.alloc(b)); tmp_bytes = b; }
        SCHK(rocprim::radix_sort_pairs(tmp.p, b, k2_a.as<uint64_t>(), k2_b.as<uint64_t>(), p_a.as<uint32_t>(), p_b.as<uint32_t>(), (size_t)m, 0u, 2 * shift, nullptr));
        hipLaunchKernelGGL(k_flags2, dim3(grid_for(m)), dim3(TPB), 0, nullptr, k2_b.as<uint64_t>(), slot_a.as<uint32_t>(), m, flag.as<uint8_t>(), start.as<uint32_t>());
        SCHK(hipGetLastError());
        b = 0;
        SCHK(rocprim::inclusive_scan(nullptr, b, start.as<uint32_t>(), start.as<uint32_t>(), (size_t)m, rocprim::maximum<uint32_t>(), nullptr));
        if (b > tmp_bytes) { SCHK(tmp.alloc(b)); tmp_bytes = b; }
        SCHK(rocprim::inclusive_scan(tmp.p, b, start.as<uint32_t>(), start.as<uint32_t>(), (size_t)m, rocprim::maximum<uint32_t>(), nullptr));
        hipLaunchKernelGGL(k_apply2, dim3(grid_for(m)), dim3(TPB), 0, nullptr, slot_a.as<uint32_t>(), p_b.as<uint32_t>(), start.as<uint32_t>(), m, d_sa, d_rank);
        SCHK(hipGetLastError());
        uint64_t m2 = 0;
        rc = count_tied(flag.as<uint8_t>(), m, start.as<uint32_t>(), &m2);
        if (rc) return rc;
        if (m2) {
            hipLaunchKernelGGL(k_compact, dim3(grid_for(m)), dim3(TPB), 0, nullptr, flag.as<uint8_t>(), start.as<uint32_t>(), m, slot_a.as<uint32_t>(), p_b.as<uint32_t>(),
                               slot_b.as<uint32_t>(), p_a.as<uint32_t>());
            SCHK(hipGetLastError());
            std::swap(slot_a.p, slot_b.p);
        }
        SCHK(hipDeviceSynchronize());
        if (verbose) fprintf(stderr, "[sufsort] round %d (%llu symbols): %llu sorted in %.2f s, %llu still tied\n", round, (unsigned long long)(2 * h), (unsigned long long)m, now_s() - t1, (unsigned long long)m2);
        m = m2;
        if (h > 2 * n) return sfail(SALT_E_INDEX, "suffix_sort did not converge");
    }
    SCHK(hipDeviceSynchronize());
    if (verbose) fprintf(stderr, "[sufsort] done in %.2f s\n", now_s() - t0);
    return SALT_OK;
}

// ---- arrays derived from the suffix array ---------------------------------------------------------------------------------------
// BWT symbol i of the $-removed BWT: row r = i + (i >= primary) of the full suffix array (row 0 = the empty suffix)
__device__ __forceinline__ uint32_t bwt_sym(const uint8_t *text, const uint32_t *sa, uint64_t n, uint32_t primary, uint64_t i)
{
    const uint64_t r = i + (i >= primary);
    const uint64_t p = r == 0 ? n : sa[r - 1];
    return text[p - 1];
}

// C index: one thread per 16-symbol word of the 2-bit BWT (first symbol in the top bits, bwt.h:57-64) + its symbol counts (4 x 8 bit)
__global__ void __launch_bounds__(TPB) k_c_words(const uint8_t *__restrict__ text, const uint32_t *__restrict__ sa, uint64_t n, uint32_t primary,
                                                  uint64_t n_words, uint32_t *__restrict__ words, uint32_t *__restrict__ wcnt)
{
    GSTRIDE(w, n_words) {
        uint32_t word = 0, cnt = 0;
        for (uint32_t q = 0; q < 16; ++q) {
            const uint64_t i = w * 16 + q;
            if (i >= n) break;
            const uint32_t s = bwt_sym(text, sa, n, primary, i);
            word |= s << ((15u - q) << 1);
            cnt += 1u << (8 * s);
        }
        words[w] = word; wcnt[w] = cnt;
    }
}
// per 128-symbol block: symbol counts (SoA, one array per base)
__global__ void __launch_bounds__(TPB) k_c_blockcnt(const uint32_t *__restrict__ wcnt, uint64_t n_words, uint64_t n_blocks, uint32_t *__restrict__ c0, uint32_t *__restrict__ c1,
                                                     uint32_t *__restrict__ c2, uint32_t *__restrict__ c3)
{
    GSTRIDE(b, n_blocks) {
        uint32_t a[4] = { 0, 0, 0, 0 };
        for (uint32_t j = 0; j < 8; ++j) {
            const uint64_t w = b * 8 + j;
            if (w >= n_words) break;
            const uint32_t c = wcnt[w];
            a[0] += c & 255u; a[1] += (c >> 8) & 255u; a[2] += (c >> 16) & 255u; a[3] += c >> 24;
        }
        c0[b] = a[0]; c1[b] = a[1]; c2[b] = a[2]; c3[b] = a[3];
    }
}
// interleave: per block 4 running counts, then its words (bwt_bwtupdate_core, bwtmisc.c:121-143); n_blocks includes the final counts-only block
__global__ void __launch_bounds__(TPB) k_c_interleave(const uint32_t *__restrict__ words, uint64_t n_words, uint64_t n_blocks, const uint32_t *__restrict__ c0,
                                                       const uint32_t *__restrict__ c1, const uint32_t *__restrict__ c2, const uint32_t *__restrict__ c3,
                                                       uint32_t *__restrict__ out)
{
    GSTRIDE(t, n_blocks * 12) {
        const uint64_t b = t / 12; const uint32_t j = (uint32_t)(t % 12);
        // block b starts at word offset 4*b + 8*b unless it is the counts-only block that ends the array
        const uint64_t full_blocks = (n_words + 7) / 8;               // blocks that hold words
        if (b < full_blocks) {
            if (j < 4) out[b * 12 + j] = j == 0 ? c0[b] : j == 1 ? c1[b] : j == 2 ? c2[b] : c3[b];
            else { const uint64_t w = b * 8 + (j - 4); if (w < n_words) out[b * 12 + j] = words[w]; }
        } else if (b == full_blocks && j < 4) {
            out[n_words + full_blocks * 4 + j] = j == 0 ? c0[b] : j == 1 ? c1[b] : j == 2 ? c2[b] : c3[b];
        }
    }
}
// suffix-array samples: sa_out[j] = SA[j * intv] of the full suffix array, sa_out[0] = -1 (bwt_cal_sa, bwt.c:48-68; bwtio.c:30-50)
__global__ void __launch_bounds__(TPB) k_sa_sample(const uint32_t *__restrict__ sa, uint64_t n_sa, uint32_t intv, uint32_t *__restrict__ out)
{
    GSTRIDE(j, n_sa) out[j] = j == 0 ? 0xFFFFFFFFu : sa[j * intv - 1];
}
// 12-mer table: one count per suffix start p in [0, n] with the tail padded with A (LKT_build_lookuptable, LookUpTable.c:70-150)
__global__ void __launch_bounds__(TPB) k_lkt_hist(const uint8_t *__restrict__ text, uint64_t n, uint32_t len, uint32_t *__restrict__ item)
{
    GSTRIDE(p, n + 1) {
        uint32_t x = 0;
        for (uint32_t j = 0; j < len; ++j) x = (x << 2) | (p + j < n ? text[p + j] : 0u);
        atomicAdd(item + x + 1, 1u);
    }
}
// R index: 8 symbols per word, first in the top nibble (rbwt.h:115-119)
__global__ void __launch_bounds__(TPB) k_r_words(const uint8_t *__restrict__ text, const uint32_t *__restrict__ sa, uint64_t n, uint32_t inv_sa0,
                                                  uint64_t n_words, uint32_t *__restrict__ words)
{
    GSTRIDE(w, n_words) {
        uint32_t word = 0;
        for (uint32_t q = 0; q < 8; ++q) {
            const uint64_t i = w * 8 + q;
            if (i >= n) break;
            word |= bwt_sym(text, sa, n, inv_sa0, i) << ((7u - q) * 4u);
        }
        words[w] = word;
    }
}
// saValueSharp (Rbwt_gen_sa with direction -1, rbwt.c:424-475): for the '#' that opens segment j the value is
// header(record of the '#' two places further on) - (length of segment j + 1)
__global__ void __launch_bounds__(TPB) k_r_sharp(const uint32_t *__restrict__ rank, const uint32_t *__restrict__ sharp_off, const uint32_t *__restrict__ sharp_hdr,
                                                  uint64_t n_sharp, uint32_t cum4, uint32_t *__restrict__ rsa)
{
    GSTRIDE(j, n_sharp - 1) {
        const uint32_t row = rank[sharp_off[j]];
        const uint32_t seg_len = sharp_off[j + 1] - sharp_off[j] - 1;
        const uint32_t hdr = j + 2 < n_sharp ? sharp_hdr[j + 2] : 0u;
        rsa[row - cum4 - 1] = hdr - (seg_len + 1);
    }
}

struct Sorted {            // device text + its suffix array, shared by the two builders
    DevBuf text, sa, rank;
};
int upload_and_sort(int device, const uint8_t *text, uint64_t n, int bits, Sorted &s, int verbose)
{
    int n_dev = 0;
    SCHK(hipGetDeviceCount(&n_dev));
    if (n_dev <= 0) return sfail(SALT_E_HIP, "no HIP device visible: the device index builder cannot run");
    SCHK(hipSetDevice(device));
    SCHK(s.text.alloc(n + 64)); SCHK(s.sa.alloc(n * 4)); SCHK(s.rank.alloc((n + 1) * 4));
    SCHK(hipMemcpy(s.text.p, text, n, hipMemcpyHostToDevice));
    SCHK(hipMemset(s.text.as<uint8_t>() + n, 0, 64));
    return suffix_sort(s.text.as<uint8_t>(), n, bits, s.sa.as<uint32_t>(), s.rank.as<uint32_t>(), verbose);
}

int verbose_env() { const char *e = getenv("SALT_IDX_VERBOSE"); return e && atoi(e) ? 1 : 0; }

} // namespace

extern "C" const char *salt_gpu_idx_last_error(void) { return s_err.c_str(); }

extern "C" int salt_gpu_suffix_array(int device, const uint8_t *text, uint64_t n, int bits, uint32_t *sa_out)
{
    if (!text || !sa_out) return sfail(SALT_E_INVAL, "null argument");
    for (uint64_t i = 0; i < n; ++i) if (text[i] >> bits) return sfail(SALT_E_INVAL, "symbol outside the alphabet");
    Sorted s;
    int rc = upload_and_sort(device, text, n, bits, s, verbose_env());
    if (rc) return rc;
    sa_out[0] = (uint32_t)n;
    if (n) SCHK(hipMemcpy(sa_out + 1, s.sa.p, n * 4, hipMemcpyDeviceToHost));
    return SALT_OK;
}

extern "C" uint64_t salt_gpu_idx_c_bwt_words(uint64_t n) { return (n + 15) / 16 + ((n + 127) / 128 + 1) * 4; }

extern "C" int salt_gpu_idx_build_c(int device, const uint8_t *text, uint64_t n, uint32_t sa_intv, uint32_t *primary_out, uint32_t L2[5],
                                    uint32_t *bwt, uint32_t *sa, uint32_t *lkt, uint32_t lkt_len)
{
    if (!text || !primary_out || !L2 || !bwt || !sa || !lkt || n == 0 || sa_intv == 0 || lkt_len < 1 || lkt_len > 14) return sfail(SALT_E_INVAL, "bad argument");
    const int verbose = verbose_env();
    Sorted s;
    int rc = upload_and_sort(device, text, n, 2, s, verbose);
    if (rc) return rc;
    const double t0 = now_s();
    uint32_t primary = 0;
    SCHK(hipMemcpy(&primary, s.rank.p, 4, hipMemcpyDeviceToHost));       // row of suffix 0
    s.rank.release();
    const uint64_t n_words = (n + 15) / 16, n_blocks = (n + 127) / 128 + 1, out_words = n_words + n_blocks * 4;
    DevBuf words, wcnt, c[4], out, tmp;
    SCHK(words.alloc(n_words * 4)); SCHK(wcnt.alloc(n_words * 4));
    hipLaunchKernelGGL(k_c_words, dim3(grid_for(n_words)), dim3(TPB), 0, nullptr, s.text.as<uint8_t>(), s.sa.as<uint32_t>(), n, primary, n_words, words.as<uint32_t>(), wcnt.as<uint32_t>());
    SCHK(hipGetLastError());
    for (auto &b : c) SCHK(b.alloc(n_blocks * 4));
    hipLaunchKernelGGL(k_c_blockcnt, dim3(grid_for(n_blocks)), dim3(TPB), 0, nullptr, wcnt.as<uint32_t>(), n_words, n_blocks, c[0].as<uint32_t>(), c[1].as<uint32_t>(), c[2].as<uint32_t>(), c[3].as<uint32_t>());
    SCHK(hipGetLastError());
    size_t tb = 0;
    SCHK(rocprim::exclusive_scan(nullptr, tb, c[0].as<uint32_t>(), c[0].as<uint32_t>(), 0u, (size_t)n_blocks, rocprim::plus<uint32_t>(), nullptr));
    SCHK(tmp.alloc(tb));
    uint32_t tot[4];
    for (int k = 0; k < 4; ++k) {
        SCHK(rocprim::exclusive_scan(tmp.p, tb, c[k].as<uint32_t>(), c[k].as<uint32_t>(), 0u, (size_t)n_blocks, rocprim::plus<uint32_t>(), nullptr));
        SCHK(hipMemcpy(&tot[k], c[k].as<uint32_t>() + (n_blocks - 1), 4, hipMemcpyDeviceToHost));   // the last block holds no symbols: its exclusive sum is the total
    }
    // the block after the last word-holding block carries the totals (when n % 128 == 0 that is block n / 128, else the partial block's successor)
    SCHK(out.alloc(out_words * 4));
    SCHK(hipMemset(out.p, 0, out_words * 4));
    hipLaunchKernelGGL(k_c_interleave, dim3(grid_for(n_blocks * 12)), dim3(TPB), 0, nullptr, words.as<uint32_t>(), n_words, n_blocks, c[0].as<uint32_t>(), c[1].as<uint32_t>(),
                       c[2].as<uint32_t>(), c[3].as<uint32_t>(), out.as<uint32_t>());
    SCHK(hipGetLastError());
    SCHK(hipMemcpy(bwt, out.p, out_words * 4, hipMemcpyDeviceToHost));
    out.release(); words.release(); wcnt.release();
    L2[0] = 0;
    for (int k = 0; k < 4; ++k) L2[k + 1] = L2[k] + tot[k];
    *primary_out = primary;
    // suffix-array samples
    const uint64_t n_sa = (n + sa_intv) / sa_intv;
    DevBuf sas;
    SCHK(sas.alloc(n_sa * 4));
    hipLaunchKernelGGL(k_sa_sample, dim3(grid_for(n_sa)), dim3(TPB), 0, nullptr, s.sa.as<uint32_t>(), n_sa, sa_intv, sas.as<uint32_t>());
    SCHK(hipGetLastError());
    SCHK(hipMemcpy(sa, sas.p, n_sa * 4, hipMemcpyDeviceToHost));
    sas.release(); s.sa.release();
    // k-mer table
    const uint64_t n_item = (1ull << (2 * lkt_len)) + 1;
    DevBuf item;
    SCHK(item.alloc(n_item * 4));
    SCHK(hipMemset(item.p, 0, n_item * 4));
    hipLaunchKernelGGL(k_lkt_hist, dim3(grid_for(n + 1)), dim3(TPB), 0, nullptr, s.text.as<uint8_t>(), n, lkt_len, item.as<uint32_t>());
    SCHK(hipGetLastError());
    tb = 0;
    SCHK(rocprim::inclusive_scan(nullptr, tb, item.as<uint32_t>(), item.as<uint32_t>(), (size_t)n_item, rocprim::plus<uint32_t>(), nullptr));
    SCHK(tmp.alloc(tb));
    SCHK(rocprim::inclusive_scan(tmp.p, tb, item.as<uint32_t>(), item.as<uint32_t>(), (size_t)n_item, rocprim::plus<uint32_t>(), nullptr));
    SCHK(hipMemcpy(lkt, item.p, n_item * 4, hipMemcpyDeviceToHost));
    SCHK(hipDeviceSynchronize());
    if (verbose) fprintf(stderr, "[idx-gpu] C index: BWT + Occ + SA samples + %u-mer table derived and copied back in %.2f s\n", lkt_len, now_s() - t0);
    return SALT_OK;
}

extern "C" int salt_gpu_idx_build_r(int device, const uint8_t *rtext, uint64_t n, const uint32_t *sharp_off, const uint32_t *sharp_hdr, uint64_t n_sharp,
                                    uint32_t cum4, uint32_t *inv_sa0_out, uint32_t *code, uint64_t code_words, uint32_t *rsa)
{
    if (!rtext || !sharp_off || !sharp_hdr || !inv_sa0_out || !code || !rsa || n == 0 || n_sharp == 0) return sfail(SALT_E_INVAL, "bad argument");
    if (code_words * 8 < n) return sfail(SALT_E_INVAL, "code buffer shorter than the text");
    const int verbose = verbose_env();
    Sorted s;
    int rc = upload_and_sort(device, rtext, n, 3, s, verbose);
    if (rc) return rc;
    const double t0 = now_s();
    uint32_t inv = 0;
    SCHK(hipMemcpy(&inv, s.rank.p, 4, hipMemcpyDeviceToHost));
    *inv_sa0_out = inv;
    DevBuf words, d_off, d_hdr, d_rsa;
    SCHK(words.alloc(code_words * 4));
    SCHK(hipMemset(words.p, 0, code_words * 4));
    hipLaunchKernelGGL(k_r_words, dim3(grid_for((n + 7) / 8)), dim3(TPB), 0, nullptr, s.text.as<uint8_t>(), s.sa.as<uint32_t>(), n, inv, (n + 7) / 8, words.as<uint32_t>());
    SCHK(hipGetLastError());
    SCHK(hipMemcpy(code, words.p, code_words * 4, hipMemcpyDeviceToHost));
    words.release(); s.sa.release();
    SCHK(d_off.alloc(n_sharp * 4)); SCHK(d_hdr.alloc(n_sharp * 4)); SCHK(d_rsa.alloc((n_sharp + 1) * 4));
    SCHK(hipMemcpy(d_off.p, sharp_off, n_sharp * 4, hipMemcpyHostToDevice));
    SCHK(hipMemcpy(d_hdr.p, sharp_hdr, n_sharp * 4, hipMemcpyHostToDevice));
    SCHK(hipMemset(d_rsa.p, 0, (n_sharp + 1) * 4));
    if (n_sharp > 1) hipLaunchKernelGGL(k_r_sharp, dim3(grid_for(n_sharp)), dim3(TPB), 0, nullptr, s.rank.as<uint32_t>(), d_off.as<uint32_t>(), d_hdr.as<uint32_t>(), n_sharp, cum4, d_rsa.as<uint32_t>());
    SCHK(hipGetLastError());
    SCHK(hipMemcpy(rsa, d_rsa.p, (n_sharp + 1) * 4, hipMemcpyDeviceToHost));
    SCHK(hipDeviceSynchronize());
    if (verbose) fprintf(stderr, "[idx-gpu] R index: BWT + '#' rows derived and copied back in %.2f s\n", now_s() - t0);
    return SALT_OK;
}
