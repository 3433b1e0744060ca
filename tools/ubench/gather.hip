// Calibration of FETCH_SIZE for the access shapes of the aligner: N independent random loads of W bytes per lane (W = 4: suffix-array
// rows, 16: W-mer table entries, 32: C Occ blocks, 64: R Occ blocks / verify windows as 4 lanes x 16 B) from a table far larger than
// the 256 MiB Infinity Cache.  Run under `rocprofv3 --pmc FETCH_SIZE`: FETCH_SIZE x 1024 / loads = bytes the fabric moves per load.
// Also prints the achieved load rate (loads/s and the sector bandwidth it implies).   usage: gather [table GiB] [Mloads] [blocks of 256 threads]
// (blocks: how many loads are in flight at once -- one per lane, the loop is not unrolled; default 8192 = every wave slot of the chip)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
__device__ __forceinline__ uint64_t mix(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
template <int W> __global__ void __launch_bounds__(256) k_gather(const uint32_t *tab, uint64_t n_rec, uint64_t n_loads, uint32_t *sink)
{
    uint32_t acc = 0;
#pragma unroll 1
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_loads; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = mix(i) % n_rec;
        if (W == 1024) {                                 // a random run of 64 consecutive 16-byte records per wave (a suffix-array interval's context rows)
            const uint64_t run = mix(i >> 6) % (n_rec / 64);
            const uint4 v = reinterpret_cast<const uint4 *>(tab)[run * 64 + (i & 63)]; acc ^= v.x ^ v.w;
        }
        else if (W == 4) acc ^= tab[r];
        else if (W == 16) { const uint4 v = reinterpret_cast<const uint4 *>(tab)[r]; acc ^= v.x ^ v.w; }
        else if (W == 32) { const uint4 *p = reinterpret_cast<const uint4 *>(tab) + 2 * r; const uint4 a = p[0], b = p[1]; acc ^= a.x ^ b.w; }
        else {                                           // 64 bytes by 4 neighbouring lanes x 16 B: lane quad q reads record mix(i / 4)
            const uint64_t rq = mix(i >> 2) % n_rec;
            const uint4 v = reinterpret_cast<const uint4 *>(tab)[rq * 4 + (i & 3)]; acc ^= v.x ^ v.w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
static uint32_t g_blocks = 256 * 32;
template <int W> void run(const uint32_t *d, uint64_t bytes, uint64_t n_loads, uint32_t *sink)
{
    const uint64_t n_rec = bytes / (W == 1024 ? 16 : W);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a); hipLaunchKernelGGL(k_gather<W>, dim3(g_blocks), dim3(256), 0, 0, d, n_rec, n_loads, sink); hipEventRecord(b);
        hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    const double per_rec = W == 64 || W == 1024 ? n_loads / 4.0 : (double)n_loads;      // sectors: four 16-byte lanes each
    printf("W = %4d B: %.0f M %s in %.3f ms = %.2f G/s; at one 64-B sector each = %.2f TB/s, useful %.2f TB/s\n", W, per_rec / 1e6, W == 1024 ? "sectors (runs of 64 rows x 16 B)" : "records", ms,
           per_rec / ms / 1e6, per_rec * 64 / ms / 1e9, per_rec * (W == 1024 ? 64 : W) / ms / 1e9);
}
int main(int argc, char **argv)
{
    const uint64_t gib = argc > 1 ? atoll(argv[1]) : 16, mloads = argc > 2 ? atoll(argv[2]) : 256;
    const uint64_t bytes = gib << 30, n_loads = mloads << 20;
    if (argc > 3) g_blocks = (uint32_t)atoi(argv[3]);
    uint32_t *d, *sink;
    if (hipMalloc(&d, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(d, 1, bytes); hipMalloc(&sink, 64);
    run<4>(d, bytes, n_loads, sink); run<16>(d, bytes, n_loads, sink); run<32>(d, bytes, n_loads, sink); run<64>(d, bytes, n_loads, sink); run<1024>(d, bytes, n_loads, sink);
    hipDeviceSynchronize();
    return 0;
}
