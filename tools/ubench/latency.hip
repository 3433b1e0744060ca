// dependent-load round-trip cost under the k_heavy-like regime: W one-wave blocks per CU, each wave
// chases a chain of coalesced 256-B rows (64 lanes x 4 B) in a table of T MiB.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void __launch_bounds__(64) chase(const unsigned *tab, unsigned n_rows, int iters, unsigned long long *cyc, unsigned *sink)
{
    unsigned row = (blockIdx.x * 2654435761u) % n_rows;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) {
        unsigned v = tab[(size_t)row * 64 + threadIdx.x];
        acc += v;
        row = (unsigned)__shfl((int)v, 0) % n_rows;      // next row depends on the loaded data
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { atomicAdd(cyc, t1 - t0); sink[blockIdx.x & 1023] = acc; }
}
int main(int argc, char **argv)
{
    int mib = argc > 1 ? atoi(argv[1]) : 160, per_cu = argc > 2 ? atoi(argv[2]) : 9, iters = 200;
    size_t n_rows = (size_t)mib * 1024 * 1024 / 256;
    unsigned *h = (unsigned *)malloc(n_rows * 256);
    unsigned x = 12345;
    for (size_t i = 0; i < n_rows * 64; ++i) { x = x * 1664525u + 1013904223u; h[i] = x >> 4; }
    unsigned *d, *sink; unsigned long long *cyc;
    hipMalloc(&d, n_rows * 256); hipMemcpy(d, h, n_rows * 256, hipMemcpyHostToDevice);
    hipMalloc(&sink, 4096); hipMalloc(&cyc, 8);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    for (int w = 1; w <= per_cu; w *= 3) {
        int blocks = p.multiProcessorCount * w;
        hipMemset(cyc, 0, 8);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a); hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, d, (unsigned)n_rows, iters, cyc, sink); hipEventRecord(b);
        hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("table %d MiB, %d waves/CU: %.0f cycles per dependent 256-B row load, kernel %.3f ms (%.2f us per hop)\n", mib, w, (double)c / blocks / iters, ms, ms * 1e3 / iters);
    }
    return 0;
}
