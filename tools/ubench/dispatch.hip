#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(64) k_empty(int *out) { if (out && blockIdx.x == 0xFFFFFFFF) out[0] = 1; }
__global__ void __launch_bounds__(64) k_lds(int *out, const int* in) { __shared__ int s[800]; s[threadIdx.x] = in[blockIdx.x & 1023]; __syncthreads(); if (s[(threadIdx.x+1)&63] == 12345) out[0] = 1; }
__global__ void __launch_bounds__(256) k_lds4(int *out, const int* in) { __shared__ int s[3200]; s[threadIdx.x] = in[(blockIdx.x*4 + (threadIdx.x>>6)) & 1023]; __syncthreads(); if (s[(threadIdx.x+1)&255] == 12345) out[0] = 1; }
int main() {
  int *d, *in; hipMalloc(&d, 4096); hipMalloc(&in, 4096); hipMemset(in, 0, 4096);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    float ms;
    hipEventRecord(a); hipLaunchKernelGGL(k_empty, dim3(1000000), dim3(64), 0, 0, d); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); printf("empty 1M x 64: %.3f ms\n", ms);
    hipEventRecord(a); hipLaunchKernelGGL(k_lds, dim3(1000000), dim3(64), 0, 0, d, in); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); printf("lds3k 1M x 64: %.3f ms\n", ms);
    hipEventRecord(a); hipLaunchKernelGGL(k_lds4, dim3(250000), dim3(256), 0, 0, d, in); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); printf("lds12k 250k x 256: %.3f ms\n", ms);
  }
  return 0;
}
