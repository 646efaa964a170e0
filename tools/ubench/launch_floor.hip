// Launch-period floor on gfx950: back-to-back launches of (a) an empty kernel, (b) a kernel that loads and stores 62 floats per
// lane with the access pattern of the step kernel's prologue / epilogue, 256 blocks x 64 threads, same stream, HIP events.
// build: hipcc -O3 --offload-arch=gfx950 -o launch_floor launch_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty() {}
__global__ void k_copy(const float *__restrict__ in, float *__restrict__ out, int n) {
    int e = blockIdx.x * 64 + threadIdx.x;
    float acc[62];
#pragma unroll
    for (int j = 0; j < 62; j++) acc[j] = in[j * n + e];
#pragma unroll
    for (int j = 0; j < 62; j++) out[j * n + e] = acc[j] * 1.0001f;
}
int main() {
    const int n = 16384, iters = 3000;
    float *a, *b; hipMalloc(&a, 62 * n * 4); hipMalloc(&b, 62 * n * 4); hipMemset(a, 0, 62 * n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int w = 0; w < 100; w++) hipLaunchKernelGGL(k_empty, dim3(256), dim3(64), 0, 0);
    hipEventRecord(e0);
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k_empty, dim3(256), dim3(64), 0, 0);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("empty kernel, 256 x 64:            %.2f us per launch\n", ms * 1e3 / iters);
    hipEventRecord(e0);
    for (int i = 0; i < iters; i++) { hipLaunchKernelGGL(k_copy, dim3(256), dim3(64), 0, 0, a, b, n); float *t = a; a = b; b = t; }
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("load 62 + store 62 floats per lane: %.2f us per launch (ping-pong buffers, each launch reads what the last one wrote)\n", ms * 1e3 / iters);
    return 0;
}
