// What a direct-to-LDS load (global_load_lds_dword / _dwordx4) costs the issuing wave, against a plain global_load_dword:
// one wave per SIMD (1024 waves), each lane issues N loads back to back; s_memrealtime around the issue and around the wait.
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/lds_dma_rate tools/ubench/lds_dma_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
#define N 24
template <int MODE>
__global__ __launch_bounds__(256) void k(const float *src, float *dst, size_t stride, unsigned long long *times) {
    __shared__ float buf[4][N * 64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    float *mine = buf[wave];
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    float v[N];
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = src[t + i * stride];
    } else if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < N; ++i) __builtin_amdgcn_global_load_lds((gptr_t *)(src + t + i * stride), (lptr_t *)(mine + i * 64), 4, 0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < N / 4; ++i) __builtin_amdgcn_global_load_lds((gptr_t *)(src + (t + i * stride) * 4), (lptr_t *)(mine + i * 256), 16, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) acc += v[i];
    } else if (MODE == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < N; ++i) acc += mine[i * 64 + lane];
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < N; ++i) acc += mine[(i / 4) * 256 + lane * 4 + (i & 3)];
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    dst[t] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) { times[0] = t1 - t0; times[1] = t2 - t1; }
}
int main() {
    const int blocks = 256;
    const size_t n = (size_t)blocks * 256, stride = n;
    float *src, *dst; unsigned long long *tm;
    hipMalloc(&src, n * N * 4 * sizeof(float)); hipMalloc(&dst, n * sizeof(float)); hipMalloc(&tm, 16);
    hipMemset(src, 0, n * N * 4 * sizeof(float));
    const char *names[3] = {"global_load_dword -> VGPR", "global_load_lds_dword", "global_load_lds_dwordx4 (same bytes)"};
    for (int mode = 0; mode < 3; ++mode) {
        unsigned long long acc[2] = {0, 0};
        for (int rep = 0; rep < 20; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, src, dst, stride, tm);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, src, dst, stride, tm);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, src, dst, stride, tm);
            unsigned long long h[2]; hipMemcpy(h, tm, 16, hipMemcpyDeviceToHost);
            if (rep >= 4) { acc[0] += h[0]; acc[1] += h[1]; }
        }
        printf("%-40s %d loads per lane: issue %6.0f ns, wait + read back %6.0f ns\n", names[mode], mode == 2 ? N / 4 : N, acc[0] * 10.0 / 16, acc[1] * 10.0 / 16);
    }
    return 0;
}
