// Does a packed FP32 instruction slow down on denormal operands?  (round 4: the one-link-per-lane kernel with ~110 v_pk_* per substep ran
// 2-3x slower than the build without them, on the same data.)  One wave per SIMD; loops of 64 independent instructions; cycles from
// s_memtime.  usage: ./pk_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(64) void k(float seed, float mul, unsigned long long *cyc, float *sink) {
    f2 a[8];
    float s[8];
    for (int i = 0; i < 8; ++i) { a[i] = f2{seed * (i + 1), seed * (i + 2)}; s[i] = seed * (i + 3); }
    f2 m = {mul, mul};
    f2 c = {seed, seed};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(m), "v"(c));
                if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s[i]) : "v"(s[i]), "v"(mul), "v"(seed));
                if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(m));
                if (MODE == 3) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(c));
            }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
    for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y + s[i];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = acc;
}
int main() {
    unsigned long long *cyc; float *sink;
    hipMalloc(&cyc, 8); hipMalloc(&sink, 1024 * 64 * 4);
    const char *names[4] = {"v_pk_fma_f32", "v_fma_f32   ", "v_pk_mul_f32", "v_pk_add_f32"};
    struct { const char *what; float seed, mul; } cases[] = {{"normal operands (1.0 .. 9.0, x 1.0)", 1.0f, 1.0f},
                                                              {"denormal operands (1e-40 .., x 1.0)", 1e-40f, 1.0f},
                                                              {"results underflow to denormal (1e-20 x 1e-20)", 1e-20f, 1e-20f},
                                                              {"zero operands", 0.f, 1.0f}};
    for (auto &cs : cases)
        for (int mode = 0; mode < 4; ++mode) {
            unsigned long long h = 0;
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(64), 0, 0, cs.seed, cs.mul, cyc, sink);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1024), dim3(64), 0, 0, cs.seed, cs.mul, cyc, sink);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1024), dim3(64), 0, 0, cs.seed, cs.mul, cyc, sink);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1024), dim3(64), 0, 0, cs.seed, cs.mul, cyc, sink);
                hipDeviceSynchronize();
                hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            }
            printf("%-48s %s  %6.2f cycles per instruction (lone wave)\n", cs.what, names[mode], (double)h / (256.0 * 64));
        }
    return 0;
}
