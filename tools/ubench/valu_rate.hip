// Issue-rate microbenchmark for gfx950: wave64 v_fma_f32 vs v_pk_fma_f32, dependent chain vs 8 independent accumulators,
// 1 or 2 waves per SIMD.  Prints cycles per instruction per wave (s_memtime / wall clock of the kernel).
// build: hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 2048
#define REP 8      // 8 x 8 = 64 fma per loop iteration
template <int MODE> __global__ __launch_bounds__(64) void k(float *out, float a, float b) {
    float x[8]; f2 y[8];
    for (int i = 0; i < 8; i++) { x[i] = threadIdx.x * 0.001f + i; y[i] = f2{x[i], x[i] + 1.f}; }
    f2 a2 = {a, a * 1.0001f}, b2 = {b, b * 0.9999f};
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) {          // dependent scalar chain: 8 fma on one accumulator
#pragma unroll
            for (int r = 0; r < REP; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) x[0] = __builtin_fmaf(x[0], a, b);
        } else if (MODE == 1) {   // 8 independent scalar chains
#pragma unroll
            for (int r = 0; r < REP; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) x[i] = __builtin_fmaf(x[i], a, b);
        } else if (MODE == 2) {   // dependent packed chain
#pragma unroll
            for (int r = 0; r < REP; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) y[0] = __builtin_elementwise_fma(y[0], a2, b2);
        } else {                  // 8 independent packed chains
#pragma unroll
            for (int r = 0; r < REP; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) y[i] = __builtin_elementwise_fma(y[i], a2, b2);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, int waves_per_simd, float *d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 1024 * waves_per_simd;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 0.999f, 0.001f);
    hipEventRecord(e0);
    for (int q = 0; q < 5; q++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 0.999f, 0.001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double inst = (double)ITERS * 8 * REP;                          // fma instructions per wave
    double cyc = ms * 1e-3 * 2.4e9;                           // at the 2.4 GHz peak clock
    printf("%-28s waves/SIMD %d  %7.3f ms  %.2f cycles per fma per wave, %.2f per SIMD issue slot\n", name, waves_per_simd, ms, cyc / inst,
           cyc / inst / waves_per_simd);
}
int main() {
    float *d; hipMalloc(&d, 4096 * 64 * 4);
    for (int w = 1; w <= 2; w++) {
        run<0>("v_fma_f32 dependent", w, d);
        run<1>("v_fma_f32 8 independent", w, d);
        run<2>("v_pk_fma_f32 dependent", w, d);
        run<3>("v_pk_fma_f32 8 independent", w, d);
    }
    return 0;
}
