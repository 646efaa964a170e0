// Issue-rate microbenchmark for gfx950: wave64 v_fma_f32 vs v_pk_fma_f32, dependent chain vs 8 independent accumulators,
// 1 or 2 waves per SIMD.  Prints cycles per instruction per wave (s_memtime / wall clock of the kernel).
// build: hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 2048
#define REP 8      // 8 x 8 = 64 fma per loop iteration
template <int MODE> __global__ __launch_bounds__(64) void k(float *out, float a, float b, unsigned long long *clk) {
    unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
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
        } else if (MODE == 4) {   // 8 independent v_mul_f32 by a literal
#pragma unroll
            for (int r = 0; r < REP; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) x[i] = x[i] * 1.0001f;
        } else if (MODE == 5) {   // 8 independent v_add_f32 of an inline constant
#pragma unroll
            for (int r = 0; r < REP; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) x[i] = x[i] + 0.5f;
        } else if (MODE == 6) {   // 8 independent fma with one VGPR and two literals (v_fmaak / v_fmamk)
#pragma unroll
            for (int r = 0; r < REP; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) x[i] = __builtin_fmaf(x[i], 0.9999f, 0.0001f);
        } else if (MODE == 7) {   // 8 independent fma with three VGPR sources
#pragma unroll
            for (int r = 0; r < REP; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) x[i] = __builtin_fmaf(x[i], x[(i + 1) & 7], x[(i + 2) & 7]);
        } else if (MODE == 10) {  // 8 independent packed chains, VGPR-pair sources only (no SGPR / constant-bus operand)
#pragma unroll
            for (int r = 0; r < REP; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) y[i] = __builtin_elementwise_fma(y[i], y[(i + 1) & 7], y[(i + 2) & 7]);
        } else if (MODE == 11) {  // 8 independent packed multiplies by an SGPR pair
#pragma unroll
            for (int r = 0; r < REP; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) y[i] = y[i] * a2;
        } else if (MODE == 12) {  // 8 independent packed multiplies, VGPR pairs only
#pragma unroll
            for (int r = 0; r < REP; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) y[i] = y[i] * y[(i + 1) & 7];
        } else if (MODE == 8) {   // dependent chain, VGPR sources only (no SGPR / constant-bus operand): x0 = fma(x0, x1, x2)
#pragma unroll
            for (int r = 0; r < REP; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) x[0] = __builtin_fmaf(x[0], x[1], x[2]);
        } else if (MODE == 9) {   // two interleaved dependent chains, VGPR sources only (ILP 2)
#pragma unroll
            for (int r = 0; r < REP; r++)
#pragma unroll
                for (int i = 0; i < 4; i++) { x[0] = __builtin_fmaf(x[0], x[2], x[3]); x[1] = __builtin_fmaf(x[1], x[2], x[3]); }
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
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = wall_clock64() - r0; }
}
template <int MODE> void run(const char *name, int waves_per_simd, float *d) {
    static unsigned long long *clk = nullptr; if (!clk) hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 1024 * waves_per_simd;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 0.999f, 0.001f, clk);
    hipEventRecord(e0);
    for (int q = 0; q < 5; q++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 0.999f, 0.001f, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double inst = (double)ITERS * 8 * REP;                          // fma instructions per wave
    double cyc = ms * 1e-3 * 2.4e9;                           // at the 2.4 GHz peak clock
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    // wave 0's own counters: s_memtime ticks and the 100 MHz constant clock over its whole run
    printf("   [wave 0: %llu s_memtime ticks in %.1f us -> %.0f MHz tick rate, %.2f ticks per fma]\n", h[0], h[1] / 100.0, h[0] / (h[1] / 100.0), (double)h[0] / inst);
    printf("%-28s waves/SIMD %d  %7.3f ms  %.2f cycles per fma per wave, %.2f per SIMD issue slot\n", name, waves_per_simd, ms, cyc / inst,
           cyc / inst / waves_per_simd);
}
int main() {
    float *d; hipMalloc(&d, 8192 * 64 * 4);
    for (int w = 1; w <= 2; w++) {      // round 2: packed FP32 with and without a constant-bus operand
        run<3>("v_pk_fma_f32 8 indep, SGPR pairs", w, d);
        run<10>("v_pk_fma_f32 8 indep, VGPR only", w, d);
        run<11>("v_pk_mul_f32 8 indep, SGPR pair", w, d);
        run<12>("v_pk_mul_f32 8 indep, VGPR only", w, d);
    }
    if (getenv("QG_UBENCH_OCC"))
    for (int w = 1; w <= 8; w++) {      // round 2: does occupancy hide dependent-issue stalls when no SGPR operand is involved?
        if (w == 5 || w == 7) continue;
        run<8>("v_fma_f32 dependent, VGPR only", w, d);
        run<9>("v_fma_f32 2 chains, VGPR only", w, d);
        run<7>("v_fma_f32 3 VGPRs 8 indep", w, d);
    }
    if (getenv("QG_UBENCH_ALL"))
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("v_fma_f32 dependent", w, d);
        run<1>("v_fma_f32 8 independent", w, d);
        run<2>("v_pk_fma_f32 dependent", w, d);
        run<3>("v_pk_fma_f32 8 independent", w, d);
        run<4>("v_mul_f32 literal 8 indep", w, d);
        run<5>("v_add_f32 inline 8 indep", w, d);
        run<6>("v_fmaak 2 literals 8 indep", w, d);
        run<7>("v_fma_f32 3 VGPRs 8 indep", w, d);
    }
    return 0;
}
