// What a RESIDENT step kernel would pay per env-step for its hand-offs, against a kernel boundary (gfx950, ROCm 7.2).
//
// A 256 x 256 grid (1024 waves, one per SIMD: the one-link-per-lane step kernel's geometry at 4096 envs) either
//   A. is launched once per step (K dependent launches on one stream), or
//   B. stays resident and is handed each step through memory: a 64-thread RING kernel on the caller's stream stores door = k
//      (sc1), the resident waves poll the door (one lane, sc1 load + s_sleep), load their 4 bytes of "action" (sc1), work, store their
//      output (sc1, write-through), drain, and one lane per workgroup adds to the done counter of its shard (blockIdx % 8); the ring
//      kernel polls the eight shards and exits -- the next kernel on the stream (a policy) sees the step's outputs;
//   C. as B, but ONE ring advances the door by the whole run (run-ahead: actions of all K steps are already in the ring of slots);
//   D. as B with hipStreamWriteValue32 / hipStreamWaitValue64 in place of the ring kernel (two-level arrival: the shard's last
//      arriver adds to one top word).
// WORK = length of a dependent FMA chain per step (0: the hand-offs alone; ~1750 at one instruction per ~4.6 cycles = ~3.4 us ...).
// Every spin in here has a deadline on the 100 MHz clock (s_memrealtime): nothing can hang the device.
// build: hipcc -O3 --offload-arch=gfx950 -o doorbell doorbell.hip ;  run: ./doorbell [K=2000] [WORK=0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define STOP 0xFFFFFFFFu
#define SLOTS 16
static const unsigned long long DEADLINE = 5000000ull;    // 50 ms of the 100 MHz clock

__device__ __forceinline__ unsigned ld_sc1(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_sc1(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1f(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1f(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ float work(float x, int n) {
    for (int i = 0; i < n; ++i) x = fmaf(x, 0.999f, 0.001f);
    return x;
}

// A: one launch per step
__global__ __launch_bounds__(256) void k_step(const float *__restrict__ act, float *__restrict__ out, int n, int w) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    out[t] = work(act[t] + out[t], w);
}

struct Mail {
    unsigned *door;                 // step sequence word (STOP ends the kernel)
    unsigned long long *done;       // [8] shards, one 128-byte line each (index 16 * s)
    unsigned long long *top;        // two-level arrival (mode D)
    unsigned long long *seq;        // the ring kernels' own count
    unsigned long long *stamps;     // [K][6] realtime stamps of wave 0 (and the ring)
    unsigned *status;               // 0 running, 1 stopped, 2 timed out
    const float *act;               // [SLOTS][n]
    float *out;                     // [SLOTS][n]
    unsigned cnt[8];                // workgroups per shard
    int n, w, two_level, nstamps;
};

__global__ __launch_bounds__(256, 1) void k_resident(Mail M) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const bool w0 = blockIdx.x == 0 && threadIdx.x < 64;
    float carry = 0.f;                // the "state" that stays in registers
    for (unsigned k = 1;; ++k) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned v;
        for (;;) {
            v = __builtin_amdgcn_readfirstlane(ld_sc1(M.door));
            if (v == STOP || v >= k) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > DEADLINE) { v = STOP - 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (v == STOP || v == STOP - 1) {
            if (t == 0) st_sc1(M.status, v == STOP ? 1u : 2u);
            return;
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        const int slot = k % SLOTS;
        const float a = ld_sc1f(M.act + (size_t)slot * M.n + t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
        carry = work(a + carry, M.w);
        const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
        st_sc1f(M.out + (size_t)slot * M.n + t, carry);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t4 = __builtin_amdgcn_s_memrealtime();
        __syncthreads();
        if (threadIdx.x == 0) {
            const int s = blockIdx.x & 7;
            if (M.two_level) {
                const unsigned long long old = __hip_atomic_fetch_add(M.done + 16 * s, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1 == (unsigned long long)k * M.cnt[s]) __hip_atomic_fetch_add(M.top, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                (void)__hip_atomic_fetch_add(M.done + 16 * s, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (w0 && threadIdx.x == 0 && (int)k <= M.nstamps) {
            unsigned long long *S = M.stamps + (size_t)(k - 1) * 8;
            S[1] = t1; S[2] = t2; S[3] = t3; S[4] = t4; S[5] = __builtin_amdgcn_s_memrealtime();
        }
    }
}

// B / C: the ring kernel -- `advance` steps at once
__global__ __launch_bounds__(64) void k_ring(Mail M, unsigned advance, unsigned stop) {
    const int lane = threadIdx.x;
    unsigned long long k = 0;
    if (lane == 0) {
        if (stop) { st_sc1(M.door, STOP); }
        else {
            k = *M.seq + advance;
            *M.seq = k;
            const unsigned long long tw = __builtin_amdgcn_s_memrealtime();
            st_sc1(M.door, (unsigned)k);
            if ((long long)k <= M.nstamps) M.stamps[(size_t)(k - 1) * 8 + 0] = tw;
        }
    }
    if (stop) return;
    k = __shfl(k, 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool ok = lane >= 8;
    for (;;) {
        if (!ok) ok = ld_sc1(M.done + 16 * lane) >= k * M.cnt[lane];
        if (__all(ok)) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > DEADLINE) { if (lane == 0) st_sc1(M.status, 3u); break; }
        __builtin_amdgcn_s_sleep(1);
    }
    if (lane == 0 && (long long)k <= M.nstamps) M.stamps[(size_t)(k - 1) * 8 + 6] = __builtin_amdgcn_s_memrealtime();
}

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 2000, W = argc > 2 ? atoi(argv[2]) : 0;
    const int G = 256, n = G * 256;
    hipStream_t sr, ss;
    CK(hipStreamCreateWithFlags(&sr, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&ss, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float *act, *out;
    CK(hipMalloc(&act, (size_t)SLOTS * n * 4)); CK(hipMalloc(&out, (size_t)SLOTS * n * 4));
    CK(hipMemset(act, 0, (size_t)SLOTS * n * 4)); CK(hipMemset(out, 0, (size_t)SLOTS * n * 4));
    float ms;
    // ---- A: one launch per step
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_step, dim3(G), dim3(256), 0, ss, act, out, n, W);
    CK(hipEventRecord(e0, ss));
    for (int i = 0; i < K; i++) hipLaunchKernelGGL(k_step, dim3(G), dim3(256), 0, ss, act + (size_t)(i % SLOTS) * n, out, n, W);
    CK(hipEventRecord(e1, ss)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("WORK %d  K %d\nA  one launch per step:                          %7.2f us per step\n", W, K, ms * 1e3 / K);

    Mail M = {};
    unsigned *sig_door = nullptr; unsigned long long *sig_top = nullptr;
    int can_wait = 0;
    (void)hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, 0);
    CK(hipMalloc(&M.door, 256)); CK(hipMalloc(&M.done, 8 * 128)); CK(hipMalloc(&M.top, 256)); CK(hipMalloc(&M.seq, 256));
    CK(hipMalloc(&M.status, 256)); CK(hipMalloc(&M.stamps, (size_t)K * 8 * 8));
    M.act = act; M.out = out; M.n = n; M.w = W; M.nstamps = K;
    for (int s = 0; s < 8; s++) M.cnt[s] = (G - s + 7) / 8;
    unsigned *door_plain = M.door; unsigned long long *top_plain = M.top;

    for (int mode = 0; mode < 3 + (can_wait ? 2 : 0); mode++) {
        // 0: B closed loop (ring kernel per step)   1: C run-ahead (one ring for K steps)   2: B again with two-level arrival
        // 3: D stream write / wait value on hipMalloc memory   4: D on signal memory
        if (mode == 4) {
            if (hipExtMallocWithFlags((void **)&sig_door, 8, hipMallocSignalMemory) != hipSuccess ||
                hipExtMallocWithFlags((void **)&sig_top, 8, hipMallocSignalMemory) != hipSuccess) { printf("D  signal memory: allocation refused\n"); break; }
            M.door = sig_door; M.top = sig_top;
        } else { M.door = door_plain; M.top = top_plain; }
        M.two_level = mode >= 2;
        CK(hipMemset(M.door, 0, 8)); CK(hipMemset(M.done, 0, 8 * 128)); CK(hipMemset(M.top, 0, 8)); CK(hipMemset(M.seq, 0, 8));
        CK(hipMemset(M.status, 0, 4)); CK(hipMemset(M.stamps, 0, (size_t)K * 8 * 8));
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_resident, dim3(G), dim3(256), 0, sr, M);
        CK(hipGetLastError());
        hipError_t werr = hipSuccess;
        CK(hipEventRecord(e0, ss));
        if (mode == 0 || mode == 2) for (int i = 0; i < K; i++) hipLaunchKernelGGL(k_ring, dim3(1), dim3(64), 0, ss, M, 1u, 0u);
        else if (mode == 1) hipLaunchKernelGGL(k_ring, dim3(1), dim3(64), 0, ss, M, (unsigned)K, 0u);
        else for (int i = 1; i <= K && werr == hipSuccess; i++) {
            werr = hipStreamWriteValue32(ss, M.door, (uint32_t)i, 0);
            if (werr == hipSuccess) werr = hipStreamWaitValue64(ss, M.top, 8ull * i, hipStreamWaitValueGte, ~0ull);
        }
        CK(hipEventRecord(e1, ss));
        hipLaunchKernelGGL(k_ring, dim3(1), dim3(64), 0, ss, M, 0u, 1u);       // STOP
        if (werr != hipSuccess) printf("   stream value op failed: %s\n", hipGetErrorString(werr));
        CK(hipStreamSynchronize(ss)); CK(hipStreamSynchronize(sr));
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned status = 9; CK(hipMemcpy(&status, M.status, 4, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> st((size_t)K * 8);
        CK(hipMemcpy(st.data(), M.stamps, st.size() * 8, hipMemcpyDeviceToHost));
        static const char *name[] = {"B  resident, ring kernel per step (closed loop):", "C  resident, one ring for the run (run-ahead):  ",
                                     "B' closed loop, two-level arrival:              ", "D  stream write/wait value, hipMalloc memory:   ",
                                     "D' stream write/wait value, signal memory:      "};
        printf("%s %7.2f us per step   (status %u)\n", name[mode], ms * 1e3 / K, status);
        if (mode == 0 || mode == 2 || mode == 1) {
            std::vector<double> d_seen, d_load, d_work, d_drain, d_arr, d_ring, d_period;
            for (int k = K / 2; k < K - 1; k++) {
                const unsigned long long *S = &st[(size_t)k * 8];
                if (!S[1]) continue;
                if (mode != 1) { d_seen.push_back((double)(long long)(S[1] - S[0]) * 0.01); d_ring.push_back((double)(long long)(S[6] - S[5]) * 0.01); }
                d_load.push_back((S[2] - S[1]) * 0.01); d_work.push_back((S[3] - S[2]) * 0.01); d_drain.push_back((S[4] - S[3]) * 0.01);
                d_arr.push_back((S[5] - S[4]) * 0.01);
                d_period.push_back((double)(long long)(st[(size_t)(k + 1) * 8 + 1] - S[1]) * 0.01);
            }
            if (!d_load.empty()) {
                printf("     wave 0, medians (us): ");
                if (mode != 1) printf("door written -> seen %.2f, ", med(d_seen));
                printf("action load %.2f, work %.2f, store drain %.2f, barrier + arrive %.2f", med(d_load), med(d_work), med(d_drain), med(d_arr));
                if (mode != 1) printf(", arrive -> ring sees all shards %.2f", med(d_ring));
                printf(", period %.2f\n", med(d_period));
            }
        }
    }
    return 0;
}
