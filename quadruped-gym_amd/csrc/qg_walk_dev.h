// qg_walk_dev.h -- per-env device functions of the walking task layer, shared by the stand-alone task kernels (qg_walk.hip)
// and by the fused walking variant of the step kernel (qg_kernels.hip, qg_step_kernel_quad<.., WALK = true>): the control-signal
// frequency / amplitude estimator (src/envs/math_utils.py:11-158), the command sampler (src/envs/control_inputs.py:74-115) and
// the eleven reward terms of input_control_reward (src/envs/walking_quad.py:352-428).
// Included from qg_kernels.hip after the counter-based random streams (uniform24s) it uses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define QG_WALK_BLOCK 16      // samples per block summary of the estimator's ring buffer

// Square roots and reciprocals of the task layer: the hardware instructions (1 ulp), not the IEEE-exact expansions (~10 instructions
// and an SGPR pair each -- on the one lane per env that evaluates the reward while its wave is alone on the SIMD).  The quantities
// are O(1) sums of squares and durations; the parity bounds against the f64 oracle are four orders of magnitude wider.
__device__ __forceinline__ float walk_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float walk_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

struct KWalkParams {
    float dt;                    // timestep * frame_skip
    float inv_dt;                // 1 / dt, rounded from double
    int32_t settle_substeps;     // data.time < settling_time  <=>  nstep < settle_substeps (f64-accumulated clock)
    int32_t window;              // estimator window size
    float ema_alpha;
    float control_cost_alpha;
    float w[10];
    float w_diff_ideal;
    float body_height;
    float joint_centers[12];
    float amp_target[12];
    float freq_target[12];
    int32_t auto_reset;
    // on-device command sampler (control_inputs.py:74-115); cmd_sample = 0: commands only change through qg_walk_set_commands
    int32_t cmd_sample;
    uint32_t cmd_fixed;
    float cmd_min_speed, cmd_max_speed, cmd_theta, cmd_alpha, cmd_speed;
    int32_t unit_zero;           // 1: unit() of an exactly zero vector makes the direction term 0 instead of the reference's NaN
};

struct KWalkState {
    // commands (control_inputs.py): local velocity xy, heading unit vector xy, global velocity xy   [2][n] each
    float *vel, *head, *gvel;
    float *ideal;            // [2][n]   ideal position (integrated commanded global velocity)
    float *prev_ctrl;        // [n][12]  walking_quad.py:260-262
    float *prev_ctrl_cost;   // [n]      set on the first step ever, never updated (:266-270)
    uint8_t *has_ctrl_cost;  // [n]
    float *prev_derive;      // [n]      previous_rewards_to_derive (:388-396)
    uint8_t *has_derive;     // [n]      cleared by every reset (:109)
    // estimator (math_utils.py): never reset between episodes (walking_quad.py:115)
    int32_t *calls;          // [n]      update() calls so far: buffer index = calls % window, samples = min(calls, window)
    float *sig;              // [window][n][12]  (the previous sample = the slot before the write index: no copy of it is kept)
    float *bmax, *bmin;      // [blocks][n][12]  max / min of the samples written into each 16-sample block during its latest pass
    float *smax, *smin;      // [17][n][12]  extrema of everything in the window EXCEPT what has been written into the block being
                             //          written during its current pass: [j] = old samples j .. 15 of that block and every other
                             //          block (j = 1 .. 15; [16] = the other blocks alone); rebuilt when the write index enters a block
    uint8_t *cross;          // [window][n][12]
    int32_t *count;          // [n][12]  4 * (running number of derivative sign changes inside the window) + (last derivative sign + 1)
    float *f_est, *a_est;    // [n][12]
    float *eff_actions;      // [n][12]  the action actually applied (joint centres while settling)
};


// what the fused walking variant of the step kernel takes as an extra by-value kernel argument.  By value on purpose: the kernarg
// segment is constant memory to the compiler, so the two dozen state pointers stay in scalar registers; behind a device-memory
// pointer every one of them was re-loaded (vector load + full vmcnt wait) after each store that might alias it.
struct KWalkLaunch {
    KWalkParams P;
    KWalkState S;
    float *comps;            // [n][11] or NULL
    int32_t sample;          // redraw the command of the envs this step auto-resets
};
struct KWalkNone {};
template <bool WALK> struct WalkArgT { typedef KWalkNone type; };
template <> struct WalkArgT<true> { typedef KWalkLaunch type; };

// ---- estimator update of NCH channels of one env with data.ctrl (math_utils.py:53-131); t[c] = env * 12 + channel ------------
// Per-channel task state is laid out ENV-MAJOR, channel-minor ([n][12]; ring buffers [slot][n][12]): the 12 channels of an env --
// and the 48 of the four envs a wave of the one-link-per-lane kernel carries -- are contiguous.  Channel-major ([12][n], round 1)
// made that kernel touch 12 separate 64-byte lines per load, 16 bytes of each: 7.4 us of task layer per walking step.
// The amplitude is max - min over a sliding window of W samples.  max / min are exact, so however they are regrouped the result is
// bit-identical to a full scan of the ring (the oracle pinned to the reference's math_utils.py checks it over window wraps).  Three
// levels keep a call's traffic at twelve values per channel:
//   * the ring is cut into blocks of 16 samples; bmax / bmin[b] hold the extrema of the samples written into block b during its
//     latest pass (a running value while the write index is inside b, the block's summary once it has moved on);
//   * the extrema over all OTHER filled blocks only change when the write index enters a new block (every 16th call), where they
//     are rebuilt from the summaries;
//   * what is left of the block being overwritten -- the OLD samples behind the write index, the oldest of the window -- enters
//     through suffix extrema of old[j .. 15], computed once, when the index enters the block (third pass of round 2; before, every
//     call re-read the block's 16 samples: 24 values per channel and call);
//   * round 3: the two are kept TOGETHER -- smax / smin[j] = extrema of (old[j .. 15] and the other blocks), [16] = the other blocks
//     alone -- the derivative sign rides in the low bits of the sign-change count, and the previous sample is read from the ring
//     itself: 33 bytes loaded and 25 stored per channel and call instead of 45 and 33.  At large batches the walking prologue is HBM
//     traffic (~1 KB of task state per env-step against 0.6 KB for the physics: 6.6 us of a 33 us step at 32 768 envs), so bytes are time.
// Written in three phases -- every load, then the arithmetic, then every store -- so that the loads of all channels are in flight
// together: inside the fused step kernel a wave is alone on its SIMD and a chain of dependent loads costs its full latency each time
// (measured: the loop form made the fused walking step 42 us, slower than three launches).
// The NCH channels a lane owns are CONSECUTIVE (t[c] = t[0] + c: one, three or six of an env's twelve), i.e. contiguous in every
// [..][n][12] array: they move as ONE 12-byte access per three channels (global_load / store_dwordx3 needs dword alignment only)
// instead of three 4-byte ones that each use a third of every cache line they touch -- at 32 768 envs the six-channel update of the
// two-legs-per-lane kernel spent 7.1 us in its 54 scalar stores per lane and 3.4 us in its loads (tools/phase_times.py).
struct WalkF3 { float a, b, c; };
struct WalkI3 { int a, b, c; };
template <int NCH> __device__ __forceinline__ void walk_ldv(const float *p, float (&o)[NCH]) {
    if constexpr (NCH % 3 == 0) {
#pragma unroll
        for (int g = 0; g < NCH / 3; ++g) { const WalkF3 v = *reinterpret_cast<const WalkF3 *>(p + 3 * g); o[3 * g] = v.a; o[3 * g + 1] = v.b; o[3 * g + 2] = v.c; }
    } else {
#pragma unroll
        for (int c = 0; c < NCH; ++c) o[c] = p[c];
    }
}
template <int NCH> __device__ __forceinline__ void walk_ldv(const int *p, int (&o)[NCH]) {
    if constexpr (NCH % 3 == 0) {
#pragma unroll
        for (int g = 0; g < NCH / 3; ++g) { const WalkI3 v = *reinterpret_cast<const WalkI3 *>(p + 3 * g); o[3 * g] = v.a; o[3 * g + 1] = v.b; o[3 * g + 2] = v.c; }
    } else {
#pragma unroll
        for (int c = 0; c < NCH; ++c) o[c] = p[c];
    }
}
template <int NCH> __device__ __forceinline__ void walk_stv(float *p, const float (&v)[NCH]) {
    if constexpr (NCH % 3 == 0) {
#pragma unroll
        for (int g = 0; g < NCH / 3; ++g) { const WalkF3 w = {v[3 * g], v[3 * g + 1], v[3 * g + 2]}; *reinterpret_cast<WalkF3 *>(p + 3 * g) = w; }
    } else {
#pragma unroll
        for (int c = 0; c < NCH; ++c) p[c] = v[c];
    }
}
template <int NCH> __device__ __forceinline__ void walk_stv(int *p, const int (&v)[NCH]) {
    if constexpr (NCH % 3 == 0) {
#pragma unroll
        for (int g = 0; g < NCH / 3; ++g) { const WalkI3 w = {v[3 * g], v[3 * g + 1], v[3 * g + 2]}; *reinterpret_cast<WalkI3 *>(p + 3 * g) = w; }
    } else {
#pragma unroll
        for (int c = 0; c < NCH; ++c) p[c] = v[c];
    }
}
#define QG_WALK_MAXBLOCKS 16              // block summaries read by the unrolled rebuild (windows up to 256 samples); longer windows add a rolled loop
#define QG_WALK_EMPTY_MAX (-3.0e38f)      // "nothing there" (finite: the device pass is compiled with -ffinite-math-only)
#define QG_WALK_EMPTY_MIN (3.0e38f)
template <int NCH> struct WalkEstIn {
    float prev[NCH], fe[NCH], ae[NCH];
    float bm[NCH], bn[NCH];               // extrema of the samples written into the current block so far (valid when j0 > 0)
    float sh[NCH], sl[NCH];               // extrema of the rest of the window: the old samples behind the write index and every other block
    int cnt[NCH], cr[NCH];                // packed count / sign (see KWalkState.count), the crossing flag of the slot being overwritten
};
// phase 1: every load of the update: call it among the caller's other loads.  (On the call that enters a new block -- every 16th --
// it also reads the block's 16 old samples and the 2 x 16 summaries and STORES the suffix extrema: a rare, slower path.)
// LOWP: the block-entry path in rolled groups of four loads, reduced as they arrive (max / min: the same bits in any order), for a caller
// that must fit in 256 registers with nothing to park -- the helper waves of qg_step_kernel_quad<.., HELP>; the unrolled form keeps
// 2 x 16 summaries and 16 old samples per channel in flight.
template <int NCH, bool LOWP = false>
__device__ __forceinline__ void walk_estimator_load_n(const KWalkParams &P, const KWalkState &S, int n, const int (&t)[NCH], int calls, WalkEstIn<NCH> &in,
                                                      bool live = true /* false: a tail lane shadowing the last env -- loads only, no stores */) {
    const int W = P.window;
    const int idx = calls % W;
    const size_t stride = (size_t)12 * n;
    const int samples = min(calls + 1, W);              // :89-90
    const int bidx = idx / QG_WALK_BLOCK, j0 = idx - bidx * QG_WALK_BLOCK, base = bidx * QG_WALK_BLOCK;
    const int nblocks = (W + QG_WALK_BLOCK - 1) / QG_WALK_BLOCK;
    const int pidx = idx > 0 ? idx - 1 : W - 1;         // the previous sample's slot (written by the previous call)
    const int t0 = t[0];                                // t[c] = t0 + c
    walk_ldv<NCH>(S.sig + (size_t)pidx * stride + t0, in.prev);
    walk_ldv<NCH>(S.count + t0, in.cnt);
    walk_ldv<NCH>(S.f_est + t0, in.fe);
    walk_ldv<NCH>(S.a_est + t0, in.ae);
    walk_ldv<NCH>(S.bmax + (size_t)bidx * stride + t0, in.bm);
    walk_ldv<NCH>(S.bmin + (size_t)bidx * stride + t0, in.bn);
    walk_ldv<NCH>(S.smax + (size_t)(j0 + 1) * stride + t0, in.sh);       // slot j0 + 1 <= 16: what stands behind the sample this call writes
    walk_ldv<NCH>(S.smin + (size_t)(j0 + 1) * stride + t0, in.sl);
#pragma unroll
    for (int c = 0; c < NCH; ++c) in.cr[c] = (int)S.cross[(size_t)idx * stride + t0 + c];     // the slot holds 0 until the buffer wraps
    if constexpr (LOWP) {
        if (calls > 0 && j0 == 0) {                     // the write index enters block bidx
            float sh[NCH], sl[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) { sh[c] = QG_WALK_EMPTY_MAX; sl[c] = QG_WALK_EMPTY_MIN; }
            const int nb = max(nblocks, QG_WALK_MAXBLOCKS);             // summary slots that exist
#pragma unroll 1
            for (int b0 = 0; b0 < nb; b0 += 4) {                        // (a) the extrema of the other blocks
                float h[4][NCH], l[4][NCH];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int b = min(b0 + u, nb - 1);
                    walk_ldv<NCH>(S.bmax + (size_t)b * stride + t0, h[u]);
                    walk_ldv<NCH>(S.bmin + (size_t)b * stride + t0, l[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int b = b0 + u;
                    const bool use = b < nblocks && b != bidx && b * QG_WALK_BLOCK < samples;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) { sh[c] = fmaxf(sh[c], use ? h[u][c] : QG_WALK_EMPTY_MAX); sl[c] = fminf(sl[c], use ? l[u][c] : QG_WALK_EMPTY_MIN); }
                }
                asm volatile("" ::: "memory");                          // the next group's loads stay behind this group's reduction
            }
            if (live) {
                walk_stv<NCH>(S.smax + (size_t)QG_WALK_BLOCK * stride + t0, sh);
                walk_stv<NCH>(S.smin + (size_t)QG_WALK_BLOCK * stride + t0, sl);
            }
#pragma unroll 1
            for (int j0g = QG_WALK_BLOCK - 4; j0g >= 0; j0g -= 4) {     // (b) joined by the suffixes of the block's old samples, from the top
                float o[4][NCH];
#pragma unroll
                for (int u = 0; u < 4; ++u) walk_ldv<NCH>(S.sig + (size_t)(base + j0g + u) * stride + t0, o[u]);
#pragma unroll
                for (int u = 3; u >= 0; --u) {
                    const int j = j0g + u;
                    if (j >= 1) {
                        const bool valid = base + j < samples;
#pragma unroll
                        for (int c = 0; c < NCH; ++c) {
                            sh[c] = fmaxf(sh[c], valid ? o[u][c] : QG_WALK_EMPTY_MAX);
                            sl[c] = fminf(sl[c], valid ? o[u][c] : QG_WALK_EMPTY_MIN);
                        }
                        if (live) {
                            walk_stv<NCH>(S.smax + (size_t)j * stride + t0, sh);
                            walk_stv<NCH>(S.smin + (size_t)j * stride + t0, sl);
                        }
                    }
                }
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) { in.sh[c] = sh[c]; in.sl[c] = sl[c]; }     // j = 1: what stands behind the sample this call writes
        }
        return;
    }
    if (calls > 0 && j0 == 0) {                         // the write index enters block bidx
        // (a) the extrema of the other blocks, from the summaries
        float hi[NCH][QG_WALK_MAXBLOCKS], lo[NCH][QG_WALK_MAXBLOCKS];
#pragma unroll
        for (int b = 0; b < QG_WALK_MAXBLOCKS; ++b) {   // every summary slot exists (at least 16 blocks are allocated whatever the window): plain loads ...
            float h[NCH], l[NCH];
            walk_ldv<NCH>(S.bmax + (size_t)b * stride + t0, h);
            walk_ldv<NCH>(S.bmin + (size_t)b * stride + t0, l);
#pragma unroll
            for (int c = 0; c < NCH; ++c) { hi[c][b] = h[c]; lo[c][b] = l[c]; }
        }
        // windows of more than 256 samples (frame_skip 1 - 3 at the reference's 2 ms timestep: 1000 / 500 / 334): the summaries
        // past the sixteenth in a rolled loop -- a chain of loads, but only on every sixteenth call and only for such windows
        float om[NCH], on[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) { om[c] = QG_WALK_EMPTY_MAX; on[c] = QG_WALK_EMPTY_MIN; }
        for (int b = QG_WALK_MAXBLOCKS; b < nblocks; ++b) {
            float h[NCH], l[NCH];
            walk_ldv<NCH>(S.bmax + (size_t)b * stride + t0, h);
            walk_ldv<NCH>(S.bmin + (size_t)b * stride + t0, l);
            const bool use = b != bidx && b * QG_WALK_BLOCK < samples;
#pragma unroll
            for (int c = 0; c < NCH; ++c) { om[c] = fmaxf(om[c], use ? h[c] : QG_WALK_EMPTY_MAX); on[c] = fminf(on[c], use ? l[c] : QG_WALK_EMPTY_MIN); }
        }
        // (b) the block's old samples: the ring is allocated in whole blocks, so the 16 loads are unconditional
        float old[NCH][QG_WALK_BLOCK];
#pragma unroll
        for (int j = 0; j < QG_WALK_BLOCK; ++j) {
            float o[NCH];
            walk_ldv<NCH>(S.sig + (size_t)(base + j) * stride + t0, o);
#pragma unroll
            for (int c = 0; c < NCH; ++c) old[c][j] = o[c];
        }
#pragma unroll
        for (int b = 0; b < QG_WALK_MAXBLOCKS; ++b) {   // ... then selects
            const bool use = b < nblocks && b != bidx && b * QG_WALK_BLOCK < samples;      // blocks that hold at least one filled slot
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                om[c] = fmaxf(om[c], use ? hi[c][b] : QG_WALK_EMPTY_MAX);
                on[c] = fminf(on[c], use ? lo[c][b] : QG_WALK_EMPTY_MIN);
            }
        }
        // the other blocks alone (slot 16), then joined by the suffixes of the old samples, filled slots only (slot < samples: none
        // before the buffer has wrapped, and the last block of a window that is not a multiple of 16 ends early); slot j is stored
        // for the call that writes sample j - 1
        {
            float sh[NCH], sl[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) { sh[c] = om[c]; sl[c] = on[c]; }
            if (live) {
                walk_stv<NCH>(S.smax + (size_t)QG_WALK_BLOCK * stride + t0, sh);
                walk_stv<NCH>(S.smin + (size_t)QG_WALK_BLOCK * stride + t0, sl);
            }
#pragma unroll
            for (int j = QG_WALK_BLOCK - 1; j >= 1; --j) {
                const bool valid = base + j < samples;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    sh[c] = fmaxf(sh[c], valid ? old[c][j] : QG_WALK_EMPTY_MAX);
                    sl[c] = fminf(sl[c], valid ? old[c][j] : QG_WALK_EMPTY_MIN);
                }
                if (live) {
                    walk_stv<NCH>(S.smax + (size_t)j * stride + t0, sh);
                    walk_stv<NCH>(S.smin + (size_t)j * stride + t0, sl);
                }
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) { in.sh[c] = sh[c]; in.sl[c] = sl[c]; }     // j = 1: what stands behind the sample this call writes
        }
    }
}
// phases 2 and 3: the arithmetic and every store; returns the new estimates (what the reward of this step reads)
template <int NCH>
__device__ __forceinline__ void walk_estimator_finish_n(const KWalkParams &P, const KWalkState &S, int n, const int (&t)[NCH], const float (&x)[NCH],
                                                        int calls, const WalkEstIn<NCH> &in, float (&f_new)[NCH], float (&a_new)[NCH]) {
    const int W = P.window;
    const int idx = calls % W;
    const size_t stride = (size_t)12 * n;
    if (calls == 0) {                                   // first call: remember the sample, estimates stay 0 (:66-72)
        // (the estimates are handed back BEFORE the stores below: with an assignment to the caller's arrays as the last statement of
        // this branch and a global store as the last one of the other, the optimiser sinks both into one store through a pointer phi
        // and the caller's array ends up in scratch memory)
#pragma unroll
        for (int c = 0; c < NCH; ++c) { f_new[c] = in.fe[c]; a_new[c] = in.ae[c]; }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            S.sig[(size_t)idx * stride + t[c]] = x[c];
            S.count[t[c]] = 1;                          // no sign change yet, derivative sign 0
            S.bmax[t[c]] = x[c];                        // block 0 holds exactly this sample
            S.bmin[t[c]] = x[c];
#pragma unroll
            for (int j = 1; j <= QG_WALK_BLOCK; ++j) {  // nothing stands behind the write index of block 0 yet, and there is no other block
                S.smax[(size_t)j * stride + t[c]] = QG_WALK_EMPTY_MAX;
                S.smin[(size_t)j * stride + t[c]] = QG_WALK_EMPTY_MIN;
            }
        }
        return;
    }
    const int samples = min(calls + 1, W);              // :89-90
    const int bidx = idx / QG_WALK_BLOCK, j0 = idx - bidx * QG_WALK_BLOCK;
    const bool enter = j0 == 0;
    const float rdur = walk_rcp((float)samples * P.dt); // :109
    const int t0 = t[0];                                // t[c] = t0 + c
    float mxs[NCH], mns[NCH], xs[NCH];
    int counts[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const float psign = (float)((in.cnt[c] & 3) - 1);                       // the previous derivative sign, packed with the count
        float d = x[c] - in.prev[c];
        float cur = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
        const bool have_sign = calls >= 2;              // a previous derivative sign exists (:78-86)
        cur = (have_sign && cur == 0.f) ? psign : cur;
        const int crossing = (have_sign && cur != psign) ? 1 : 0;
        const int count = (in.cnt[c] >> 2) - in.cr[c] + crossing;               // :94-96
        const float f_cur = (0.5f * (float)count) * rdur;                       // :113-114
        // the samples written into this block so far, the new one included ...
        const float mx = enter ? x[c] : fmaxf(in.bm[c], x[c]);
        const float mn = enter ? x[c] : fminf(in.bn[c], x[c]);
        // ... and the rest of the window: the old samples still standing behind it and every other block
        const float amp = fmaxf(mx, in.sh[c]) - fminf(mn, in.sl[c]);           // :121-126
        f_new[c] = P.ema_alpha * in.fe[c] + (1.f - P.ema_alpha) * f_cur;        // :117
        a_new[c] = P.ema_alpha * in.ae[c] + (1.f - P.ema_alpha) * amp;          // :129
        S.cross[(size_t)idx * stride + t0 + c] = (uint8_t)crossing;
        counts[c] = 4 * count + ((int)cur + 1); mxs[c] = mx; mns[c] = mn; xs[c] = x[c];
    }
    walk_stv<NCH>(S.count + t0, counts);
    walk_stv<NCH>(S.sig + (size_t)idx * stride + t0, xs);                       // :99 (and the next call's previous sample, :105-106)
    walk_stv<NCH>(S.f_est + t0, f_new);
    walk_stv<NCH>(S.bmax + (size_t)bidx * stride + t0, mxs);                    // the block's summary once the write index has moved on
    walk_stv<NCH>(S.bmin + (size_t)bidx * stride + t0, mns);
    walk_stv<NCH>(S.a_est + t0, a_new);
}

// ---- per-channel contributions to the reward sums; also moves previous_ctrl on (walking_quad.py:249-285) ------------------
struct WalkSums { float cost, posture, amp, frq; };
// the three per-channel targets.  A channel index that differs from lane to lane makes these vector loads from the kernel-argument
// segment: fetch them up front (in the fused kernels: in the prologue) -- read where they are used, in the epilogue of a wave that
// is alone on its SIMD, each sat behind every store issued before it (vmcnt counts in order): three exposed round trips.
struct WalkChanTargets { float center, amp, frq; };
__device__ __forceinline__ WalkChanTargets walk_channel_targets(const KWalkParams &P, int j) {
    WalkChanTargets t = {P.joint_centers[j], P.amp_target[j], P.freq_target[j]};
    return t;
}
__device__ __forceinline__ void walk_channel_terms(const KWalkState &S, int env, int j, const WalkChanTargets &T, float c /* data.ctrl, clipped */,
                                                   float prev_ctrl, float f_est, float a_est, WalkSums &a) {
    const float inv_nu = 1.f / 12.f;
    float dc = c - prev_ctrl;                                        // control_cost (:254-270); the CALLER moves previous_ctrl on (walk_stv)
    a.cost = fmaf(dc, dc, a.cost);
    float pj = (c - T.center) * inv_nu;                              // :249-253
    a.posture = fmaf(pj, pj, a.posture);
    float aj = (a_est - T.amp) * inv_nu;                             // :279-285
    a.amp = fmaf(aj, aj, a.amp);
    float fj = (f_est - T.frq) * inv_nu;                             // :272-277
    a.frq = fmaf(fj, fj, a.frq);
}

// VelocityHeadingControls.sample (control_inputs.py:74-115) for one env: the command of the episode that has just begun.
// `episode_key` is the counter of that episode in the env's reset streams -- the one its reset yaw used.
__device__ __forceinline__ void walk_sample_command(const KWalkParams &P, const KWalkState &S, int n, int env, uint64_t seed,
                                                    uint64_t env_index_base, int episode_key) {
    const uint64_t g = env_index_base + (uint64_t)env, c = (uint64_t)episode_key;
    const float pi = 3.14159265358979323846f;
    float theta = P.cmd_theta, alpha = P.cmd_alpha, speed = P.cmd_speed;
    if (!(P.cmd_fixed & 1u)) theta = pi * (2.f * uniform24s(seed, g, c, QG_STREAM_COMMAND + 0u) - 1.f);      // :97-100
    if (!(P.cmd_fixed & 2u)) alpha = pi * (2.f * uniform24s(seed, g, c, QG_STREAM_COMMAND + 1u) - 1.f);      // :106-109
    if (!(P.cmd_fixed & 4u)) speed = fmaf(P.cmd_max_speed - P.cmd_min_speed, uniform24s(seed, g, c, QG_STREAM_COMMAND + 2u), P.cmd_min_speed);   // :112-115
    float st, ct, sa, ca;
    sincosf(theta, &st, &ct);
    sincosf(alpha, &sa, &ca);
    const float vx = speed * ca, vy = speed * sa;          // set_velocity_speed_alpha (:45-51)
    S.vel[env] = vx; S.vel[n + env] = vy;
    S.head[env] = ct; S.head[n + env] = st;                // set_orientation (:29-36)
    S.gvel[env] = ct * vx - st * vy;                       // :14-27
    S.gvel[n + env] = st * vx + ct * vy;
}

// what the per-env part of the reward reads from the task state; loaded up front (in the fused kernel: in the prologue, so that
// the latency hides behind the physics instead of sitting exposed in the epilogue of a wave that is alone on its SIMD)
struct WalkEnvIn {
    float cvx, cvy, hx, hy, gvx, gvy, ideal_x, ideal_y, first_cost, prev_derive;
    int has_cost, has_derive;
    int calls;               // update() calls of the estimator before this step
    int episode_key;         // counter of the episode that begins if this one ends (the caller fills it in: it lives in the simulator's state)
};
__device__ __forceinline__ WalkEnvIn walk_env_load(const KWalkState &S, int n, int env) {
    WalkEnvIn in;
    in.calls = S.calls[env];
    in.episode_key = 0;
    in.cvx = S.vel[env]; in.cvy = S.vel[n + env];
    in.hx = S.head[env]; in.hy = S.head[n + env];
    in.gvx = S.gvel[env]; in.gvy = S.gvel[n + env];
    in.ideal_x = S.ideal[env]; in.ideal_y = S.ideal[n + env];
    in.first_cost = S.prev_ctrl_cost[env];
    in.prev_derive = S.prev_derive[env];
    in.has_cost = S.has_ctrl_cost[env];
    in.has_derive = S.has_derive[env];
    return in;
}

// ---- the per-env remainder of input_control_reward (walking_quad.py:352-428) once the four sums over the 12 channels are
// known: the eleven weighted terms on the step's sensordata `s` (33 values, any address space), their total, the episode
// bookkeeping of an env the physics has just auto-reset.  `in` = the task state as it stood BEFORE this step (walk_env_load);
// the ideal position is integrated here (walking_quad.py:93,133: before the reward looks at it).
__device__ __forceinline__ void walk_reward_env(const KWalkParams &P, const KWalkState &S, int n, int env, const float *s, const WalkSums &sum,
                                                const WalkEnvIn &in, bool finished, float *__restrict__ reward,
                                                float *__restrict__ comps /* [n][11] or NULL */, int sample_here, uint64_t seed,
                                                uint64_t env_index_base) {
    // No multiply-add contraction in here: every product and sum below rounds on its own, as in the reference's NumPy expressions --
    // and, what matters on the device, the same way in EVERY kernel this function is inlined into.  Left to the backend, a*b + c*d
    // became fma(a, b, c*d) in one kernel and fma(c, d, a*b) in another (the helper wave of the walking launch against the physics
    // wave of the fused observation-pack launch: rewards one ulp of their largest term apart, round 3).  One lane per env runs this.
#pragma clang fp contract(off)
    const float px = s[18], py = s[19], pz = s[20];                  // body_pos
    const float xax = s[24], xay = s[25];                            // body_xaxis
    const float zaz = s[29];                                         // body_zaxis z
    const float vx = s[30], vy = s[31];                              // body_vel (velocimeter, local)
    const float cvx = in.cvx, cvy = in.cvy;
    const float hx = in.hx, hy = in.hy;
    const float ideal_x = fmaf(in.gvx, P.dt, in.ideal_x), ideal_y = fmaf(in.gvy, P.dt, in.ideal_y);    // :93,133
    // control_cost (:254-270): EMA against the FIRST cost ever seen, which is never updated
    float first_cost = in.first_cost;
    if (!in.has_cost) {
        first_cost = sum.cost;
        S.prev_ctrl_cost[env] = sum.cost;
        S.has_ctrl_cost[env] = 1;
    }
    const float control_cost = P.control_cost_alpha * first_cost + (1.f - P.control_cost_alpha) * sum.cost;
    // progress terms on the local (velocimeter) velocity (:197-218).  unit() of a zero vector is NaN in the reference
    // (math_utils.py:7-8) and that NaN reaches the direction term and the total.  The device pass is compiled with
    // -ffinite-math-only, under which 0/0 is formally undefined, so the documented NaN is produced explicitly: the
    // division is guarded and the quiet-NaN bit pattern is stored through integer selects below.
    const float nv = walk_sqrt(vx * vx + vy * vy), nc = walk_sqrt(cvx * cvx + cvy * cvy);
    const bool zero_norm = (nv == 0.f) || (nc == 0.f);
    const bool degenerate = zero_norm && !P.unit_zero;              // qg_walk_params.unit_zero = 1: the direction term is 0 there instead
    const float dv = zero_norm ? 1.f : nv, dc = zero_norm ? 1.f : nc;
    const float direction = (vx * cvx + vy * cvy) * walk_rcp(dv * dc);   // = unit(v) . unit(c): one reciprocal instead of four divisions
                                                                         // (a zero norm means a zero numerator: 0 * 1 = 0)
    const float dsp = nv - nc;
    const float speed_cost = dsp * dsp;
    const float heading = xax * hx + xay * hy;                       // :231-235
    const float height = fabsf(pz - P.body_height);                  // :243-247
    float v[11];
    v[0] = P.w[0];
    v[1] = P.w[1] * control_cost;
    v[2] = P.w[2] * direction;
    v[3] = P.w[3] * speed_cost;
    v[4] = P.w[4] * (__expf(heading) - 1.f);                          // exp_dist, math_utils.py:4-5
    v[5] = P.w[5] * (__expf(zaz) - 1.f);
    v[6] = P.w[6] * (__expf(height) - 1.f);
    v[7] = P.w[7] * walk_sqrt(sum.posture);
    v[8] = P.w[8] * walk_sqrt(sum.amp);
    v[9] = P.w[9] * walk_sqrt(sum.frq);
    // derived term (:383-396): d/dt of -20 * |pos_xy - ideal_xy|, zero on the first step after a reset
    const float ex = px - ideal_x, ey = py - ideal_y;
    const float derive = P.w_diff_ideal * walk_sqrt(ex * ex + ey * ey);
    const float prev = in.has_derive ? in.prev_derive : derive;
    v[10] = (derive - prev) * P.inv_dt;
    S.prev_derive[env] = derive;
    S.has_derive[env] = 1;
    float total = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) total += v[k];                       // :422 sum(values), in order
    const unsigned qnan = 0x7FC00000u;
    reinterpret_cast<unsigned *>(reward)[env] = degenerate ? qnan : __builtin_bit_cast(unsigned, total);
    if (comps) {
        unsigned *cu = reinterpret_cast<unsigned *>(comps) + (size_t)env * 11;
#pragma unroll
        for (int k = 0; k < 11; ++k) cu[k] = (k == 2 && degenerate) ? qnan : __builtin_bit_cast(unsigned, v[k]);
    }
    S.calls[env] = in.calls + 1;                                      // the estimator update of this step is complete
    // episode bookkeeping of envs the physics has just auto-reset (walking_quad.py:96-126)
    const bool restart = P.auto_reset && finished;
    S.ideal[env] = restart ? 0.f : ideal_x;
    S.ideal[n + env] = restart ? 0.f : ideal_y;
    if (restart) {
#pragma unroll
        for (int j = 0; j < 12; ++j) S.prev_ctrl[env * 12 + j] = P.joint_centers[j];
        S.has_derive[env] = 0;
        if (sample_here) walk_sample_command(P, S, n, env, seed, env_index_base, in.episode_key);  // walking_quad.py:121-122
    }
}
