// qg_walk.hip -- device side of the walking task layer: what
// WalkingQuadrupedEnv adds around QuadrupedEnv.step() in antopio26/quadruped-gym
// (src/envs/walking_quad.py:96-148 step/reset bookkeeping, :162-428 the reward stack,
// src/envs/math_utils.py:11-158 the online frequency/amplitude estimator of the control signal,
// src/envs/control_inputs.py the velocity/heading command).
//
// Per env-step, in the reference's order (walking_quad.py:128-148):
//   qg_walk_pre_kernel   one thread per (channel, env): the estimator takes data.ctrl (the PREVIOUS applied
//                        action, :136), channel 0 integrates the commanded global velocity into the ideal
//                        position (:93,133), every thread writes its entry of the effective action (joint centres
//                        while data.time < settling_time, :142-143);
//   qg_step_kernel*      the physics, with flip + time-limit terminations (:156-166);
//   qg_walk_post_kernel  one thread per env: the 11 reward terms of input_control_reward (:352-428) on the step's
//                        sensordata and data.ctrl, their sum, episode bookkeeping of envs that finished.
// The estimator's amplitude is max - min over a sliding window of 2 / (min_freq * dt) samples (250 at frame_skip
// 4).  Scanning the window every step streams 12 KB per env (measured 62.7 us per launch at 4096 envs, 3x the
// physics), so the ring buffer carries per-block (16 samples) max / min summaries: a step re-reduces the one block
// that received the new sample and combines it with the other blocks' summaries -- 16 + 2*15 values instead of
// 250, bit-identical results (max / min are exact).  Buffers are laid out [slot][channel][env] so that consecutive
// threads touch consecutive addresses.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define QG_WALK_BLOCK 16      // samples per block summary of the estimator's ring buffer

struct KWalkParams {
    float dt;                    // timestep * frame_skip
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
};

struct KWalkState {
    // commands (control_inputs.py): local velocity xy, heading unit vector xy, global velocity xy   [2][n] each
    float *vel, *head, *gvel;
    float *ideal;            // [2][n]   ideal position (integrated commanded global velocity)
    float *prev_ctrl;        // [12][n]  walking_quad.py:260-262
    float *prev_ctrl_cost;   // [n]      set on the first step ever, never updated (:266-270)
    uint8_t *has_ctrl_cost;  // [n]
    float *prev_derive;      // [n]      previous_rewards_to_derive (:388-396)
    uint8_t *has_derive;     // [n]      cleared by every reset (:109)
    // estimator (math_utils.py): never reset between episodes (walking_quad.py:115)
    int32_t *calls;          // [n]      update() calls so far: buffer index = calls % window, samples = min(calls, window)
    float *sig;              // [window][12][n]
    float *bmax, *bmin;      // [blocks][12][n]  max / min of each 16-sample block of the ring buffer
    uint8_t *cross;          // [window][12][n]
    int32_t *count;          // [12][n]  running number of derivative sign changes inside the window
    float *prev;             // [12][n]
    float *sign;             // [12][n]  -1 / 0 / +1
    float *f_est, *a_est;    // [12][n]
    float *eff_actions;      // [n][12]  the action actually applied (joint centres while settling)
};

// one thread per (channel, env): thread t = channel * n + env
__global__ void qg_walk_pre_kernel(KWalkParams P, KWalkState S, int n, const float *__restrict__ actions, const float *__restrict__ data_ctrl,
                                   const int32_t *__restrict__ nstep) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 12 * n) return;
    const int ch = t / n, env = t - ch * n;
    // effective action: the joint centres while the robot settles (walking_quad.py:142-143)
    const bool settle = nstep[env] < P.settle_substeps;
    S.eff_actions[(size_t)env * 12 + ch] = settle ? P.joint_centers[ch] : actions[(size_t)env * 12 + ch];
    if (ch < 2) S.ideal[t] = fmaf(S.gvel[t], P.dt, S.ideal[t]);          // :93,133  (ch 0 -> x, ch 1 -> y)

    // ---- estimator update with data.ctrl (math_utils.py:53-131) ----------------------------------------
    const float x = data_ctrl[t];
    const int calls = S.calls[env];
    const int W = P.window;
    const int idx = calls % W;
    const size_t slot = (size_t)idx * 12 * n + t;
    if (calls == 0) {                                   // first call: remember the sample, estimates stay 0 (:66-72)
        S.prev[t] = x;
        S.sig[slot] = x;
        S.bmax[t] = x;                                  // block 0 holds exactly this sample
        S.bmin[t] = x;
        return;
    }
    float d = x - S.prev[t];
    float cur = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
    int crossing = 0;
    if (calls >= 2) {                                   // a previous derivative sign exists (:78-86)
        float ps = S.sign[t];
        if (cur == 0.f) cur = ps;
        crossing = (cur != ps) ? 1 : 0;
    }
    const int samples = min(calls + 1, W);              // :89-90
    int cnt = S.count[t] - (int)S.cross[slot] + crossing;          // :94-96 (the slot holds 0 until the buffer wraps)
    S.cross[slot] = (uint8_t)crossing;
    S.count[t] = cnt;
    S.sig[slot] = x;                                    // :99
    S.prev[t] = x;                                      // :105-106
    S.sign[t] = cur;
    const float dur = (float)samples * P.dt;            // :109
    const float f_cur = (0.5f * (float)cnt) / dur;      // :113-114
    S.f_est[t] = P.ema_alpha * S.f_est[t] + (1.f - P.ema_alpha) * f_cur;           // :117
    // amplitude = max - min over the filled part of the window (:121-126), through the block summaries
    const size_t stride = (size_t)12 * n;
    const int bidx = idx / QG_WALK_BLOCK;
    const int nblocks = (W + QG_WALK_BLOCK - 1) / QG_WALK_BLOCK;
    float mx = x, mn = x;
    {
        const float *col = S.sig + t + (size_t)bidx * QG_WALK_BLOCK * stride;
        const int base = bidx * QG_WALK_BLOCK;
#pragma unroll
        for (int j = 0; j < QG_WALK_BLOCK; ++j) {
            const int slot_j = base + j;
            if (slot_j < samples && slot_j != idx) {        // filled slots only (samples == W once the buffer has wrapped)
                float v = col[(size_t)j * stride];
                mx = fmaxf(mx, v);
                mn = fminf(mn, v);
            }
        }
        S.bmax[(size_t)bidx * stride + t] = mx;
        S.bmin[(size_t)bidx * stride + t] = mn;
    }
    for (int b = 0; b < nblocks; ++b) {
        if (b != bidx && b * QG_WALK_BLOCK < samples) {      // blocks that hold at least one filled slot
            mx = fmaxf(mx, S.bmax[(size_t)b * stride + t]);
            mn = fminf(mn, S.bmin[(size_t)b * stride + t]);
        }
    }
    S.a_est[t] = P.ema_alpha * S.a_est[t] + (1.f - P.ema_alpha) * (mx - mn);       // :129
}

// VelocityHeadingControls.sample (control_inputs.py:74-115) for one env: the command of the episode that has just begun.
// The physics reset has already advanced the env's episode counter, so the key of this episode -- the one its reset yaw
// used -- is episode - 1.
__device__ __forceinline__ void walk_sample_command(const KWalkParams &P, const KWalkState &S, int n, int env, uint64_t seed,
                                                    uint64_t env_index_base, int episode_now) {
    const uint64_t g = env_index_base + (uint64_t)env, c = (uint64_t)(episode_now - 1);
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

// one thread per env.  `sample_here`: redraw the command of the envs the step has auto-reset (random_controls on the device);
// off when a partially observable pack follows, which still has to show the old command and redraws afterwards itself.
__global__ void qg_walk_post_kernel(KWalkParams P, KWalkState S, int n, const float *__restrict__ obs /* [n][33] */,
                                    const uint8_t *__restrict__ done, float *__restrict__ reward, float *__restrict__ comps /* [n][11] or NULL */,
                                    int sample_here, uint64_t seed, uint64_t env_index_base, const int32_t *__restrict__ episode) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= n) return;
    const float *s = obs + (size_t)env * 33;
    const float px = s[18], py = s[19], pz = s[20];                  // body_pos
    const float xax = s[24], xay = s[25];                            // body_xaxis
    const float zaz = s[29];                                         // body_zaxis z
    const float vx = s[30], vy = s[31];                              // body_vel (velocimeter, local)
    const float cvx = S.vel[env], cvy = S.vel[n + env];
    const float hx = S.head[env], hy = S.head[n + env];

    // control_cost (walking_quad.py:254-270): EMA against the FIRST cost ever seen, which is never updated
    float cost = 0.f, posture = 0.f, amp = 0.f, frq = 0.f;
    const float inv_nu = 1.f / 12.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        float c = S.eff_actions[(size_t)env * 12 + j];               // data.ctrl after the step (already clipped to +-1 by the caller's contract)
        c = fminf(fmaxf(c, -1.f), 1.f);                              // quadruped.py:160
        float dc = c - S.prev_ctrl[j * n + env];
        S.prev_ctrl[j * n + env] = c;
        cost = fmaf(dc, dc, cost);
        float pj = (c - P.joint_centers[j]) * inv_nu;                // :249-253
        posture = fmaf(pj, pj, posture);
        float aj = (S.a_est[j * n + env] - P.amp_target[j]) * inv_nu;   // :279-285
        amp = fmaf(aj, aj, amp);
        float fj = (S.f_est[j * n + env] - P.freq_target[j]) * inv_nu;  // :272-277
        frq = fmaf(fj, fj, frq);
    }
    float first_cost = S.prev_ctrl_cost[env];
    if (!S.has_ctrl_cost[env]) {
        first_cost = cost;
        S.prev_ctrl_cost[env] = cost;
        S.has_ctrl_cost[env] = 1;
    }
    const float control_cost = P.control_cost_alpha * first_cost + (1.f - P.control_cost_alpha) * cost;
    // progress terms on the local (velocimeter) velocity (:197-218).  unit() of a zero vector is NaN in the reference
    // (math_utils.py:7-8) and that NaN reaches the direction term and the total.  The device pass is compiled with
    // -ffinite-math-only, under which 0/0 is formally undefined, so the documented NaN is produced explicitly: the
    // division is guarded and the quiet-NaN bit pattern is stored through integer selects below.
    const float nv = __builtin_sqrtf(vx * vx + vy * vy), nc = __builtin_sqrtf(cvx * cvx + cvy * cvy);
    const bool degenerate = (nv == 0.f) || (nc == 0.f);
    const float dv = degenerate ? 1.f : nv, dc = degenerate ? 1.f : nc;
    const float direction = (vx / dv) * (cvx / dc) + (vy / dv) * (cvy / dc);
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
    v[7] = P.w[7] * __builtin_sqrtf(posture);
    v[8] = P.w[8] * __builtin_sqrtf(amp);
    v[9] = P.w[9] * __builtin_sqrtf(frq);
    // derived term (:383-396): d/dt of -20 * |pos_xy - ideal_xy|, zero on the first step after a reset
    const float ex = px - S.ideal[env], ey = py - S.ideal[n + env];
    const float derive = P.w_diff_ideal * __builtin_sqrtf(ex * ex + ey * ey);
    const float prev = S.has_derive[env] ? S.prev_derive[env] : derive;
    v[10] = (derive - prev) / P.dt;
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
    S.calls[env] += 1;                                                // the estimator update of this step is complete
    // episode bookkeeping of envs the physics kernel has just auto-reset (walking_quad.py:96-126)
    if (P.auto_reset && done[env]) {
        S.ideal[env] = 0.f; S.ideal[n + env] = 0.f;
#pragma unroll
        for (int j = 0; j < 12; ++j) S.prev_ctrl[j * n + env] = P.joint_centers[j];
        S.has_derive[env] = 0;
        if (sample_here) walk_sample_command(P, S, n, env, seed, env_index_base, episode[env]);     // walking_quad.py:121-122
    }
}

// the same draw for the envs `select` marks (NULL = all), as its own launch: explicit resets
__global__ void qg_walk_command_kernel(KWalkParams P, KWalkState S, int n, const uint8_t *__restrict__ select, uint64_t seed,
                                       uint64_t env_index_base, const int32_t *__restrict__ episode) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= n) return;
    if (select && !select[env]) return;
    walk_sample_command(P, S, n, env, seed, env_index_base, episode[env]);
}

// explicit (masked) episode reset of the walking state
__global__ void qg_walk_reset_kernel(KWalkParams P, KWalkState S, int n, const uint8_t *mask) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= n) return;
    if (mask && !mask[env]) return;
    S.ideal[env] = 0.f; S.ideal[n + env] = 0.f;
    for (int j = 0; j < 12; ++j) S.prev_ctrl[j * n + env] = P.joint_centers[j];
    S.has_derive[env] = 0;
}
