// qg_walk.hip -- device side of the walking task layer: what
// WalkingQuadrupedEnv adds around QuadrupedEnv.step() in antopio26/quadruped-gym
// (src/envs/walking_quad.py:96-148 step/reset bookkeeping, :162-428 the reward stack,
// src/envs/math_utils.py:11-158 the online frequency/amplitude estimator of the control signal,
// src/envs/control_inputs.py the velocity/heading command).
//
// Per env-step, in the reference's order (walking_quad.py:128-148):
//   qg_walk_pre_kernel   one thread per (channel, env): the estimator takes data.ctrl (the PREVIOUS applied
//                        action, :136), every thread writes its entry of the effective action (joint centres
//                        while data.time < settling_time, :142-143);
//   qg_step_kernel*      the physics, with flip + time-limit terminations (:156-166);
//   qg_walk_post_kernel  one thread per env: the ideal position integrates the commanded global velocity (:93,133; nothing
//                        between reads it), the 11 reward terms of input_control_reward (:352-428) on the step's
//                        sensordata and data.ctrl, their sum, episode bookkeeping of envs that finished.
// These three launches serve the one-env-per-lane and two-legs-per-lane mappings; with the default one-leg-per-lane mapping the
// whole walking env-step is ONE launch: qg_step_kernel_quad<.., WALK = true> (qg_kernels.hip) runs the estimator update of
// its three channels per lane in the prologue and the reward in the epilogue, through the same device functions
// (qg_walk_dev.h).
// The estimator's amplitude is max - min over a sliding window of 2 / (min_freq * dt) samples (250 at frame_skip
// 4).  Scanning the window every step streams 12 KB per env (measured 62.7 us per launch at 4096 envs, 3x the
// physics), so the ring buffer carries per-block (16 samples) max / min summaries: a step re-reduces the one block
// that received the new sample and combines it with the other blocks' summaries -- 16 + 2*15 values instead of
// 250, bit-identical results (max / min are exact).  Buffers are laid out [slot][channel][env] so that consecutive
// threads touch consecutive addresses.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qg_walk_dev.h"     // KWalkParams / KWalkState and the per-env device functions (shared with the fused step kernel)

// one thread per (env, channel): thread t = env * 12 + channel
__global__ void qg_walk_pre_kernel(KWalkParams P, KWalkState S, int n, const float *__restrict__ actions, const float *__restrict__ data_ctrl,
                                   const int32_t *__restrict__ nstep) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 12 * n) return;
    const int env = t / 12, ch = t - env * 12;
    // effective action: the joint centres while the robot settles (walking_quad.py:142-143)
    const bool settle = nstep[env] < P.settle_substeps;
    S.eff_actions[(size_t)env * 12 + ch] = settle ? P.joint_centers[ch] : actions[(size_t)env * 12 + ch];
    const int tt[1] = {t};
    const float xx[1] = {data_ctrl[ch * n + env]};     // data.ctrl is physics state: [12][n]
    const int calls = S.calls[env];
    WalkEstIn<1> in;
    float f_new[1], a_new[1];
    walk_estimator_load_n<1>(P, S, n, tt, calls, in);                     // math_utils.py:53-131 with data.ctrl (:136)
    walk_estimator_finish_n<1>(P, S, n, tt, xx, calls, in, f_new, a_new);
}

// one thread per env.  `sample_here`: redraw the command of the envs the step has auto-reset (random_controls on the device);
// off when a partially observable pack follows, which still has to show the old command and redraws afterwards itself.
__global__ void qg_walk_post_kernel(KWalkParams P, KWalkState S, int n, const float *__restrict__ obs /* [n][33] */,
                                    const uint8_t *__restrict__ done, float *__restrict__ reward, float *__restrict__ comps /* [n][11] or NULL */,
                                    int sample_here, uint64_t seed, uint64_t env_index_base, const int32_t *__restrict__ episode) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= n) return;
    WalkSums sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        float c = S.eff_actions[(size_t)env * 12 + j];               // data.ctrl after the step
        c = fminf(fmaxf(c, -1.f), 1.f);                              // quadruped.py:160
        walk_channel_terms(S, env, j, walk_channel_targets(P, j), c, S.prev_ctrl[env * 12 + j], S.f_est[env * 12 + j], S.a_est[env * 12 + j], sum);
        S.prev_ctrl[env * 12 + j] = c;
    }
    // the physics reset has already advanced the env's episode counter: the key of the episode that begins is episode - 1
    WalkEnvIn in = walk_env_load(S, n, env);
    in.episode_key = episode[env] - 1;
    walk_reward_env(P, S, n, env, obs + (size_t)env * 33, sum, in, done[env] != 0, reward, comps, sample_here, seed, env_index_base);
}

// the same draw for the envs `select` marks (NULL = all), as its own launch: explicit resets
__global__ void qg_walk_command_kernel(KWalkParams P, KWalkState S, int n, const uint8_t *__restrict__ select, uint64_t seed,
                                       uint64_t env_index_base, const int32_t *__restrict__ episode) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= n) return;
    if (select && !select[env]) return;
    walk_sample_command(P, S, n, env, seed, env_index_base, episode[env] - 1);
}

// explicit (masked) episode reset of the walking state
__global__ void qg_walk_reset_kernel(KWalkParams P, KWalkState S, int n, const uint8_t *mask) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= n) return;
    if (mask && !mask[env]) return;
    S.ideal[env] = 0.f; S.ideal[n + env] = 0.f;
    for (int j = 0; j < 12; ++j) S.prev_ctrl[env * 12 + j] = P.joint_centers[j];
    S.has_derive[env] = 0;
}
