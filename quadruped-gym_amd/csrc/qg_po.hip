// qg_po.hip -- stand-alone kernels of the partially observable observation pack; the per-env
// arithmetic and the row output live in qg_po_dev.h, which the fused step kernel shares.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qg_po_dev.h"

// Runs after the physics (and walking-reward) kernels of the step when the step kernel in use has no fused form of it.  A block of
// QG_PO_THREADS threads owns QG_PO_ENVS envs:
//   phase 1  one thread per env: orientation filter, the new 26-value frame and (for an env the step has just auto-reset) the
//            reset frame, into LDS (po_frame_env);
//   phase 2  all threads, 16 per env: the env's row of `out`, the new frame into the env's ring slot (po_emit_rows).
__global__ __launch_bounds__(QG_PO_THREADS) void qg_po_frame_kernel(KPoParams P, KPoState S, int n, const float *__restrict__ obs33,
                                   const float *__restrict__ eff_actions, const float *__restrict__ qpos /* [19][n] */,
                                   KWalkParams WP, KWalkState WS /* the commands live here */, const uint8_t *__restrict__ done,
                                   float *__restrict__ out /* [n][window*26] */, float *__restrict__ term_out /* [n][window*26] or NULL */,
                                   int sample_cmd, uint64_t seed, uint64_t env_index_base, const int32_t *__restrict__ episode) {
    __shared__ float s_new[QG_PO_ENVS][QG_PO_FRAME];     // the frame of this step
    __shared__ float s_rst[QG_PO_ENVS][QG_PO_FRAME];     // the frame reset() would return (only for envs that finished)
    __shared__ int s_slot[QG_PO_ENVS];                   // ring slot of the newest frame
    __shared__ int s_fin[QG_PO_ENVS];                    // the env finished and was auto-reset by the physics kernel
    const int env0 = blockIdx.x * QG_PO_ENVS;
    if (threadIdx.x < QG_PO_ENVS && env0 + threadIdx.x < n) {
        const int le = threadIdx.x, env = env0 + le;
        const PoEnvIn in = po_env_load(S, n, env);
        float *fr = s_new[le];
        for (int j = 0; j < 12; ++j) fr[11 + j] = fminf(fmaxf(eff_actions[(size_t)env * 12 + j], -1.f), 1.f);
        int slot, fin;
        po_frame_env(P, S, n, env, in, obs33 + (size_t)env * 33, qpos[3 * n + env], qpos[4 * n + env], qpos[5 * n + env], qpos[6 * n + env],
                     WS.vel[env], WS.vel[n + env], WS.head[env], WS.head[n + env], done[env] != 0, fr, s_rst[le], slot, fin);
        s_slot[le] = slot;
        s_fin[le] = fin;
        // random_controls on the device: the new episode's command, drawn only now that both frames show the old one
        if (fin && sample_cmd) walk_sample_command(WP, WS, n, env, seed, env_index_base, episode[env] - 1);
    }
    __syncthreads();
    po_emit_rows(P, S, n, env0, threadIdx.x >> 4, threadIdx.x & 15, s_new, s_rst, s_slot, s_fin, out, term_out);
}

// explicit (masked) reset: the stack is filled with the reset frame, the estimate aliases data.qpos[3:7] from now on
__global__ void qg_po_reset_kernel(KPoParams P, KPoState S, int n, const uint8_t *mask, const float *__restrict__ vel,
                                   const float *__restrict__ head, float *__restrict__ out /* nullable */) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= n) return;
    if (mask && !mask[env]) return;
    float q[4] = {S.orient[env], S.orient[n + env], S.orient[2 * n + env], S.orient[3 * n + env]};
    if (S.alias[env]) { q[0] = 1.f; q[1] = q[2] = q[3] = 0.f; }
    float roll, pitch, yaw;
    po_euler(q[0], q[1], q[2], q[3], roll, pitch, yaw);
    float rf[QG_PO_FRAME];
    for (int i = 0; i < 6; ++i) rf[i] = 0.f;
    rf[6] = roll; rf[7] = pitch; rf[8] = yaw; rf[9] = 0.f; rf[10] = 0.f;
    for (int j = 0; j < 12; ++j) rf[11 + j] = P.default_ctrl[j];
    rf[23] = vel[env]; rf[24] = vel[n + env]; rf[25] = po_atan2(head[n + env], head[env]);
    float *st = S.stack + 2 * (size_t)env * P.window * QG_PO_FRAME;       // the ring holds every frame twice (KPoState.stack)
    for (int f = 0; f < P.window; ++f)
        for (int i = 0; i < QG_PO_FRAME; ++i) {
            st[f * QG_PO_FRAME + i] = rf[i];
            st[(P.window + f) * QG_PO_FRAME + i] = rf[i];
            if (out) out[(size_t)env * P.window * QG_PO_FRAME + f * QG_PO_FRAME + i] = rf[i];
        }
    S.alias[env] = 1;
    S.nstep[env] = 0;
}
