// qg_po.hip -- device side of the partially observable observation pack:
// POWalkingQuadrupedEnv of antopio26/quadruped-gym (src/envs/po_walking_quad.py:10-90).
// Per env and step one 26-value frame [gyro 3, accel 3, Madgwick-IMU Euler angles 3, body_vel xy 2, data.ctrl 12,
// command vx vy, heading angle] (:48-56), stacked over `obs_window` steps as a FIFO (:65,80-88).
// The orientation filter is ahrs.filters.Madgwick (third party, not available offline): restated from the
// published IMU form of the algorithm (eqs. 12, 13, 25, 26, 33, 34; gain 0.033) -- parity unpinned.
// Reference quirks kept: the filter only runs while data.time > settling_time / 2 (:37); after a reset the
// estimate IS the live data.qpos[3:7] (a NumPy view, :67) until the first filter update replaces it; the frame
// reset() returns shows zero sensors, the PREVIOUS estimate and the PREVIOUS command (:59-69).
#include <hip/hip_runtime.h>
#include <stdint.h>

#define QG_PO_FRAME 26

struct KPoParams {
    float dt;                    // timestep * frame_skip (Madgwick Dt, :18)
    float gain;                  // 0.033
    int32_t half_settle_substeps;// data.time > settling_time / 2  <=>  substeps since reset >= this (f64 clock)
    int32_t window;
    int32_t frame_skip;
    int32_t auto_reset;
    float default_ctrl[12];
};

struct KPoState {
    float *orient;       // [4][n]  computed_orientation
    uint8_t *alias;      // [n]     the estimate is the live data.qpos[3:7]
    int32_t *nstep;      // [n]     substeps since the last reset (data.time of the step being observed)
    float *stack;        // [n][window][26]  ring of the last `window` frames
    int32_t *head;       // [n]              ring slot of the newest frame
};

__device__ __forceinline__ void po_euler(float w, float x, float y, float z, float &roll, float &pitch, float &yaw) {
    float inv = 1.f / __builtin_sqrtf(w * w + x * x + y * y + z * z);
    w *= inv; x *= inv; y *= inv; z *= inv;
    roll = atan2f(2.f * (w * x + y * z), 1.f - 2.f * (x * x + y * y));
    pitch = asinf(fminf(fmaxf(2.f * (w * y - z * x), -1.f), 1.f));
    yaw = atan2f(2.f * (w * z + x * y), 1.f - 2.f * (y * y + z * z));
}

// Runs after the physics (and walking-reward) kernels of the step.  A block of QG_PO_THREADS threads owns QG_PO_ENVS envs:
//   phase 1  one thread per env: orientation filter, the new 26-value frame and (for an env the step has just auto-reset) the
//            reset frame, into LDS; the new frame also goes into the env's ring slot;
//   phase 2  all threads, 16 per env: the env's row of `out` (window x 26 floats) is written in 64-byte segments,
//            frame f coming from ring slot (head + 1 + f) mod window -- the FIFO of po_walking_quad.py:80-83
//            without moving 9 of 10 frames every step (a per-env shift of the stack measured 78 us per launch at 4096 envs
//            and window 10, four times the physics).
#define QG_PO_ENVS 16
#define QG_PO_THREADS 256
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
    const int W = P.window;
    if (threadIdx.x < QG_PO_ENVS && env0 + threadIdx.x < n) {
        const int le = threadIdx.x, env = env0 + le;
        const float *s = obs33 + (size_t)env * 33;
        const float gx = s[15], gy = s[16], gz = s[17], ax = s[12], ay = s[13], az = s[14];
        const int nstep = S.nstep[env] + P.frame_skip;
        float qw, qx, qy, qz;
        const bool alias = S.alias[env] != 0;
        if (alias) { qw = qpos[3 * n + env]; qx = qpos[4 * n + env]; qy = qpos[5 * n + env]; qz = qpos[6 * n + env]; }
        else { qw = S.orient[env]; qx = S.orient[n + env]; qy = S.orient[2 * n + env]; qz = S.orient[3 * n + env]; }
        bool still_alias = alias;
        if (nstep >= P.half_settle_substeps) {                             // po_walking_quad.py:37
            const float gn2 = gx * gx + gy * gy + gz * gz;
            if (gn2 > 0.f) {                                                // the library returns q unchanged for a zero gyro reading
                // qDot = 0.5 * q (x) [0, gyr]   (eq. 12)
                float dw = 0.5f * (-qx * gx - qy * gy - qz * gz);
                float dx = 0.5f * (qw * gx + qy * gz - qz * gy);
                float dy = 0.5f * (qw * gy - qx * gz + qz * gx);
                float dz = 0.5f * (qw * gz + qx * gy - qy * gx);
                const float an2 = ax * ax + ay * ay + az * az;
                if (an2 > 0.f) {
                    const float ia = 1.f / __builtin_sqrtf(an2);
                    const float iq = 1.f / __builtin_sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
                    const float w = qw * iq, x = qx * iq, y = qy * iq, z = qz * iq;
                    const float f0 = 2.f * (x * z - w * y) - ax * ia;     // eq. 25
                    const float f1 = 2.f * (w * x + y * z) - ay * ia;
                    const float f2 = 2.f * (0.5f - x * x - y * y) - az * ia;
                    if (f0 * f0 + f1 * f1 + f2 * f2 > 0.f) {
                        float g0 = -2.f * y * f0 + 2.f * x * f1;                 // J^T f  (eqs. 26, 34)
                        float g1 = 2.f * z * f0 + 2.f * w * f1 - 4.f * x * f2;
                        float g2 = -2.f * w * f0 + 2.f * z * f1 - 4.f * y * f2;
                        float g3 = 2.f * x * f0 + 2.f * y * f1;
                        const float gn2g = g0 * g0 + g1 * g1 + g2 * g2 + g3 * g3;
                        if (gn2g > 0.f) {      // a vanishing gradient (f along the null space of J^T) would divide 0 by 0: no correction
                            const float ig = P.gain / __builtin_sqrtf(gn2g);
                            dw -= ig * g0; dx -= ig * g1; dy -= ig * g2; dz -= ig * g3;   // eq. 33
                        }
                    }
                }
                qw += dw * P.dt; qx += dx * P.dt; qy += dy * P.dt; qz += dz * P.dt;   // eq. 13
                const float inv = 1.f / __builtin_sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
                qw *= inv; qx *= inv; qy *= inv; qz *= inv;
            }
            S.orient[env] = qw; S.orient[n + env] = qx; S.orient[2 * n + env] = qy; S.orient[3 * n + env] = qz;
            still_alias = false;
        }
        float roll, pitch, yaw;
        po_euler(qw, qx, qy, qz, roll, pitch, yaw);
        const float cvx = WS.vel[env], cvy = WS.vel[n + env];
        const float theta = atan2f(WS.head[n + env], WS.head[env]);       // control_inputs.py:69-73
        float *fr = s_new[le];
        fr[0] = gx; fr[1] = gy; fr[2] = gz; fr[3] = ax; fr[4] = ay; fr[5] = az;
        fr[6] = roll; fr[7] = pitch; fr[8] = yaw;
        fr[9] = s[30]; fr[10] = s[31];
        for (int j = 0; j < 12; ++j) fr[11 + j] = fminf(fmaxf(eff_actions[(size_t)env * 12 + j], -1.f), 1.f);
        fr[23] = cvx; fr[24] = cvy; fr[25] = theta;
        // FIFO: the newest frame replaces the oldest one (:80-83)
        int slot = S.head[env] + 1;
        if (slot >= W) slot = 0;
        s_slot[le] = slot;
        float *ring = S.stack + ((size_t)env * W + slot) * QG_PO_FRAME;
        for (int i = 0; i < QG_PO_FRAME; ++i) ring[i] = fr[i];
        const bool fin = P.auto_reset && done[env];
        s_fin[le] = fin ? 1 : 0;
        if (!fin) {
            S.head[env] = slot;
            S.alias[env] = still_alias ? 1 : 0;
            S.nstep[env] = nstep;
        } else {
            // frame of reset() (:59-69): zero sensors, the estimate as it stands (mj_resetData has put [1,0,0,0] into qpos
            // if the estimate still aliases it), default ctrl, the command of the episode that just ended
            float rq[4] = {qw, qx, qy, qz};
            if (still_alias) { rq[0] = 1.f; rq[1] = rq[2] = rq[3] = 0.f; }
            po_euler(rq[0], rq[1], rq[2], rq[3], roll, pitch, yaw);
            float *rf = s_rst[le];
            for (int i = 0; i < 6; ++i) rf[i] = 0.f;
            rf[6] = roll; rf[7] = pitch; rf[8] = yaw; rf[9] = 0.f; rf[10] = 0.f;
            for (int j = 0; j < 12; ++j) rf[11 + j] = P.default_ctrl[j];
            rf[23] = cvx; rf[24] = cvy; rf[25] = theta;
            S.head[env] = slot;
            S.alias[env] = 1;                                              // :67 computed_orientation = data.qpos[3:7]
            S.nstep[env] = 0;
            // random_controls on the device: the new episode's command, drawn only now that both frames show the old one
            if (sample_cmd) walk_sample_command(WP, WS, n, env, seed, env_index_base, episode[env] - 1);
        }
    }
    __syncthreads();
    // phase 2: 16 threads per env walk its row in steps of 16 floats (64-byte segments; no division by the run-time width)
    const int envs = min(QG_PO_ENVS, n - env0);
    const int width = W * QG_PO_FRAME;
    const int le = threadIdx.x >> 4, l16 = threadIdx.x & 15;
    if (le < envs) {
        const size_t row = (size_t)(env0 + le) * width;
        float *__restrict__ o = out + row;
        float *__restrict__ t_o = term_out ? term_out + row : nullptr;
        const float *__restrict__ st = S.stack + row;
        const int slot = s_slot[le];
        const bool fin = s_fin[le] != 0;
        // four segments per trip: the four ring reads are issued before the first store
        for (int r0 = l16; r0 < width; r0 += 64) {
            float v[4];
            int ii[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = r0 + 16 * u;
                const int f = r / QG_PO_FRAME, i = r - f * QG_PO_FRAME;
                int src = slot + 1 + f;                                    // oldest frame first
                if (src >= W) src -= W;
                ii[u] = i;
                v[u] = 0.f;
                if (r < width) v[u] = (f == W - 1) ? s_new[le][i] : st[src * QG_PO_FRAME + i];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = r0 + 16 * u;
                if (r >= width) break;
                if (!fin) { o[r] = v[u]; continue; }
                // the env finished: hand out the terminal stack, then the reset stack
                if (t_o) t_o[r] = v[u];
                o[r] = s_rst[le][ii[u]];
            }
        }
    }
    __syncthreads();                                                       // every read of the old ring contents is done
    if (le < envs && s_fin[le]) {                                          // restart the FIFO from the reset frame
        float *st = S.stack + (size_t)(env0 + le) * width;
        for (int r = l16; r < width; r += 16) st[r] = s_rst[le][r % QG_PO_FRAME];
    }
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
    rf[23] = vel[env]; rf[24] = vel[n + env]; rf[25] = atan2f(head[n + env], head[env]);
    float *st = S.stack + (size_t)env * P.window * QG_PO_FRAME;
    for (int f = 0; f < P.window; ++f)
        for (int i = 0; i < QG_PO_FRAME; ++i) {
            st[f * QG_PO_FRAME + i] = rf[i];
            if (out) out[(size_t)env * P.window * QG_PO_FRAME + f * QG_PO_FRAME + i] = rf[i];
        }
    S.alias[env] = 1;
    S.nstep[env] = 0;
}
