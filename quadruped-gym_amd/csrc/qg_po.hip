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
    float *stack;        // [n][window][26]
};

__device__ __forceinline__ void po_euler(float w, float x, float y, float z, float &roll, float &pitch, float &yaw) {
    float inv = 1.f / __builtin_sqrtf(w * w + x * x + y * y + z * z);
    w *= inv; x *= inv; y *= inv; z *= inv;
    roll = atan2f(2.f * (w * x + y * z), 1.f - 2.f * (x * x + y * y));
    pitch = asinf(fminf(fmaxf(2.f * (w * y - z * x), -1.f), 1.f));
    yaw = atan2f(2.f * (w * z + x * y), 1.f - 2.f * (y * y + z * z));
}

// one thread per env; runs after the physics (and walking-reward) kernels of the step
__global__ void qg_po_frame_kernel(KPoParams P, KPoState S, int n, const float *__restrict__ obs33, const float *__restrict__ eff_actions,
                                   const float *__restrict__ qpos /* [19][n] */, const float *__restrict__ vel, const float *__restrict__ head,
                                   const uint8_t *__restrict__ done, float *__restrict__ out /* [n][window*26] */,
                                   float *__restrict__ term_out /* [n][window*26] or NULL */) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= n) return;
    const float *s = obs33 + (size_t)env * 33;
    const float gx = s[15], gy = s[16], gz = s[17], ax = s[12], ay = s[13], az = s[14];
    const int nstep = S.nstep[env] + P.frame_skip;
    float qw, qx, qy, qz;
    const bool alias = S.alias[env] != 0;
    if (alias) { qw = qpos[3 * n + env]; qx = qpos[4 * n + env]; qy = qpos[5 * n + env]; qz = qpos[6 * n + env]; }
    else { qw = S.orient[env]; qx = S.orient[n + env]; qy = S.orient[2 * n + env]; qz = S.orient[3 * n + env]; }
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
        S.alias[env] = 0;
    }
    float roll, pitch, yaw;
    po_euler(qw, qx, qy, qz, roll, pitch, yaw);
    const float cvx = vel[env], cvy = vel[n + env];
    const float theta = atan2f(head[n + env], head[env]);             // control_inputs.py:69-73
    // FIFO: drop the oldest frame, append the new one (:80-83)
    float *st = S.stack + (size_t)env * P.window * QG_PO_FRAME;
    for (int i = 0; i < (P.window - 1) * QG_PO_FRAME; ++i) st[i] = st[i + QG_PO_FRAME];
    float *fr = st + (size_t)(P.window - 1) * QG_PO_FRAME;
    fr[0] = gx; fr[1] = gy; fr[2] = gz; fr[3] = ax; fr[4] = ay; fr[5] = az;
    fr[6] = roll; fr[7] = pitch; fr[8] = yaw;
    fr[9] = s[30]; fr[10] = s[31];
    for (int j = 0; j < 12; ++j) fr[11 + j] = fminf(fmaxf(eff_actions[(size_t)env * 12 + j], -1.f), 1.f);
    fr[23] = cvx; fr[24] = cvy; fr[25] = theta;
    const int width = P.window * QG_PO_FRAME;
    float *o = out + (size_t)env * width;
    const bool rst = P.auto_reset && done[env];
    if (!rst) {
        for (int i = 0; i < width; ++i) o[i] = st[i];
        S.nstep[env] = nstep;
        return;
    }
    // the env finished and was auto-reset by the physics kernel: hand out the terminal stack, then the reset stack
    if (term_out)
        for (int i = 0; i < width; ++i) term_out[(size_t)env * width + i] = st[i];
    // frame of reset() (:59-69): zero sensors, the estimate as it stands (mj_resetData has put [1,0,0,0] into qpos
    // if the estimate still aliases it), default ctrl, the command of the episode that just ended
    float rq[4] = {qw, qx, qy, qz};
    if (S.alias[env]) { rq[0] = 1.f; rq[1] = rq[2] = rq[3] = 0.f; }
    po_euler(rq[0], rq[1], rq[2], rq[3], roll, pitch, yaw);
    float rf[QG_PO_FRAME];
    for (int i = 0; i < 6; ++i) rf[i] = 0.f;
    rf[6] = roll; rf[7] = pitch; rf[8] = yaw; rf[9] = 0.f; rf[10] = 0.f;
    for (int j = 0; j < 12; ++j) rf[11 + j] = P.default_ctrl[j];
    rf[23] = cvx; rf[24] = cvy; rf[25] = theta;
    for (int f = 0; f < P.window; ++f)
        for (int i = 0; i < QG_PO_FRAME; ++i) { st[f * QG_PO_FRAME + i] = rf[i]; o[f * QG_PO_FRAME + i] = rf[i]; }
    S.alias[env] = 1;                                                  // :67 computed_orientation = data.qpos[3:7]
    S.nstep[env] = 0;
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
