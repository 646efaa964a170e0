// qg_po_dev.h -- per-env device functions of the partially observable observation pack, shared by
// the stand-alone kernel (qg_po.hip) and by the fused variant of the one-link-per-lane step kernel (qg_kernel_link.hip,
// qg_step_kernel_link<WALK, PO>): POWalkingQuadrupedEnv of antopio26/quadruped-gym (src/envs/po_walking_quad.py:10-90).
// Per env and step one 26-value frame [gyro 3, accel 3, Madgwick-IMU Euler angles 3, body_vel xy 2, data.ctrl 12,
// command vx vy, heading angle] (:48-56), stacked over `obs_window` steps as a FIFO (:65,80-88).
// The orientation filter is ahrs.filters.Madgwick (third party, not available offline): restated from the
// published IMU form of the algorithm (eqs. 12, 13, 25, 26, 33, 34; gain 0.033) -- parity unpinned.
// Reference quirks kept: the filter only runs while data.time > settling_time / 2 (:37); after a reset the
// estimate IS the live data.qpos[3:7] (a NumPy view, :67) until the first filter update replaces it; the frame
// reset() returns shows zero sensors, the PREVIOUS estimate and the PREVIOUS command (:59-69).
#pragma once
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
    // the history copy of the wave-level fused forms (po_wave_copy_*), precomputed on the host: 16-byte groups per row of history
    // (ceil(26 (window - 1) / 4); 0 for window 1) and 64 divided by it (quotient, remainder): how a lane's next group follows from
    // its last one without a division
    int32_t hist_groups, hg_q64, hg_r64;
};

struct KPoState {
    float *orient;       // [4][n]  computed_orientation
    uint8_t *alias;      // [n]     the estimate is the live data.qpos[3:7]
    int32_t *nstep;      // [n]     substeps since the last reset (data.time of the step being observed)
    float *stack;        // [n][window][26]  ring of the last `window` frames
    int32_t *head;       // [n]              ring slot of the newest frame
};

// 1 / sqrt(x) of the filter's normalisations: the hardware instruction (1 ulp) instead of an IEEE square root followed by an IEEE
// division (~20 instructions, six times per frame on lanes that have nothing else to do); every argument is a guarded, O(1) sum of squares
__device__ __forceinline__ float po_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }

// atan2 / asin of the frame's Euler angles and command heading: a ~25-instruction form (the math library's are ~70 each, evaluated on
// lanes that have nothing else to do).  |y| / |x| reduced to [0, 1], then Cephes' single-precision reduction at tan(pi/8) and its
// degree-7 odd polynomial (abs. error < 3e-7 rad over the plane, checked against numpy in tests/test_po_env.py); asin(v) = atan2(v,
// sqrt((1 - v)(1 + v))).
__device__ __forceinline__ float po_atan2(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float a = mn * __builtin_amdgcn_rcpf(mx);
    const bool big = a > 0.41421356237f;
    const float t = big ? (a - 1.f) * __builtin_amdgcn_rcpf(a + 1.f) : a;
    const float z = t * t;
    const float p = fmaf(fmaf(fmaf(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f);
    float r = fmaf(p * z, t, t) + (big ? 0.78539816339f : 0.f);
    r = ay > ax ? 1.57079632679f - r : r;
    r = x < 0.f ? 3.14159265359f - r : r;
    r = mx == 0.f ? 0.f : r;
    return __builtin_copysignf(r, y);
}
__device__ __forceinline__ float po_asin(float v) {
    return po_atan2(v, __builtin_amdgcn_sqrtf(fmaxf((1.f - v) * (1.f + v), 0.f)));
}

__device__ __forceinline__ void po_euler(float w, float x, float y, float z, float &roll, float &pitch, float &yaw) {
    float inv = po_rsqrt(w * w + x * x + y * y + z * z);
    w *= inv; x *= inv; y *= inv; z *= inv;
    roll = po_atan2(2.f * (w * x + y * z), 1.f - 2.f * (x * x + y * y));
    pitch = po_asin(fminf(fmaxf(2.f * (w * y - z * x), -1.f), 1.f));
    yaw = po_atan2(2.f * (w * z + x * y), 1.f - 2.f * (y * y + z * z));
}


// A workgroup of QG_PO_THREADS threads owns QG_PO_ENVS envs (thread = 16 * local env + l16) in both users of this header.
#define QG_PO_ENVS 16
#define QG_PO_THREADS 256

// what the fused step kernel takes as an extra by-value kernel argument (see KWalkLaunch for why by value)
struct KPoLaunch {
    KPoParams P;
    KPoState S;
    float *out;              // [n][window * 26]
    float *term_out;         // [n][window * 26] or NULL
    int32_t sample;          // redraw the command of the envs this step auto-resets (after both frames show the old one)
};
struct KPoNone {};
template <bool PO> struct PoArgT { typedef KPoNone type; };
template <> struct PoArgT<true> { typedef KPoLaunch type; };

// the per-env filter state as it stood before this step: loaded up front (in the fused kernel among the state loads)
struct PoEnvIn {
    float ow, ox, oy, oz;
    int alias, nstep, head;
};
__device__ __forceinline__ PoEnvIn po_env_load(const KPoState &S, int n, int env) {
    PoEnvIn in;
    in.ow = S.orient[env]; in.ox = S.orient[n + env]; in.oy = S.orient[2 * n + env]; in.oz = S.orient[3 * n + env];
    in.alias = S.alias[env];
    in.nstep = S.nstep[env];
    in.head = S.head[env];
    return in;
}

// One filter update (ahrs Madgwick.updateIMU) of the estimate q with the step's gyro / accelerometer readings.
__device__ __forceinline__ void po_filter_update(const KPoParams &P, float gx, float gy, float gz, float ax, float ay, float az,
                                                 float &qw, float &qx, float &qy, float &qz) {
    const float gn2 = gx * gx + gy * gy + gz * gz;
    if (gn2 > 0.f) {                                                // the library returns q unchanged for a zero gyro reading
        // qDot = 0.5 * q (x) [0, gyr]   (eq. 12)
        float dw = 0.5f * (-qx * gx - qy * gy - qz * gz);
        float dx = 0.5f * (qw * gx + qy * gz - qz * gy);
        float dy = 0.5f * (qw * gy - qx * gz + qz * gx);
        float dz = 0.5f * (qw * gz + qx * gy - qy * gx);
        const float an2 = ax * ax + ay * ay + az * az;
        if (an2 > 0.f) {
            const float ia = po_rsqrt(an2);
            const float iq = po_rsqrt(qw * qw + qx * qx + qy * qy + qz * qz);
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
                    const float ig = P.gain * po_rsqrt(gn2g);
                    dw -= ig * g0; dx -= ig * g1; dy -= ig * g2; dz -= ig * g3;   // eq. 33
                }
            }
        }
        qw += dw * P.dt; qx += dx * P.dt; qy += dy * P.dt; qz += dz * P.dt;   // eq. 13
        const float inv = po_rsqrt(qw * qw + qx * qx + qy * qy + qz * qz);
        qw *= inv; qx *= inv; qy *= inv; qz *= inv;
    }
}

// Phase 1, one thread per env: orientation filter, the env's new 26-value frame `fr` (all but the 12 data.ctrl values fr[11..22],
// which the caller provides) and -- for an env the physics has just auto-reset -- the frame `rf` reset() would return.
//   s            the step's 33 sensor values (any address space)
//   live_q       data.qpos[3:7] as the step leaves it (after an auto-reset: the reset pose), what an aliasing estimate shows
//   cvx.. hy     the command of the episode the step belongs to
// Returns the ring slot of the new frame and whether the env finished; updates the env's filter state.
__device__ __forceinline__ void po_frame_env(const KPoParams &P, const KPoState &S, int n, int env, const PoEnvIn &in, const float *s,
                                             float lqw, float lqx, float lqy, float lqz, float cvx, float cvy, float hx, float hy,
                                             bool done, float *fr, float *rf, int &slot_out, int &fin_out) {
    const float gx = s[15], gy = s[16], gz = s[17], ax = s[12], ay = s[13], az = s[14];
    const int nstep = in.nstep + P.frame_skip;
    const bool alias = in.alias != 0;
    float qw = alias ? lqw : in.ow, qx = alias ? lqx : in.ox, qy = alias ? lqy : in.oy, qz = alias ? lqz : in.oz;
    bool still_alias = alias;
    if (nstep >= P.half_settle_substeps) {                             // po_walking_quad.py:37
        po_filter_update(P, gx, gy, gz, ax, ay, az, qw, qx, qy, qz);
        S.orient[env] = qw; S.orient[n + env] = qx; S.orient[2 * n + env] = qy; S.orient[3 * n + env] = qz;
        still_alias = false;
    }
    float roll, pitch, yaw;
    po_euler(qw, qx, qy, qz, roll, pitch, yaw);
    const float theta = po_atan2(hy, hx);                               // control_inputs.py:69-73
    fr[0] = gx; fr[1] = gy; fr[2] = gz; fr[3] = ax; fr[4] = ay; fr[5] = az;
    fr[6] = roll; fr[7] = pitch; fr[8] = yaw;
    fr[9] = s[30]; fr[10] = s[31];
    fr[23] = cvx; fr[24] = cvy; fr[25] = theta;
    // FIFO: the newest frame replaces the oldest one (:80-83)
    int slot = in.head + 1;
    if (slot >= P.window) slot = 0;
    slot_out = slot;
    const bool fin = P.auto_reset && done;
    fin_out = fin ? 1 : 0;
    S.head[env] = slot;
    if (!fin) {
        S.alias[env] = still_alias ? 1 : 0;
        S.nstep[env] = nstep;
    } else {
        // frame of reset() (:59-69): zero sensors, the estimate as it stands (mj_resetData has put [1,0,0,0] into qpos
        // if the estimate still aliases it), default ctrl, the command of the episode that just ended
        float rq[4] = {qw, qx, qy, qz};
        if (still_alias) { rq[0] = 1.f; rq[1] = rq[2] = rq[3] = 0.f; }
        po_euler(rq[0], rq[1], rq[2], rq[3], roll, pitch, yaw);
        for (int i = 0; i < 6; ++i) rf[i] = 0.f;
        rf[6] = roll; rf[7] = pitch; rf[8] = yaw; rf[9] = 0.f; rf[10] = 0.f;
        for (int j = 0; j < 12; ++j) rf[11 + j] = P.default_ctrl[j];
        rf[23] = cvx; rf[24] = cvy; rf[25] = theta;
        S.alias[env] = 1;                                              // :67 computed_orientation = data.qpos[3:7]
        S.nstep[env] = 0;
    }
}

// Phases 2 and 3, every thread of the workgroup, 16 per env (le = local env, l16 = lane within it), after a barrier that follows
// phase 1: the env's row of `out` (window x 26 floats) is written in 64-byte segments, frame f coming from ring slot
// (head + 1 + f) mod window -- the FIFO of po_walking_quad.py:80-83 without moving 9 of 10 frames every step (a per-env shift of
// the stack measured 78 us per launch at 4096 envs and window 10, four times the physics); the new frame goes into its ring slot;
// an env that finished hands out the terminal stack and restarts its FIFO from the reset frame.
__device__ __forceinline__ void po_emit_rows(const KPoParams &P, const KPoState &S, int n, int env0, int le, int l16,
                                             const float (*s_new)[QG_PO_FRAME], const float (*s_rst)[QG_PO_FRAME], const int *s_slot,
                                             const int *s_fin, float *__restrict__ out, float *__restrict__ term_out) {
    const int envs = min(QG_PO_ENVS, n - env0);
    const int W = P.window;
    const int width = W * QG_PO_FRAME;
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
        if (!fin) {                                                        // nobody reads the slot of the newest frame above
            float *ring = S.stack + row + (size_t)slot * QG_PO_FRAME;
            for (int i = l16; i < QG_PO_FRAME; i += 16) ring[i] = s_new[le][i];
        }
    }
    __syncthreads();                                                       // every read of the old ring contents is done
    if (le < envs && s_fin[le]) {                                          // restart the FIFO from the reset frame
        float *st = S.stack + (size_t)(env0 + le) * width;
        for (int r = l16; r < width; r += 16) st[r] = s_rst[le][r % QG_PO_FRAME];
    }
}

// ---- the fused form (qg_step_kernel_link<WALK, PO>): the same output in two parts -------------------------------------------------
// Part 1, in the step kernel's PROLOGUE: the W - 1 frames the new stack keeps are known before the physics runs, so they are copied
// ring -> out there, their loads in flight together with the state loads (a wave that is alone on its SIMD would otherwise sit
// through their latency in the epilogue, behind the stores of the step).  `slot` = ring slot the NEW frame will take.  One trip of 16 loads + 16 stores per lane covers obs_window <= 10.
__device__ __forceinline__ void po_copy_history(const KPoParams &P, const KPoState &S, size_t row, int slot, int l16,
                                                float *__restrict__ out, bool live = true) {
    // the new row is the ring rotated: out[r] = ring[(r + 26 * (slot + 1)) mod (26 W)] for r < 26 (W - 1).  Lanes past the end
    // repeat element 26 (W - 1) - 1 (same value to the same address: harmless), so that no load is predicated; the lanes of a
    // wave's tail (no env of their own, `live` false) shadow the last env's loads and store nothing -- a whole env is live or not,
    // so the predicate costs no divergence inside an env, and nothing then orders a shadow copy against the epilogue of the wave
    // that owns the env.
    const int W = P.window, width = W * QG_PO_FRAME, hist = width - QG_PO_FRAME;
    const int off = (slot + 1 >= W ? 0 : slot + 1) * QG_PO_FRAME;
    const float *__restrict__ st = S.stack + row;
    float *__restrict__ o = out + row;
    for (int r0 = l16; r0 < hist; r0 += 256) {
        float v[16];
        int rr[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            rr[u] = min(r0 + 16 * u, hist - 1);
            int src = rr[u] + off;
            if (src >= width) src -= width;
            v[u] = st[src];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) if (live) o[rr[u]] = v[u];
    }
}
// Phase 1 of the fused form: the 16 lanes of the env run it together.  The filter update is redundant in all of them (no lane has
// anything else to do); the four inverse trigonometric evaluations of the frame -- roll, yaw, the heading angle of the command
// (atan2f) and pitch (asinf) -- which a single thread does one after the other (~60 instructions each, most of phase 1), are ONE
// atan2f on per-lane arguments plus one asinf; the same expressions as po_euler, so the same bits.  Everything else (the frame's
// plain values, the filter state, the rare reset frame) is the env's lead lane's.  `in`, `done`, live_q: identical in the 16
// lanes; cvx .. hy: needed in the lead lane and (hx, hy) in lane 2.
__device__ __forceinline__ void po_frame_env16(const KPoParams &P, const KPoState &S, int n, int env, int l16, bool lead, const PoEnvIn &in,
                                               const float *s, float lqw, float lqx, float lqy, float lqz, float cvx, float cvy, float hx,
                                               float hy, bool done, float *fr, float *rf, int *slot_out, int *fin_out) {
    const float gx = s[15], gy = s[16], gz = s[17], ax = s[12], ay = s[13], az = s[14];
    const int nstep = in.nstep + P.frame_skip;
    const bool alias = in.alias != 0;
    float qw = alias ? lqw : in.ow, qx = alias ? lqx : in.ox, qy = alias ? lqy : in.oy, qz = alias ? lqz : in.oz;
    bool still_alias = alias;
    if (nstep >= P.half_settle_substeps) {                             // po_walking_quad.py:37
        po_filter_update(P, gx, gy, gz, ax, ay, az, qw, qx, qy, qz);
        if (lead) { S.orient[env] = qw; S.orient[n + env] = qx; S.orient[2 * n + env] = qy; S.orient[3 * n + env] = qz; }
        still_alias = false;
    }
    {
        const float inv = po_rsqrt(qw * qw + qx * qx + qy * qy + qz * qz);      // po_euler, spread over lanes 0..3
        const float w = qw * inv, x = qx * inv, y = qy * inv, z = qz * inv;
        const float ya = l16 == 0 ? 2.f * (w * x + y * z) : (l16 == 1 ? 2.f * (w * z + x * y) : hy);
        const float xa = l16 == 0 ? 1.f - 2.f * (x * x + y * y) : (l16 == 1 ? 1.f - 2.f * (y * y + z * z) : hx);
        const float ang = po_atan2(ya, xa);
        const float pit = po_asin(fminf(fmaxf(2.f * (w * y - z * x), -1.f), 1.f));
        if (l16 == 0) fr[6] = ang;                                     // roll
        if (l16 == 1) fr[8] = ang;                                     // yaw
        if (l16 == 2) { fr[25] = ang; rf[25] = ang; }                  // heading angle of the command, control_inputs.py:69-73
        if (l16 == 3) fr[7] = pit;
    }
    if (!lead) return;
    fr[0] = gx; fr[1] = gy; fr[2] = gz; fr[3] = ax; fr[4] = ay; fr[5] = az;
    fr[9] = s[30]; fr[10] = s[31];
    fr[23] = cvx; fr[24] = cvy;
    int slot = in.head + 1;                                            // FIFO: the newest frame replaces the oldest one (:80-83)
    if (slot >= P.window) slot = 0;
    *slot_out = slot;
    const bool fin = P.auto_reset && done;
    *fin_out = fin ? 1 : 0;
    S.head[env] = slot;
    if (!fin) {
        S.alias[env] = still_alias ? 1 : 0;
        S.nstep[env] = nstep;
    } else {                                                           // frame of reset() (:59-69), see po_frame_env
        float rq[4] = {qw, qx, qy, qz};
        if (still_alias) { rq[0] = 1.f; rq[1] = rq[2] = rq[3] = 0.f; }
        float roll, pitch, yaw;
        po_euler(rq[0], rq[1], rq[2], rq[3], roll, pitch, yaw);
        for (int i = 0; i < 6; ++i) rf[i] = 0.f;
        rf[6] = roll; rf[7] = pitch; rf[8] = yaw; rf[9] = 0.f; rf[10] = 0.f;
        for (int j = 0; j < 12; ++j) rf[11 + j] = P.default_ctrl[j];
        rf[23] = cvx; rf[24] = cvy;
        S.alias[env] = 1;                                              // :67 computed_orientation = data.qpos[3:7]
        S.nstep[env] = 0;
    }
}

// Part 2, in the EPILOGUE after phase 1 and a barrier: the new frame into the last 26 values of the row and into its ring slot; an
// env that finished (rare) hands out the terminal stack, shows the reset stack and restarts its FIFO from the reset frame.
__device__ __forceinline__ void po_emit_new(const KPoParams &P, const KPoState &S, int n, int env0, int le, int l16,
                                            const float (*s_new)[QG_PO_FRAME], const float (*s_rst)[QG_PO_FRAME], const int *s_slot,
                                            const int *s_fin, float *__restrict__ out, float *__restrict__ term_out) {
    const int envs = min(QG_PO_ENVS, n - env0);
    const int W = P.window;
    const int width = W * QG_PO_FRAME;
    if (le < envs) {
        const size_t row = (size_t)(env0 + le) * width;
        float *__restrict__ o = out + row;
        float *__restrict__ stw = S.stack + row;
        const int slot = s_slot[le];
        if (!s_fin[le]) {
            for (int i = l16; i < QG_PO_FRAME; i += 16) {
                const float x = s_new[le][i];
                o[width - QG_PO_FRAME + i] = x;
                stw[slot * QG_PO_FRAME + i] = x;
            }
        } else {
            float *__restrict__ t_o = term_out ? term_out + row : nullptr;
            for (int r = l16; r < width; r += 16) {
                const int f = r / QG_PO_FRAME, i = r - f * QG_PO_FRAME;
                int src = slot + 1 + f;
                if (src >= W) src -= W;
                const float x = (f == W - 1) ? s_new[le][i] : stw[src * QG_PO_FRAME + i];
                if (t_o) t_o[r] = x;
                o[r] = s_rst[le][i];
            }
        }
    }
    wave_sync();                                                           // every read of the old ring contents is done
    if (le < envs && s_fin[le]) {
        float *st = S.stack + (size_t)(env0 + le) * width;
        for (int r = l16; r < width; r += 16) st[r] = s_rst[le][r % QG_PO_FRAME];
    }
}


// ---- the fused form for the step kernels whose WAVE owns a block of consecutive envs (one leg per lane: 16 envs x 4 lanes, two legs
// per lane: 32 envs x 2 lanes; round 3).  At the batch sizes these kernels serve the observation pack is HBM traffic -- per env and
// step 936 B of ring read and 1040 B of row written at window 10, 65 MB per launch at 32 768 envs, 15 us as a kernel of its own at
// the ~4.5 TB/s it reached -- next to a physics launch that leaves the memory system idle.  So the copy ring -> out of the W - 1
// frames the new stack keeps is spread over the SUBSTEP LOOP: at the head of a substep a lane loads its next K 16-byte groups, at its
// tail (2 500 instructions later: the latency has long passed) it stores them.  The wave's rows are contiguous in `out` and in the
// ring (ENVS x width floats); a group is four consecutive floats of one row's history, group g of the block belongs to lane g mod 64,
// so a wave-instruction moves 1 KB of consecutive addresses (rows permitting).  The ring is rotated by whole frames (26 floats: even,
// not a multiple of 4), so a group's source is two 8-byte halves, each wrapped on its own; a row's last group is a half when
// 26 (W - 1) is not a multiple of 4.  What does not fit into frame_skip x K groups is copied after the loop.
struct PoF2 { float x, y; };
struct PoF4 { float x, y, z, w; };
struct PoCopyState { int c, el, left; };          // this lane's next group: column group within the row's history, row, groups left
template <int K> struct PoCopyRegs { PoF2 a[K], b[K]; };

// rotation offset (floats) of the oldest frame the new stack keeps, for a ring whose NEWEST frame sits in slot `head`
__device__ __forceinline__ int po_hist_offset(const KPoParams &P, int head) {
    int slot = head + 1;                           // the slot the new frame will take (the oldest frame's)
    if (slot >= P.window) slot = 0;
    int first = slot + 1;                          // the oldest frame that stays
    if (first >= P.window) first = 0;
    return first * QG_PO_FRAME;
}
__device__ __forceinline__ void po_wave_copy_init(const KPoParams &P, int lane, int live_envs, PoCopyState &st) {
    const int hq = P.hist_groups;
    const int total = live_envs * hq;
    st.left = lane < total ? (total - lane + 63) >> 6 : 0;
    st.el = hq > 0 ? lane / hq : 0;
    st.c = lane - st.el * hq;
}
__device__ __forceinline__ void po_copy_advance(const KPoParams &P, int &c, int &el) {
    c += P.hg_r64; el += P.hg_q64;
    if (c >= P.hist_groups) { c -= P.hist_groups; el += 1; }
}
// loads of this lane's next K groups (unpredicated: a lane past its last group repeats group (0, 0) of the block)
template <int K>
__device__ __forceinline__ void po_wave_copy_load(const KPoParams &P, const float *__restrict__ ring_block, const int *s_off, const PoCopyState &st,
                                                  PoCopyRegs<K> &R) {
    const int width = P.window * QG_PO_FRAME;
    int c = st.c, el = st.el;
#pragma unroll
    for (int u = 0; u < K; ++u) {
        const bool valid = u < st.left;
        const int ec = valid ? el : 0, cc = valid ? c : 0;
        const int x = 4 * cc + s_off[ec], y = x + 2;
        const int xa = x >= width ? x - width : x, yb = y >= width ? y - width : y;
        const float *rowp = ring_block + (size_t)ec * width;
        R.a[u] = *reinterpret_cast<const PoF2 *>(rowp + xa);
        R.b[u] = *reinterpret_cast<const PoF2 *>(rowp + yb);
        po_copy_advance(P, c, el);
    }
}
template <int K>
__device__ __forceinline__ void po_wave_copy_store(const KPoParams &P, float *__restrict__ out_block, PoCopyState &st, const PoCopyRegs<K> &R) {
    const int width = P.window * QG_PO_FRAME, hist = width - QG_PO_FRAME;
    int c = st.c, el = st.el;
#pragma unroll
    for (int u = 0; u < K; ++u) {
        if (u < st.left) {
            float *dst = out_block + (size_t)el * width + 4 * c;
            if (4 * c + 4 <= hist) {
                const PoF4 v = {R.a[u].x, R.a[u].y, R.b[u].x, R.b[u].y};
                *reinterpret_cast<PoF4 *>(dst) = v;
            } else {
                *reinterpret_cast<PoF2 *>(dst) = R.a[u];                  // a row's last group when 26 (W - 1) is not a multiple of 4
            }
        }
        po_copy_advance(P, c, el);
    }
    st.c = c; st.el = el;
    st.left = st.left > K ? st.left - K : 0;
}
// what the substep loop did not get to (long windows, small frame_skip): the same groups, load -> store
template <int K>
__device__ __forceinline__ void po_wave_copy_rest(const KPoParams &P, const float *__restrict__ ring_block, float *__restrict__ out_block,
                                                  const int *s_off, PoCopyState &st) {
    while (__any(st.left > 0)) {
        PoCopyRegs<K> R;
        po_wave_copy_load<K>(P, ring_block, s_off, st, R);
        po_wave_copy_store<K>(P, out_block, st, R);
    }
}
// Epilogue of the wave-level fused forms, after every lead lane has built its env's frame (po_frame_env) into s_new / s_rst /
// s_slot / s_fin and a wave-level fence: the new frames go into the last 26 values of the rows and into their ring slots; an env
// that finished (rare) hands out the terminal stack, shows the reset stack and restarts its FIFO from the reset frame.  The whole
// wave works on its block (env0 .. env0 + live_envs).
__device__ __forceinline__ void po_wave_emit(const KPoParams &P, const KPoState &S, int env0, int live_envs, int lane,
                                             const float (*s_new)[QG_PO_FRAME], const float (*s_rst)[QG_PO_FRAME], const int *s_slot,
                                             const int *s_fin, float *__restrict__ out, float *__restrict__ term_out) {
    const int W = P.window, width = W * QG_PO_FRAME, hist = width - QG_PO_FRAME;
    const size_t block = (size_t)env0 * width;
    for (int idx = lane; idx < live_envs * QG_PO_FRAME; idx += 64) {
        const int el = idx / QG_PO_FRAME, i = idx - el * QG_PO_FRAME;
        if (!s_fin[el]) {
            const float x = s_new[el][i];
            const size_t row = block + (size_t)el * width;
            out[row + hist + i] = x;
            S.stack[row + s_slot[el] * QG_PO_FRAME + i] = x;
        }
    }
    unsigned long long fins = __ballot(lane < live_envs && s_fin[lane < live_envs ? lane : 0] != 0);
    if (fins == 0) return;                                                 // the usual case
    for (unsigned long long m = fins; m; m &= m - 1) {
        const int el = __ffsll((long long)m) - 1;
        const size_t row = block + (size_t)el * width;
        const int slot = s_slot[el];
        for (int r = lane; r < width; r += 64) {
            const int f = r / QG_PO_FRAME, i = r - f * QG_PO_FRAME;
            int src = slot + 1 + f;                                        // oldest frame first
            if (src >= W) src -= W;
            const float x = (f == W - 1) ? s_new[el][i] : S.stack[row + src * QG_PO_FRAME + i];
            if (term_out) term_out[row + r] = x;
            out[row + r] = s_rst[el][i];
        }
    }
    wave_sync();                                                           // every read of the old ring contents is done
    for (unsigned long long m = fins; m; m &= m - 1) {
        const int el = __ffsll((long long)m) - 1;
        const size_t row = block + (size_t)el * width;
        for (int r = lane; r < width; r += 64) S.stack[row + r] = s_rst[el][r % QG_PO_FRAME];
    }
}
