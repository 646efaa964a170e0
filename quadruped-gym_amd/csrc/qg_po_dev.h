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
};

struct KPoState {
    float *orient;       // [4][n]  computed_orientation
    uint8_t *alias;      // [n]     the estimate is the live data.qpos[3:7]
    int32_t *nstep;      // [n]     substeps since the last reset (data.time of the step being observed)
    float *stack;        // [n][2 window][26]  ring of the last `window` frames, kept TWICE: slot s and slot s + window hold the same
                         // frame, so any `window` consecutive frames -- oldest first, as the observation wants them -- are one
                         // contiguous run starting at the oldest frame's slot, whatever the ring's rotation (round 3: the fused
                         // forms copy a row with plain 16-byte loads instead of wrapping every 8 bytes; one more frame written per step)
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
        const float *__restrict__ st = S.stack + 2 * row;                  // the ring's rows are two windows long (second copy behind the first)
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
            float *ring = S.stack + 2 * row + (size_t)slot * QG_PO_FRAME;
            for (int i = l16; i < QG_PO_FRAME; i += 16) { const float x = s_new[le][i]; ring[i] = x; ring[width + i] = x; }
        }
    }
    __syncthreads();                                                       // every read of the old ring contents is done
    if (le < envs && s_fin[le]) {                                          // restart the FIFO from the reset frame
        float *st = S.stack + 2 * (size_t)(env0 + le) * width;
        for (int r = l16; r < 2 * width; r += 16) st[r] = s_rst[le][r % QG_PO_FRAME];
    }
}

// 8- and 16-byte accesses of the copy: vector types (a struct cannot be assigned across address spaces), declared with the alignment
// the data really has -- rows are 8-byte aligned within their buffers, the caller's `out` is only promised to be 4-byte aligned
typedef float PoF2 __attribute__((ext_vector_type(2), aligned(4)));
typedef float PoF4 __attribute__((ext_vector_type(4), aligned(4)));
// ---- the fused form (qg_step_kernel_link<WALK, PO>): the same output in two parts -------------------------------------------------
// Part 1, in the step kernel's PROLOGUE: the W - 1 frames the new stack keeps are known before the physics runs, so they are copied
// ring -> out there, their loads in flight together with the state loads (a wave that is alone on its SIMD would otherwise sit
// through their latency in the epilogue, behind the stores of the step).  `slot` = ring slot the NEW frame will take.  With the ring
// kept twice (KPoState.stack) the new row is ONE contiguous run of it, from the oldest frame that stays: the env's 16 lanes copy it in
// 16-byte groups, group g to lane g mod 16 -- five loads and five stores per lane at window 10 (round 2: sixteen 4-byte ones each,
// with a wrap test per element; the 65th group of window 10 cost a second, one-lane pass until the batch became five).  The row is copied WHOLE: its last 26 values receive the frame that is about to be dropped and are
// overwritten by the new frame in the epilogue (po_emit_new: same wave, program order).
// The first 80 groups (five per lane: a whole row up to window 12) are LOADED in the prologue and STORED after the substep loop
// (po_copy_history_store), so that the copy's memory round trip runs under the physics instead of in front of it; longer rows copy
// the rest on the spot.
#define QG_PO_RING_SLACK 256   // bytes allocated behind the frame ring (see the static_assert next to QG_PO_COPY_K, qg_kernels.hip)
#define QG_PO_HIST_K 5
struct PoHistRegs { PoF4 v[QG_PO_HIST_K]; };
__device__ __forceinline__ const char *po_hist_src(const KPoParams &P, const KPoState &S, size_t row, int slot) {
    const int first = slot + 1 >= P.window ? 0 : slot + 1;               // the oldest frame that stays
    return reinterpret_cast<const char *>(S.stack + 2 * row) + first * (QG_PO_FRAME * 4);
}
__device__ __forceinline__ void po_copy_history_load(const KPoParams &P, const KPoState &S, size_t row, int slot, int l16,
                                                     float *__restrict__ out, bool live, PoHistRegs &H) {
    // Lanes past the end repeat the row's last group (same value to the same address: harmless), so that no load is predicated; the
    // lanes of a wave's tail (no env of their own, `live` false) shadow the last env's loads and store nothing -- a whole env is live
    // or not, so the predicate costs no divergence inside an env, and nothing then orders a shadow copy against the epilogue of the
    // wave that owns the env.
    const int wbytes = P.window * QG_PO_FRAME * 4;
    const char *src = po_hist_src(P, S, row, slot);
    char *dst = reinterpret_cast<char *>(out + row);
    const int groups = wbytes >> 4;
#pragma unroll
    for (int u = 0; u < QG_PO_HIST_K; ++u) H.v[u] = *reinterpret_cast<const PoF4 *>(src + 16 * min(l16 + 16 * u, groups - 1));
    for (int g0 = l16 + 16 * QG_PO_HIST_K; g0 < groups; g0 += 16 * QG_PO_HIST_K) {      // windows past 12: the rest, now
        PoF4 v[QG_PO_HIST_K];
        int gg[QG_PO_HIST_K];
#pragma unroll
        for (int u = 0; u < QG_PO_HIST_K; ++u) {
            gg[u] = min(g0 + 16 * u, groups - 1);
            v[u] = *reinterpret_cast<const PoF4 *>(src + 16 * gg[u]);
        }
#pragma unroll
        for (int u = 0; u < QG_PO_HIST_K; ++u) if (live) *reinterpret_cast<PoF4 *>(dst + 16 * gg[u]) = v[u];
    }
    if ((wbytes & 8) && l16 == 15 && live)                                  // odd windows: the row's last 8 bytes
        *reinterpret_cast<PoF2 *>(dst + wbytes - 8) = *reinterpret_cast<const PoF2 *>(src + wbytes - 8);
}
// The whole copy at once, LPE lanes per env (the helper waves of qg_step_kernel_quad<.., PO, HELP>: four lanes per env, batches of
// four 16-byte groups per lane; they have the whole physics to finish in)
template <int LPE>
__device__ __forceinline__ void po_copy_history_now(const KPoParams &P, const KPoState &S, size_t row, int slot, int j, float *__restrict__ out, bool live) {
    const int wbytes = P.window * QG_PO_FRAME * 4;
    const char *src = po_hist_src(P, S, row, slot);
    char *dst = reinterpret_cast<char *>(out + row);
    const int groups = wbytes >> 4;
#pragma unroll 1
    for (int g0 = j; g0 < groups; g0 += 4 * LPE) {
        PoF4 v[4];
        int gg[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            gg[u] = min(g0 + LPE * u, groups - 1);
            v[u] = *reinterpret_cast<const PoF4 *>(src + 16 * gg[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (live) *reinterpret_cast<PoF4 *>(dst + 16 * gg[u]) = v[u];
    }
    if ((wbytes & 8) && j == LPE - 1 && live)                               // odd windows: the row's last 8 bytes
        *reinterpret_cast<PoF2 *>(dst + wbytes - 8) = *reinterpret_cast<const PoF2 *>(src + wbytes - 8);
}
// after the substep loop and BEFORE po_emit_new (same wave, program order: the new frame then lands on the row's tail, and an env
// that finished overwrites the whole row)
__device__ __forceinline__ void po_copy_history_store(const KPoParams &P, size_t row, int l16, float *__restrict__ out, bool live,
                                                      const PoHistRegs &H) {
    char *dst = reinterpret_cast<char *>(out + row);
    const int groups = (P.window * QG_PO_FRAME * 4) >> 4;
#pragma unroll
    for (int u = 0; u < QG_PO_HIST_K; ++u) if (live) *reinterpret_cast<PoF4 *>(dst + 16 * min(l16 + 16 * u, groups - 1)) = H.v[u];
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
        float *__restrict__ stw = S.stack + 2 * row;                       // rows of the ring are two windows long
        const int slot = s_slot[le];
        if (!s_fin[le]) {
            for (int i = l16; i < QG_PO_FRAME; i += 16) {
                const float x = s_new[le][i];
                o[width - QG_PO_FRAME + i] = x;
                stw[slot * QG_PO_FRAME + i] = x;
                stw[width + slot * QG_PO_FRAME + i] = x;
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
        float *st = S.stack + 2 * (size_t)(env0 + le) * width;
        for (int r = l16; r < 2 * width; r += 16) st[r] = s_rst[le][r % QG_PO_FRAME];
    }
}


// ---- the fused form for the step kernels whose WAVE owns a block of consecutive envs (one leg per lane: 16 envs x 4 lanes, two legs
// per lane: 32 envs x 2 lanes; round 3).  At the batch sizes these kernels serve the observation pack is HBM traffic -- per env and
// step ~1 KB of ring read and ~1 KB of row written at window 10, 65 MB per launch at 32 768 envs, 15 us as a kernel of its own at
// the ~4.5 TB/s it reached -- next to a physics launch that leaves the memory system idle.  So the copy ring -> out is spread over
// the SUBSTEP LOOP: at the head of a substep a lane loads its next K 16-byte groups, at its tail (2 500 instructions later: the
// latency has long passed) it stores them.  The LPE lanes of an env copy THEIR env's row, group g of the row going to lane g mod LPE.
// With the ring kept twice (KPoState.stack) the row of the new stack is ONE contiguous run of the ring, from the oldest frame that
// stays: source and destination of a lane's groups are a base each plus compile-time offsets, no wrap, no table.  The row is copied
// WHOLE: its last 26 values receive the frame that is about to be dropped and are overwritten by the new frame in the epilogue (same
// wave, program order).  What does not fit into frame_skip x K groups per lane is copied after the loop; the one group (and, for
// odd windows, the 8-byte half) that only some of the env's lanes have travels with the first substep's batch.
// a pointer every lane of the wave agrees on, into scalar registers -- and typed as GLOBAL memory: after the round trip through
// integers the compiler no longer knows the address space and would fall back to flat_load / flat_store with 64-bit vector addresses;
// with it, accesses take the form  global_load v, v_offset32, s[base] offset:imm
typedef const char __attribute__((address_space(1))) *po_gcptr;
typedef char __attribute__((address_space(1))) *po_gptr;
typedef const PoF2 __attribute__((address_space(1))) *po_gcf2;
typedef PoF2 __attribute__((address_space(1))) *po_gf2;
typedef const PoF4 __attribute__((address_space(1))) *po_gcf4;
typedef PoF4 __attribute__((address_space(1))) *po_gf4;
__device__ __forceinline__ int po_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ unsigned long long po_uniform_addr(const void *p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// this lane's part of its env's row: byte offsets of its next group within the wave's block of the ring / of `out`; `left` = groups
// every lane of the env still has (wave-uniform); row_src / row_dst: the row's start; xa / xh: the leftovers (below)
struct PoCopyState { unsigned src, dst, row_src, row_dst; int left; };
template <int K> struct PoCopyRegs { PoF4 v[K]; PoF4 xa; PoF2 xh; };

// el: the env's row within the wave's block (the tail lanes of a ragged last wave pass the last live row: they load, never store);
// head: ring slot of the env's newest frame; j: this lane's index among the env's LPE lanes
template <int LPE>
__device__ __forceinline__ void po_row_copy_init(const KPoParams &P, int el, int head, int j, PoCopyState &st) {
    const int wbytes = P.window * QG_PO_FRAME * 4;
    int slot = head + 1;                           // the slot the new frame will take (the oldest frame's)
    if (slot >= P.window) slot = 0;
    int first = slot + 1;                          // the oldest frame that stays
    if (first >= P.window) first = 0;
    st.row_src = (unsigned)(el * 2 * wbytes + first * QG_PO_FRAME * 4);
    st.row_dst = (unsigned)(el * wbytes);
    st.src = st.row_src + 16u * j;
    st.dst = st.row_dst + 16u * j;
    st.left = P.window > 1 ? po_uniform(((wbytes >> 4) / LPE)) : 0;        // 16-byte groups of a row that every lane of the env has
}
// loads of this lane's next K groups (unpredicated within a batch: the last batch of a row may read up to K - 1 groups past it, into
// the ring's second copy, the next row or the allocation's slack -- never stored).  `first`: the substep loop's first batch also
// fetches the LEFTOVERS -- the one group that only some of the env's lanes have (the row's group count is no multiple of LPE) and,
// for odd windows, the row's last 8 bytes -- so that they too spend a substep in flight instead of a round trip after the loop
// (fetched in the prologue they cost more: their address depends on the ring position, a load of its own, and what the compiler then
// parks in AGPRs has to have arrived first: +2.8 us of prologue at 32 768 envs).
template <int K, int LPE>
__device__ __forceinline__ void po_row_copy_load(const KPoParams &P, unsigned long long ring_block, int j, bool first, PoCopyState &st, PoCopyRegs<K> &R) {
    const po_gcptr base = (po_gcptr)ring_block;
    if (st.left > 0) {                             // wave-uniform: once the row is done the substeps that follow copy nothing
#pragma unroll
        for (int u = 0; u < K; ++u) R.v[u] = *(po_gcf4)(base + (st.src + 16u * LPE * u));
        st.src += 16u * LPE * K;
    }
    if (first && P.window > 1) {                   // wave-uniform
        const int wbytes = P.window * QG_PO_FRAME * 4;
        const int groups = wbytes >> 4, g = (groups / LPE) * LPE + j;
        R.xa = *(po_gcf4)(base + (st.row_src + 16u * (unsigned)(g < groups ? g : 0)));
        R.xh = *(po_gcf2)(base + (st.row_src + (unsigned)wbytes - 8u));
    }
}
template <int K, int LPE>
__device__ __forceinline__ void po_row_copy_store(const KPoParams &P, unsigned long long out_block, bool live, int j, bool first, PoCopyState &st,
                                                  const PoCopyRegs<K> &R) {
    const po_gptr base = (po_gptr)out_block;
    if (live) {
#pragma unroll
        for (int u = 0; u < K; ++u) {
            if (u < st.left) *(po_gf4)(base + (st.dst + 16u * LPE * u)) = R.v[u];       // wave-uniform condition: a scalar branch
        }
        if (first && P.window > 1) {
            const int wbytes = P.window * QG_PO_FRAME * 4;
            const int groups = wbytes >> 4, g = (groups / LPE) * LPE + j;
            if (g < groups) *(po_gf4)(base + (st.row_dst + 16u * g)) = R.xa;
            if ((wbytes & 8) && j == LPE - 1) *(po_gf2)(base + (st.row_dst + (unsigned)wbytes - 8u)) = R.xh;
        }
    }
    st.dst += 16u * LPE * K;
    st.left = st.left > K ? st.left - K : 0;
}
// what the substep loop did not get to (long windows, small frame_skip): the same groups, load -> store
template <int K, int LPE>
__device__ __forceinline__ void po_row_copy_rest(const KPoParams &P, unsigned long long ring_block, unsigned long long out_block, bool live, int j,
                                                 PoCopyState &st) {
    while (st.left > 0) {
        PoCopyRegs<K> R;
        po_row_copy_load<K, LPE>(P, ring_block, j, false, st, R);
        po_row_copy_store<K, LPE>(P, out_block, live, j, false, st, R);
    }
}
// Epilogue of the wave-level fused forms, after every lead lane has built its env's frame (po_frame_env) into s_new / s_rst /
// s_slot / s_fin and a wave-level fence: the new frames go into the last 26 values of the rows and into their ring slots (both
// copies); an env that finished (rare) hands out the terminal stack, shows the reset stack and restarts its FIFO from the reset
// frame.  The LPE lanes of an env write their env's frame in 8-byte pieces (the frame is 13 of them); el = this lane's env within
// the wave, j = its index among the env's lanes.
template <int LPE>
__device__ __forceinline__ void po_wave_emit(const KPoParams &P, const KPoState &S, int env0, int live_envs, int lane, int el, int j,
                                             const float (*s_new)[QG_PO_FRAME], const float (*s_rst)[QG_PO_FRAME], const int *s_slot,
                                             const int *s_fin, float *__restrict__ out, float *__restrict__ term_out) {
    const int W = P.window, width = W * QG_PO_FRAME, hist = width - QG_PO_FRAME;
    const size_t block = (size_t)env0 * width;
    if (el < live_envs && !s_fin[el]) {
        const size_t row = block + (size_t)el * width;
        float *__restrict__ o = out + row + hist;
        float *__restrict__ r0 = S.stack + 2 * row + s_slot[el] * QG_PO_FRAME;
        const float *fr = s_new[el];
#pragma unroll
        for (int t = 0; t < (QG_PO_FRAME / 2 + LPE - 1) / LPE; ++t) {
            const int pr = j + LPE * t;                                    // 8-byte piece of the frame
            if (pr < QG_PO_FRAME / 2) {
                PoF2 v;
                v.x = fr[2 * pr]; v.y = fr[2 * pr + 1];
                *reinterpret_cast<PoF2 *>(o + 2 * pr) = v;
                *reinterpret_cast<PoF2 *>(r0 + 2 * pr) = v;
                *reinterpret_cast<PoF2 *>(r0 + width + 2 * pr) = v;
            }
        }
    }
    unsigned long long fins = __ballot(lane < live_envs && s_fin[lane < live_envs ? lane : 0] != 0);
    if (fins == 0) return;                                                 // the usual case
    for (unsigned long long m = fins; m; m &= m - 1) {
        const int fe = __ffsll((long long)m) - 1;
        const size_t row = block + (size_t)fe * width;
        const int slot = s_slot[fe];
        for (int r = lane; r < width; r += 64) {
            const int f = r / QG_PO_FRAME, i = r - f * QG_PO_FRAME;
            int src = slot + 1 + f;                                        // oldest frame first
            if (src >= W) src -= W;
            const float x = (f == W - 1) ? s_new[fe][i] : S.stack[2 * row + src * QG_PO_FRAME + i];
            if (term_out) term_out[row + r] = x;
            out[row + r] = s_rst[fe][i];
        }
    }
    wave_sync();                                                           // every read of the old ring contents is done
    for (unsigned long long m = fins; m; m &= m - 1) {
        const int fe = __ffsll((long long)m) - 1;
        const size_t row = block + (size_t)fe * width;
        for (int r = lane; r < 2 * width; r += 64) S.stack[2 * row + r] = s_rst[fe][r % QG_PO_FRAME];
    }
}
