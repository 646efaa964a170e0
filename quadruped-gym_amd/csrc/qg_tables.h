// qg_tables.h -- host-side conversion of the double-precision qg_model / qg_task (include/quadgym.h)
// into the single-precision tables the kernels read (qg_device.h).  Pure C++: shared by the C ABI
// (qg_capi.hip) and by the generator of the baked-in default table (gen_baked.cpp).
#pragma once
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/quadgym.h"
#include "qg_device.h"

int qg_fail(int code, const char *fmt, ...);   // records the message for qg_last_error()
#define fail qg_fail

static inline int64_t qg_time_limit_substeps_impl(double timestep, double max_time) {
    if (!(timestep > 0)) return -1;
    if (!(max_time / timestep < 2.0e9)) return INT32_MAX;    // beyond the int32 substep counter: never reached
    double t = 0;
    int64_t n = 0;
    while (!(t >= max_time)) {     // quadruped.py:151 with the engine's `time += timestep` (f64)
        t += timestep;
        n++;
        if (n >= INT32_MAX) break;
    }
    return n;
}

static inline void quat_to_rowmajor(const double q[4], float R[9]) {
    double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    double w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
    R[0] = (float)(1 - 2 * (y * y + z * z)); R[1] = (float)(2 * (x * y - w * z));     R[2] = (float)(2 * (x * z + w * y));
    R[3] = (float)(2 * (x * y + w * z));     R[4] = (float)(1 - 2 * (x * x + z * z)); R[5] = (float)(2 * (y * z - w * x));
    R[6] = (float)(2 * (x * z - w * y));     R[7] = (float)(2 * (y * z + w * x));     R[8] = (float)(1 - 2 * (x * x + y * y));
    for (int i = 0; i < 9; i++) {            // axis-aligned mounts (quadruped.xml:71-141): exact 0 / +-1, no round-off dust
        if (std::fabs(R[i]) < 1e-7f) R[i] = 0.0f;
        if (std::fabs(R[i] - 1.0f) < 1e-7f) R[i] = 1.0f;
        if (std::fabs(R[i] + 1.0f) < 1e-7f) R[i] = -1.0f;
    }
}

static inline int build_tables(const qg_model *m, const qg_task *t, KModel *km, KTask *kt) {
    if (!(m->timestep > 0)) return fail(QG_ERR_ARG, "model.timestep must be positive");
    if (!(m->limit_ramp > 0) || !(m->contact_ramp > 0)) return fail(QG_ERR_ARG, "model.limit_ramp and model.contact_ramp must be positive");
    if (m->body_parent[0] != -1) return fail(QG_ERR_ARG, "body 0 must be the free-floating FRAME");
    if (m->ncp[0] != QGK_CP_FRAME) return fail(QG_ERR_ARG, "FRAME must carry %d contact points", QGK_CP_FRAME);
    for (int b = 1; b < QG_NBODY; b++) {
        int expect = ((b - 1) % 3 == 0) ? 0 : b - 1;
        if (m->body_parent[b] != expect) return fail(QG_ERR_ARG, "body %d: parent %d, expected %d (4 chains of 3 links)", b, m->body_parent[b], expect);
        if (m->ncp[b] != QGK_CP_LINK) return fail(QG_ERR_ARG, "body %d must carry %d contact points", b, QGK_CP_LINK);
        const double *ax = m->jnt_axis[b - 1];
        double an = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
        if (!(std::fabs(ax[0]) < 1e-12 * an && std::fabs(ax[1]) < 1e-12 * an && ax[2] > 0))
            return fail(QG_ERR_ARG, "joint %d: the kernels assume the hinge axis is the link's +z (quadruped.xml:9)", b - 1);
    }
    if (t->frame_skip < 1) return fail(QG_ERR_ARG, "task.frame_skip must be >= 1");
    if (!(t->reset_joint_jitter >= 0) || !(t->reset_joint_jitter < 10)) return fail(QG_ERR_ARG, "task.reset_joint_jitter must be in [0, 10) rad");
    if (t->obs_mode != QG_OBS_FULL && t->obs_mode != QG_OBS_IMU) return fail(QG_ERR_ARG, "task.obs_mode invalid");

    memset(km, 0, sizeof *km);
    km->h = (float)m->timestep;
    for (int i = 0; i < 3; i++) km->g[i] = (float)m->gravity[i];
    // FRAME: rigid inertia about its own origin
    {
        double ms = m->body_mass[0];
        const double *c = m->body_ipos[0], *I = m->body_inertia[0];
        double cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
        km->m0 = (float)ms;
        for (int i = 0; i < 3; i++) km->h0[i] = (float)(ms * c[i]);
        km->I0[0] = (float)(I[0] + ms * (cc - c[0] * c[0]));
        km->I0[1] = (float)(I[1] + ms * (cc - c[1] * c[1]));
        km->I0[2] = (float)(I[2] + ms * (cc - c[2] * c[2]));
        km->I0[3] = (float)(I[3] - ms * c[0] * c[1]);
        km->I0[4] = (float)(I[4] - ms * c[0] * c[2]);
        km->I0[5] = (float)(I[5] - ms * c[1] * c[2]);
        for (int i = 0; i < QGK_CP_FRAME; i++)
            for (int d = 0; d < 3; d++) km->cp0[i][d] = (float)m->cp[0][i][d];
    }
    km->free_damping = (float)m->free_damping;
    km->free_armature = (float)m->free_armature;
    km->contact_k = (float)m->contact_stiffness;
    km->contact_c = (float)m->contact_damping;
    km->contact_margin = (float)m->contact_margin;
    km->contact_mu = (float)m->contact_friction;
    km->limit_k = (float)m->limit_stiffness;
    km->limit_b = (float)m->limit_damping;
    km->limit_inv_ramp = (float)(1.0 / m->limit_ramp);
    km->contact_inv_ramp = (float)(1.0 / m->contact_ramp);
    for (int i = 0; i < QG_NQ; i++) km->qpos0[i] = (float)m->qpos0[i];
    for (int j = 0; j < QGK_NLINK; j++) {
        KLink &L = km->link[j];
        int b = j + 1;
        for (int d = 0; d < 3; d++) { L.pos[d] = (float)m->body_pos[b][d]; L.ipos[d] = (float)m->body_ipos[b][d]; }
        quat_to_rowmajor(m->body_quat[b], L.Q);
        L.mass = (float)m->body_mass[b];
        for (int d = 0; d < 6; d++) L.inertia[d] = (float)m->body_inertia[b][d];
        for (int i = 0; i < QGK_CP_LINK; i++)
            for (int d = 0; d < 3; d++) L.cp[i][d] = (float)m->cp[b][i][d];
        L.ref = (float)m->jnt_ref[j];
        L.lo = (float)m->jnt_range[j][0];
        L.hi = (float)m->jnt_range[j][1];
        L.damping = (float)m->jnt_damping[j];
        L.armature = (float)m->jnt_armature[j];
        L.kp = (float)m->act_kp[j];
        L.kv = (float)m->act_kv[j];
        L.gear = (float)m->act_gear[j];
        L.ctrl_lo = (float)m->act_ctrlrange[j][0];
        L.ctrl_hi = (float)m->act_ctrlrange[j][1];
        L.force_lo = (float)m->act_forcerange[j][0];
        L.force_hi = (float)m->act_forcerange[j][1];
        double tau = m->act_timeconst[j];
        L.act_decay = (float)(tau > 0 ? 1.0 - std::exp(-m->timestep / tau) : 1.0);   // filterexact
    }

    memset(kt, 0, sizeof *kt);
    kt->frame_skip = t->frame_skip;
    int64_t lim = t->use_time_limit ? qg_time_limit_substeps_impl(m->timestep, t->max_time) : (int64_t)INT32_MAX;
    kt->limit_substeps = (int32_t)(lim > INT32_MAX ? INT32_MAX : lim);
    kt->use_fall = t->use_fall;
    kt->fall_height = (float)t->fall_height;
    kt->use_flip = t->use_flip;
    kt->w_forward = (float)t->w_forward;
    kt->w_ctrl = (float)t->w_ctrl;
    kt->alive_bonus = (float)t->alive_bonus;
    kt->obs_mode = t->obs_mode;
    kt->sensor_lag = t->sensor_lag;
    kt->auto_reset = t->auto_reset;
    kt->reset_flags = t->reset_flags;
    for (int i = 0; i < QG_NU; i++) kt->default_ctrl[i] = (float)t->default_ctrl[i];
    kt->reset_joint_jitter = (float)t->reset_joint_jitter;
    return QG_OK;
}

#undef fail
