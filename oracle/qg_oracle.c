/* qg_oracle.c -- CPU restatement (double precision) of the hot path
 * QuadrupedEnv.step()/reset() of antopio26/quadruped-gym.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (quadruped-gym_amd/,
 * libquadgym.so) may link, import or call this file; only tests/, the smoke
 * check in __graft_entry__.py and the cpu_baseline leg of bench.py use it, as
 * the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED for the physics: the arithmetic of the path lives in the
 * third-party `mujoco` package (requirements.txt:1, unpinned, not vendored, not
 * installable offline) and the reference holds no tests, golden vectors or
 * fixtures for it.  This file restates
 *   - src/envs/quadruped.py:115-139  reset   (qgo_reset)
 *   - src/envs/quadruped.py:153-182  step    (qgo_step)
 *   - src/envs/quadruped.py:165      mj_step with integrator=implicitfast
 *                                    (qgo_substep; semantics per SURVEY.md
 *                                    Appendix A items 1-8,10-12)
 *   - src/envs/quadruped.py:141-143 + quadruped.xml:174-217 sensor pack
 *   - README.md:64-90 reward / termination set
 * from the published rigid-body algorithms (composite-rigid-body mass matrix,
 * recursive Newton-Euler bias, linearly-implicit velocity integration), and it
 * is pinned by the known-answer tests in tests/test_oracle_physics.py (energy
 * and momentum conservation, M*a + c == RNE(q, v, a), servo filter closed form,
 * 4-fold symmetry, exact discrete free fall) and, loosely, by the one recorded
 * output of the real engine the reference holds: the joint-angle plot stored in
 * src/quadruped_model.ipynb, whose envelope and slew rates the same protocol run
 * through this file reproduces (tools/digitize_notebook_plot.py).  Constraints (ground contact and
 * joint limits) are this project's own LCP-free penalty model, not
 * a restatement of the engine's convex solver (Appendix A.5, A.9).
 *
 * Formulation: every spatial vector is expressed in WORLD coordinates about the
 * WORLD origin -- a textbook generic-tree formulation with a dense 18x18
 * Cholesky solve, on purpose different from the HIP kernel (base-frame,
 * leg-structured, sparse elimination) so the two check each other.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../include/quadgym.h"
#include "../include/qg_model_data.h"

#define NB QG_NBODY
#define NV QG_NV
#define NQ QG_NQ
#define NU QG_NU

typedef struct qgo_env {
    double qpos[NQ];
    double qvel[NV];
    double act[NU];
    double ctrl[NU];
    int32_t nstep;
} qgo_env;

/* intermediate results of one substep, for the known-answer tests */
typedef struct qgo_diag {
    double M[NV * NV];      /* CRBA + armature */
    double A[NV * NV];      /* M + h*D (what is factorised) */
    double bias[NV];        /* RNE(q, v, 0) incl. gravity */
    double f_passive[NV];
    double f_act[NV];
    double f_limit[NV];
    double f_contact[NV];
    double qacc[NV];
    double act_force[NU];   /* scalar servo force after the forcerange clamp */
    double contact_W[NB];   /* summed spring force per body */
    double contact_F[NB][3];/* contact force per body, world */
    double contact_P[NB][3];/* centre of pressure, world */
} qgo_diag;

/* ------------------------------------------------------------------ vec3 */
static void cross3(const double a[3], const double b[3], double o[3]) {
    double x = a[1] * b[2] - a[2] * b[1];
    double y = a[2] * b[0] - a[0] * b[2];
    double z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void matvec3(const double R[9], const double v[3], double o[3]) {
    double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
    double y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
    double z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static void quat_mul(const double a[4], const double b[4], double o[4]) {
    double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
static void quat_normalize(double q[4]) {
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n < 1e-300) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static void quat_to_mat(const double q[4], double R[9]) {
    double w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}

/* ------------------------------------------------------------ kinematics */
typedef struct kin {
    double xpos[NB][3];   /* body frame origin, world */
    double xquat[NB][4];
    double R[NB][9];      /* body -> world */
    double com[NB][3];    /* centre of mass, world */
    double Iw[NB][9];     /* inertia about the COM, world axes */
    /* spatial inertia about the world origin: mass, h = m*com, Ibar_O */
    double h[NB][3];
    double IO[NB][9];
    /* motion subspace of each DoF, world coords about the world origin: [ang; lin] */
    double S[NV][6];
    int dof_body[NV];
} kin;

static void kinematics(const qg_model *m, const double *qpos, kin *k) {
    for (int b = 0; b < NB; b++) {
        if (b == 0) {
            for (int i = 0; i < 3; i++) k->xpos[0][i] = qpos[i];
            for (int i = 0; i < 4; i++) k->xquat[0][i] = qpos[3 + i];
            quat_normalize(k->xquat[0]);      /* the engine normalises the free-joint quaternion */
        } else {
            int p = m->body_parent[b];
            int j = b - 1;
            double off[3];
            matvec3(k->R[p], m->body_pos[b], off);
            for (int i = 0; i < 3; i++) k->xpos[b][i] = k->xpos[p][i] + off[i];
            double q1[4];
            quat_mul(k->xquat[p], m->body_quat[b], q1);
            /* hinge: rotation about the joint axis by (qpos - ref), Appendix A.4 */
            double ang = qpos[7 + j] - m->jnt_ref[j];
            const double *ax = m->jnt_axis[j];
            double an = sqrt(dot3(ax, ax));
            double s = sin(0.5 * ang) / an;
            double qj[4] = {cos(0.5 * ang), s * ax[0], s * ax[1], s * ax[2]};
            quat_mul(q1, qj, k->xquat[b]);
            quat_normalize(k->xquat[b]);
        }
        quat_to_mat(k->xquat[b], k->R[b]);
        double c[3];
        matvec3(k->R[b], m->body_ipos[b], c);
        for (int i = 0; i < 3; i++) k->com[b][i] = k->xpos[b][i] + c[i];
        /* Iw = R I R^T */
        const double *I6 = m->body_inertia[b];
        double Ib[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
        double T[9];
        for (int r = 0; r < 3; r++)
            for (int c2 = 0; c2 < 3; c2++) {
                double s = 0;
                for (int t = 0; t < 3; t++) s += k->R[b][3 * r + t] * Ib[3 * t + c2];
                T[3 * r + c2] = s;
            }
        for (int r = 0; r < 3; r++)
            for (int c2 = 0; c2 < 3; c2++) {
                double s = 0;
                for (int t = 0; t < 3; t++) s += T[3 * r + t] * k->R[b][3 * c2 + t];
                k->Iw[b][3 * r + c2] = s;
            }
        double mass = m->body_mass[b];
        const double *cm = k->com[b];
        double cc = dot3(cm, cm);
        for (int i = 0; i < 3; i++) k->h[b][i] = mass * cm[i];
        for (int r = 0; r < 3; r++)
            for (int c2 = 0; c2 < 3; c2++)
                k->IO[b][3 * r + c2] = k->Iw[b][3 * r + c2] + mass * ((r == c2 ? cc : 0.0) - cm[r] * cm[c2]);
    }
    /* motion subspaces.  Free joint (Appendix A.3): DoF 0-2 translate along the
     * world axes, DoF 3-5 rotate about the BODY axes through the body origin. */
    for (int d = 0; d < 3; d++) {
        for (int i = 0; i < 6; i++) k->S[d][i] = 0;
        k->S[d][3 + d] = 1.0;
        k->dof_body[d] = 0;
    }
    for (int d = 0; d < 3; d++) {
        double a[3] = {k->R[0][d], k->R[0][3 + d], k->R[0][6 + d]};
        double l[3];
        cross3(k->xpos[0], a, l);
        for (int i = 0; i < 3; i++) { k->S[3 + d][i] = a[i]; k->S[3 + d][3 + i] = l[i]; }
        k->dof_body[3 + d] = 0;
    }
    for (int j = 0; j < QG_NJNT; j++) {
        int b = j + 1;
        double an = sqrt(dot3(m->jnt_axis[j], m->jnt_axis[j]));
        double al[3] = {m->jnt_axis[j][0] / an, m->jnt_axis[j][1] / an, m->jnt_axis[j][2] / an};
        double a[3], l[3];
        matvec3(k->R[b], al, a);
        cross3(k->xpos[b], a, l);
        for (int i = 0; i < 3; i++) { k->S[6 + j][i] = a[i]; k->S[6 + j][3 + i] = l[i]; }
        k->dof_body[6 + j] = b;
    }
}

/* spatial inertia (mass, h, IO) times motion vector [w; v] -> force [n; f] */
static void inertia_mul(double mass, const double h[3], const double IO[9], const double mv[6], double out[6]) {
    double n[3], t[3];
    matvec3(IO, mv, n);
    cross3(h, mv + 3, t);
    for (int i = 0; i < 3; i++) out[i] = n[i] + t[i];
    cross3(h, mv, t);
    for (int i = 0; i < 3; i++) out[3 + i] = mass * mv[3 + i] - t[i];
}
/* motion cross product  crm(v) m */
static void crm(const double v[6], const double mv[6], double out[6]) {
    double a[3], b[3], c[3];
    cross3(v, mv, a);
    cross3(v, mv + 3, b);
    cross3(v + 3, mv, c);
    for (int i = 0; i < 3; i++) { out[i] = a[i]; out[3 + i] = b[i] + c[i]; }
}
/* force cross product  crf(v) f */
static void crf(const double v[6], const double f[6], double out[6]) {
    double a[3], b[3], c[3];
    cross3(v, f, a);
    cross3(v + 3, f + 3, b);
    cross3(v, f + 3, c);
    for (int i = 0; i < 3; i++) { out[i] = a[i] + b[i]; out[3 + i] = c[i]; }
}
static double dot6(const double a[6], const double b[6]) {
    double s = 0;
    for (int i = 0; i < 6; i++) s += a[i] * b[i];
    return s;
}
/* is body `anc` an ancestor of (or equal to) body b */
static int supports(const qg_model *m, int anc, int b) {
    while (b >= 0) {
        if (b == anc) return 1;
        b = m->body_parent[b];
    }
    return 0;
}

/* ---------------------------------------------------- composite rigid body */
static void crba(const qg_model *m, const kin *k, double *M) {
    double cm[NB], ch[NB][3], cI[NB][9];
    for (int b = 0; b < NB; b++) {
        cm[b] = m->body_mass[b];
        memcpy(ch[b], k->h[b], sizeof ch[b]);
        memcpy(cI[b], k->IO[b], sizeof cI[b]);
    }
    for (int b = NB - 1; b > 0; b--) {
        int p = m->body_parent[b];
        cm[p] += cm[b];
        for (int i = 0; i < 3; i++) ch[p][i] += ch[b][i];
        for (int i = 0; i < 9; i++) cI[p][i] += cI[b][i];
    }
    memset(M, 0, sizeof(double) * NV * NV);
    for (int i = 0; i < NV; i++) {
        int bi = k->dof_body[i];
        double F[6];
        inertia_mul(cm[bi], ch[bi], cI[bi], k->S[i], F);
        for (int j = 0; j < NV; j++) {
            int bj = k->dof_body[j];
            if (!supports(m, bj, bi)) continue;   /* DoF j must move body bi */
            double v = dot6(k->S[j], F);
            M[i * NV + j] = v;
            M[j * NV + i] = v;
        }
    }
    /* armature: rotor inertia on the diagonal (quadruped.xml:9; free joint inherits it) */
    for (int d = 0; d < 6; d++) M[d * NV + d] += m->free_armature;
    for (int j = 0; j < QG_NJNT; j++) M[(6 + j) * NV + 6 + j] += m->jnt_armature[j];
}

/* ------------------------------------------------ recursive Newton-Euler */
/* tau = RNE(q, qvel, qacc) with gravity (no armature, no passive forces) */
static void rne(const qg_model *m, const kin *k, const double *qvel, const double *qacc, double *tau) {
    double v[NB][6], a[NB][6], f[NB][6];
    double a0[6] = {0, 0, 0, -m->gravity[0], -m->gravity[1], -m->gravity[2]};
    for (int b = 0; b < NB; b++) {
        int p = m->body_parent[b];
        if (p < 0) {
            for (int i = 0; i < 6; i++) { v[b][i] = 0; a[b][i] = a0[i]; }
        } else {
            memcpy(v[b], v[p], sizeof v[b]);
            memcpy(a[b], a[p], sizeof a[b]);
        }
        int d0 = (b == 0) ? 0 : 5 + b, nd = (b == 0) ? 6 : 1;
        /* velocity first (all DoFs of the joint), then S-dot with the full body velocity */
        for (int d = d0; d < d0 + nd; d++)
            for (int i = 0; i < 6; i++) v[b][i] += k->S[d][i] * qvel[d];
        for (int d = d0; d < d0 + nd; d++) {
            double sd[6] = {0, 0, 0, 0, 0, 0};
            if (!(b == 0 && d < 3)) crm(v[b], k->S[d], sd);   /* world-fixed translation axes: S-dot = 0 */
            for (int i = 0; i < 6; i++) a[b][i] += k->S[d][i] * (qacc ? qacc[d] : 0.0) + sd[i] * qvel[d];
        }
        double Iv[6], Ia[6], t[6];
        inertia_mul(m->body_mass[b], k->h[b], k->IO[b], v[b], Iv);
        inertia_mul(m->body_mass[b], k->h[b], k->IO[b], a[b], Ia);
        crf(v[b], Iv, t);
        for (int i = 0; i < 6; i++) f[b][i] = Ia[i] + t[i];
    }
    for (int b = NB - 1; b >= 0; b--) {
        int d0 = (b == 0) ? 0 : 5 + b, nd = (b == 0) ? 6 : 1;
        for (int d = d0; d < d0 + nd; d++) tau[d] = dot6(k->S[d], f[b]);
        int p = m->body_parent[b];
        if (p >= 0)
            for (int i = 0; i < 6; i++) f[p][i] += f[b][i];
    }
}

/* dense Cholesky solve A x = b (A symmetric positive definite, n <= NV); returns 0 if ok */
static int chol_solve(const double *A, const double *b, double *x, int n) {
    double L[NV * NV];
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) {
            double s = A[i * n + j];
            for (int t = 0; t < j; t++) s -= L[i * n + t] * L[j * n + t];
            if (i == j) {
                if (!(s > 0)) return -1;
                L[i * n + i] = sqrt(s);
            } else
                L[i * n + j] = s / L[j * n + j];
        }
    double y[NV];
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int t = 0; t < i; t++) s -= L[i * n + t] * y[t];
        y[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = y[i];
        for (int t = i + 1; t < n; t++) s -= L[t * n + i] * x[t];
        x[i] = s / L[i * n + i];
    }
    return 0;
}

/* ------------------------------------------------------------ one mj_step */
/* Restates mujoco.mj_step(model, data) at quadruped.py:165 (Appendix A.1-A.3,
 * A.7): forward pass at (qpos, qvel, act) -> linearly-implicit velocity update
 * -> semi-implicit position update -> activation filter -> time.  `sens` (if
 * not NULL) receives the 33 sensor values of THIS forward pass, i.e. before
 * integrating -- the lag the reference observes because it never calls
 * mj_forward after mj_step. */
int qgo_substep(const qg_model *m, qgo_env *e, const double *ctrl, double *sens, qgo_diag *dg) {
    const double h = m->timestep;
    kin k;
    kinematics(m, e->qpos, &k);

    double M[NV * NV], A[NV * NV];
    crba(m, &k, M);
    memcpy(A, M, sizeof A);

    double bias[NV];
    rne(m, &k, e->qvel, NULL, bias);

    double rhs[NV], fpas[NV], fact[NV], flim[NV], fcon[NV];
    memset(fact, 0, sizeof fact); memset(flim, 0, sizeof flim); memset(fcon, 0, sizeof fcon);

    /* passive: viscous joint damping on all 18 DoFs (Appendix A.6) */
    for (int d = 0; d < NV; d++) {
        double dmp = d < 6 ? m->free_damping : m->jnt_damping[d - 6];
        fpas[d] = -dmp * e->qvel[d];
        A[d * NV + d] += h * dmp;
    }

    /* position servos (quadruped.xml:10-37; Appendix A.7).  Force uses the
     * PRE-update activation; the velocity derivative -kv*gear^2 enters the
     * implicit matrix unless the force sits on its forcerange clamp (A.2). */
    double uclamp[NU];
    for (int i = 0; i < NU; i++) {
        int d = 6 + i;
        double g = m->act_gear[i];
        double u = ctrl[i];
        if (u < m->act_ctrlrange[i][0]) u = m->act_ctrlrange[i][0];
        if (u > m->act_ctrlrange[i][1]) u = m->act_ctrlrange[i][1];
        uclamp[i] = u;
        double len = g * e->qpos[7 + i], vel = g * e->qvel[d];
        double force = m->act_kp[i] * e->act[i] - m->act_kp[i] * len - m->act_kv[i] * vel;
        int clamped = 0;
        if (force <= m->act_forcerange[i][0]) { force = m->act_forcerange[i][0]; clamped = 1; }
        if (force >= m->act_forcerange[i][1]) { force = m->act_forcerange[i][1]; clamped = 1; }
        fact[d] = g * force;
        if (!clamped) A[d * NV + d] += h * m->act_kv[i] * g * g;
        if (dg) dg->act_force[i] = force;
    }

    /* soft joint limits: one-sided penalty spring-damper, damping implicit */
    for (int j = 0; j < QG_NJNT; j++) {
        int d = 6 + j;
        double q = e->qpos[7 + j], qd = e->qvel[d];
        double lo = m->jnt_range[j][0], hi = m->jnt_range[j][1];
        double kl = m->limit_stiffness;
        double pen = q < lo ? lo - q : (q > hi ? q - hi : 0.0);
        /* the damper ramps in with the penetration so the torque is continuous at the limit */
        double ramp = pen / m->limit_ramp;
        double bl = m->limit_damping * (ramp < 1.0 ? ramp : 1.0);
        if (q < lo) {
            double spring = kl * pen;
            double t = spring - bl * qd;
            double beff = bl;
            if (t < 0) { t = 0; beff = spring / qd; }     /* qd > 0 here: leaving the limit fast */
            flim[d] = t;
            A[d * NV + d] += h * beff;
        } else if (q > hi) {
            double spring = kl * pen;
            double t = spring + bl * qd;
            double beff = bl;
            if (t < 0) { t = 0; beff = -spring / qd; }    /* qd < 0 */
            flim[d] = -t;
            A[d * NV + d] += h * beff;
        }
    }

    /* LCP-free soft ground contact (plane z = 0), one aggregated contact per body:
     *   W   = sum_i k * max(0, margin - z_i)          spring force
 *   c   = contact_damping * min(1, sum_i pen_i / contact_ramp)   damper ramps in with depth
     *   P   = centre of pressure of the spring forces
     *   F_n = max(0, W - c * v_n(P))                  no adhesion
     *   F_t = -min(c, mu F_n / |v_t|) * v_t(P)        viscous, Coulomb-limited
     * The damper is linear in velocity with the secant coefficients (c_n, c_t), and
     * h * J^T diag(c_t, c_t, c_n) J is added to the implicit matrix. */
    for (int b = 0; b < NB; b++) {
        double W = 0, s[3] = {0, 0, 0}, pensum = 0;
        for (int i = 0; i < m->ncp[b]; i++) {
            double r[3];
            matvec3(k.R[b], m->cp[b][i], r);
            double z = k.xpos[b][2] + r[2];
            double pen = m->contact_margin - z;
            if (pen > 0) {
                double w = m->contact_stiffness * pen;
                pensum += pen;
                W += w;
                for (int t = 0; t < 3; t++) s[t] += w * r[t];
            }
        }
        if (dg) { dg->contact_W[b] = W; for (int t = 0; t < 3; t++) { dg->contact_F[b][t] = 0; dg->contact_P[b][t] = 0; } }
        if (!(W > 0)) continue;
        double P[3];
        for (int t = 0; t < 3; t++) P[t] = k.xpos[b][t] + s[t] / W;
        /* Jacobian of the point P on body b */
        double J[3][NV];
        for (int d = 0; d < NV; d++) {
            if (supports(m, k.dof_body[d], b)) {
                double wxP[3];
                cross3(k.S[d], P, wxP);
                for (int t = 0; t < 3; t++) J[t][d] = k.S[d][3 + t] + wxP[t];
            } else
                for (int t = 0; t < 3; t++) J[t][d] = 0;
        }
        double vP[3] = {0, 0, 0};
        for (int d = 0; d < NV; d++)
            for (int t = 0; t < 3; t++) vP[t] += J[t][d] * e->qvel[d];
        double ramp = pensum / m->contact_ramp;
        double c = m->contact_damping * (ramp < 1.0 ? ramp : 1.0), mu = m->contact_friction;
        double cn = c, Fn = W - c * vP[2];
        if (Fn < 0) { Fn = 0; cn = W / vP[2]; }
        double speed = sqrt(vP[0] * vP[0] + vP[1] * vP[1]);
        double ct = c;
        if (c * speed > mu * Fn) ct = mu * Fn / speed;
        double F[3] = {-ct * vP[0], -ct * vP[1], Fn};
        for (int d = 0; d < NV; d++) fcon[d] += J[0][d] * F[0] + J[1][d] * F[1] + J[2][d] * F[2];
        for (int i = 0; i < NV; i++)
            for (int j = 0; j < NV; j++)
                A[i * NV + j] += h * (ct * (J[0][i] * J[0][j] + J[1][i] * J[1][j]) + cn * J[2][i] * J[2][j]);
        if (dg) for (int t = 0; t < 3; t++) { dg->contact_F[b][t] = F[t]; dg->contact_P[b][t] = P[t]; }
    }

    for (int d = 0; d < NV; d++) rhs[d] = fpas[d] + fact[d] + flim[d] + fcon[d] - bias[d];
    double qacc[NV];
    if (chol_solve(A, rhs, qacc, NV) != 0) return -1;

    /* sensors of this forward pass (quadruped.xml:174-217; DOCS.md:365-400) */
    if (sens) {
        const double *R = k.R[0];
        for (int j = 0; j < 12; j++) sens[j] = e->qpos[7 + j];                        /* jointpos */
        /* accelerometer: site-frame proper acceleration.  With the penalty contact the
         * acceleration realised by the implicit update is used. */
        double aw[3] = {qacc[0] - m->gravity[0], qacc[1] - m->gravity[1], qacc[2] - m->gravity[2]};
        for (int i = 0; i < 3; i++) sens[12 + i] = R[i] * aw[0] + R[3 + i] * aw[1] + R[6 + i] * aw[2];
        for (int i = 0; i < 3; i++) sens[15 + i] = e->qvel[3 + i];                    /* gyro: local omega */
        for (int i = 0; i < 3; i++) sens[18 + i] = k.xpos[0][i];                      /* framepos */
        for (int i = 0; i < 3; i++) sens[21 + i] = e->qvel[i];                        /* framelinvel (world) */
        for (int i = 0; i < 3; i++) sens[24 + i] = R[3 * i + 0];                      /* framexaxis */
        for (int i = 0; i < 3; i++) sens[27 + i] = R[3 * i + 2];                      /* framezaxis */
        for (int i = 0; i < 3; i++)                                                   /* velocimeter: R^T v */
            sens[30 + i] = R[i] * e->qvel[0] + R[3 + i] * e->qvel[1] + R[6 + i] * e->qvel[2];
    }
    if (dg) {
        memcpy(dg->M, M, sizeof M); memcpy(dg->A, A, sizeof A); memcpy(dg->bias, bias, sizeof bias);
        memcpy(dg->f_passive, fpas, sizeof fpas); memcpy(dg->f_act, fact, sizeof fact);
        memcpy(dg->f_limit, flim, sizeof flim); memcpy(dg->f_contact, fcon, sizeof fcon);
        memcpy(dg->qacc, qacc, sizeof qacc);
    }

    /* integrate (mj_advance): activations, velocity, then position with the NEW velocity */
    for (int i = 0; i < NU; i++) {
        double tau = m->act_timeconst[i];
        if (tau > 0)
            e->act[i] += (uclamp[i] - e->act[i]) * (1.0 - exp(-h / tau));   /* filterexact */
        else
            e->act[i] = uclamp[i];
    }
    for (int d = 0; d < NV; d++) e->qvel[d] += h * qacc[d];
    for (int i = 0; i < 3; i++) e->qpos[i] += h * e->qvel[i];
    {   /* quaternion: q <- q * exp(h * omega_local), normalised (Appendix A.3) */
        double w[3] = {e->qvel[3], e->qvel[4], e->qvel[5]};
        double n = sqrt(dot3(w, w));
        double q0[4] = {k.xquat[0][0], k.xquat[0][1], k.xquat[0][2], k.xquat[0][3]};
        if (n > 0) {
            double ang = h * n, s = sin(0.5 * ang) / n;
            double dq[4] = {cos(0.5 * ang), s * w[0], s * w[1], s * w[2]};
            double qn[4];
            quat_mul(q0, dq, qn);
            quat_normalize(qn);
            for (int i = 0; i < 4; i++) e->qpos[3 + i] = qn[i];
        } else
            for (int i = 0; i < 4; i++) e->qpos[3 + i] = q0[i];
    }
    for (int j = 0; j < QG_NJNT; j++) e->qpos[7 + j] += h * e->qvel[6 + j];
    for (int i = 0; i < NU; i++) e->ctrl[i] = ctrl[i];
    e->nstep += 1;
    return 0;
}

/* ------------------------------------------------------------ reset / step */
static uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
/* counter-based uniform in [0,1) with 24 random bits: exact in f32 and f64 */
double qgo_uniform(uint64_t seed, uint64_t env_index, uint64_t counter) {
    uint64_t x = seed + 0x9E3779B97F4A7C15ull * (env_index + 1) + 0xD1B54A32D192ED03ull * (counter + 1);
    x = mix64(mix64(x));
    return (double)(x >> 40) * (1.0 / 16777216.0);
}

/* independent streams of one (seed, env, episode) key: 0 = reset yaw, 1..12 = hinge jitter, 13..15 = walking command */
double qgo_uniform_stream(uint64_t seed, uint64_t env_index, uint64_t counter, uint32_t stream) {
    return qgo_uniform(seed + 0xA0761D6478BD642Full * (uint64_t)stream, env_index, counter);
}

/* QuadrupedEnv.reset (quadruped.py:115-139): mj_resetData, time = 0, ctrl = default.
 * With QG_RESET_RANDOM_YAW the heading of walking_quad.py:68-75 is applied. */
int qgo_reset(const qg_model *m, const qg_task *t, qgo_env *e, uint64_t seed, uint64_t env_index, uint64_t counter,
              uint32_t flags) {
    memcpy(e->qpos, m->qpos0, sizeof e->qpos);
    memset(e->qvel, 0, sizeof e->qvel);
    memset(e->act, 0, sizeof e->act);
    memcpy(e->ctrl, t->default_ctrl, sizeof e->ctrl);
    e->nstep = 0;
    if (flags & QG_RESET_RANDOM_YAW) {
        double a = 6.283185307179586 * qgo_uniform(seed, env_index, counter);
        e->qpos[3] = cos(0.5 * a); e->qpos[4] = 0; e->qpos[5] = 0; e->qpos[6] = sin(0.5 * a);
    }
    if (flags & QG_RESET_JOINT_JITTER) {   /* TODO.md:8 "RANDOMIZE ENVIRONMENT - Starting pose, joints": hinge j = qpos0 + jitter * U(-1,1), in range */
        for (int j = 0; j < QG_NJNT; j++) {
            double u = qgo_uniform_stream(seed, env_index, counter, 1u + (uint32_t)j);
            double q = e->qpos[7 + j] + t->reset_joint_jitter * (2.0 * u - 1.0);
            if (q < m->jnt_range[j][0]) q = m->jnt_range[j][0];
            if (q > m->jnt_range[j][1]) q = m->jnt_range[j][1];
            e->qpos[7 + j] = q;
        }
    }
    return 0;
}

int64_t qgo_time_limit_substeps(double timestep, double max_time) {
    if (!(max_time / timestep < 2.0e9)) return 2147483647;   /* beyond the int32 substep counter: never reached */
    double t = 0;
    int64_t n = 0;
    while (!(t >= max_time)) {   /* quadruped.py:151 `data.time >= max_time`, f64 accumulation */
        t += timestep;
        n++;
        if (n > (1ll << 40)) break;
    }
    return n;
}

/* QuadrupedEnv.step (quadruped.py:153-182) for one env.  obs has 33 entries
 * (QG_OBS_FULL) or 21 (QG_OBS_IMU).  `limit_substeps` = qgo_time_limit_substeps().
 * Returns 0, or -1 if the linear solve failed. */
int qgo_step(const qg_model *m, const qg_task *t, qgo_env *e, const double *action, int64_t limit_substeps,
             double *obs, double *reward, int32_t *done, double *comps) {
    double a[NU];
    for (int i = 0; i < NU; i++) {                 /* quadruped.py:160 np.clip to the action space */
        a[i] = action[i];
        if (a[i] < -1.0) a[i] = -1.0;
        if (a[i] > 1.0) a[i] = 1.0;
    }
    double sens[QG_NSENSOR];
    memset(sens, 0, sizeof sens);
    for (int s = 0; s < t->frame_skip; s++) {     /* quadruped.py:163-165 */
        int last = (s == t->frame_skip - 1);
        if (qgo_substep(m, e, a, (last && t->sensor_lag) ? sens : NULL, NULL) != 0) return -1;
    }
    if (!t->sensor_lag) {                          /* un-lagged variant: sensors of the final state */
        qgo_env tmp = *e;
        if (qgo_substep(m, &tmp, a, sens, NULL) != 0) return -1;
    }
    if (t->obs_mode == QG_OBS_IMU) {
        for (int i = 0; i < 18; i++) obs[i] = sens[i];
        for (int i = 0; i < 3; i++) obs[18 + i] = sens[30 + i];
    } else
        for (int i = 0; i < QG_NSENSOR; i++) obs[i] = sens[i];
    /* rewards (README.md:65-78): on the post-step state and the env-clipped action */
    double sq = 0;
    for (int i = 0; i < NU; i++) sq += a[i] * a[i];
    double c0 = t->w_forward * e->qvel[0], c1 = t->w_ctrl * sq, c2 = t->alive_bonus;
    if (comps) { comps[0] = c0; comps[1] = c1; comps[2] = c2; }
    *reward = c0 + c1 + c2;
    /* terminations (quadruped.py:149-151,178; README.md:86-89) */
    int d = 0;
    if (t->use_time_limit && e->nstep >= limit_substeps) d = 1;
    if (t->use_fall && e->qpos[2] < t->fall_height) d = 1;
    if (t->use_flip && sens[29] < 0) d = 1;        /* walking_quad.py:156-160 */
    *done = d;
    return 0;
}

/* batched convenience: n independent envs, env-major arrays (same layouts as quadgym.h) */
int qgo_step_batch(const qg_model *m, const qg_task *t, qgo_env *envs, int32_t n, const double *actions,
                   int64_t limit_substeps, double *obs, double *reward, int32_t *done, double *comps) {
    int od = t->obs_mode == QG_OBS_IMU ? 21 : QG_NSENSOR;
    for (int i = 0; i < n; i++)
        if (qgo_step(m, t, &envs[i], actions + (size_t)i * NU, limit_substeps, obs + (size_t)i * od, reward + i,
                     done + i, comps ? comps + (size_t)i * 3 : NULL) != 0)
            return -1 - i;
    return 0;
}

/* the same over `nthreads` host threads (OpenMP, static partition of the env range): the all-cores row of the CPU baseline.
 * Envs are independent, so the results are those of qgo_step_batch bit for bit. */
int qgo_step_batch_mt(const qg_model *m, const qg_task *t, qgo_env *envs, int32_t n, const double *actions,
                      int64_t limit_substeps, double *obs, double *reward, int32_t *done, double *comps, int32_t nthreads) {
    int od = t->obs_mode == QG_OBS_IMU ? 21 : QG_NSENSOR;
    int bad = 0;
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int i = 0; i < n; i++)
        if (qgo_step(m, t, &envs[i], actions + (size_t)i * NU, limit_substeps, obs + (size_t)i * od, reward + i,
                     done + i, comps ? comps + (size_t)i * 3 : NULL) != 0) {
#pragma omp atomic write
            bad = 1 + i;
        }
    return bad ? -bad : 0;
}

/* ------------------------------------------------- probes for the KAT suite */
int qgo_default_model(qg_model *out) {
    static const qg_model def = QG_MODEL_DEFAULT_INIT;
    *out = def;
    return 0;
}
int qgo_default_task(qg_task *out) {
    memset(out, 0, sizeof *out);
    out->frame_skip = 4; out->max_time = 10.0; out->use_time_limit = 1; out->use_fall = 0; out->fall_height = 0.2;
    out->w_forward = 1.0; out->w_ctrl = -0.1; out->alive_bonus = 1.0; out->obs_mode = QG_OBS_FULL;
    out->sensor_lag = 1; out->auto_reset = 0; out->reset_flags = 0;
    for (int i = 0; i < NU; i++) out->default_ctrl[i] = (i % 3 == 2) ? -0.5 : 0.0;
    out->reset_joint_jitter = 0.1;
    return 0;
}
int qgo_mass_matrix(const qg_model *m, const double *qpos, double *M) {
    kin k;
    kinematics(m, qpos, &k);
    crba(m, &k, M);
    return 0;
}
int qgo_rne(const qg_model *m, const double *qpos, const double *qvel, const double *qacc, double *tau) {
    kin k;
    kinematics(m, qpos, &k);
    rne(m, &k, qvel, qacc, tau);
    return 0;
}
int qgo_kinematics(const qg_model *m, const double *qpos, double *xpos, double *xmat, double *xcom) {
    kin k;
    kinematics(m, qpos, &k);
    memcpy(xpos, k.xpos, sizeof k.xpos);
    memcpy(xmat, k.R, sizeof k.R);
    memcpy(xcom, k.com, sizeof k.com);
    return 0;
}
/* kinetic energy summed over bodies from the spatial velocities (independent of crba) and potential energy */
int qgo_energy(const qg_model *m, const double *qpos, const double *qvel, double *kinetic, double *potential) {
    kin k;
    kinematics(m, qpos, &k);
    double v[NB][6];
    double T = 0, V = 0;
    for (int b = 0; b < NB; b++) {
        int p = m->body_parent[b];
        if (p < 0) memset(v[b], 0, sizeof v[b]); else memcpy(v[b], v[p], sizeof v[b]);
        int d0 = (b == 0) ? 0 : 5 + b, nd = (b == 0) ? 6 : 1;
        for (int d = d0; d < d0 + nd; d++)
            for (int i = 0; i < 6; i++) v[b][i] += k.S[d][i] * qvel[d];
        /* COM velocity and body angular velocity */
        double wxc[3], vc[3], Iw[3];
        cross3(v[b], k.com[b], wxc);
        for (int i = 0; i < 3; i++) vc[i] = v[b][3 + i] + wxc[i];
        matvec3(k.Iw[b], v[b], Iw);
        T += 0.5 * m->body_mass[b] * dot3(vc, vc) + 0.5 * dot3(v[b], Iw);
        V -= m->body_mass[b] * dot3(m->gravity, k.com[b]);
    }
    for (int d = 0; d < 6; d++) T += 0.5 * m->free_armature * qvel[d] * qvel[d];
    for (int j = 0; j < QG_NJNT; j++) T += 0.5 * m->jnt_armature[j] * qvel[6 + j] * qvel[6 + j];
    *kinetic = T; *potential = V;
    return 0;
}
/* total linear momentum and angular momentum about the world origin */
int qgo_momentum(const qg_model *m, const double *qpos, const double *qvel, double *lin, double *ang) {
    kin k;
    kinematics(m, qpos, &k);
    double v[NB][6];
    for (int i = 0; i < 3; i++) { lin[i] = 0; ang[i] = 0; }
    for (int b = 0; b < NB; b++) {
        int p = m->body_parent[b];
        if (p < 0) memset(v[b], 0, sizeof v[b]); else memcpy(v[b], v[p], sizeof v[b]);
        int d0 = (b == 0) ? 0 : 5 + b, nd = (b == 0) ? 6 : 1;
        for (int d = d0; d < d0 + nd; d++)
            for (int i = 0; i < 6; i++) v[b][i] += k.S[d][i] * qvel[d];
        double f[6];
        inertia_mul(m->body_mass[b], k.h[b], k.IO[b], v[b], f);
        for (int i = 0; i < 3; i++) { ang[i] += f[i]; lin[i] += f[3 + i]; }
    }
    return 0;
}
int qgo_sizeof_env(void) { return (int)sizeof(qgo_env); }
int qgo_sizeof_diag(void) { return (int)sizeof(qgo_diag); }
int qgo_sizeof_model(void) { return (int)sizeof(qg_model); }
int qgo_sizeof_task(void) { return (int)sizeof(qg_task); }
