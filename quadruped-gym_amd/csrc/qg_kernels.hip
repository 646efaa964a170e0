// qg_kernels.hip -- gfx950 kernels of the batched quadruped simulator.
//
// Replaces the arithmetic behind QuadrupedEnv.step() (src/envs/quadruped.py:153-182 of
// antopio26/quadruped-gym): frame_skip x mj_step (quadruped.py:163-165, integrator
// implicitfast, quadruped.xml:4), the sensor pack (quadruped.py:141-143, quadruped.xml:174-217),
// the README reward / termination set (README.md:64-90) and reset (quadruped.py:115-139).
//
// Mapping: ONE ENVIRONMENT PER WAVEFRONT LANE, one 64-lane wave per workgroup.  State is
// struct-of-arrays in HBM ([field][env]) so every load/store of a wave is one 256-byte
// segment.  All frame_skip substeps run inside one launch; state is read once and written
// once per env-step.  No MFMA: the path is a chain of small (3x3 / 6x6) per-env solves.
//
// Formulation (different on purpose from the CPU oracle, which works in world coordinates
// with a dense 18x18 solve): everything is expressed in the FRAME's own axes about the
// FRAME's origin.  Per substep
//   A. base prelude: rotation from the quaternion, base velocity / bias acceleration,
//      FRAME body force and ground contact, start of the base 6x6 block;
//   B. loop over the 4 legs (rolled): kinematics of fema/shin/foot, recursive Newton-Euler
//      bias forces, ground contact, composite-rigid-body inertia (mass matrix columns), servo
//      / limit / damping terms, then block elimination of the leg's 3x3 joint block into the
//      base block (the mass matrix is base 6x6 + four 3x3 leg blocks + four 6x3 couplings);
//      the per-leg factors (Y = F H^-1, u = H^-1 b) and the joint state live in LDS, one
//      column per lane, because the leg loop indexes them at run time;
//   C. 6x6 base solve (LDL^T);
//   D. loop over the legs: back-substitution, semi-implicit integration of the hinges,
//      servo activation filter;
//   E. base integration (position, quaternion).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qg_device.h"
#include "qg_model_baked.h"

#define DEV __device__ __forceinline__

// ------------------------------------------------------------------------------------------
// model tables.  Two kernel variants: BAKED reads the default robot's constants from a `const`
// device table with compile-time indices, so they become instruction literals (zeros and ones
// fold away, nothing is loaded); the generic variant reads the same struct from a device buffer
// through scalar loads and serves any other numbers.  The legs of the reference robot are
// identical up to the mounting transform of the fema (quadruped.xml:71,89,107,125), so the baked
// variant uses leg 0's link constants for every leg and only the mount (pos, Q) is read per leg.
// ------------------------------------------------------------------------------------------
static __device__ const KModel QG_BAKED_MODEL = {QG_BAKED_FLOATS};
template <bool BAKED> DEV const KModel &table(const KModel *__restrict__ M) {
    if constexpr (BAKED) return QG_BAKED_MODEL; else return *M;
}
template <bool BAKED> DEV const KLink &link_of(const KModel &C, int k, int i) { return C.link[BAKED ? i : 3 * k + i]; }

// ------------------------------------------------------------------------------------------
// small fixed-size algebra
// ------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
DEV V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
DEV float dot(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
DEV V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DEV V3 fma3(float s, V3 a, V3 b) { return v3(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z)); }
DEV V3 ld3(const float *p) { return v3(p[0], p[1], p[2]); }

// orthonormal frame given by its three axes (columns) expressed in the working frame
struct Fr { V3 ex, ey, ez; };
DEV V3 rot(const Fr &E, V3 r) { return fma3(r.x, E.ex, fma3(r.y, E.ey, r.z * E.ez)); }   // local -> working
DEV V3 rotT(const Fr &E, V3 r) { return v3(dot(E.ex, r), dot(E.ey, r), dot(E.ez, r)); }  // working -> local

struct Sym3 { float xx, yy, zz, xy, xz, yz; };
DEV V3 mul(const Sym3 &S, V3 v) {
    return v3(fmaf(S.xx, v.x, fmaf(S.xy, v.y, S.xz * v.z)), fmaf(S.xy, v.x, fmaf(S.yy, v.y, S.yz * v.z)),
              fmaf(S.xz, v.x, fmaf(S.yz, v.y, S.zz * v.z)));
}
DEV void add(Sym3 &a, const Sym3 &b) { a.xx += b.xx; a.yy += b.yy; a.zz += b.zz; a.xy += b.xy; a.xz += b.xz; a.yz += b.yz; }
DEV void rank1(Sym3 &a, float w, V3 u, V3 v) {  // a += w * (u v^T), caller guarantees symmetry (u == v)
    a.xx = fmaf(w * u.x, v.x, a.xx); a.yy = fmaf(w * u.y, v.y, a.yy); a.zz = fmaf(w * u.z, v.z, a.zz);
    a.xy = fmaf(w * u.x, v.y, a.xy); a.xz = fmaf(w * u.x, v.z, a.xz); a.yz = fmaf(w * u.y, v.z, a.yz);
}

struct M3 { V3 r0, r1, r2; };  // rows
DEV V3 mul(const M3 &A, V3 v) { return v3(dot(A.r0, v), dot(A.r1, v), dot(A.r2, v)); }
DEV V3 mulT(const M3 &A, V3 v) { return fma3(v.x, A.r0, fma3(v.y, A.r1, v.z * A.r2)); }
DEV void add(M3 &a, const M3 &b) { a.r0 = a.r0 + b.r0; a.r1 = a.r1 + b.r1; a.r2 = a.r2 + b.r2; }

// spatial vectors [angular; linear] about the FRAME origin, FRAME axes
struct SV { V3 a, l; };
DEV SV operator+(SV p, SV q) { SV r = {p.a + q.a, p.l + q.l}; return r; }
DEV float dot(SV p, SV q) { return dot(p.a, q.a) + dot(p.l, q.l); }

// rigid-body spatial inertia about the FRAME origin: mass, first moment h = m*c, rotational inertia
struct Rigid { float m; V3 h; Sym3 I; };
DEV SV mul(const Rigid &B, SV v) {
    SV f;
    f.a = mul(B.I, v.a) + cross(B.h, v.l);
    f.l = B.m * v.l - cross(B.h, v.a);
    return f;
}

// general symmetric 6x6 (rigid inertia + implicit contact damping): [[AA, AL], [AL^T, LL]]
struct Sym6 { Sym3 AA; M3 AL; Sym3 LL; };
DEV SV mul(const Sym6 &A, SV s) {
    SV f;
    f.a = mul(A.AA, s.a) + mul(A.AL, s.l);
    f.l = mulT(A.AL, s.a) + mul(A.LL, s.l);
    return f;
}
DEV void add(Sym6 &a, const Sym6 &b) { add(a.AA, b.AA); add(a.AL, b.AL); add(a.LL, b.LL); }
DEV Sym6 sym6_of(const Rigid &B) {
    Sym6 A;
    A.AA = B.I;
    A.AL.r0 = v3(0.f, -B.h.z, B.h.y);   // [h]x
    A.AL.r1 = v3(B.h.z, 0.f, -B.h.x);
    A.AL.r2 = v3(-B.h.y, B.h.x, 0.f);
    A.LL.xx = A.LL.yy = A.LL.zz = B.m;
    A.LL.xy = A.LL.xz = A.LL.yz = 0.f;
    return A;
}
// A += m * (point mass at r)  +  w * a a^T with a = [r x n; n]
DEV void add_contact_damping(Sym6 &A, float m, float w, V3 r, V3 n) {
    float rr = dot(r, r);
    A.AA.xx += m * (rr - r.x * r.x); A.AA.yy += m * (rr - r.y * r.y); A.AA.zz += m * (rr - r.z * r.z);
    A.AA.xy -= m * r.x * r.y; A.AA.xz -= m * r.x * r.z; A.AA.yz -= m * r.y * r.z;
    V3 h = m * r;
    A.AL.r0 = A.AL.r0 + v3(0.f, -h.z, h.y);
    A.AL.r1 = A.AL.r1 + v3(h.z, 0.f, -h.x);
    A.AL.r2 = A.AL.r2 + v3(-h.y, h.x, 0.f);
    A.LL.xx += m; A.LL.yy += m; A.LL.zz += m;
    V3 ra = cross(r, n);
    rank1(A.AA, w, ra, ra);
    A.AL.r0 = fma3(w * ra.x, n, A.AL.r0);
    A.AL.r1 = fma3(w * ra.y, n, A.AL.r1);
    A.AL.r2 = fma3(w * ra.z, n, A.AL.r2);
    rank1(A.LL, w, n, n);
}

DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// sin and cos with Cody-Waite reduction to [-pi/4, pi/4] and minimax polynomials (~1 ulp for |x| < 1e4)
DEV void sincos_f(float x, float &s, float &c) {
    float k = rintf(x * 0.63661977236758134f);
    float r = fmaf(-k, 1.57079625129699707031f, x);
    r = fmaf(-k, 7.54978941586159635335e-08f, r);
    float r2 = r * r;
    float sp = fmaf(r2, fmaf(r2, fmaf(r2, 2.718311493989822e-06f, -1.984090162742e-04f), 8.333329385889463e-03f), -1.666666597127914e-01f);
    float sr = fmaf(r * r2, sp, r);
    float cp = fmaf(r2, fmaf(r2, fmaf(r2, 2.443315711809948e-05f, -1.388731625493765e-03f), 4.166664568298827e-02f), -0.5f);
    float cr = fmaf(r2, cp, 1.0f);
    int q = (int)k;
    float s0 = (q & 1) ? cr : sr;
    float c0 = (q & 1) ? sr : cr;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
}

// counter-based uniform in [0,1) with 24 random bits (same stream as the oracle's qgo_uniform)
DEV uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
DEV float uniform24(uint64_t seed, uint64_t env_index, uint64_t counter) {
    uint64_t x = seed + 0x9E3779B97F4A7C15ull * (env_index + 1) + 0xD1B54A32D192ED03ull * (counter + 1);
    x = mix64(mix64(x));
    return (float)(uint32_t)(x >> 40) * (1.0f / 16777216.0f);
}

// ------------------------------------------------------------------------------------------
// ground contact of one body: LCP-free penalty model, one aggregated contact per body
//   W   = sum_i k * max(0, margin - z_i)      spring force of the sample points below the margin
//   P   = centre of pressure of those spring forces
//   c   = contact_c * min(1, sum_i pen_i / ramp)      the damper ramps in with depth (continuous force)
//   F_n = max(0, W - c v_n(P));   F_t = -min(c, mu F_n / |v_t|) v_t(P)
// The damper is linear in the velocity with secant coefficients (c_n, c_t); it enters the
// system matrix as h * (c_t * point-mass(P) + (c_n - c_t) a a^T), a = [P x n; n].
// ------------------------------------------------------------------------------------------
template <int NCP>
DEV void body_contact(const float (*cp)[3], const Fr &E, V3 p, float z_origin, V3 n, SV v, float kc, float cmax, float inv_ramp,
                      float margin, float mu, float h, SV &f_ext, Sym6 &A) {
    V3 nl = rotT(E, n);                 // world up in the body's own axes
    float wsum = 0.f;
    V3 s = v3(0.f, 0.f, 0.f);
    float zb = margin - z_origin;
#pragma unroll
    for (int i = 0; i < NCP; ++i) {
        V3 r = ld3(cp[i]);
        float pen = fmaxf(zb - dot(nl, r), 0.f);
        wsum += pen;
        s = fma3(pen, r, s);
    }
    bool active = wsum > 0.f;
    float W = kc * wsum;
    float cc = cmax * fminf(wsum * inv_ramp, 1.f);
    float inv = rcp(active ? wsum : 1.f);
    V3 P = p + rot(E, inv * s);         // centre of pressure, FRAME axes about the FRAME origin
    V3 vP = v.l + cross(v.a, P);
    float vn = dot(n, vP);
    V3 vt = vP - vn * n;
    float Fn = W - cc * vn;
    float cn = cc;
    if (Fn < 0.f) { Fn = 0.f; cn = W * rcp(vn); }
    float speed = __builtin_amdgcn_sqrtf(dot(vt, vt));
    float ct = cc;
    if (cc * speed > mu * Fn) ct = mu * Fn * rcp(speed);
    if (!active) { Fn = 0.f; cn = 0.f; ct = 0.f; }
    V3 F = Fn * n - ct * vt;
    f_ext.a = cross(P, F);
    f_ext.l = F;
    add_contact_damping(A, h * ct, h * (cn - ct), P, n);
}

// ------------------------------------------------------------------------------------------
// per-lane scratch columns in LDS: slot s of lane l lives at lds[s * 64 + l] (conflict-free)
// ------------------------------------------------------------------------------------------
#define LQ(j) (0 + (j))        // hinge positions      12
#define LQD(j) (12 + (j))      // hinge velocities     12
#define LACT(j) (24 + (j))     // servo activations    12
#define LU(j) (36 + (j))       // servo-clamped ctrl   12
#define LY(k, i) (48 + 21 * (k) + (i))   // per leg: Y (6x3, 18) then u (3)
#define QG_LDS_SLOTS (48 + 21 * 4)
#define QG_OBS_TILE_FLOATS (64 * 35)

struct BaseState { V3 pw; float qw, qx, qy, qz; V3 vw; V3 wb; };

struct SensorOut { float accel[3]; V3 pw, vw, wb, vb, xaxis, zaxis; float jpos[12]; };

// one physics substep for the env of this lane (mj_step of quadruped.py:165)
template <bool BAKED>
DEV void substep(const KModel *__restrict__ Mp, float *__restrict__ lds, int lane, BaseState &B, bool want_sensors, SensorOut &so) {
    const KModel &C = table<BAKED>(Mp);
    const KModel *M = &C;
    const float h = M->h;
    // ---- A. base prelude -----------------------------------------------------------------
    float qn = rcp(__builtin_amdgcn_sqrtf(B.qw * B.qw + B.qx * B.qx + B.qy * B.qy + B.qz * B.qz));
    float w = B.qw * qn, x = B.qx * qn, y = B.qy * qn, z = B.qz * qn;
    // R (FRAME -> world), stored by columns: cx, cy, cz are the FRAME axes in world coordinates
    V3 cx = v3(1.f - 2.f * (y * y + z * z), 2.f * (x * y + w * z), 2.f * (x * z - w * y));
    V3 cy = v3(2.f * (x * y - w * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z + w * x));
    V3 cz = v3(2.f * (x * z + w * y), 2.f * (y * z - w * x), 1.f - 2.f * (x * x + y * y));
    V3 n = v3(cx.z, cy.z, cz.z);                                   // world up in FRAME axes
    V3 gw = ld3(M->g);
    V3 gb = v3(dot(cx, gw), dot(cy, gw), dot(cz, gw));             // gravity in FRAME axes
    V3 vb = v3(dot(cx, B.vw), dot(cy, B.vw), dot(cz, B.vw));       // base linear velocity in FRAME axes
    SV V0 = {B.wb, vb};
    // unknowns are (d/dt w_b, classical acceleration of the FRAME origin); with both zero the
    // spatial acceleration of the FRAME is [0; -w x v - g]
    SV A0 = {v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 0.f) - cross(B.wb, vb) - gb};

    if (want_sensors) {
        so.pw = B.pw; so.vw = B.vw; so.wb = B.wb; so.vb = vb;
        so.xaxis = cx; so.zaxis = cz;
#pragma unroll
        for (int j = 0; j < 12; ++j) so.jpos[j] = lds[LQ(j) * 64 + lane];
    }

    Rigid I0 = {M->m0, ld3(M->h0), {M->I0[0], M->I0[1], M->I0[2], M->I0[3], M->I0[4], M->I0[5]}};
    SV Iv0 = mul(I0, V0), Ia0 = mul(I0, A0);
    SV p0;                                                          // bias force on the base rows
    p0.a = Ia0.a + cross(V0.a, Iv0.a) + cross(V0.l, Iv0.l);
    p0.l = Ia0.l + cross(V0.a, Iv0.l);
    Sym6 Ic0 = sym6_of(I0);
    {
        Fr E0 = {v3(1.f, 0.f, 0.f), v3(0.f, 1.f, 0.f), v3(0.f, 0.f, 1.f)};
        SV fe;
        body_contact<QGK_CP_FRAME>(M->cp0, E0, v3(0.f, 0.f, 0.f), B.pw.z, n, V0, M->contact_k, M->contact_c, M->contact_inv_ramp,
                                   M->contact_margin, M->contact_mu, h, fe, Ic0);
        p0.a = p0.a - fe.a;
        p0.l = p0.l - fe.l;
    }
    // free-joint damping and armature act on all six base DoFs (quadruped.xml:9,62-63)
    {
        float dg = M->free_armature + h * M->free_damping;
        Ic0.AA.xx += dg; Ic0.AA.yy += dg; Ic0.AA.zz += dg;
        Ic0.LL.xx += dg; Ic0.LL.yy += dg; Ic0.LL.zz += dg;
        p0.a = fma3(M->free_damping, V0.a, p0.a);
        p0.l = fma3(M->free_damping, V0.l, p0.l);
    }
    SV rhs0 = {v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 0.f)};               // accumulates -Y b of the legs

    // ---- B. legs: dynamics terms and elimination into the base block ---------------------
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        Fr Ep = {v3(1.f, 0.f, 0.f), v3(0.f, 1.f, 0.f), v3(0.f, 0.f, 1.f)};
        V3 pp = v3(0.f, 0.f, 0.f);
        SV vp = V0, ap = A0;
        SV S[3], f[3];
        Sym6 Ag[3];
        float qd[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const KLink &L = link_of<BAKED>(C, k, i);
            const KLink &Lm = (i == 0) ? C.link[3 * k] : L;   // the fema's mounting transform differs per leg
            const int j = 3 * k + i;
            float th = lds[LQ(j) * 64 + lane] - L.ref;      // rotation applied = qpos - ref
            qd[i] = lds[LQD(j) * 64 + lane];
            float sn, cs;
            sincos_f(th, sn, cs);
            V3 p = pp + rot(Ep, ld3(Lm.pos));
            V3 tx = fma3(Lm.Q[0], Ep.ex, fma3(Lm.Q[3], Ep.ey, Lm.Q[6] * Ep.ez));
            V3 ty = fma3(Lm.Q[1], Ep.ex, fma3(Lm.Q[4], Ep.ey, Lm.Q[7] * Ep.ez));
            V3 tz = fma3(Lm.Q[2], Ep.ex, fma3(Lm.Q[5], Ep.ey, Lm.Q[8] * Ep.ez));
            Fr E = {fma3(cs, tx, sn * ty), fma3(cs, ty, (-sn) * tx), tz};
            S[i].a = E.ez;
            S[i].l = cross(p, E.ez);
            SV v = {fma3(qd[i], S[i].a, vp.a), fma3(qd[i], S[i].l, vp.l)};
            // a = a_parent + (v x S) qd
            SV a;
            a.a = fma3(qd[i], cross(v.a, S[i].a), ap.a);
            a.l = fma3(qd[i], cross(v.a, S[i].l) + cross(v.l, S[i].a), ap.l);
            // rigid inertia of the link about the FRAME origin, FRAME axes
            Rigid Bi;
            Bi.m = L.mass;
            V3 c = p + rot(E, ld3(L.ipos));
            Bi.h = L.mass * c;
            {
                V3 ux = fma3(L.inertia[0], E.ex, fma3(L.inertia[3], E.ey, L.inertia[4] * E.ez));
                V3 uy = fma3(L.inertia[3], E.ex, fma3(L.inertia[1], E.ey, L.inertia[5] * E.ez));
                V3 uz = fma3(L.inertia[4], E.ex, fma3(L.inertia[5], E.ey, L.inertia[2] * E.ez));
                float hc = dot(Bi.h, c);
                Bi.I.xx = fmaf(ux.x, E.ex.x, fmaf(uy.x, E.ey.x, uz.x * E.ez.x)) + hc - Bi.h.x * c.x;
                Bi.I.yy = fmaf(ux.y, E.ex.y, fmaf(uy.y, E.ey.y, uz.y * E.ez.y)) + hc - Bi.h.y * c.y;
                Bi.I.zz = fmaf(ux.z, E.ex.z, fmaf(uy.z, E.ey.z, uz.z * E.ez.z)) + hc - Bi.h.z * c.z;
                Bi.I.xy = fmaf(ux.x, E.ex.y, fmaf(uy.x, E.ey.y, uz.x * E.ez.y)) - Bi.h.x * c.y;
                Bi.I.xz = fmaf(ux.x, E.ex.z, fmaf(uy.x, E.ey.z, uz.x * E.ez.z)) - Bi.h.x * c.z;
                Bi.I.yz = fmaf(ux.y, E.ex.z, fmaf(uy.y, E.ey.z, uz.y * E.ez.z)) - Bi.h.y * c.z;
            }
            SV Iv = mul(Bi, v), Ia = mul(Bi, a);
            f[i].a = Ia.a + cross(v.a, Iv.a) + cross(v.l, Iv.l);
            f[i].l = Ia.l + cross(v.a, Iv.l);
            Ag[i] = sym6_of(Bi);
            SV fe;
            body_contact<QGK_CP_LINK>(L.cp, E, p, B.pw.z + dot(n, p), n, v, M->contact_k, M->contact_c, M->contact_inv_ramp,
                                      M->contact_margin, M->contact_mu, h, fe, Ag[i]);
            f[i].a = f[i].a - fe.a;
            f[i].l = f[i].l - fe.l;
            Ep = E; pp = p; vp = v; ap = a;
        }
        // backward pass: composite inertias (mass-matrix columns) and bias torques
        Sym6 Ic = Ag[2];
        SV fc = f[2];
        SV F2 = mul(Ic, S[2]);
        float H22 = dot(S[2], F2), H12 = dot(S[1], F2), H02 = dot(S[0], F2), t2 = dot(S[2], fc);
        add(Ic, Ag[1]);
        fc = fc + f[1];
        SV F1 = mul(Ic, S[1]);
        float H11 = dot(S[1], F1), H01 = dot(S[0], F1), t1 = dot(S[1], fc);
        add(Ic, Ag[0]);
        fc = fc + f[0];
        SV F0 = mul(Ic, S[0]);
        float H00 = dot(S[0], F0), t0 = dot(S[0], fc);
        add(Ic0, Ic);
        p0 = p0 + fc;

        // joint-space terms: damping, armature, servo, soft limits
        float bj[3], Hd[3] = {H00, H11, H22}, tb[3] = {t0, t1, t2};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const KLink &L = link_of<BAKED>(C, k, i);
            const int j = 3 * k + i;
            float q = lds[LQ(j) * 64 + lane];
            float act = lds[LACT(j) * 64 + lane];
            // position servo (quadruped.xml:10-37): force from the PRE-update activation
            float force = L.kp * (act - L.gear * q) - L.kv * L.gear * qd[i];
            bool clamped = (force <= L.force_lo) || (force >= L.force_hi);
            force = fminf(fmaxf(force, L.force_lo), L.force_hi);
            float dimp = L.damping + (clamped ? 0.f : L.kv * L.gear * L.gear);
            float tau = L.gear * force - L.damping * qd[i];
            // soft joint limits
            float below = L.lo - q, above = q - L.hi;
            float pen = fmaxf(fmaxf(below, above), 0.f);
            float bl = M->limit_b * fminf(pen * M->limit_inv_ramp, 1.f);   // damper ramps in: continuous torque
            if (below > 0.f) {
                float spring = M->limit_k * below;
                float t = spring - bl * qd[i];
                float be = bl;
                if (t < 0.f) { t = 0.f; be = spring * rcp(qd[i]); }
                tau += t;
                dimp += be;
            } else if (above > 0.f) {
                float spring = M->limit_k * above;
                float t = spring + bl * qd[i];
                float be = bl;
                if (t < 0.f) { t = 0.f; be = -spring * rcp(qd[i]); }
                tau -= t;
                dimp += be;
            }
            Hd[i] += L.armature + h * dimp;
            bj[i] = tau - tb[i];
        }
        // LDL^T of the leg block [hip, knee, ankle]
        float d0 = Hd[0], id0 = rcp(d0);
        float l10 = H01 * id0, l20 = H02 * id0;
        float d1 = fmaf(-l10, H01, Hd[1]), id1 = rcp(d1);
        float t21 = fmaf(-l20, H01, H12);
        float l21 = t21 * id1;
        float d2 = fmaf(-l21, t21, fmaf(-l20, H02, Hd[2])), id2 = rcp(d2);
        // y = H^-1 r for r = rows of F (6) and r = b
        float Fr0[6] = {F0.a.x, F0.a.y, F0.a.z, F0.l.x, F0.l.y, F0.l.z};
        float Fr1[6] = {F1.a.x, F1.a.y, F1.a.z, F1.l.x, F1.l.y, F1.l.z};
        float Fr2[6] = {F2.a.x, F2.a.y, F2.a.z, F2.l.x, F2.l.y, F2.l.z};
        float Y0[6], Y1[6], Y2[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            float z0 = Fr0[r];
            float z1 = fmaf(-l10, z0, Fr1[r]);
            float z2 = fmaf(-l21, z1, fmaf(-l20, z0, Fr2[r]));
            float y2 = z2 * id2;
            float y1 = fmaf(-l21, y2, z1 * id1);
            float y0 = fmaf(-l20, y2, fmaf(-l10, y1, z0 * id0));
            Y0[r] = y0; Y1[r] = y1; Y2[r] = y2;
        }
        float u0, u1, u2;
        {
            float z0 = bj[0];
            float z1 = fmaf(-l10, z0, bj[1]);
            float z2 = fmaf(-l21, z1, fmaf(-l20, z0, bj[2]));
            u2 = z2 * id2;
            u1 = fmaf(-l21, u2, z1 * id1);
            u0 = fmaf(-l20, u2, fmaf(-l10, u1, z0 * id0));
        }
        // Schur complement: Ic0 -= Y F^T (symmetric), rhs0 -= F u
        {
#define YF(r, c) (Y0[r] * Fr0[c] + Y1[r] * Fr1[c] + Y2[r] * Fr2[c])
            Ic0.AA.xx -= YF(0, 0); Ic0.AA.yy -= YF(1, 1); Ic0.AA.zz -= YF(2, 2);
            Ic0.AA.xy -= YF(0, 1); Ic0.AA.xz -= YF(0, 2); Ic0.AA.yz -= YF(1, 2);
            Ic0.AL.r0.x -= YF(0, 3); Ic0.AL.r0.y -= YF(0, 4); Ic0.AL.r0.z -= YF(0, 5);
            Ic0.AL.r1.x -= YF(1, 3); Ic0.AL.r1.y -= YF(1, 4); Ic0.AL.r1.z -= YF(1, 5);
            Ic0.AL.r2.x -= YF(2, 3); Ic0.AL.r2.y -= YF(2, 4); Ic0.AL.r2.z -= YF(2, 5);
            Ic0.LL.xx -= YF(3, 3); Ic0.LL.yy -= YF(4, 4); Ic0.LL.zz -= YF(5, 5);
            Ic0.LL.xy -= YF(3, 4); Ic0.LL.xz -= YF(3, 5); Ic0.LL.yz -= YF(4, 5);
#undef YF
            rhs0.a.x -= Fr0[0] * u0 + Fr1[0] * u1 + Fr2[0] * u2;
            rhs0.a.y -= Fr0[1] * u0 + Fr1[1] * u1 + Fr2[1] * u2;
            rhs0.a.z -= Fr0[2] * u0 + Fr1[2] * u1 + Fr2[2] * u2;
            rhs0.l.x -= Fr0[3] * u0 + Fr1[3] * u1 + Fr2[3] * u2;
            rhs0.l.y -= Fr0[4] * u0 + Fr1[4] * u1 + Fr2[4] * u2;
            rhs0.l.z -= Fr0[5] * u0 + Fr1[5] * u1 + Fr2[5] * u2;
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            lds[LY(k, r) * 64 + lane] = Y0[r];
            lds[LY(k, 6 + r) * 64 + lane] = Y1[r];
            lds[LY(k, 12 + r) * 64 + lane] = Y2[r];
        }
        lds[LY(k, 18) * 64 + lane] = u0;
        lds[LY(k, 19) * 64 + lane] = u1;
        lds[LY(k, 20) * 64 + lane] = u2;
    }

    // ---- C. base solve: Ic0 x = rhs0 - p0, x = [d/dt w_b; classical acceleration] ---------
    float x6[6];
    {
        float A[6][6];
        A[0][0] = Ic0.AA.xx; A[1][1] = Ic0.AA.yy; A[2][2] = Ic0.AA.zz;
        A[1][0] = Ic0.AA.xy; A[2][0] = Ic0.AA.xz; A[2][1] = Ic0.AA.yz;
        A[3][0] = Ic0.AL.r0.x; A[4][0] = Ic0.AL.r0.y; A[5][0] = Ic0.AL.r0.z;
        A[3][1] = Ic0.AL.r1.x; A[4][1] = Ic0.AL.r1.y; A[5][1] = Ic0.AL.r1.z;
        A[3][2] = Ic0.AL.r2.x; A[4][2] = Ic0.AL.r2.y; A[5][2] = Ic0.AL.r2.z;
        A[3][3] = Ic0.LL.xx; A[4][4] = Ic0.LL.yy; A[5][5] = Ic0.LL.zz;
        A[4][3] = Ic0.LL.xy; A[5][3] = Ic0.LL.xz; A[5][4] = Ic0.LL.yz;
        float b[6] = {rhs0.a.x - p0.a.x, rhs0.a.y - p0.a.y, rhs0.a.z - p0.a.z,
                      rhs0.l.x - p0.l.x, rhs0.l.y - p0.l.y, rhs0.l.z - p0.l.z};
        // in-place LDL^T on the lower triangle (A[i][j], i >= j); L below the diagonal, 1/D kept apart
        float dg[6], idg[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float ld[6];                      // L[j][t] * D[t]
            float d = A[j][j];
#pragma unroll
            for (int t = 0; t < j; ++t) { ld[t] = A[j][t] * dg[t]; d = fmaf(-A[j][t], ld[t], d); }
            dg[j] = d;
            idg[j] = rcp(d);
#pragma unroll
            for (int i = j + 1; i < 6; ++i) {
                float s = A[i][j];
#pragma unroll
                for (int t = 0; t < j; ++t) s = fmaf(-A[i][t], ld[t], s);
                A[i][j] = s * idg[j];
            }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int t = 0; t < i; ++t) b[i] = fmaf(-A[i][t], b[t], b[i]);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) b[i] *= idg[i];
#pragma unroll
        for (int i = 5; i >= 0; --i) {
#pragma unroll
            for (int t = i + 1; t < 6; ++t) b[i] = fmaf(-A[t][i], b[t], b[i]);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) x6[i] = b[i];
    }
    V3 wdot = v3(x6[0], x6[1], x6[2]);
    V3 acl = v3(x6[3], x6[4], x6[5]);
    if (want_sensors) {
        // accelerometer (quadruped.xml:200): proper acceleration in the site frame = a_c - g_b
        so.accel[0] = acl.x - gb.x; so.accel[1] = acl.y - gb.y; so.accel[2] = acl.z - gb.z;
    }

    // ---- D. legs: back-substitution, hinge integration, servo filter ---------------------
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const KLink &L = link_of<BAKED>(C, k, i);
            const int j = 3 * k + i;
            float acc = lds[LY(k, 18 + i) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 6; ++r) acc = fmaf(-lds[LY(k, 6 * i + r) * 64 + lane], x6[r], acc);
            float qdn = fmaf(h, acc, lds[LQD(j) * 64 + lane]);
            lds[LQD(j) * 64 + lane] = qdn;
            lds[LQ(j) * 64 + lane] = fmaf(h, qdn, lds[LQ(j) * 64 + lane]);
            float act = lds[LACT(j) * 64 + lane];
            lds[LACT(j) * 64 + lane] = fmaf(lds[LU(j) * 64 + lane] - act, L.act_decay, act);
        }
    }

    // ---- E. base integration ---------------------------------------------------------------
    // d/dt v_world = R * (classical acceleration in FRAME axes)
    V3 aw = fma3(acl.x, cx, fma3(acl.y, cy, acl.z * cz));
    B.vw = fma3(h, aw, B.vw);
    B.wb = fma3(h, wdot, B.wb);
    B.pw = fma3(h, B.vw, B.pw);
    {
        // q <- q * exp(h w): half-angle series (|h w| / 2 stays far below 0.5 rad)
        float hh = 0.5f * h;
        float x2 = hh * hh * dot(B.wb, B.wb);
        float sc = hh * fmaf(x2, fmaf(x2, fmaf(x2, -1.f / 5040.f, 1.f / 120.f), -1.f / 6.f), 1.f);
        float cw = fmaf(x2, fmaf(x2, fmaf(x2, -1.f / 720.f, 1.f / 24.f), -0.5f), 1.f);
        V3 dv = sc * B.wb;
        float nw = w * cw - x * dv.x - y * dv.y - z * dv.z;
        float nx = w * dv.x + x * cw + y * dv.z - z * dv.y;
        float ny = w * dv.y - x * dv.z + y * cw + z * dv.x;
        float nz = w * dv.z + x * dv.y - y * dv.x + z * cw;
        float inv = rcp(__builtin_amdgcn_sqrtf(nw * nw + nx * nx + ny * ny + nz * nz));
        B.qw = nw * inv; B.qx = nx * inv; B.qy = ny * inv; B.qz = nz * inv;
    }
}

// ------------------------------------------------------------------------------------------
// env-step kernel: one launch = frame_skip substeps + sensor pack + rewards + terminations
// (+ auto-reset) for every env.  grid = ceil(n / 64) workgroups of one wave.
// ------------------------------------------------------------------------------------------
template <bool BAKED>
__global__ __launch_bounds__(QGK_WAVE) void qg_step_kernel(const KModel *__restrict__ Mp, const KTask *__restrict__ T, KStepArgs P) {
    const KModel *M = &table<BAKED>(Mp);
    __shared__ float lds[QG_LDS_SLOTS * 64];
    __shared__ float tile[QG_OBS_TILE_FLOATS];
    const int lane = threadIdx.x;
    const int env0 = blockIdx.x * QGK_WAVE;
    const int n = P.n;
    const bool live = env0 + lane < n;
    const int env = live ? env0 + lane : n - 1;   // tail lanes shadow the last env; their stores are masked

    // ---- load state (coalesced: lane i reads base + i*4 of every field) -------------------
    BaseState B;
    B.pw = v3(P.st.qpos[0 * n + env], P.st.qpos[1 * n + env], P.st.qpos[2 * n + env]);
    B.qw = P.st.qpos[3 * n + env]; B.qx = P.st.qpos[4 * n + env]; B.qy = P.st.qpos[5 * n + env]; B.qz = P.st.qpos[6 * n + env];
    B.vw = v3(P.st.qvel[0 * n + env], P.st.qvel[1 * n + env], P.st.qvel[2 * n + env]);
    B.wb = v3(P.st.qvel[3 * n + env], P.st.qvel[4 * n + env], P.st.qvel[5 * n + env]);
    int nstep = P.st.nstep[env];
    // action: clip to the action space [-1, 1] (quadruped.py:160), then to the servo's ctrlrange
    float ssq = 0.f;
    float aclip[12];
    {
        const float4 *ap = reinterpret_cast<const float4 *>(P.actions + (size_t)env * 12);
        float4 a0 = ap[0], a1 = ap[1], a2 = ap[2];
        float av[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            float a = fminf(fmaxf(av[j], -1.f), 1.f);
            aclip[j] = a;
            ssq = fmaf(a, a, ssq);
            lds[LU(j) * 64 + lane] = fminf(fmaxf(a, M->link[BAKED ? j % 3 : j].ctrl_lo), M->link[BAKED ? j % 3 : j].ctrl_hi);
            lds[LQ(j) * 64 + lane] = P.st.qpos[(7 + j) * n + env];
            lds[LQD(j) * 64 + lane] = P.st.qvel[(6 + j) * n + env];
            lds[LACT(j) * 64 + lane] = P.st.act[j * n + env];
        }
    }

    // ---- frame_skip physics substeps (quadruped.py:163-165) ----------------------------------
    SensorOut so;
    const int fs = T->frame_skip;
    const bool lag = T->sensor_lag != 0;
#pragma unroll 1
    for (int s = 0; s < fs; ++s) substep<BAKED>(Mp, lds, lane, B, lag && (s == fs - 1), so);
    nstep += fs;
    if (!lag) {   // un-lagged sensors: one extra forward pass on a scratch copy of the state
        BaseState B2 = B;
        float keep[36];
#pragma unroll
        for (int j = 0; j < 36; ++j) keep[j] = lds[j * 64 + lane];
        substep<BAKED>(Mp, lds, lane, B2, true, so);
#pragma unroll
        for (int j = 0; j < 36; ++j) lds[j * 64 + lane] = keep[j];
    }

    // ---- rewards and terminations on the post-step state (README.md:64-90) ----------------
    float c_fwd = T->w_forward * B.vw.x;
    float c_ctl = T->w_ctrl * ssq;
    float c_alive = T->alive_bonus;
    float reward = c_fwd + c_ctl + c_alive;
    bool done = nstep >= T->limit_substeps;
    if (T->use_fall) done = done || (B.pw.z < T->fall_height);

    // ---- outputs: stage rows in LDS, then store the wave's contiguous chunk coalesced ------
    const int od = T->obs_mode == 1 ? 21 : 33;
    const int row = P.packed ? od + 2 : od;
    {
        float *r = tile + lane * row;
#pragma unroll
        for (int j = 0; j < 12; ++j) r[j] = so.jpos[j];
        r[12] = so.accel[0]; r[13] = so.accel[1]; r[14] = so.accel[2];
        r[15] = so.wb.x; r[16] = so.wb.y; r[17] = so.wb.z;
        if (od == 33) {
            r[18] = so.pw.x; r[19] = so.pw.y; r[20] = so.pw.z;
            r[21] = so.vw.x; r[22] = so.vw.y; r[23] = so.vw.z;
            r[24] = so.xaxis.x; r[25] = so.xaxis.y; r[26] = so.xaxis.z;
            r[27] = so.zaxis.x; r[28] = so.zaxis.y; r[29] = so.zaxis.z;
            r[30] = so.vb.x; r[31] = so.vb.y; r[32] = so.vb.z;
        } else {
            r[18] = so.vb.x; r[19] = so.vb.y; r[20] = so.vb.z;
        }
        if (P.packed) { r[od] = reward; r[od + 1] = done ? 1.f : 0.f; }
    }
    __syncthreads();
    {
        const int live_envs = min(QGK_WAVE, n - env0);
        const int total = live_envs * row;
        float *dst = (P.packed ? P.packed : P.obs) + (size_t)env0 * row;
        for (int e = lane; e < total; e += QGK_WAVE) dst[e] = tile[e];
    }
    if (live && !P.packed) {
        P.reward[env] = reward;
        P.done[env] = done ? 1 : 0;
    }
    if (live && P.comps) {
        P.comps[(size_t)env * 3 + 0] = c_fwd;
        P.comps[(size_t)env * 3 + 1] = c_ctl;
        P.comps[(size_t)env * 3 + 2] = c_alive;
    }

    // ---- auto-reset (VecEnv semantics) and state write-back ---------------------------------
    const bool rst = done && T->auto_reset;
    if (rst) {
        B.pw = v3(M->qpos0[0], M->qpos0[1], M->qpos0[2]);
        B.qw = M->qpos0[3]; B.qx = M->qpos0[4]; B.qy = M->qpos0[5]; B.qz = M->qpos0[6];
        if (T->reset_flags & 1u) {   // random heading (walking_quad.py:68-75)
            float a = 6.283185307179586f * uniform24(P.seed, P.env_index_base + (uint64_t)env, P.step_index);
            float sn, cs;
            sincos_f(0.5f * a, sn, cs);
            B.qw = cs; B.qx = 0.f; B.qy = 0.f; B.qz = sn;
        }
        B.vw = v3(0.f, 0.f, 0.f);
        B.wb = v3(0.f, 0.f, 0.f);
        nstep = 0;
    }
    if (live) {
        P.st.qpos[0 * n + env] = B.pw.x; P.st.qpos[1 * n + env] = B.pw.y; P.st.qpos[2 * n + env] = B.pw.z;
        P.st.qpos[3 * n + env] = B.qw; P.st.qpos[4 * n + env] = B.qx; P.st.qpos[5 * n + env] = B.qy; P.st.qpos[6 * n + env] = B.qz;
        P.st.qvel[0 * n + env] = B.vw.x; P.st.qvel[1 * n + env] = B.vw.y; P.st.qvel[2 * n + env] = B.vw.z;
        P.st.qvel[3 * n + env] = B.wb.x; P.st.qvel[4 * n + env] = B.wb.y; P.st.qvel[5 * n + env] = B.wb.z;
        P.st.nstep[env] = nstep;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            P.st.qpos[(7 + j) * n + env] = rst ? M->qpos0[7 + j] : lds[LQ(j) * 64 + lane];
            P.st.qvel[(6 + j) * n + env] = rst ? 0.f : lds[LQD(j) * 64 + lane];
            P.st.act[j * n + env] = rst ? 0.f : lds[LACT(j) * 64 + lane];
        }
        if (P.track_ctrl) {
#pragma unroll
            for (int j = 0; j < 12; ++j) P.st.ctrl[j * n + env] = rst ? T->default_ctrl[j] : aclip[j];
        }
    }
}

// ------------------------------------------------------------------------------------------
// reset (quadruped.py:115-139): mj_resetData, time = 0, ctrl = default; optional random yaw
// ------------------------------------------------------------------------------------------
__global__ void qg_reset_kernel(const KModel *__restrict__ M, const KTask *__restrict__ T, KState st, int n, const uint8_t *mask,
                                uint64_t seed, uint64_t env_index_base, uint64_t counter, uint32_t flags) {
    int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= n) return;
    if (mask && !mask[env]) return;
    for (int j = 0; j < 19; ++j) st.qpos[j * n + env] = M->qpos0[j];
    if (flags & 1u) {
        float a = 6.283185307179586f * uniform24(seed, env_index_base + (uint64_t)env, counter);
        float sn, cs;
        sincos_f(0.5f * a, sn, cs);
        st.qpos[3 * n + env] = cs; st.qpos[4 * n + env] = 0.f; st.qpos[5 * n + env] = 0.f; st.qpos[6 * n + env] = sn;
    }
    for (int j = 0; j < 18; ++j) st.qvel[j * n + env] = 0.f;
    for (int j = 0; j < 12; ++j) { st.act[j * n + env] = 0.f; st.ctrl[j * n + env] = T->default_ctrl[j]; }
    st.nstep[env] = 0;
}

// env-major [n][w] <-> field-major [w][n] (state snapshot / restore at the ABI)
__global__ void qg_transpose_in(const float *__restrict__ src, float *__restrict__ dst, int n, int w) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * w) return;
    int f = i / n, e = i - f * n;
    dst[i] = src[(size_t)e * w + f];
}
__global__ void qg_transpose_out(const float *__restrict__ src, float *__restrict__ dst, int n, int w) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * w) return;
    int e = i / w, f = i - e * w;
    dst[i] = src[(size_t)f * n + e];
}
