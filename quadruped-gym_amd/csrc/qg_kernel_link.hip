// qg_kernel_link.hip -- env-step kernel, ONE LINK PER LANE: sixteen lanes per environment, four environments per wave.
//
// Why: up to 4096 envs the step time is the instruction count of ONE wave (a wave alone on its SIMD issues one VALU instruction per
// >= 4.2 cycles whatever it is; 256 waves of the one-leg-per-lane kernel leave three quarters of the chip's 1024 SIMDs idle).  With
// lane = 16 * env + 4 * leg + r, lane r in {0,1,2} owns LINK r of its leg (fema / shin / foot): the per-link work of a leg -- rigid
// inertia about the FRAME origin, bias force, ground contact: two thirds of the leg pass -- runs once instead of three times per
// wave, and 4096 envs fill all 1024 SIMDs.  Lane r = 3 of each leg is a spare: it carries the leg's quarter of the FRAME's
// contact sample points.
//   * a lane r < 3 also owns HINGE r of its leg (state, servo filter, integration; round 3 -- every lane of the leg used to carry all
//     three); the sines / cosines of the leg's three hinge rotations are broadcast over the leg's lanes at the head of a substep;
//   * kinematic chain: the three FRAMES are built in every lane (literal mounts: the quarter-turn frame makes every leg "leg 0"), the
//     lane picks its own link's; velocity and bias acceleration are PREFIX sums over the leg's lanes of one term per lane
//     (qd_r S_r, qd_r v_r x S_r; two fused DPP adds per value);
//   * link body: the same code in every lane on per-lane constants (mass, inertia, eight contact points of link r: held in
//     registers, loaded once per launch);
//   * backward pass inside the leg's four lanes with DPP quad_perm: composite inertia / force = suffix sums over r (two DPP adds per
//     value), lane r forms column r of the leg's 3x3 joint block and its joint's servo / limit / damping terms, the nine numbers of
//     the block and right-hand side are broadcast and every lane factors the same 3x3; the Schur complement is the sum of the rank-1
//     terms z_r z_r^T / d_r, z = L^-1 F, one per lane;
//   * the 33 base-block numbers (+ 4 of the FRAME contact, whose twelve sample points are shared out one per link lane) are summed
//     over the env's 16 lanes with a symmetric DPP butterfly (quad_perm, quad_perm, row_half_mirror, row_mirror: every lane gets the
//     bit-identical sum); base prelude, 6x6 solve and base integration run redundantly in the 16 lanes.
// Lagged sensors only (the reference's), at most one wave per SIMD: the launcher uses it for n <= 4096.  Two variants as for the
// one-leg-per-lane kernel: the compiled-in robot with literal constants, any other robot with the model tables staged in LDS.
#define QGK_LINK_ENVS 4     // envs per wave
#include "qg_po_dev.h"      // partially observable observation pack: per-env device functions of the fused <WALK, PO> variant

// State accesses of this kernel: a scalar base plus a 32-bit BYTE offset per lane (global_load / store v, voff, s[base]) instead of a
// 64-bit address computed per access -- two to three instructions fewer in front of every load, i.e. the prologue's loads go out
// sooner (the launcher keeps this mapping to n <= 2^24 envs: 19 n floats stay below 4 GiB).
typedef const __attribute__((address_space(1))) char *lk_gcbytes;
typedef __attribute__((address_space(1))) char *lk_gbytes;
template <class T> DEV T lk_ld(const T *base, unsigned byte_off) { return *(const __attribute__((address_space(1))) T *)((lk_gcbytes)base + byte_off); }
template <class T> DEV void lk_st(T *base, unsigned byte_off, T v) { *(__attribute__((address_space(1))) T *)((lk_gbytes)base + byte_off) = v; }

template <int CTRL> DEV float dpp_any(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
// sum over the 16 lanes of an env (one DPP row); symmetric at every level, so all 16 lanes hold identical bits
// (fp contract off inside these helpers: with the default "fast" contraction the backend folds a multiply that feeds `x` into the
// first add -- fma(a, b, dpp(a * b)) -- which keeps the product alive twice and leaves the DPP move unfused: mul + mov_dpp + fma
// instead of mul + add_dpp)
DEV float env_sum(float x) {
#pragma clang fp contract(off)
    x += dpp_any<0xB1>(x);    // quad_perm [1,0,3,2]
    x += dpp_any<0x4E>(x);    // quad_perm [2,3,0,1]
    x += dpp_any<0x141>(x);   // row_half_mirror
    x += dpp_any<0x140>(x);   // row_mirror
    return x;
}
DEV V3 env_sum(V3 a) { return v3(env_sum(a.x), env_sum(a.y), env_sum(a.z)); }
// Sums of 4 G values over the env's 16 lanes in 2 G + 2 DPP instructions per group of four instead of 16, the butterfly TRANSPOSED
// through the bank mask of the DPP adds (a bank = 4 consecutive lanes; a masked DPP add writes the enabled banks only and leaves the
// rest of vdst alone -- the compiler's DPP combiner cannot express that, hence the assembly): of a group (a, b, c, d)
//   row_half_mirror (banks 0<->1, 2<->3): a's partial sums into banks 0 and 2 of one register, b's into its banks 1 and 3 (c, d alike);
//   row_ror:8 (banks 0<->2, 1<->3):       banks 0/1 of the result from the (a|b) register, banks 2/3 from the (c|d) register;
//   the two quad_perm levels on that ONE register.
// v[4g] returns the group's sums: bank k (lanes 4k .. 4k+3) holds the sum of v[4g + k], the same bits in its four lanes; the consumer
// fetches them with row_newbcast:4k as the DPP operand of the add that accumulates them (banked_sum<k>).  v[4g + 2] is clobbered.
// Hazards (a DPP read needs 2 wait states behind a VALU write of the same register; the hazard recogniser does not look into inline
// assembly): s_nop 1 at both ends, and inside the block the order keeps >= 2 instructions between a write and its DPP read (G >= 3).
#define QG_T_HM(d, s, bm) "v_add_f32_dpp %[" d "], %[" s "], %[" s "] row_half_mirror row_mask:0xf bank_mask:" bm "\n\t"
#define QG_T_ROR(d, s, bm) "v_add_f32_dpp %[" d "], %[" s "], %[" s "] row_ror:8 row_mask:0xf bank_mask:" bm "\n\t"
#define QG_T_QP(d, qp) "v_add_f32_dpp %[" d "], %[" d "], %[" d "] quad_perm:" qp " row_mask:0xf bank_mask:0xf\n\t"
#define QG_T_L1(g) QG_T_HM("a" g, "a" g, "0x5") QG_T_HM("c" g, "c" g, "0x5") QG_T_HM("a" g, "b" g, "0xa") QG_T_HM("c" g, "d" g, "0xa")
#define QG_T_L2(g) QG_T_ROR("a" g, "a" g, "0x3") QG_T_ROR("a" g, "c" g, "0xc")
#define QG_T_Q1(g) QG_T_QP("a" g, "[1,0,3,2]")
#define QG_T_Q2(g) QG_T_QP("a" g, "[2,3,0,1]")
// (every output is TIED, "+v": an input-output operand can never be allocated on top of one of the plain inputs b / d, which is what
// an early-clobber marker would otherwise have to say -- a and c are read before and written by the block, b and d only read)
#define QG_T_OUT(g, v) [a##g] "+v"(v[4 * g]), [c##g] "+v"(v[4 * g + 2])
#define QG_T_IN(g, v) [b##g] "v"(v[4 * g + 1]), [d##g] "v"(v[4 * g + 3])
DEV void env_sum_banked16(float (&v)[16]) {
    asm("s_nop 1\n\t" QG_T_L1("0") QG_T_L1("1") QG_T_L1("2") QG_T_L1("3") QG_T_L2("0") QG_T_L2("1") QG_T_L2("2") QG_T_L2("3")
        QG_T_Q1("0") QG_T_Q1("1") QG_T_Q1("2") QG_T_Q1("3") QG_T_Q2("0") QG_T_Q2("1") QG_T_Q2("2") QG_T_Q2("3") "s_nop 1"
        : QG_T_OUT(0, v), QG_T_OUT(1, v), QG_T_OUT(2, v), QG_T_OUT(3, v)
        : QG_T_IN(0, v), QG_T_IN(1, v), QG_T_IN(2, v), QG_T_IN(3, v));
}
DEV void env_sum_banked12(float (&v)[12]) {
    asm("s_nop 1\n\t" QG_T_L1("0") QG_T_L1("1") QG_T_L1("2") QG_T_L2("0") QG_T_L2("1") QG_T_L2("2")
        QG_T_Q1("0") QG_T_Q1("1") QG_T_Q1("2") QG_T_Q2("0") QG_T_Q2("1") QG_T_Q2("2") "s_nop 1"
        : QG_T_OUT(0, v), QG_T_OUT(1, v), QG_T_OUT(2, v)
        : QG_T_IN(0, v), QG_T_IN(1, v), QG_T_IN(2, v));
}
// the sum a banked group register holds in bank K, for every lane of the env (a row_newbcast DPP operand of the consuming add)
template <int K> DEV float banked_sum(float r) { return dpp_any<0x150 + 4 * K>(r); }
// suffix sum over the links of a leg, lanes r = 0,1,2 (lane 3 must hold 0): lane r gets x_r + x_{r+1} + ... + x_2
DEV float leg_suffix(float x) {
#pragma clang fp contract(off)
    x += dpp_any<0xF9>(x);    // quad_perm [1,2,3,3]
    x += dpp_any<0xFE>(x);    // quad_perm [2,3,3,3] of the partial sums: lane 0 adds (x2 + x3), lane 1 adds x3 = 0
    return x;
}
DEV float leg_bcast0(float x) { return dpp_any<0x00>(x); }
DEV float leg_bcast1(float x) { return dpp_any<0x55>(x); }
DEV float leg_bcast2(float x) { return dpp_any<0xAA>(x); }

// inclusive prefix sum over the links of a leg, lanes r = 0,1,2 (lane 3 must hold 0): lane r gets x_0 + ... + x_r
DEV float leg_prefix(float x) {
#pragma clang fp contract(off)
    x += dpp_any<0xD3>(x);    // quad_perm [3,0,1,3]: the link below (lane 0 reads the spare lane's zero, which stays zero)
    x += dpp_any<0x4F>(x);    // quad_perm [3,3,0,1] of the partial sums: lane 2 adds x_0
    return x;
}
DEV V3 leg_prefix(V3 a) { return v3(leg_prefix(a.x), leg_prefix(a.y), leg_prefix(a.z)); }

// what link r needs as per-lane data (registers)
struct LinkRegs {
    float mass, ipos[3], inertia[6], cp[QGK_CP_LINK][3];
    f2 cp2[QGK_CP_LINK / 2][3];     // the same sample points as packed pairs (i, i + 4): what the compiled-in robot's substep reads
    float lo, hi, damping, armature, kp, kv, gear, force_lo, force_hi, act_decay;
    float cpF[3];       // this lane's sample point of the FRAME (one of its twelve; the spare lanes carry none)
    float ml;           // 1 in the lanes that own a link, 0 in the spare lane
    float fb[8];        // compiled-in robot: the FRAME block's non-zero literals, held in registers for the substep's DPP adds
};
// this lane's hinge: lane r < 3 of leg k owns hinge 3k + r; the spare lane shadows hinge 3k + 2 (its copy is never stored)
struct HingeLane { float q, qd, act, u, sn, cs; };

DEV float sel3(int r, float a, float b, float c) { return r == 0 ? a : (r == 1 ? b : c); }
DEV V3 sel3(int r, V3 a, V3 b, V3 c) { return v3(sel3(r, a.x, b.x, c.x), sel3(r, a.y, b.y, c.y), sel3(r, a.z, b.z, c.z)); }

// BAKED: the compiled-in robot, whose legs are quarter-turn copies of one another: the chain's constants are literals and the lane
// works in its leg's quarter-turn frame (cm, sm).  Otherwise `C` is the model staged in LDS and the chain reads leg kleg's own links.
//
// Round 3 ("instruction diet", 1 517 -> see DESIGN section 4 for the count): every lane used to carry the leg's three hinges and to run
// the whole kinematic chain -- frames, velocities, bias accelerations of all three links -- committing what it met on the way under
// its execution mask.  Now
//   * a lane owns ONE hinge (state, servo filter, integration: once instead of three times per lane); the sines / cosines of the
//     leg's three hinge rotations are broadcast over the leg's lanes at the head of the substep (six DPP moves);
//   * only the frames of the chain are built in sequence (with literal mounts that is ~50 instructions for the three links); the
//     lane then picks its own link's frame (selects), and
//   * velocity and bias acceleration are PREFIX SUMS over the leg's lanes of one term per lane, qd_r S_r and qd_r (v_r x S_r):
//     two fused DPP adds per value instead of every lane walking the chain;
//   * the FRAME's twelve contact sample points are shared out one per link lane (the spare lane alone used to evaluate three).
template <bool BAKED>
DEV void substep_link(const KModel &C, float cm, float sm, int r, bool lead_env, BaseState &B, HingeLane &J, const LinkRegs &K,
                      bool want_sensors, float *__restrict__ row, int kleg, float &zaxis_z) {
    using namespace pk3;
    const float h = C.h;
    const BaseCtx bc = pk3::base_prelude_unit(C, B); // the quaternion is of unit length here (normalised at load, then by base_integrate)
    const V3 nb = bc.n;
    if (want_sensors) {              // the step's sensordata describes the state at the start of its last substep
        zaxis_z = bc.cz.z;
        if (r < 3) row[3 * kleg + r] = J.q;
        if (lead_env) {
            row[15] = B.wb.x; row[16] = B.wb.y; row[17] = B.wb.z;
            row[18] = B.pw.x; row[19] = B.pw.y; row[20] = B.pw.z;
            row[21] = B.vw.x; row[22] = B.vw.y; row[23] = B.vw.z;
            row[24] = bc.cx.x; row[25] = bc.cx.y; row[26] = bc.cx.z;
            row[27] = bc.cz.x; row[28] = bc.cz.y; row[29] = bc.cz.z;
            row[30] = bc.vb.x; row[31] = bc.vb.y; row[32] = bc.vb.z;
        }
    }
    const int rr = r < 2 ? r : 2;                    // the spare lane shadows link 2
    // ---- kinematic chain: the three frames in every lane of the leg ----------------------------------------------------------
    const float snj[3] = {leg_bcast0(J.sn), leg_bcast1(J.sn), leg_bcast2(J.sn)};
    const float csj[3] = {leg_bcast0(J.cs), leg_bcast1(J.cs), leg_bcast2(J.cs)};
    Fr E[3];
    V3 p[3];
    SV S[3];
    {
        Fr Ep = {v3(BAKED ? cm : 1.f, BAKED ? sm : 0.f, 0.f), v3(BAKED ? -sm : 0.f, BAKED ? cm : 1.f, 0.f), v3(0.f, 0.f, 1.f)};
        V3 pp = v3(0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const KLink &L = link_of<BAKED>(C, kleg, i);
            const float sn = snj[i], cs = csj[i];
            p[i] = pp + rot(Ep, ld3(L.pos));
            V3 tx = fma3(L.Q[0], Ep.ex, fma3(L.Q[3], Ep.ey, L.Q[6] * Ep.ez));
            V3 ty = fma3(L.Q[1], Ep.ex, fma3(L.Q[4], Ep.ey, L.Q[7] * Ep.ez));
            V3 tz = fma3(L.Q[2], Ep.ex, fma3(L.Q[5], Ep.ey, L.Q[8] * Ep.ez));
            E[i].ex = fma3(cs, tx, sn * ty);
            E[i].ey = fma3(cs, ty, (-sn) * tx);
            E[i].ez = tz;
            S[i].a = E[i].ez;
            S[i].l = cross(p[i], E[i].ez);
            Ep = E[i]; pp = p[i];
        }
    }
    // this lane's link: frame, origin, joint axis
    const Fr Ep = {sel3(rr, E[0].ex, E[1].ex, E[2].ex), sel3(rr, E[0].ey, E[1].ey, E[2].ey), sel3(rr, E[0].ez, E[1].ez, E[2].ez)};
    const V3 pp = sel3(rr, p[0], p[1], p[2]);
    const SV So = {Ep.ez, cross(pp, Ep.ez)};
    // velocity and bias acceleration of the link: prefix sums over the leg's lanes (the spare lane contributes nothing)
    SV vp, ap;
    {
        const float qdm = K.ml * J.qd;
        const SV Sq = {qdm * So.a, qdm * So.l};                      // the joint's velocity S qd: the prefix sum's term, and
        vp.a = leg_prefix(Sq.a) + bc.V0.a;                             // the second factor of the bias acceleration below
        vp.l = leg_prefix(Sq.l) + bc.V0.l;
        // a = a_parent + v x (S qd), v the link's own velocity
        const V3 ca = cross(vp.a, Sq.a);
        const V3 cl = cross_add(vp.l, Sq.a, cross(vp.a, Sq.l));
        ap.a = leg_prefix(ca) + bc.A0.a;
        ap.l = leg_prefix(cl) + bc.A0.l;
    }
    // ---- this lane's link: rigid inertia about the FRAME origin (FRAME axes), bias force, ground contact -------------------------
    Rigid Bi;
    Bi.m = K.mass;
    {
        const Fr &E = Ep;
        V3 c = pp + rot(E, v3(K.ipos[0], K.ipos[1], K.ipos[2]));
        Bi.h = Bi.m * c;
        V3 ux = fma3(K.inertia[0], E.ex, fma3(K.inertia[3], E.ey, K.inertia[4] * E.ez));
        V3 uy = fma3(K.inertia[3], E.ex, fma3(K.inertia[1], E.ey, K.inertia[5] * E.ez));
        V3 uz = fma3(K.inertia[4], E.ex, fma3(K.inertia[5], E.ey, K.inertia[2] * E.ez));
        float hc = dot(Bi.h, c);
        // (xx, yy) and (xz, yz) as packed pairs: the (x, y) halves of ux, uy, uz, the axes, h and c are the pairs the operators above made
        const f2 uxp = {ux.x, ux.y}, uyp = {uy.x, uy.y}, uzp = {uz.x, uz.y}, hp = {Bi.h.x, Bi.h.y}, cp2 = {c.x, c.y};
        const f2 d2 = __builtin_elementwise_fma(uxp, f2{E.ex.x, E.ex.y}, __builtin_elementwise_fma(uyp, f2{E.ey.x, E.ey.y}, uzp * f2{E.ez.x, E.ez.y}))
                      + f2{hc, hc} - hp * cp2;
        const f2 o2 = __builtin_elementwise_fma(uxp, f2{E.ex.z, E.ex.z}, __builtin_elementwise_fma(uyp, f2{E.ey.z, E.ey.z}, uzp * f2{E.ez.z, E.ez.z}))
                      - hp * f2{c.z, c.z};
        Bi.I.xx = d2.x; Bi.I.yy = d2.y; Bi.I.xz = o2.x; Bi.I.yz = o2.y;
        Bi.I.zz = fmaf(ux.z, E.ex.z, fmaf(uy.z, E.ey.z, uz.z * E.ez.z)) + hc - Bi.h.z * c.z;
        Bi.I.xy = fmaf(ux.x, E.ex.y, fmaf(uy.x, E.ey.y, uz.x * E.ez.y)) - Bi.h.x * c.y;
    }
    SV f;
    {
        SV Iv = mul(Bi, vp), Ia = mul(Bi, ap);
        f.a = cross_add(vp.l, Iv.l, cross_add(vp.a, Iv.a, Ia.a));
        f.l = cross_add(vp.a, Iv.l, Ia.l);
    }
    Sym6 A = sym6_of(Bi);
    {
        const float zo = B.pw.z + dot(nb, pp);
        V3 nl = rotT(Ep, nb);
        float wsum = 0.f;
        V3 s = v3(0.f, 0.f, 0.f);
        const float zb = C.contact_margin - zo;
        if constexpr (BAKED) {   // the eight sample points two at a time as packed FP32 (points i and i + 4 of the lane's register-held constants share an
            // instruction; the two partial sums are added at the end): 36 + 4 instructions for 64
            static_assert(QGK_CP_LINK % 2 == 0, "sample points are taken in pairs");
            f2 w2 = {0.f, 0.f};
            V3T<f2> s2 = v3<f2>(w2, w2, w2);
            const V3T<f2> nl2 = v3<f2>(f2{nl.x, nl.x}, f2{nl.y, nl.y}, f2{nl.z, nl.z});
            const f2 zb2 = {zb, zb};
#pragma unroll
            for (int i = 0; i < QGK_CP_LINK / 2; ++i) contact_point(v3<f2>(K.cp2[i][0], K.cp2[i][1], K.cp2[i][2]), nl2, zb2, w2, s2);
            wsum = w2.x + w2.y;
            s = v3(s2.x.x + s2.x.y, s2.y.x + s2.y.y, s2.z.x + s2.z.y);
        } else {
#pragma unroll
            for (int i = 0; i < QGK_CP_LINK; ++i) contact_point(v3(K.cp[i][0], K.cp[i][1], K.cp[i][2]), nl, zb, wsum, s);
        }
        wsum *= K.ml;                               // the spare lane is no link (its mass and inertia are zero as well)
        SV fe;
        ContactDampT<float> cd;
        pk3::contact_eval(wsum, s, Ep, pp, nb, vp, C.contact_k, C.contact_c, C.contact_inv_ramp, C.contact_mu, h, fe, cd);
        f.a = f.a - fe.a;
        f.l = f.l - fe.l;
        pk3::add_contact_damping(A, cd.mc, cd.w, cd.P, nb);
    }
    // FRAME contact: every link lane evaluates one of the FRAME's twelve sample points
    float wsumF = 0.f;
    V3 sF = v3(0.f, 0.f, 0.f);
    contact_point(v3(K.cpF[0], K.cpF[1], K.cpF[2]), nb, C.contact_margin - B.pw.z, wsumF, sF);
    wsumF *= K.ml; sF = K.ml * sF;
    // ---- composite inertia and force of the subtree rooted at this link: suffix sums over the leg's lanes ----------------------------
    Sym6 Ic;
    SV fc;
    Ic.AA.xx = leg_suffix(A.AA.xx); Ic.AA.yy = leg_suffix(A.AA.yy); Ic.AA.zz = leg_suffix(A.AA.zz);
    Ic.AA.xy = leg_suffix(A.AA.xy); Ic.AA.xz = leg_suffix(A.AA.xz); Ic.AA.yz = leg_suffix(A.AA.yz);
    Ic.AL.r0 = v3(leg_suffix(A.AL.r0.x), leg_suffix(A.AL.r0.y), leg_suffix(A.AL.r0.z));
    Ic.AL.r1 = v3(leg_suffix(A.AL.r1.x), leg_suffix(A.AL.r1.y), leg_suffix(A.AL.r1.z));
    Ic.AL.r2 = v3(leg_suffix(A.AL.r2.x), leg_suffix(A.AL.r2.y), leg_suffix(A.AL.r2.z));
    Ic.LL.xx = leg_suffix(A.LL.xx); Ic.LL.yy = leg_suffix(A.LL.yy); Ic.LL.zz = leg_suffix(A.LL.zz);
    Ic.LL.xy = leg_suffix(A.LL.xy); Ic.LL.xz = leg_suffix(A.LL.xz); Ic.LL.yz = leg_suffix(A.LL.yz);
    fc.a = v3(leg_suffix(f.a.x), leg_suffix(f.a.y), leg_suffix(f.a.z));
    fc.l = v3(leg_suffix(f.l.x), leg_suffix(f.l.y), leg_suffix(f.l.z));
    // ---- column r of the leg's joint block, the joint's own terms -------------------------------------------------------------------
    const SV F = mul(Ic, So);
    const float Hc0 = dot(S[0], F), Hc1 = dot(S[1], F);          // rows 0, 1 of column r (used by the lanes below the diagonal only)
    const float Hown = dot(So, F);                                // the diagonal entry
    const float tb = dot(So, fc);
    float Hd_o, b_o;
    {
        const float q = J.q, qd = J.qd, act = J.act;
        // position servo (quadruped.xml:10-37): force from the PRE-update activation
        float force = K.kp * (act - K.gear * q) - (K.kv * K.gear) * qd;
        const bool clamped = force <= K.force_lo || force >= K.force_hi;
        force = fminf(fmaxf(force, K.force_lo), K.force_hi);
        float dimp = K.damping + (clamped ? 0.f : K.kv * K.gear * K.gear);
        float tau = K.gear * force - K.damping * qd;
        // soft joint limits: one-sided spring + damper that ramps in with the penetration (continuous torque).  Written once for
        // both limits in the frame of the limit that is violated (sg = +1 below the lower one, -1 above the upper one; lo < hi, so at
        // most one is): t = k pen - b (sg qd) is the torque towards the range, max(t, 0) what is applied (leaving fast: no
        // pull-back, and the secant k pen / (sg qd) as its damping).  Inside the range pen = 0, hence b = 0, t = 0, nothing added:
        // no select on "active".  Same products and sums as the two-sided form it replaces (28 -> 19 instructions).
        const float below = K.lo - q, above = q - K.hi;
        const float pen = fmaxf(fmaxf(below, above), 0.f);
        const float bl = C.limit_b * fminf(pen * C.limit_inv_ramp, 1.f);
        const float sg = below > above ? 1.f : -1.f;
        const float sqd = sg * qd;
        const float spring = C.limit_k * pen;
        const float tq = spring - bl * sqd;
        const bool leaving = tq < 0.f;
        tau = fmaf(sg, fmaxf(tq, 0.f), tau);
        dimp = dimp + (leaving ? spring * rcp(sqd) : bl);
        Hd_o = Hown + K.armature + h * dimp;
        b_o = tau - tb;
    }
    // ---- the 3x3 block and its right-hand side in every lane of the leg; LDL^T ----------------------------------------------------------
    const float H00 = leg_bcast0(Hd_o), H01 = leg_bcast1(Hc0), H11 = leg_bcast1(Hd_o);
    const float H02 = leg_bcast2(Hc0), H12 = leg_bcast2(Hc1), H22 = leg_bcast2(Hd_o);
    const float b0 = leg_bcast0(b_o), b1 = leg_bcast1(b_o), b2 = leg_bcast2(b_o);
    const float id0 = rcp(H00);
    const float l10 = H01 * id0, l20 = H02 * id0;
    const float d1 = fmaf(-l10, H01, H11), id1 = rcp(d1);
    const float t21 = fmaf(-l20, H01, H12);
    const float l21 = t21 * id1;
    const float d2 = fmaf(-l21, t21, fmaf(-l20, H02, H22)), id2 = rcp(d2);
    const float y0 = b0, y1 = fmaf(-l10, y0, b1), y2 = fmaf(-l21, y1, fmaf(-l20, y0, b2));     // y = L^-1 b
    // ---- z = L^-1 F, row r in lane r: z0 = F0, z1 = F1 - l10 z0, z2 = F2 - l20 z0 - l21 z1 -----------------------------------------------
    float z[6] = {F.a.x, F.a.y, F.a.z, F.l.x, F.l.y, F.l.z};
    {
        // (negated coefficients in registers and each broadcast next to the multiply-add that consumes it: v_fmac_f32_dpp)
        const float nc0 = r == 1 ? -l10 : (r == 2 ? -l20 : 0.f), nc1 = r == 2 ? -l21 : 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) z[i] = fmaf(leg_bcast0(z[i]), nc0, z[i]);
#pragma unroll
        for (int i = 0; i < 6; ++i) z[i] = fmaf(leg_bcast1(z[i]), nc1, z[i]);
    }
    // ---- this lane's share of the base block: its own link's inertia and force (summed over the env's lanes they are the legs' composite
    // inertias and forces) minus z z^T / d_r;  F u = sum_r z_r y_r / d_r ----
    const float idr = sel3(rr, id0, id1, id2), yr = sel3(rr, y0, y1, y2);
    // Packed FP32 (a lone wave pays per instruction issued).  The pairs follow the (x, y) + z split of the packed V3 operators:
    // P = (z0, z1) and L = (z3, z4) are the xy halves of the angular and linear part, Z = (z2, z5) their z components; no value is
    // in two pairs, so nothing is copied to form one, and every product takes its halves from ONE pair per operand (op_sel).
    const f2 P = {z[0], z[1]}, L = {z[3], z[4]}, Z = {z[2], z[5]};
    const f2 idr2 = {idr, idr};
    const f2 WP = idr2 * P, WL = idr2 * L, WZ = idr2 * Z;
    Sym6 Cn;
    {
        const f2 Zlo = {Z.x, Z.x}, Zhi = {Z.y, Z.y};
        const f2 WPlo = {WP.x, WP.x}, WPhi = {WP.y, WP.y}, WZlo = {WZ.x, WZ.x};
        const f2 aa0 = __builtin_elementwise_fma(-WP, P, f2{A.AA.xx, A.AA.yy});            // w0 z0, w1 z1
        const f2 aa1 = __builtin_elementwise_fma(-WP, Zlo, f2{A.AA.xz, A.AA.yz});          // w0 z2, w1 z2
        Cn.AA.xx = aa0.x; Cn.AA.yy = aa0.y; Cn.AA.xz = aa1.x; Cn.AA.yz = aa1.y;
        Cn.AA.xy = fmaf(-WP.x, P.y, A.AA.xy); Cn.AA.zz = fmaf(-WZ.x, Z.x, A.AA.zz);
        const f2 c0 = __builtin_elementwise_fma(-WPlo, L, f2{A.AL.r0.x, A.AL.r0.y});       // w0 z3, w0 z4
        const f2 c1 = __builtin_elementwise_fma(-WPhi, L, f2{A.AL.r1.x, A.AL.r1.y});       // w1 z3, w1 z4
        const f2 c2 = __builtin_elementwise_fma(-WZlo, L, f2{A.AL.r2.x, A.AL.r2.y});       // w2 z3, w2 z4
        const f2 c3 = __builtin_elementwise_fma(-WP, Zhi, f2{A.AL.r0.z, A.AL.r1.z});       // w0 z5, w1 z5
        Cn.AL.r0 = v3(c0.x, c0.y, c3.x);
        Cn.AL.r1 = v3(c1.x, c1.y, c3.y);
        Cn.AL.r2 = v3(c2.x, c2.y, fmaf(-WZ.x, Z.y, A.AL.r2.z));
        const f2 l0 = __builtin_elementwise_fma(-WL, L, f2{A.LL.xx, A.LL.yy});             // w3 z3, w4 z4
        const f2 l1 = __builtin_elementwise_fma(-WL, Zhi, f2{A.LL.xz, A.LL.yz});           // w3 z5, w4 z5
        Cn.LL.xx = l0.x; Cn.LL.yy = l0.y; Cn.LL.xz = l1.x; Cn.LL.yz = l1.y;
        Cn.LL.xy = fmaf(-WL.x, L.y, A.LL.xy); Cn.LL.zz = fmaf(-WZ.y, Z.y, A.LL.zz);
    }
    // right-hand side share: -(f_own + z_r y_r / d_r)
    const f2 nyr = {-yr, -yr};
    const f2 rh0 = __builtin_elementwise_fma(nyr, WP, f2{-f.a.x, -f.a.y}), rh1 = __builtin_elementwise_fma(nyr, WL, f2{-f.l.x, -f.l.y}),
             rh2 = __builtin_elementwise_fma(nyr, WZ, f2{-f.a.z, -f.l.z});
    SV rhn = {v3(rh0.x, rh0.y, rh2.x), v3(rh1.x, rh1.y, rh2.y)};
    // ---- base block: FRAME body, the sums over the env's 16 lanes, the FRAME's contact (a wave-uniform branch that updates the block
    // in place: on the usual path, no contact, nothing has to be moved), the 6x6 solve -- all redundant in the 16 lanes.  Every DPP
    // move sits in one basic block with the add that consumes it (the compiler fuses them into v_add_f32_dpp only then).
    float x6[6];
    {
        SV p0;
        Sym6 Ic0;
        pk3::frame_body_pk(C, bc, h, p0, Ic0);
        // the block's non-zero literals (the FRAME's inertia + armature) come from registers filled once per launch (K.fb): a DPP add
        // takes no literal, so each of the eight sums below was a DPP move and an add with a literal; with a register operand it is
        // one v_add_f32_dpp
        if constexpr (BAKED) {
            Ic0.AA.xx = K.fb[0]; Ic0.AA.yy = K.fb[1]; Ic0.AA.zz = K.fb[2]; Ic0.LL.xx = K.fb[3]; Ic0.LL.yy = K.fb[4]; Ic0.LL.zz = K.fb[5];
            Ic0.AL.r0.y = K.fb[6]; Ic0.AL.r1.x = K.fb[7];
        }
        // the 28 sums over the env's lanes (21 of the block, 6 of the right-hand side, the FRAME's contact weight), banked
        float t16[16] = {Cn.AA.xx, Cn.AA.yy, Cn.AA.zz, Cn.AA.xy, Cn.AA.xz, Cn.AA.yz, Cn.AL.r0.x, Cn.AL.r0.y,
                         Cn.AL.r0.z, Cn.AL.r1.x, Cn.AL.r1.y, Cn.AL.r1.z, Cn.AL.r2.x, Cn.AL.r2.y, Cn.AL.r2.z, wsumF};
        float t12[12] = {Cn.LL.xx, Cn.LL.yy, Cn.LL.zz, Cn.LL.xy, Cn.LL.xz, Cn.LL.yz, rhn.a.x, rhn.a.y, rhn.a.z, rhn.l.x, rhn.l.y, rhn.l.z};
        env_sum_banked16(t16);
        env_sum_banked12(t12);
        Ic0.AA.xx += banked_sum<0>(t16[0]); Ic0.AA.yy += banked_sum<1>(t16[0]); Ic0.AA.zz += banked_sum<2>(t16[0]); Ic0.AA.xy += banked_sum<3>(t16[0]);
        Ic0.AA.xz += banked_sum<0>(t16[4]); Ic0.AA.yz += banked_sum<1>(t16[4]); Ic0.AL.r0.x += banked_sum<2>(t16[4]); Ic0.AL.r0.y += banked_sum<3>(t16[4]);
        Ic0.AL.r0.z += banked_sum<0>(t16[8]); Ic0.AL.r1.x += banked_sum<1>(t16[8]); Ic0.AL.r1.y += banked_sum<2>(t16[8]); Ic0.AL.r1.z += banked_sum<3>(t16[8]);
        Ic0.AL.r2.x += banked_sum<0>(t16[12]); Ic0.AL.r2.y += banked_sum<1>(t16[12]); Ic0.AL.r2.z += banked_sum<2>(t16[12]);
        wsumF = banked_sum<3>(t16[12]);
        Ic0.LL.xx += banked_sum<0>(t12[0]); Ic0.LL.yy += banked_sum<1>(t12[0]); Ic0.LL.zz += banked_sum<2>(t12[0]); Ic0.LL.xy += banked_sum<3>(t12[0]);
        Ic0.LL.xz += banked_sum<0>(t12[4]); Ic0.LL.yz += banked_sum<1>(t12[4]);
        SV b = {v3(banked_sum<2>(t12[4]) - p0.a.x, banked_sum<3>(t12[4]) - p0.a.y, banked_sum<0>(t12[8]) - p0.a.z),
                v3(banked_sum<1>(t12[8]) - p0.l.x, banked_sum<2>(t12[8]) - p0.l.y, banked_sum<3>(t12[8]) - p0.l.z)};
        if (__any(wsumF > 0.f)) {         // wave-uniform skip, as in the one-leg-per-lane kernel
            sF = env_sum(sF);
            Fr E0 = {v3(1.f, 0.f, 0.f), v3(0.f, 1.f, 0.f), v3(0.f, 0.f, 1.f)};
            SV fe;
            contact_finish(wsumF, sF, E0, v3(0.f, 0.f, 0.f), bc.n, bc.V0, C.contact_k, C.contact_c, C.contact_inv_ramp, C.contact_mu, h, fe, Ic0);
            b.a = b.a + fe.a;
            b.l = b.l + fe.l;
        }
        pk3::base_solve(Ic0, b, x6);
    }
    const V3 wdot = v3(x6[0], x6[1], x6[2]);
    const V3 acl = v3(x6[3], x6[4], x6[5]);
    if (want_sensors && lead_env) {
        row[12] = acl.x - bc.gb.x; row[13] = acl.y - bc.gb.y; row[14] = acl.z - bc.gb.z;   // accelerometer
    }
    // ---- hinge accelerations  qdd = H^-1 (b - F^T x)  with the factors every lane of the leg holds; the lane integrates ITS hinge ---------
    {
        const float g_o = fmaf(F.a.x, x6[0], fmaf(F.a.y, x6[1], fmaf(F.a.z, x6[2], fmaf(F.l.x, x6[3], fmaf(F.l.y, x6[4], F.l.z * x6[5])))));
        const float r0 = b0 - leg_bcast0(g_o), r1 = b1 - leg_bcast1(g_o), r2 = b2 - leg_bcast2(g_o);
        const float yy0 = r0, yy1 = fmaf(-l10, yy0, r1), yy2 = fmaf(-l21, yy1, fmaf(-l20, yy0, r2));
        const float qdd2 = yy2 * id2;
        const float qdd1 = fmaf(-l21, qdd2, yy1 * id1);
        const float qdd0 = fmaf(-l20, qdd2, fmaf(-l10, qdd1, yy0 * id0));
        J.qd = fmaf(h, sel3(rr, qdd0, qdd1, qdd2), J.qd);
        J.q = fmaf(h, J.qd, J.q);
        hinge_advance(h * J.qd, J.sn, J.cs);
        J.act = fmaf(J.u - J.act, K.act_decay, J.act);
    }
    pk3::base_integrate_unit(bc, h, wdot, acl, B);
}

// Workgroups of four waves (one per SIMD of a CU): a grid of 1024 one-wave workgroups measured 1.65 us more fixed time per launch
// than 256 (kernel time against frame_skip, tools/fs_sweep.sh: intercept 5.75 vs 4.1 us) -- the dispatch of the workgroups
// themselves; the waves do not interact (own tile rows, own envs).
#define QGK_LINK_WAVES 4
// WALK: the walking task layer fused in as in qg_step_kernel_quad<.., WALK>, one control channel per lane: lane r < 3 of leg k owns
// channel 3k + r (estimator update in the prologue, its terms of the reward sums in the epilogue), the env's lead lane evaluates
// the reward.
// PO (with WALK): the partially observable observation pack fused in as well (qg_po_dev.h) -- the whole POWalkingQuadrupedEnv.step
// (po_walking_quad.py:29-57,71-90) is this one launch.  The workgroup's 16 envs x 16 lanes are exactly the stand-alone kernel's
// layout (thread = 16 * local env + l16): the frames the new stack keeps are copied ring -> out in the prologue, the env's lead lane
// runs the orientation filter on the step's sensors in the epilogue, the 16 lanes of the env write the new frame.  The 33 sensors
// themselves are not written to memory at all.
// HELP (with WALK): four more waves per workgroup, one beside each physics wave on its SIMD, run what the step needs that does not depend
// on its physics -- the estimator update of the env's twelve channels (and the history copy of the observation pack) -- while the
// physics wave, which alone uses its SIMD's issue slots every fourth cycle only, goes straight into the substep loop; the estimates
// come back through LDS behind one workgroup barrier after the loop.  Same lane mapping in both waves (lane r < 3 of leg k = channel
// 3k + r of env el of wave w).
template <bool WALK = false, bool PO = false, bool BAKED = true, bool HELP = false>
__global__ __launch_bounds__(QGK_WAVE * QGK_LINK_WAVES * (HELP ? 2 : 1), 1) void qg_step_kernel_link(const KModel *__restrict__ Mp, const KTask *__restrict__ T, KStepArgs P,
                                                                                  const typename WalkArgT<WALK>::type WK,
                                                                                  const typename PoArgT<PO>::type PK) {
    static_assert(WALK || !PO, "the observation pack rides on the walking task layer");
    static_assert(WALK || !HELP, "the helper waves carry the walking task layer's estimator");
    static_assert(QGK_LINK_ENVS * QGK_LINK_WAVES == QG_PO_ENVS && QGK_WAVE * QGK_LINK_WAVES == QG_PO_THREADS, "workgroup layout of qg_po_dev.h");
    __shared__ float tile_all[QGK_LINK_WAVES][QGK_LINK_ENVS * 35];
    __shared__ KModel smodel;
    if constexpr (!BAKED) {                         // any other robot: the model tables staged in LDS, read with per-lane (leg) addresses
        const float *src = reinterpret_cast<const float *>(Mp);
        float *dst = reinterpret_cast<float *>(&smodel);
        for (int i = threadIdx.x; i < (int)(sizeof(KModel) / sizeof(float)); i += QGK_WAVE * QGK_LINK_WAVES * (HELP ? 2 : 1)) dst[i] = src[i];
        __syncthreads();
    }
    const KModel &C = BAKED ? QG_BAKED_MODEL : smodel;
    // the task constants into scalar registers up front: read where they are used, every read in the epilogue was its own
    // scalar-load round trip in front of a wave that has nothing else to do
    struct { int32_t frame_skip, limit_substeps, use_fall, use_flip, obs_mode, auto_reset; uint32_t reset_flags; float fall_height, w_forward, w_ctrl, alive_bonus;
             const float *default_ctrl; } Tk = {T->frame_skip, T->limit_substeps, T->use_fall, T->use_flip, T->obs_mode, T->auto_reset, T->reset_flags,
                                               T->fall_height, T->w_forward, T->w_ctrl, T->alive_bonus, T->default_ctrl};
    QG_MARK(0);
    const int lane = threadIdx.x & (QGK_WAVE - 1);
    const int wave = (threadIdx.x >> 6) & (QGK_LINK_WAVES - 1);      // HELP: waves 4 .. 7 shadow waves 0 .. 3
    const bool helper = HELP && (threadIdx.x >> 6) >= QGK_LINK_WAVES;
    float *tile = tile_all[wave];
    const int r = lane & 3;                         // link of this lane (3: spare)
    const int k = (lane >> 2) & 3;                  // leg
    const int el = lane >> 4;                       // env within the wave
    const int env0 = (blockIdx.x * QGK_LINK_WAVES + wave) * QGK_LINK_ENVS;
    const int n = P.n;
    const bool live = env0 + el < n;
    const int env = live ? env0 + el : n - 1;       // tail lanes shadow the last env; their stores are masked
    const bool lead_env = (lane & 15) == 0;
    const float cm = (k == 0) ? 1.f : (k == 2) ? -1.f : 0.f;
    const float sm = (k == 1) ? 1.f : (k == 3) ? -1.f : 0.f;

    constexpr bool RWDH = HELP && !PO;            // walking without the observation pack: the helper wave also evaluates the reward
    __shared__ float s_est[HELP ? QGK_LINK_WAVES : 1][QGK_WAVE][2];      // HELP: (f_est, a_est) of the lane's channel, helper -> physics wave
    __shared__ float s_done[RWDH ? QGK_LINK_WAVES : 1][QGK_LINK_ENVS];   // RWDH: the step's termination flags, physics -> helper
    __shared__ float s_new[PO ? QG_PO_ENVS : 1][QG_PO_FRAME];     // PO: the frame of this step
    __shared__ float s_rst[PO ? QG_PO_ENVS : 1][QG_PO_FRAME];     // PO: the frame reset() would return (only for envs that finished)
    __shared__ int s_slot[PO ? QG_PO_ENVS : 1], s_fin[PO ? QG_PO_ENVS : 1];
    __shared__ float s_hand[(HELP && PO) ? QG_PO_ENVS : 1][5];    // HELP + PO: done, data.qpos[3:7] as the step leaves it, physics -> helper
    if constexpr (HELP) {
        if (helper) {
            const int rk = r < 3 ? r : 2, jch = 3 * k + rk;
            const int tt[1] = {env * 12 + jch};
            float wf[1] = {0.f}, wa[1] = {0.f};
            // RWDH: what the reward needs and does not come out of the physics -- the env-clipped action (settling mask included), the
            // previous control, the channel's targets, the env's task state -- loaded now, ahead of the estimator's stores
            float h_aclip = 0.f, h_wprev = 0.f;
            WalkChanTargets h_wtg = {0.f, 0.f, 0.f};
            WalkEnvIn hwin = {};
            if constexpr (RWDH) {
                h_wtg = walk_channel_targets(WK.P, jch);
                float a_in = P.actions[(size_t)env * 12 + jch];
                if (P.st.nstep[env] < WK.P.settle_substeps) a_in = h_wtg.center;      // data.time < settling_time (walking_quad.py:142-143)
                h_aclip = fminf(fmaxf(a_in, -1.f), 1.f);                                // quadruped.py:160
                if (r < 3) h_wprev = WK.S.prev_ctrl[tt[0]];
                if (lead_env) {
                    hwin = walk_env_load(WK.S, n, env);
                    hwin.episode_key = P.st.episode[env];     // not advanced yet: the physics wave does that behind the barrier
                }
            }
            if (r < 3) {
                const int calls = WK.S.calls[env];
                const float xx[1] = {P.st.ctrl[jch * n + env]};       // data.ctrl of the PREVIOUS step (walking_quad.py:136)
                WalkEstIn<1> west;
                walk_estimator_load_n<1>(WK.P, WK.S, n, tt, calls, west, live);
                if (live) walk_estimator_finish_n<1>(WK.P, WK.S, n, tt, xx, calls, west, wf, wa);    // math_utils.py:53-131
            }
            if constexpr (!RWDH) {
                s_est[wave][lane][0] = wf[0];
                s_est[wave][lane][1] = wa[0];
            }
            if constexpr (PO) {
                const PoEnvIn pin0 = po_env_load(PK.S, n, env);
                int slot = pin0.head + 1;
                if (slot >= PK.P.window) slot = 0;
                if (PK.P.window > 1) {
                    PoHistRegs h;
                    po_copy_history_load(PK.P, PK.S, (size_t)env * (PK.P.window * QG_PO_FRAME), slot, lane & 15, PK.out, live, h);
                    po_copy_history_store(PK.P, (size_t)env * (PK.P.window * QG_PO_FRAME), lane & 15, PK.out, live, h);
                }
            }
            if constexpr (PO) {
                // ... and the observation pack's new frame: the physics wave hands over `done` and the base orientation behind the
                // barrier and goes on with the state stores and the reward while this wave runs the orientation filter and writes
                // the rows (the command is the one loaded at kernel entry: a command re-drawn by the physics wave for an env that
                // finished does not show in this step's frames, as in the one-role kernel)
                WalkEnvIn hw = {};
                if (lead_env) hw = walk_env_load(WK.S, n, env);
                const float hx = __shfl(hw.hx, lane & ~15), hy = __shfl(hw.hy, lane & ~15);
                const PoEnvIn pin = po_env_load(PK.S, n, env);
                __syncthreads();      // THE barrier of this role (helper, PO): its partner is the physics waves' __syncthreads() behind
                                      // their auto-reset block.  Every wave of the workgroup passes exactly one barrier on every path;
                                      // a second one in either role, or a return in front of it, deadlocks the workgroup.
                const int le = 4 * wave + el;
                if (live)
                    po_frame_env16(PK.P, PK.S, n, env, lane & 15, lead_env, pin, tile + el * 35, s_hand[le][1], s_hand[le][2], s_hand[le][3],
                                   s_hand[le][4], hw.cvx, hw.cvy, hx, hy, s_hand[le][0] != 0.f, s_new[le], s_rst[le], &s_slot[le], &s_fin[le]);
                wave_sync();
                po_emit_new(PK.P, PK.S, n, blockIdx.x * QG_PO_ENVS, le, lane & 15, s_new, s_rst, s_slot, s_fin, PK.out, PK.term_out);
                return;
            }
            __syncthreads();      // the one barrier of the workgroup (helper role without the observation pack; partner: the physics
                                  // waves' __syncthreads() behind their auto-reset block): behind it the physics waves write what this
                                  // wave read at entry.  Exactly one barrier per wave on every path -- see the note at the other site.
            if constexpr (RWDH) {
                // the reward of the step, on the sensor tile the physics wave has finished (LDS), while that wave stores the state
                // and writes the observation rows
                const bool hdone = s_done[wave][el] != 0.f;
                WalkSums sum = {0.f, 0.f, 0.f, 0.f};
                if (live && r < 3) {
                    walk_channel_terms(WK.S, env, jch, h_wtg, h_aclip, h_wprev, wf[0], wa[0], sum);
                    WK.S.prev_ctrl[tt[0]] = h_aclip;
                }
                sum.cost = env_sum(sum.cost); sum.posture = env_sum(sum.posture); sum.amp = env_sum(sum.amp); sum.frq = env_sum(sum.frq);
                if (live && lead_env)
                    walk_reward_env(WK.P, WK.S, n, env, tile + el * 35, sum, hwin, hdone, P.reward, WK.comps, WK.sample, P.seed, P.env_index_base);
            }
            return;
        }
    }

#include "qg_link_regs.inc"

    BaseState B;
    const unsigned n4 = 4u * (unsigned)n, e4 = 4u * (unsigned)env;        // byte strides of the [field][n] state arrays
    B.pw = v3(lk_ld(P.st.qpos, e4), lk_ld(P.st.qpos, n4 + e4), lk_ld(P.st.qpos, 2 * n4 + e4));
    B.qw = lk_ld(P.st.qpos, 3 * n4 + e4); B.qx = lk_ld(P.st.qpos, 4 * n4 + e4); B.qy = lk_ld(P.st.qpos, 5 * n4 + e4); B.qz = lk_ld(P.st.qpos, 6 * n4 + e4);
    B.vw = v3(lk_ld(P.st.qvel, e4), lk_ld(P.st.qvel, n4 + e4), lk_ld(P.st.qvel, 2 * n4 + e4));
    B.wb = v3(lk_ld(P.st.qvel, 3 * n4 + e4), lk_ld(P.st.qvel, 4 * n4 + e4), lk_ld(P.st.qvel, 5 * n4 + e4));
    quat_unit(B);   // unit quaternion once per launch (qg_set_state may hand in any length); the substeps keep it normalised
    const int nstep0 = lk_ld(P.st.nstep, e4);
    // this lane's hinge (the spare lane shadows hinge 2 of its leg: same loads, nothing of it is ever stored)
    const int rk = r < 3 ? r : 2;
    const int jch = 3 * k + rk;                      // hinge = control channel of this lane
    HingeLane J;
    float aclip;
    // WALK: every load of the task layer goes out among the state loads, every store of its prologue part after the last of them
    // (see qg_step_kernel_quad)
    const bool wch = live && r < 3;                 // this lane owns control channel 3k + r
    bool settle = false;
    int calls = 0;
    WalkEnvIn win = {};
    WalkChanTargets wtg = {0.f, 0.f, 0.f};
    const int tt[1] = {env * 12 + jch};               // task state: [n][12]
    float xx[1] = {0.f}, wprev = 0.f, wf[1] = {0.f}, wa[1] = {0.f}, a_eff = 0.f;
    WalkEstIn<1> west;
    if constexpr (WALK) {
        settle = nstep0 < WK.P.settle_substeps;                     // data.time < settling_time (walking_quad.py:142-143)
        if constexpr (!HELP) calls = WK.S.calls[env];
        wtg = walk_channel_targets(WK.P, jch);
        if (r < 3) {
            if constexpr (!RWDH) wprev = WK.S.prev_ctrl[tt[0]];            // previous_ctrl of the control cost (:260-262)
            if constexpr (!HELP) {
                xx[0] = P.st.ctrl[jch * n + env];     // data.ctrl of the PREVIOUS step: what the estimator takes (walking_quad.py:136)
                walk_estimator_load_n<1>(WK.P, WK.S, n, tt, calls, west, live);
            }
        }
        if (!RWDH && lead_env) {
            win = walk_env_load(WK.S, n, env);
            win.episode_key = P.st.episode[env];      // not advanced yet: the key of the episode that begins if this one ends
        }
    }
    PoEnvIn pin = {};
    if constexpr (PO) pin = po_env_load(PK.S, n, env);      // every lane of the env: the history copy below needs the ring position
    {
        float a_in = lk_ld(P.actions, 12u * e4 + 4u * (unsigned)jch);
        if constexpr (WALK) {
            if (settle) a_in = wtg.center;                          // the joint centres while the robot settles
            a_eff = a_in;
        }
        aclip = fminf(fmaxf(a_in, -1.f), 1.f);       // quadruped.py:160
        const KLink &Lj = link_of<BAKED>(C, k, rk);
        const float clo = BAKED ? sel3(rk, C.link[0].ctrl_lo, C.link[1].ctrl_lo, C.link[2].ctrl_lo) : Lj.ctrl_lo;
        const float chi = BAKED ? sel3(rk, C.link[0].ctrl_hi, C.link[1].ctrl_hi, C.link[2].ctrl_hi) : Lj.ctrl_hi;
        const float ref = BAKED ? sel3(rk, C.link[0].ref, C.link[1].ref, C.link[2].ref) : Lj.ref;
        J.u = fminf(fmaxf(aclip, clo), chi);
        const unsigned j4 = (unsigned)jch * n4 + e4;                        // hinge jch of this env within a [12][n] block
        J.q = lk_ld(P.st.qpos, 7 * n4 + j4);
        J.qd = lk_ld(P.st.qvel, 6 * n4 + j4);
        J.act = lk_ld(P.st.act, j4);
        sincos_f(J.q - ref, J.sn, J.cs);
    }
    if constexpr (WALK) {
        asm volatile("" :: "v"(B.pw.x), "v"(B.pw.y), "v"(B.pw.z), "v"(B.qw), "v"(B.qx), "v"(B.qy), "v"(B.qz), "v"(B.vw.x), "v"(B.vw.y), "v"(B.vw.z),
                     "v"(B.wb.x), "v"(B.wb.y), "v"(B.wb.z), "v"(J.q), "v"(J.qd), "v"(J.act) : "memory");
        if (wch) {
            if constexpr (!HELP) walk_estimator_finish_n<1>(WK.P, WK.S, n, tt, xx, calls, west, wf, wa);    // math_utils.py:53-131
            WK.S.eff_actions[(size_t)env * 12 + jch] = a_eff;   // the action actually applied (the PO pack reads it)
        }
    }
    PoHistRegs phist = {};
    if constexpr (PO) {
        int slot = pin.head + 1;
        if (slot >= PK.P.window) slot = 0;
        if (!HELP && PK.P.window > 1) po_copy_history_load(PK.P, PK.S, (size_t)env * (PK.P.window * QG_PO_FRAME), slot, lane & 15, PK.out, live, phist);
    }

    float *srow = tile + el * 35;
    float zaxis_z = 1.f;
    const int fs = Tk.frame_skip;
#ifdef QG_PHASE_TIMES
    asm volatile("" :: "v"(B.pw.x), "v"(B.qw), "v"(B.vw.x), "v"(B.wb.x), "v"(J.q), "v"(J.qd), "v"(J.act), "v"(J.sn), "v"(J.cs));
#endif
    QG_MARK(1);                                      // state in registers, prologue stores issued
    asm volatile(".p2align 6");
#pragma unroll 1
    for (int s = 0; s < fs; ++s) substep_link<BAKED>(C, cm, sm, r, lead_env, B, J, K, s == fs - 1, srow, k, zaxis_z);
    int nstep = nstep0 + fs;
    QG_MARK(2);                                      // physics done

    const float ssq = env_sum(r < 3 ? aclip * aclip : 0.f);
    float c_fwd = Tk.w_forward * B.vw.x;
    float c_ctl = Tk.w_ctrl * ssq;
    float c_alive = Tk.alive_bonus;
    float reward = reward_total(c_fwd, c_ctl, c_alive);
    bool done = nstep >= Tk.limit_substeps;
    if (Tk.use_fall) done = done || (B.pw.z < Tk.fall_height);
    {
        float probe = J.q + J.qd;
        probe = env_sum(r < 3 ? probe : 0.f) + B.pw.x + B.pw.y + B.pw.z + B.qw + B.vw.x + B.vw.y + B.vw.z + B.wb.x + B.wb.y + B.wb.z;
        done = done || state_is_bad(probe);
    }
    const int od = Tk.obs_mode == 1 ? 21 : 33;
    const int row = P.packed ? od + 2 : od;
    if (Tk.use_flip) done = done || (zaxis_z < 0.f);              // walking_quad.py:156-160, on the step's sensordata
    // The state goes out FIRST (two dozen of the launch's ~30 store instructions): it drains while the rest of the epilogue computes.
    const bool lead = live && lead_env;
    const bool rst = done && Tk.auto_reset;
    if (rst) {
        B.pw = v3(C.qpos0[0], C.qpos0[1], C.qpos0[2]);
        B.qw = C.qpos0[3]; B.qx = C.qpos0[4]; B.qy = C.qpos0[5]; B.qz = C.qpos0[6];
        if (Tk.reset_flags & 1u) {
            float a = 6.283185307179586f * uniform24(P.seed, P.env_index_base + (uint64_t)env, (uint64_t)P.st.episode[env]);
            float sn, cs;
            sincos_f(0.5f * a, sn, cs);
            B.qw = cs; B.qx = 0.f; B.qy = 0.f; B.qz = sn;
        }
        B.vw = v3(0.f, 0.f, 0.f);
        B.wb = v3(0.f, 0.f, 0.f);
        nstep = 0;
    }
    if constexpr (HELP) {
        if constexpr (PO) {
            const int le = 4 * wave + el;
            if (wch) s_new[le][11 + jch] = aclip;            // data.ctrl of the frame: the env-clipped action this step applied
            if (lead_env) {                                  // an aliasing estimate shows data.qpos[3:7] as the step leaves it: B after the auto-reset
                s_hand[le][0] = done ? 1.f : 0.f;
                s_hand[le][1] = B.qw; s_hand[le][2] = B.qx; s_hand[le][3] = B.qy; s_hand[le][4] = B.qz;
            }
        }
        // the helper wave of this SIMD finished its first job long ago (its estimator stores and history copy have landed: the barrier's
        // wait covers them); from here on this wave may overwrite what the helper read at entry (data.ctrl, the task state), and the
        // helper builds the observation pack's frame from the sensor tile, which nothing changes any more
        if constexpr (RWDH) {
            if (lead_env) {
                s_done[wave][el] = done ? 1.f : 0.f;
                if (od == 21) { srow[18] = srow[30]; srow[19] = srow[31]; srow[20] = srow[32]; }   // (the tile is final before the helper reads it)
            }
        }
        __syncthreads();          // partner of the helper waves' one barrier (either of its two sites above)
        if constexpr (!RWDH) {
            wf[0] = s_est[wave][lane][0];
            wa[0] = s_est[wave][lane][1];
        }
    }
    if (lead) {
        lk_st(P.st.qpos, e4, B.pw.x); lk_st(P.st.qpos, n4 + e4, B.pw.y); lk_st(P.st.qpos, 2 * n4 + e4, B.pw.z);
        lk_st(P.st.qpos, 3 * n4 + e4, B.qw); lk_st(P.st.qpos, 4 * n4 + e4, B.qx); lk_st(P.st.qpos, 5 * n4 + e4, B.qy); lk_st(P.st.qpos, 6 * n4 + e4, B.qz);
        lk_st(P.st.qvel, e4, B.vw.x); lk_st(P.st.qvel, n4 + e4, B.vw.y); lk_st(P.st.qvel, 2 * n4 + e4, B.vw.z);
        lk_st(P.st.qvel, 3 * n4 + e4, B.wb.x); lk_st(P.st.qvel, 4 * n4 + e4, B.wb.y); lk_st(P.st.qvel, 5 * n4 + e4, B.wb.z);
        lk_st(P.st.nstep, e4, nstep);
        if (rst) P.st.episode[env] += 1;
    }
    if (wch) {                                      // every link lane stores its own hinge
        const float q0 = BAKED ? sel3(rk, C.qpos0[7], C.qpos0[8], C.qpos0[9]) : C.qpos0[7 + jch];
        const unsigned j4 = (unsigned)jch * n4 + e4;
        lk_st(P.st.qpos, 7 * n4 + j4, rst ? q0 : J.q);
        lk_st(P.st.qvel, 6 * n4 + j4, rst ? 0.f : J.qd);
        lk_st(P.st.act, j4, rst ? 0.f : J.act);
        if (P.track_ctrl) lk_st(P.st.ctrl, j4, rst ? Tk.default_ctrl[jch] : aclip);
    }
    if (lead_env) {
        if (!RWDH && od == 21) { srow[18] = srow[30]; srow[19] = srow[31]; srow[20] = srow[32]; }   // IMU pack: velocimeter follows the gyro
        if (P.packed) { srow[od] = reward; srow[od + 1] = done ? 1.f : 0.f; }
    }
    wave_sync();                                                    // the tile is this wave's own
    if constexpr (!PO) {
        const int live_envs = max(0, min(QGK_LINK_ENVS, n - env0));     // a whole wave may lie past the last env
        const int total = live_envs * row;
        float *dst = (P.packed ? P.packed : P.obs) + (size_t)env0 * row;
        // every LDS read of the copy first, then the stores: in a loop with a run-time bound each pass waited out its own LDS latency
        if (row == 35) {
            float v[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) v[u] = tile[min(lane + u * QGK_WAVE, QGK_LINK_ENVS * 35 - 1)];
#pragma unroll
            for (int u = 0; u < 3; ++u) if (lane + u * QGK_WAVE < total) dst[lane + u * QGK_WAVE] = v[u];
        } else {
            // a row of the tile per pass (row < 64 lanes; the rows of a wave's envs are consecutive in `dst`): no division by `row`
            float v[QGK_LINK_ENVS];
#pragma unroll
            for (int er = 0; er < QGK_LINK_ENVS; ++er) v[er] = tile[er * 35 + min(lane, 34)];
#pragma unroll
            for (int er = 0; er < QGK_LINK_ENVS; ++er) if (er < live_envs && lane < row) dst[er * row + lane] = v[er];
        }
    }
    QG_MARK(3);                                      // obs tile written out
    if (lead && !P.packed) {
        if constexpr (!WALK) P.reward[env] = reward;
        P.done[env] = done ? 1 : 0;
    }
    if constexpr (WALK && !RWDH) {
        WalkSums sum = {0.f, 0.f, 0.f, 0.f};
        if (wch) {
            walk_channel_terms(WK.S, env, jch, wtg, aclip, wprev, wf[0], wa[0], sum);
            WK.S.prev_ctrl[tt[0]] = aclip;
        }
        sum.cost = env_sum(sum.cost); sum.posture = env_sum(sum.posture); sum.amp = env_sum(sum.amp); sum.frq = env_sum(sum.frq);
        QG_MARK(4);                                  // channel terms + sums
        if (lead) walk_reward_env(WK.P, WK.S, n, env, tile + el * 35, sum, win, done, P.reward, WK.comps, WK.sample, P.seed, P.env_index_base);
        QG_MARK(5);                                  // reward
    }
    if (lead && P.comps) {
        P.comps[(size_t)env * 3 + 0] = c_fwd;
        P.comps[(size_t)env * 3 + 1] = c_ctl;
        P.comps[(size_t)env * 3 + 2] = c_alive;
    }
    QG_MARK(6);
    if constexpr (PO && HELP) {
        // the frame and the rows are the helper wave's; the new episode's command is drawn here (the helper shows the old one)
        if (live && lead_env && done && Tk.auto_reset && PK.sample)
            walk_sample_command(WK.P, WK.S, n, env, P.seed, P.env_index_base, win.episode_key);
    } else if constexpr (PO) {
        const int le = threadIdx.x >> 4;                     // = 4 * wave + el
        if (wch) s_new[le][11 + jch] = aclip;                // data.ctrl of the frame: the env-clipped action this step applied
        if (live) {
            // the heading of the command, which the env's lead lane holds, into lane 2 as well (it evaluates the angle)
            const float hx = __shfl(win.hx, lane & ~15), hy = __shfl(win.hy, lane & ~15);
            // an aliasing estimate shows data.qpos[3:7] as the step leaves it: B after the auto-reset above
            po_frame_env16(PK.P, PK.S, n, env, lane & 15, lead_env, pin, srow, B.qw, B.qx, B.qy, B.qz, win.cvx, win.cvy, hx, hy, done,
                           s_new[le], s_rst[le], &s_slot[le], &s_fin[le]);
            // random_controls on the device: the new episode's command, drawn only now that both frames show the old one; the
            // env's episode counter has not been advanced yet
            if (lead_env && done && Tk.auto_reset && PK.sample)
                walk_sample_command(WK.P, WK.S, n, env, P.seed, P.env_index_base, win.episode_key);
        }
        wave_sync();                                         // the four envs of a wave are its own in every phase
        QG_MARK(7);                                  // frame built
        if (!HELP && PK.P.window > 1) po_copy_history_store(PK.P, (size_t)env * (PK.P.window * QG_PO_FRAME), lane & 15, PK.out, live, phist);
        po_emit_new(PK.P, PK.S, n, blockIdx.x * QG_PO_ENVS, le, lane & 15, s_new, s_rst, s_slot, s_fin, PK.out, PK.term_out);
    }
    QG_MARK(8);
    QG_MARK(9);                                      // every store issued
}
