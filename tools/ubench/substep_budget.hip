// Static VALU budget of the pieces of one substep (baked one-leg-per-lane form): each piece in its own probe kernel.
// build + count (from quadruped-gym_amd/csrc): hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffinite-math-only -fno-signed-zeros -fno-slp-vectorize \
//   -mllvm -amdgpu-sched-strategy=max-ilp --cuda-device-only -I. -S -o /tmp/budget.s ../../tools/ubench/substep_budget.hip ; count v_* per kernel
#include "qg_device.h"
#include "qg_kernels.hip"
// isolated pieces of one substep (baked quad form): static instruction budget of each
#define LD(i) in[(i) * 64 + threadIdx.x]
__device__ __forceinline__ BaseState ldB(const float *in) { BaseState B; B.pw = v3(LD(0), LD(1), LD(2)); B.qw = LD(3); B.qx = LD(4); B.qy = LD(5); B.qz = LD(6); B.vw = v3(LD(7), LD(8), LD(9)); B.wb = v3(LD(10), LD(11), LD(12)); return B; }
__global__ void p_prelude(const float *in, float *out) {
    const KModel &C = QG_BAKED_MODEL; BaseState B = ldB(in); BaseCtx c = base_prelude(C, B);
    float s = c.cx.x + c.cx.y + c.cx.z + c.cy.x + c.cy.y + c.cy.z + c.cz.x + c.cz.y + c.cz.z + c.n.x + c.n.y + c.n.z + c.gb.x + c.gb.y + c.gb.z + c.vb.x + c.vb.y + c.vb.z
            + c.V0.a.x + c.V0.a.y + c.V0.a.z + c.V0.l.x + c.V0.l.y + c.V0.l.z + c.A0.a.x + c.A0.a.y + c.A0.a.z + c.A0.l.x + c.A0.l.y + c.A0.l.z;
    out[threadIdx.x] = s;
}
__device__ __forceinline__ float sum6(const Sym6 &A) { return A.AA.xx + A.AA.yy + A.AA.zz + A.AA.xy + A.AA.xz + A.AA.yz + A.LL.xx + A.LL.yy + A.LL.zz + A.LL.xy + A.LL.xz + A.LL.yz + A.AL.r0.x + A.AL.r0.y + A.AL.r0.z + A.AL.r1.x + A.AL.r1.y + A.AL.r1.z + A.AL.r2.x + A.AL.r2.y + A.AL.r2.z; }
__device__ __forceinline__ BaseCtx ldC(const float *in) { BaseCtx c; float *p = (float *)&c; for (int i = 0; i < (int)(sizeof(BaseCtx) / 4); i++) p[i] = LD(20 + i); return c; }
__global__ void p_frame(const float *in, float *out) {
    const KModel &C = QG_BAKED_MODEL; BaseCtx c = ldC(in); SV p0; Sym6 I; frame_body(C, c, C.h, p0, I);
    out[threadIdx.x] = sum6(I) + p0.a.x + p0.a.y + p0.a.z + p0.l.x + p0.l.y + p0.l.z;
}
__global__ void p_legpass(const float *in, float *out) {
    const KModel &C = QG_BAKED_MODEL; BaseCtx c = ldC(in);
    float q[3] = {LD(0), LD(1), LD(2)}, qd[3] = {LD(3), LD(4), LD(5)}, act[3] = {LD(6), LD(7), LD(8)};
    Fr Ek = {v3(LD(9), LD(10), 0.f), v3(-LD(10), LD(9), 0.f), v3(0.f, 0.f, 1.f)};
    Sym6 Ic; SV fc, F[3]; float Hd[3], H01, H02, H12, bj[3];
    leg_pass<float, true, true, false>(C, 0, Ek, q, qd, act, c, LD(11), C.h, Ic, fc, F, Hd, H01, H02, H12, bj);
    float s = sum6(Ic) + fc.a.x + fc.a.y + fc.a.z + fc.l.x + fc.l.y + fc.l.z + Hd[0] + Hd[1] + Hd[2] + H01 + H02 + H12 + bj[0] + bj[1] + bj[2];
    for (int i = 0; i < 3; i++) s += F[i].a.x + F[i].a.y + F[i].a.z + F[i].l.x + F[i].l.y + F[i].l.z;
    out[threadIdx.x] = s;
}
__global__ void p_elim(const float *in, float *out) {
    SV F[3]; float Hd[3], bj[3];
    for (int i = 0; i < 3; i++) { F[i].a = v3(LD(6 * i), LD(6 * i + 1), LD(6 * i + 2)); F[i].l = v3(LD(6 * i + 3), LD(6 * i + 4), LD(6 * i + 5)); Hd[i] = LD(18 + i); bj[i] = LD(21 + i); }
    float Y0[6], Y1[6], Y2[6], u[3]; Sym6 YFt; SV Fu;
    leg_eliminate(F, Hd, LD(24), LD(25), LD(26), bj, Y0, Y1, Y2, u, YFt, Fu);
    float s = sum6(YFt) + Fu.a.x + Fu.a.y + Fu.a.z + Fu.l.x + Fu.l.y + Fu.l.z + u[0] + u[1] + u[2];
    for (int r = 0; r < 6; r++) s += Y0[r] + Y1[r] + Y2[r];
    out[threadIdx.x] = s;
}
__global__ void p_solve(const float *in, float *out) {
    Sym6 A; float *p = (float *)&A; for (int i = 0; i < 21; i++) p[i] = LD(i);
    SV b = {v3(LD(21), LD(22), LD(23)), v3(LD(24), LD(25), LD(26))}; float x[6]; base_solve(A, b, x);
    out[threadIdx.x] = x[0] + x[1] + x[2] + x[3] + x[4] + x[5];
}
__global__ void p_integrate(const float *in, float *out) {
    BaseState B = ldB(in); BaseCtx c = ldC(in); base_integrate(c, 0.002f, v3(LD(60), LD(61), LD(62)), v3(LD(63), LD(64), LD(65)), B);
    out[threadIdx.x] = B.pw.x + B.pw.y + B.pw.z + B.qw + B.qx + B.qy + B.qz + B.vw.x + B.vw.y + B.vw.z + B.wb.x + B.wb.y + B.wb.z;
}
__global__ void p_quadsum(const float *in, float *out) {
    Sym6 A; float *p = (float *)&A; for (int i = 0; i < 21; i++) p[i] = LD(i);
    quad_sum(A); V3 a = quad_sum(v3(LD(21), LD(22), LD(23))), b = quad_sum(v3(LD(24), LD(25), LD(26))), c = quad_sum(v3(LD(27), LD(28), LD(29))), d = quad_sum(v3(LD(30), LD(31), LD(32)));
    out[threadIdx.x] = sum6(A) + a.x + a.y + a.z + b.x + b.y + b.z + c.x + c.y + c.z + d.x + d.y + d.z;
}
__global__ void p_contact(const float *in, float *out) {
    const KModel &C = QG_BAKED_MODEL; Fr E = {v3(LD(0), LD(1), LD(2)), v3(LD(3), LD(4), LD(5)), v3(LD(6), LD(7), LD(8))};
    SV v = {v3(LD(9), LD(10), LD(11)), v3(LD(12), LD(13), LD(14))}; SV fe; Sym6 A; float *p = (float *)&A; for (int i = 0; i < 21; i++) p[i] = 0.f;
    body_contact<float, QGK_CP_LINK>(C.link[2].cp, E, v3(LD(15), LD(16), LD(17)), LD(18), v3(LD(19), LD(20), LD(21)), v, C.contact_k, C.contact_c, C.contact_inv_ramp, C.contact_margin, C.contact_mu, C.h, fe, A);
    out[threadIdx.x] = sum6(A) + fe.a.x + fe.a.y + fe.a.z + fe.l.x + fe.l.y + fe.l.z;
}
__global__ void p_base(const float *in, float *out) { out[threadIdx.x] = LD(0); }
