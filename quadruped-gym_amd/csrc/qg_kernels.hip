// qg_kernels.hip -- gfx950 kernels of the batched quadruped simulator.
//
// Replaces the arithmetic behind QuadrupedEnv.step() (src/envs/quadruped.py:153-182 of
// antopio26/quadruped-gym): frame_skip x mj_step (quadruped.py:163-165, integrator
// implicitfast, quadruped.xml:4), the sensor pack (quadruped.py:141-143, quadruped.xml:174-217),
// the README reward / termination set (README.md:64-90) and reset (quadruped.py:115-139).
//
// Two work mappings of the same physics (shared device functions below):
//   qg_step_kernel       ONE ENVIRONMENT PER WAVEFRONT LANE, 64 envs per wave; the legs are a rolled loop, the joint
//                        state and the per-leg factors live in per-lane LDS columns (they are indexed by the run-time
//                        leg index);
//   qg_step_kernel_quad  ONE LEG PER LANE, four lanes per environment, 16 envs per wave; leg terms are reduced over
//                        the quad with DPP adds, everything stays in registers.  ~3.5x fewer instructions per wave,
//                        which is what sets the time of an env-step at batch sizes that leave most SIMDs idle.
// State is struct-of-arrays in HBM ([field][env]) so a wave's loads and stores of a field are contiguous.  All frame_skip
// substeps run inside one launch; state is read once and written once per env-step.  No MFMA: the path is a chain of
// small (3x3 / 6x6) per-env solves, bound by VALU issue, not by HBM.
// Constants: BAKED variants read the compiled-in robot from a const table with compile-time indices (instruction
// literals); generic variants read any model numbers from device memory (scalar loads / an LDS copy).
//
// Formulation (different on purpose from the CPU oracle, which works in world coordinates with a dense 18x18 solve):
// everything is expressed in the FRAME's own axes about the FRAME's origin.  Per substep
//   A. base prelude: rotation from the quaternion, base velocity / bias acceleration, FRAME body force and ground
//      contact, start of the base 6x6 block;
//   B. per leg (leg_pass): kinematics of fema/shin/foot, recursive Newton-Euler bias forces, ground contact,
//      composite-rigid-body inertia (mass matrix columns), servo / limit / damping terms, then block elimination of the
//      leg's 3x3 joint block into the base block (leg_eliminate) -- the mass matrix is base 6x6 + four 3x3 leg blocks +
//      four 6x3 couplings and is never formed densely;
//   C. 6x6 base solve (LDL^T);
//   D. back-substitution, semi-implicit integration of the hinges, servo activation filter;
//   E. base integration (position, quaternion).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qg_device.h"
#include "qg_model_baked.h"

#define DEV __device__ __forceinline__
#define QG_STR2(x) #x
#define QG_STR(x) QG_STR2(x)

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
// scalar layer.  Every physics function below is a template on the scalar type T:
//   T = float  one leg (or one env) per lane -- the lane and quad kernels;
//   T = f2     a 2-wide ext-vector: TWO LEGS per lane.  hipcc turns fma / mul / add on f2 into v_pk_fma_f32 /
//              v_pk_mul_f32 / v_pk_add_f32 -- two legs for the issue slot of one, with no shuffles because every
//              quantity of the pair lives in an aligned register pair from the start (measured: identical instruction
//              count to the float build of the same code, all packed, zero v_mov).  max / min / select / rcp / sqrt /
//              compares have no packed f32 form on gfx950 and run once per component.
// Conditionals are written as masks + select so that the same source serves both.
// ------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));

DEV float fma_(float a, float b, float c) { return fmaf(a, b, c); }
DEV f2 fma_(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
DEV float max_(float a, float b) { return fmaxf(a, b); }
DEV f2 max_(f2 a, f2 b) { f2 r = {fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; return r; }
DEV float min_(float a, float b) { return fminf(a, b); }
DEV f2 min_(f2 a, f2 b) { f2 r = {fminf(a.x, b.x), fminf(a.y, b.y)}; return r; }
DEV float abs_(float a) { return fabsf(a); }
DEV f2 abs_(f2 a) { f2 r = {fabsf(a.x), fabsf(a.y)}; return r; }
DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
DEV f2 rcp(f2 x) { f2 r = {__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; return r; }
DEV float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
DEV f2 sqrt_(f2 x) { f2 r = {__builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y)}; return r; }
// masks: bool for float, i2 (-1 / 0 per component, what a vector compare yields) for f2
DEV float sel(bool m, float a, float b) { return m ? a : b; }
DEV f2 sel(i2 m, f2 a, f2 b) { f2 r = {m.x ? a.x : b.x, m.y ? a.y : b.y}; return r; }
DEV bool m_and(bool a, bool b) { return a && b; }
DEV i2 m_and(i2 a, i2 b) { return a & b; }
DEV bool m_or(bool a, bool b) { return a || b; }
DEV i2 m_or(i2 a, i2 b) { return a | b; }
DEV bool m_not(bool a) { return !a; }
DEV i2 m_not(i2 a) { return ~a; }
DEV bool m_any(bool a) { return a; }
DEV bool m_any(i2 a) { return (a.x | a.y) != 0; }
template <class T> DEV T splat(float x) { return T(x); }

// ------------------------------------------------------------------------------------------
// small fixed-size algebra
// ------------------------------------------------------------------------------------------
template <class T> struct V3T { T x, y, z; };
typedef V3T<float> V3;
template <class T> DEV V3T<T> v3(T x, T y, T z) { V3T<T> r = {x, y, z}; return r; }
template <class T> DEV V3T<T> operator+(V3T<T> a, V3T<T> b) { return v3<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class T> DEV V3T<T> operator-(V3T<T> a, V3T<T> b) { return v3<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class T> DEV V3T<T> operator*(T s, V3T<T> a) { return v3<T>(s * a.x, s * a.y, s * a.z); }
template <class T> DEV T dot(V3T<T> a, V3T<T> b) { return fma_(a.x, b.x, fma_(a.y, b.y, a.z * b.z)); }
template <class T> DEV V3T<T> cross(V3T<T> a, V3T<T> b) { return v3<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
template <class T> DEV V3T<T> fma3(T s, V3T<T> a, V3T<T> b) { return v3<T>(fma_(s, a.x, b.x), fma_(s, a.y, b.y), fma_(s, a.z, b.z)); }
// Packed FP32 for the kernel that is ALONE on its SIMD (one link per lane, qg_kernel_link.hip): a lone wave issues one VALU instruction
// per >= 4 cycles whatever it is, so (x, y) of a 3-vector operation as ONE v_pk_*_f32 is an instruction saved (z stays a plain one).
// At two waves per SIMD a packed instruction costs the two plain ones it replaces (tools/ubench/vgpr_bank), and with the generic
// robot's register budget the aligned pairs spill -- hence a namespace that only substep_link() opens (`using namespace pk3`: these
// non-template overloads then win over the templates above), not a change of the templates.  Same arithmetic, same rounding: a
// packed multiply-add is two IEEE multiply-adds; the compiler takes a broadcast or a swap of a pair's halves as op_sel modifiers.
namespace pk3 {
DEV V3 fma3(float s, V3 a, V3 b) { f2 A = {a.x, a.y}, Bv = {b.x, b.y}, S = {s, s}; f2 r = __builtin_elementwise_fma(S, A, Bv); return v3<float>(r.x, r.y, fmaf(s, a.z, b.z)); }
DEV V3 operator+(V3 a, V3 b) { f2 A = {a.x, a.y}, Bv = {b.x, b.y}; f2 r = A + Bv; return v3<float>(r.x, r.y, a.z + b.z); }
DEV V3 operator-(V3 a, V3 b) { f2 A = {a.x, a.y}, Bv = {b.x, b.y}; f2 r = A - Bv; return v3<float>(r.x, r.y, a.z - b.z); }
DEV V3 operator*(float s, V3 a) { f2 A = {a.x, a.y}, S = {s, s}; f2 r = S * A; return v3<float>(r.x, r.y, s * a.z); }
// a x b with its (x, y) half as two packed instructions, (a.y b.z - a.z b.y, a.z b.x - a.x b.z) = a.z (-b.y, b.x) + (a.y, -a.x) b.z:
// the swaps and signs are operand modifiers (written out: the compiler folds a swap only now and then); a.z and b.z ride in the
// low half of a pair whose other half is never read and is left unset on purpose (any initialiser is a v_mov).  Four instructions
// for six; the product a.z b is rounded before the multiply-add (the compiler's own contraction rounds one of the two products too).
// (an operand with a component known at compile time -- the compiled-in robot's mount axes -- takes the plain form, which folds)
DEV bool has_literal(V3 a) { return __builtin_constant_p(a.x) || __builtin_constant_p(a.y) || __builtin_constant_p(a.z); }
DEV V3 cross(V3 a, V3 b) {
    if (has_literal(a) || has_literal(b)) return ::cross<float>(a, b);
    f2 A = {a.x, a.y}, Bv = {b.x, b.y}, az, bz, m, r;
    az.x = a.z; bz.x = b.z;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[1,0]" : "=v"(m) : "v"(az), "v"(Bv));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(A), "v"(bz), "v"(m));
    return v3<float>(r.x, r.y, a.x * b.y - a.y * b.x);
}
// acc + a x b: the same two packed instructions with the sum as their addend (acc - a x b is cross_add(b, a, acc))
DEV V3 cross_add(V3 a, V3 b, V3 acc) {
    if (has_literal(a) || has_literal(b)) return ::operator+<float>(acc, ::cross<float>(a, b));
    f2 A = {a.x, a.y}, Bv = {b.x, b.y}, C2 = {acc.x, acc.y}, az, bz, m, r;
    az.x = a.z; bz.x = b.z;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]" : "=v"(m) : "v"(az), "v"(Bv), "v"(C2));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(A), "v"(bz), "v"(m));
    return v3<float>(r.x, r.y, fmaf(a.x, b.y, fmaf(-a.y, b.x, acc.z)));
}
}  // namespace pk3
DEV V3 ld3(const float *p) { return v3<float>(p[0], p[1], p[2]); }
template <class T> DEV V3T<T> ld3t(const float *p) { return v3<T>(T(p[0]), T(p[1]), T(p[2])); }     // a model constant, same for every component
template <class T> DEV V3T<T> splat3(V3 a) { return v3<T>(T(a.x), T(a.y), T(a.z)); }

// orthonormal frame given by its three axes (columns) expressed in the working frame
template <class T> struct FrT { V3T<T> ex, ey, ez; };
typedef FrT<float> Fr;
template <class T> DEV V3T<T> rot(const FrT<T> &E, V3T<T> r) { return fma3(r.x, E.ex, fma3(r.y, E.ey, r.z * E.ez)); }   // local -> working
template <class T> DEV V3T<T> rotT(const FrT<T> &E, V3T<T> r) { return v3<T>(dot(E.ex, r), dot(E.ey, r), dot(E.ez, r)); }  // working -> local

template <class T> struct Sym3T { T xx, yy, zz, xy, xz, yz; };
typedef Sym3T<float> Sym3;
template <class T> DEV V3T<T> mul(const Sym3T<T> &S, V3T<T> v) {
    return v3<T>(fma_(S.xx, v.x, fma_(S.xy, v.y, S.xz * v.z)), fma_(S.xy, v.x, fma_(S.yy, v.y, S.yz * v.z)),
                 fma_(S.xz, v.x, fma_(S.yz, v.y, S.zz * v.z)));
}
template <class T> DEV void add(Sym3T<T> &a, const Sym3T<T> &b) { a.xx += b.xx; a.yy += b.yy; a.zz += b.zz; a.xy += b.xy; a.xz += b.xz; a.yz += b.yz; }
template <class T> DEV void rank1(Sym3T<T> &a, T w, V3T<T> u, V3T<T> v) {  // a += w * (u v^T), caller guarantees symmetry (u == v)
    a.xx = fma_(w * u.x, v.x, a.xx); a.yy = fma_(w * u.y, v.y, a.yy); a.zz = fma_(w * u.z, v.z, a.zz);
    a.xy = fma_(w * u.x, v.y, a.xy); a.xz = fma_(w * u.x, v.z, a.xz); a.yz = fma_(w * u.y, v.z, a.yz);
}

template <class T> struct M3T { V3T<T> r0, r1, r2; };  // rows
typedef M3T<float> M3;
template <class T> DEV V3T<T> mul(const M3T<T> &A, V3T<T> v) { return v3<T>(dot(A.r0, v), dot(A.r1, v), dot(A.r2, v)); }
template <class T> DEV V3T<T> mulT(const M3T<T> &A, V3T<T> v) { return fma3(v.x, A.r0, fma3(v.y, A.r1, v.z * A.r2)); }
template <class T> DEV void add(M3T<T> &a, const M3T<T> &b) { a.r0 = a.r0 + b.r0; a.r1 = a.r1 + b.r1; a.r2 = a.r2 + b.r2; }

// spatial vectors [angular; linear] about the FRAME origin, FRAME axes
template <class T> struct SVT { V3T<T> a, l; };
typedef SVT<float> SV;
template <class T> DEV SVT<T> operator+(SVT<T> p, SVT<T> q) { SVT<T> r = {p.a + q.a, p.l + q.l}; return r; }
template <class T> DEV T dot(SVT<T> p, SVT<T> q) { return dot(p.a, q.a) + dot(p.l, q.l); }
template <class T> DEV SVT<T> splat6(SV v) { SVT<T> r = {splat3<T>(v.a), splat3<T>(v.l)}; return r; }

// rigid-body spatial inertia about the FRAME origin: mass, first moment h = m*c, rotational inertia
template <class T> struct RigidT { T m; V3T<T> h; Sym3T<T> I; };
typedef RigidT<float> Rigid;
template <class T> DEV SVT<T> mul(const RigidT<T> &B, SVT<T> v) {
    SVT<T> f;
    f.a = mul(B.I, v.a) + cross(B.h, v.l);
    f.l = B.m * v.l - cross(B.h, v.a);
    return f;
}

// general symmetric 6x6 (rigid inertia + implicit contact damping): [[AA, AL], [AL^T, LL]]
template <class T> struct Sym6T { Sym3T<T> AA; M3T<T> AL; Sym3T<T> LL; };
typedef Sym6T<float> Sym6;
template <class T> DEV SVT<T> mul(const Sym6T<T> &A, SVT<T> s) {
    SVT<T> f;
    f.a = mul(A.AA, s.a) + mul(A.AL, s.l);
    f.l = mulT(A.AL, s.a) + mul(A.LL, s.l);
    return f;
}
template <class T> DEV void add(Sym6T<T> &a, const Sym6T<T> &b) { add(a.AA, b.AA); add(a.AL, b.AL); add(a.LL, b.LL); }
namespace pk3 {      // (see above) matrix-vector products on the (x, y) + z split: (xx, yy) and (xz, yz) are pairs, xy enters through the swapped vector
// six-term dot product of two spatial vectors: the (x, y) products of both halves packed, five instructions for six
DEV float dot(SV p, SV q) {
    if (has_literal(p.a) || has_literal(p.l) || has_literal(q.a) || has_literal(q.l)) return ::dot<float>(p, q);
    const f2 t = __builtin_elementwise_fma(f2{p.l.x, p.l.y}, f2{q.l.x, q.l.y}, f2{p.a.x, p.a.y} * f2{q.a.x, q.a.y});
    return fmaf(p.l.z, q.l.z, fmaf(p.a.z, q.a.z, t.x)) + t.y;
}
// s * (v.y, v.x) + acc in one instruction: the swap of a pair's halves is an op_sel modifier (the compiler folds a broadcast into
// op_sel, a swap only now and then -- two v_mov otherwise); s rides in the low half of a pair whose high half is never read
DEV f2 fma_swapped(float s, f2 v, f2 acc) {
    f2 sp, r;                                                  // (s, don't care): only the low half is selected below.  Left
    sp.x = s;                                                  // unset on purpose: __builtin_nondeterministic_value (frozen poison)
                                                               // becomes a v_mov from a zero register per call
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(sp), "v"(v), "v"(acc));
    return r;
}
DEV V3 mul(const Sym3 &S, V3 v) {
    const f2 d = {S.xx, S.yy}, o = {S.xz, S.yz}, vxy = {v.x, v.y}, vz = {v.z, v.z};
    const f2 r = __builtin_elementwise_fma(d, vxy, fma_swapped(S.xy, vxy, o * vz));
    return v3<float>(r.x, r.y, fmaf(S.xz, v.x, fmaf(S.yz, v.y, S.zz * v.z)));
}
DEV V3 mulT(const M3 &A, V3 v) { return fma3(v.x, A.r0, fma3(v.y, A.r1, v.z * A.r2)); }
DEV V3 rot(const Fr &E, V3 r) { return fma3(r.x, E.ex, fma3(r.y, E.ey, r.z * E.ez)); }   // local -> working
DEV SV mul(const Rigid &B, SV v) {
    SV f;
    f.a = cross_add(B.h, v.l, mul(B.I, v.a));
    f.l = cross_add(v.a, B.h, B.m * v.l);               // m v - h x w
    return f;
}
DEV SV mul(const Sym6 &A, SV s) {
    SV f;
    f.a = mul(A.AA, s.a) + ::mul(A.AL, s.l);
    f.l = mulT(A.AL, s.a) + mul(A.LL, s.l);
    return f;
}
}  // namespace pk3
template <class T> DEV Sym6T<T> sym6_of(const RigidT<T> &B) {
    Sym6T<T> A;
    const T z = T(0.f);
    A.AA = B.I;
    A.AL.r0 = v3<T>(z, -B.h.z, B.h.y);   // [h]x
    A.AL.r1 = v3<T>(B.h.z, z, -B.h.x);
    A.AL.r2 = v3<T>(-B.h.y, B.h.x, z);
    A.LL.xx = A.LL.yy = A.LL.zz = B.m;
    A.LL.xy = A.LL.xz = A.LL.yz = z;
    return A;
}
// A += m * (point mass at r)  +  w * a a^T with a = [r x n; n]
template <class T> DEV void add_contact_damping(Sym6T<T> &A, T m, T w, V3T<T> r, V3T<T> n) {
    T rr = dot(r, r);
    A.AA.xx += m * (rr - r.x * r.x); A.AA.yy += m * (rr - r.y * r.y); A.AA.zz += m * (rr - r.z * r.z);
    A.AA.xy -= m * r.x * r.y; A.AA.xz -= m * r.x * r.z; A.AA.yz -= m * r.y * r.z;
    V3T<T> h = m * r;
    const T z = T(0.f);
    A.AL.r0 = A.AL.r0 + v3<T>(z, -h.z, h.y);
    A.AL.r1 = A.AL.r1 + v3<T>(h.z, z, -h.x);
    A.AL.r2 = A.AL.r2 + v3<T>(-h.y, h.x, z);
    A.LL.xx += m; A.LL.yy += m; A.LL.zz += m;
    V3T<T> ra = cross(r, n);
    rank1(A.AA, w, ra, ra);
    A.AL.r0 = fma3(w * ra.x, n, A.AL.r0);
    A.AL.r1 = fma3(w * ra.y, n, A.AL.r1);
    A.AL.r2 = fma3(w * ra.z, n, A.AL.r2);
    rank1(A.LL, w, n, n);
}

// sin and cos with Cody-Waite reduction to [-pi/4, pi/4] and minimax polynomials (~1 ulp for |x| < 1e4)
DEV void sincos_quadrant(float k, float sr, float cr, float &s, float &c) {
    int q = (int)k;
    float s0 = (q & 1) ? cr : sr;
    float c0 = (q & 1) ? sr : cr;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
}
DEV float rint_(float x) { return rintf(x); }
DEV f2 rint_(f2 x) { f2 r = {rintf(x.x), rintf(x.y)}; return r; }
template <class T> DEV void sincos_f(T x, T &s, T &c) {
    T k = rint_(x * 0.63661977236758134f);
    T r = fma_(-k, T(1.57079625129699707031f), x);
    r = fma_(-k, T(7.54978941586159635335e-08f), r);
    T r2 = r * r;
    T sp = fma_(r2, fma_(r2, fma_(r2, T(2.718311493989822e-06f), T(-1.984090162742e-04f)), T(8.333329385889463e-03f)), T(-1.666666597127914e-01f));
    T sr = fma_(r * r2, sp, r);
    T cp = fma_(r2, fma_(r2, fma_(r2, T(2.443315711809948e-05f), T(-1.388731625493765e-03f)), T(4.166664568298827e-02f)), T(-0.5f));
    T cr = fma_(r2, cp, T(1.0f));
    if constexpr (sizeof(T) == sizeof(float)) {
        sincos_quadrant(k, sr, cr, s, c);
    } else {
        float s0, c0, s1, c1;
        sincos_quadrant(k.x, sr.x, cr.x, s0, c0);
        sincos_quadrant(k.y, sr.y, cr.y, s1, c1);
        s.x = s0; s.y = s1; c.x = c0; c.y = c1;
    }
}

// sin / cos of a hinge rotation advanced by the substep's own increment d = h * qd_new (|d| < 0.05 rad: the 5th-order series is
// exact to f32 rounding) instead of re-evaluated from the angle: 10 instructions per hinge and substep against 25.  The env-step
// starts from the polynomial values, so the rounding of at most frame_skip products accumulates (measured inside the stated
// tolerances at frame_skip 4 and 20).
template <class T> DEV void hinge_advance(T d, T &sn, T &cs) {
    T d2 = d * d;
    T sd = d * fma_(d2, fma_(d2, T(1.f / 120.f), T(-1.f / 6.f)), T(1.f));
    T cd = fma_(d2, fma_(d2, T(1.f / 24.f), T(-0.5f)), T(1.f));
    T s1 = fma_(sn, cd, cs * sd);
    cs = fma_(cs, cd, -(sn * sd));
    sn = s1;
}

// Divergence guard.  The engine being replaced checks positions / velocities / accelerations every step and resets
// a simulation that produced NaN / Inf or huge values (mj_checkPos / mj_checkVel / mj_checkAcc); here such an env is
// reported as done (and reset when auto_reset is on).  The device pass is compiled with -ffinite-math-only, so the
// test reads the exponent bits of a sum of state values instead of using isnan / isinf.
DEV bool state_is_bad(float probe) {
    unsigned bits = __builtin_bit_cast(unsigned, probe);
    return ((bits >> 23) & 0xFFu) >= 0x9Eu;          // NaN, Inf, or magnitude >= 2^31
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
// independent streams of the same (seed, env, episode) key: 0 = reset yaw, 1..12 = hinge jitter, 13..15 = walking command
#define QG_STREAM_HINGE 1u
#define QG_STREAM_COMMAND 13u
DEV float uniform24s(uint64_t seed, uint64_t env_index, uint64_t counter, uint32_t stream) {
    return uniform24(seed + 0xA0761D6478BD642Full * (uint64_t)stream, env_index, counter);
}
// In-kernel phase clock (development builds only: -DQG_PHASE_TIMES, tools/phase_times.sh).  Wave 0 of workgroup 0 stamps
// s_memrealtime (100 MHz) at the marks below into qg_phase_times[]; qg_debug_phase_times() copies them out and tools/phase_times.py
// prints the deltas.  Compiled out otherwise: the production kernels carry none of it.
#ifdef QG_PHASE_TIMES
__device__ unsigned long long qg_phase_times[16];
#define QG_MARK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { __builtin_amdgcn_sched_barrier(0); qg_phase_times[i] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define QG_MARK(i) do { } while (0)
#endif

// Hand-off through LDS between the lanes of ONE wave (a tile no other wave touches): the wave's LDS operations execute in order, so
// no s_barrier is needed -- in a four-wave workgroup that would also make every wave wait for the slowest of the four -- only the
// compiler has to be told that other lanes read what this lane wrote.
DEV void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#include "qg_walk_dev.h"     // walking task layer: per-env device functions used by the fused walking variant of the quad kernel
#include "qg_po_dev.h"       // partially observable observation pack: per-env / per-wave device functions of the fused <WALK, PO> variants

// start value of hinge j at a reset with QG_RESET_JOINT_JITTER: qpos0 + jitter * U(-1, 1), kept inside the joint range
DEV float jittered_hinge(float q0, float lo, float hi, float jitter, uint64_t seed, uint64_t env_index, int episode, int j) {
    float u = uniform24s(seed, env_index, (uint64_t)episode, QG_STREAM_HINGE + (uint32_t)j);
    return fminf(fmaxf(fmaf(jitter, 2.f * u - 1.f, q0), lo), hi);
}

// ------------------------------------------------------------------------------------------
// ground contact of one body: LCP-free penalty model, one aggregated contact per body
//   W   = sum_i k * max(0, margin - z_i)      spring force of the sample points below the margin
//   P   = centre of pressure of those spring forces
//   c   = contact_c * min(1, sum_i pen_i / ramp)      the damper ramps in with depth (continuous force)
//   F_n = max(0, W - c v_n(P));   F_t = -min(c, mu F_n / |v_t|) v_t(P)
// The damper is linear in the velocity with secant coefficients (c_n, c_t); it enters the
// system matrix as h * (c_t * point-mass(P) + (c_n - c_t) a a^T), a = [P x n; n].
// Split in two so that the sample points of one body can be shared out over several lanes.
// ------------------------------------------------------------------------------------------
template <class T> DEV void contact_point(V3T<T> r, V3T<T> nl, T zb, T &wsum, V3T<T> &s) {
    // zb - nl . r as three multiply-adds onto zb (round 3: one instruction fewer per sample point than dot + subtract)
    T pen = max_(fma_(-nl.x, r.x, fma_(-nl.y, r.y, fma_(-nl.z, r.z, zb))), T(0.f));
    wsum += pen;
    s = fma3(pen, r, s);
}
// The implicit (velocity-proportional) part of one aggregated contact: h * (ct * point-mass(P) + (cn - ct) * a a^T), a = [P x n; n],
// kept as five numbers until the composite inertia it belongs to is assembled.
template <class T> struct ContactDampT { T mc, w; V3T<T> P; };
#define QG_TEMPLATE_T template <class T>
#define QG_CROSS_ADD(a, b, acc) ((acc) + cross((a), (b)))
#include "qg_contact_eval.inc"
#undef QG_CROSS_ADD
#undef QG_TEMPLATE_T
namespace pk3 {
typedef float T;
#define QG_TEMPLATE_T
#define QG_CROSS_ADD(a, b, acc) cross_add((a), (b), (acc))          // the packed accumulate form
#include "qg_contact_eval.inc"
#undef QG_CROSS_ADD
#undef QG_TEMPLATE_T
// add_contact_damping on the pairs the packed code keeps a symmetric block in: (xx, yy), (xz, yz) and the (x, y) halves of AL's rows
DEV void add_contact_damping(Sym6 &A, float m, float w, V3 r, V3 n) {
    const float rr = dot(r, r);
    const V3 h = m * r;
    const f2 hxy = {h.x, h.y}, rxy = {r.x, r.y}, nxy = {n.x, n.y};
    const float mrr = m * rr;
    // point mass at r: m (r.r I - r r^T), [h]x, m I
    const f2 aad = f2{A.AA.xx, A.AA.yy} + __builtin_elementwise_fma(-hxy, rxy, f2{mrr, mrr});
    const f2 aao = __builtin_elementwise_fma(-hxy, f2{r.z, r.z}, f2{A.AA.xz, A.AA.yz});
    A.AA.zz += fmaf(-h.z, r.z, mrr);
    A.AA.xy = fmaf(-h.x, r.y, A.AA.xy);
    A.AL.r0.y -= h.z; A.AL.r0.z += h.y;
    A.AL.r1.x += h.z; A.AL.r1.z -= h.x;
    const f2 r2xy = f2{A.AL.r2.x, A.AL.r2.y} + f2{-h.y, h.x};
    const f2 lld = f2{A.LL.xx, A.LL.yy} + f2{m, m};
    A.LL.zz += m;
    // w a a^T, a = [r x n; n]
    const V3 ra = cross(r, n), wra = w * ra, wn = w * n;
    const f2 raxy = {ra.x, ra.y}, wraxy = {wra.x, wra.y}, wnxy = {wn.x, wn.y};
    const f2 aad2 = __builtin_elementwise_fma(wraxy, raxy, aad);
    const f2 aao2 = __builtin_elementwise_fma(wraxy, f2{ra.z, ra.z}, aao);
    A.AA.xx = aad2.x; A.AA.yy = aad2.y; A.AA.xz = aao2.x; A.AA.yz = aao2.y;
    A.AA.xy = fmaf(wra.x, ra.y, A.AA.xy); A.AA.zz = fmaf(wra.z, ra.z, A.AA.zz);
    A.AL.r0 = fma3(wra.x, n, A.AL.r0);
    A.AL.r1 = fma3(wra.y, n, A.AL.r1);
    A.AL.r2 = fma3(wra.z, n, v3<float>(r2xy.x, r2xy.y, A.AL.r2.z));
    const f2 lld2 = __builtin_elementwise_fma(wnxy, nxy, lld);
    const f2 llo = __builtin_elementwise_fma(wnxy, f2{n.z, n.z}, f2{A.LL.xz, A.LL.yz});
    A.LL.xx = lld2.x; A.LL.yy = lld2.y; A.LL.xz = llo.x; A.LL.yz = llo.y;
    A.LL.xy = fmaf(wn.x, n.y, A.LL.xy); A.LL.zz = fmaf(wn.z, n.z, A.LL.zz);
}
}  // namespace pk3
template <class T>
DEV void contact_finish(T wsum, V3T<T> s, const FrT<T> &E, V3T<T> p, V3T<T> n, SVT<T> v, float kc, float cmax, float inv_ramp, float mu, float h,
                        SVT<T> &f_ext, Sym6T<T> &A) {
    ContactDampT<T> cd;
    contact_eval(wsum, s, E, p, n, v, kc, cmax, inv_ramp, mu, h, f_ext, cd);
    add_contact_damping(A, cd.mc, cd.w, cd.P, n);
}
// the sample points of one body: penetration sum and penetration-weighted position sum.  (Tried in round 2 for one leg per lane:
// taking the points in PAIRS as packed FP32 with the pair's constants in scalar register pairs -- 108 instructions fewer per
// substep, 182 v_pk_* in the kernel, and no gain: 18.30 vs 18.22 us at 4096 envs, 21.0 vs 20.9 at 16 384; a v_pk_* with a
// constant-bus operand costs a lone wave about two plain issue slots.)
template <class T, int NCP>
DEV void contact_points(const float (*cp)[3], V3T<T> nl, T zb, T &wsum, V3T<T> &s) {
    wsum = T(0.f);
    s = v3<T>(T(0.f), T(0.f), T(0.f));
#pragma unroll
    for (int i = 0; i < NCP; ++i) contact_point(ld3t<T>(cp[i]), nl, zb, wsum, s);
}
template <class T, int NCP>
DEV void body_contact(const float (*cp)[3], const FrT<T> &E, V3T<T> p, T z_origin, V3T<T> n, SVT<T> v, float kc, float cmax, float inv_ramp,
                      float margin, float mu, float h, SVT<T> &f_ext, Sym6T<T> &A) {
    V3T<T> nl = rotT(E, n);                 // world up in the body's own axes
    T wsum;
    V3T<T> s;
    contact_points<T, NCP>(cp, nl, T(margin) - z_origin, wsum, s);
    contact_finish(wsum, s, E, p, n, v, kc, cmax, inv_ramp, mu, h, f_ext, A);
}
template <class T, int NCP>
DEV void body_contact_compact(const float (*cp)[3], const FrT<T> &E, V3T<T> p, T z_origin, V3T<T> n, SVT<T> v, float kc, float cmax, float inv_ramp,
                              float margin, float mu, float h, SVT<T> &f_ext, ContactDampT<T> &cd) {
    V3T<T> nl = rotT(E, n);                 // world up in the body's own axes
    T wsum;
    V3T<T> s;
    contact_points<T, NCP>(cp, nl, T(margin) - z_origin, wsum, s);
    contact_eval(wsum, s, E, p, n, v, kc, cmax, inv_ramp, mu, h, f_ext, cd);
}
// A += (rigid inertia B about the FRAME origin)
template <class T> DEV void add_rigid(Sym6T<T> &A, const RigidT<T> &B) {
    add(A.AA, B.I);
    A.AL.r0.y -= B.h.z; A.AL.r0.z += B.h.y;
    A.AL.r1.x += B.h.z; A.AL.r1.z -= B.h.x;
    A.AL.r2.x -= B.h.y; A.AL.r2.y += B.h.x;
    A.LL.xx += B.m; A.LL.yy += B.m; A.LL.zz += B.m;
}

struct BaseState { V3 pw; float qw, qx, qy, qz; V3 vw; V3 wb; };

// forward + control_cost + alive_bonus (README.md:65-72) -- `c_fwd` = w_forward * qvel[0] and `c_ctl` = w_ctrl * sum(ctrl^2) are products,
// and a sum of two products is where the compiler's contraction has a choice (fma(a, b, c*d) or fma(c, d, a*b)): the per-launch and the
// many-steps-per-launch forms of the two-legs-per-lane kernel made different ones (round 4: rewards one ulp apart in a fifth of the envs).
// Written out: no contraction in here, every kernel adds the two rounded products and then the bonus.
DEV float reward_total(float c_fwd, float c_ctl, float c_alive) {
#pragma clang fp contract(off)
    return (c_fwd + c_ctl) + c_alive;
}

// The quaternion scaled to unit length at the head of an env-step.  The sum of squares is written out as explicit multiply-adds: left
// to the compiler's contraction, w*w + x*x may become fma(w, w, x*x) in one kernel and fma(x, x, w*w) in another (it did: the
// per-launch kernel and the many-steps-per-launch kernel of qg_kernel_resident.hip differed by one ulp in a few envs per step),
// and those two kernels must leave the same bits.
DEV void quat_unit(BaseState &B) {
    float d = B.qx * B.qx;
    d = fmaf(B.qw, B.qw, d);
    d = fmaf(B.qy, B.qy, d);
    d = fmaf(B.qz, B.qz, d);
    const float qn = __builtin_amdgcn_rsqf(d);
    B.qw *= qn; B.qx *= qn; B.qy *= qn; B.qz *= qn;
}

// quantities of one substep that depend on the base state only
struct BaseCtx {
    float w, x, y, z;       // normalised quaternion
    V3 cx, cy, cz;          // FRAME axes in world coordinates (columns of R)
    V3 n, gb, vb;           // world up, gravity and base linear velocity in FRAME axes
    SV V0, A0;              // spatial velocity of the FRAME; its spatial acceleration when the unknowns are zero
};
// UNIT: the caller guarantees a quaternion of unit length (the one-link-per-lane kernel normalises the loaded state once per launch;
// base_integrate hands a normalised quaternion from substep to substep) -- 13 instructions per substep that only repeated it.
template <bool UNIT = false>
DEV BaseCtx base_prelude(const KModel &C, const BaseState &B) {
    BaseCtx c;
    float w = B.qw, x = B.qx, y = B.qy, z = B.qz;
    if constexpr (!UNIT) {
        float qn = rcp(__builtin_amdgcn_sqrtf(B.qw * B.qw + B.qx * B.qx + B.qy * B.qy + B.qz * B.qz));
        w = B.qw * qn; x = B.qx * qn; y = B.qy * qn; z = B.qz * qn;
    }
    c.w = w; c.x = x; c.y = y; c.z = z;
    c.cx = v3(1.f - 2.f * (y * y + z * z), 2.f * (x * y + w * z), 2.f * (x * z - w * y));
    c.cy = v3(2.f * (x * y - w * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z + w * x));
    c.cz = v3(2.f * (x * z + w * y), 2.f * (y * z - w * x), 1.f - 2.f * (x * x + y * y));
    c.n = v3(c.cx.z, c.cy.z, c.cz.z);
    V3 gw = ld3(C.g);
    c.gb = v3(dot(c.cx, gw), dot(c.cy, gw), dot(c.cz, gw));
    c.vb = v3(dot(c.cx, B.vw), dot(c.cy, B.vw), dot(c.cz, B.vw));
    c.V0.a = B.wb; c.V0.l = c.vb;
    // unknowns are (d/dt w_b, classical acceleration of the FRAME origin); with both zero the
    // spatial acceleration of the FRAME is [0; -w x v - g]
    c.A0.a = v3(0.f, 0.f, 0.f);
    c.A0.l = v3(0.f, 0.f, 0.f) - cross(B.wb, c.vb) - c.gb;
    return c;
}
// FRAME body: bias force, rigid inertia, free-joint damping and armature on all six base DoFs
// (quadruped.xml:9,62-63).  Contact of the FRAME is added by the caller.
DEV void frame_body(const KModel &C, const BaseCtx &c, float h, SV &p0, Sym6 &Ic0) {
    Rigid I0 = {C.m0, ld3(C.h0), {C.I0[0], C.I0[1], C.I0[2], C.I0[3], C.I0[4], C.I0[5]}};
    SV Iv0 = mul(I0, c.V0), Ia0 = mul(I0, c.A0);
    p0.a = Ia0.a + cross(c.V0.a, Iv0.a) + cross(c.V0.l, Iv0.l);
    p0.l = Ia0.l + cross(c.V0.a, Iv0.l);
    Ic0 = sym6_of(I0);
    float dg = C.free_armature + h * C.free_damping;
    Ic0.AA.xx += dg; Ic0.AA.yy += dg; Ic0.AA.zz += dg;
    Ic0.LL.xx += dg; Ic0.LL.yy += dg; Ic0.LL.zz += dg;
    p0.a = fma3(C.free_damping, c.V0.a, p0.a);
    p0.l = fma3(C.free_damping, c.V0.l, p0.l);
}
namespace pk3 {
// the same with the velocity-product terms as packed cross products (the inertia products keep the plain templates: the compiled-in
// robot's FRAME inertia is literals, which fold there and would have to be materialised for a packed operand)
DEV void frame_body_pk(const KModel &C, const BaseCtx &c, float h, SV &p0, Sym6 &Ic0) {
    Rigid I0 = {C.m0, ld3(C.h0), {C.I0[0], C.I0[1], C.I0[2], C.I0[3], C.I0[4], C.I0[5]}};
    SV Iv0 = ::mul<float>(I0, c.V0), Ia0 = ::mul<float>(I0, c.A0);
    p0.a = cross_add(c.V0.l, Iv0.l, cross_add(c.V0.a, Iv0.a, Ia0.a));
    p0.l = cross_add(c.V0.a, Iv0.l, Ia0.l);
    Ic0 = sym6_of(I0);
    float dg = C.free_armature + h * C.free_damping;
    Ic0.AA.xx += dg; Ic0.AA.yy += dg; Ic0.AA.zz += dg;
    Ic0.LL.xx += dg; Ic0.LL.yy += dg; Ic0.LL.zz += dg;
    p0.a = fma3(C.free_damping, c.V0.a, p0.a);
    p0.l = fma3(C.free_damping, c.V0.l, p0.l);
}
}  // namespace pk3

// ------------------------------------------------------------------------------------------
// one leg: kinematics of fema / shin / foot, recursive Newton-Euler bias forces, ground contact,
// composite (augmented) inertias = mass-matrix columns, servo / limit / damping terms.
// `Ep0` is the frame the fema is mounted in: the identity (mount read from the table, leg k) in
// the one-env-per-lane kernel; the quarter turn of leg k in the one-leg-per-lane kernel, where
// the mount is then leg 0's (the legs are quarter-turn copies of one another).
// Out: leg composite inertia Ic and force fc (to be added to the base rows), the base coupling
// columns F[j], the leg block H = [[Hd0,H01,H02],[.,Hd1,H12],[.,.,Hd2]] and the right-hand side b.
// ------------------------------------------------------------------------------------------
// COMPACT: keep (rigid inertia, contact-damping numbers) per link -- 14 live values instead of a 21-value 6x6 -- and assemble the
// 6x6 forms in the backward pass straight into the composite.  Same arithmetic per term, different summation order; used by the
// register-capped (3 and 4 waves per SIMD) instantiations of the one-leg-per-lane kernel.
// sc: sin / cos of the three hinge rotations carried by the caller (advanced by each substep's small rotation, see hinge_advance),
// or NULL: evaluate the polynomials here.
template <class T, bool BAKED, bool QUAD, bool CULL_FEMUR = false, bool COMPACT = false>
DEV void leg_pass(const KModel &C, int k, FrT<T> Ep, const T q[3], const T qd[3], const T act[3], const BaseCtx &bc, float zbase_f,
                  float h, Sym6T<T> &Ic, SVT<T> &fc, SVT<T> F[3], T Hd[3], T &H01, T &H02, T &H12, T bj[3], const T *sc = nullptr) {
    const T zero = T(0.f);
    V3T<T> pp = v3<T>(zero, zero, zero);
    const V3T<T> nb = splat3<T>(bc.n);
    const T zbase = T(zbase_f);
    SVT<T> vp = splat6<T>(bc.V0), ap = splat6<T>(bc.A0);
    SVT<T> S[3], f[3];
    // per link: rigid inertia (h, I; the mass is a constant) and the five numbers of its contact's implicit damping -- the 6x6
    // forms are assembled in the backward pass straight into the composite (14 live values per link instead of 21)
    RigidT<T> Bs[3];
    ContactDampT<T> Cd[3];
    Sym6T<T> Ag[3];                 // only one of the two forms is live in an instantiation
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const KLink &L = link_of<BAKED>(C, k, i);
        const KLink &Lm = QUAD ? C.link[i] : ((i == 0) ? C.link[3 * k] : L);   // the fema's mounting transform differs per leg
        T sn, cs;
        if (sc) { sn = sc[2 * i]; cs = sc[2 * i + 1]; }
        else sincos_f(q[i] - T(L.ref), sn, cs);      // rotation applied = qpos - ref
        V3T<T> p = pp + rot(Ep, ld3t<T>(Lm.pos));
        V3T<T> tx = fma3(T(Lm.Q[0]), Ep.ex, fma3(T(Lm.Q[3]), Ep.ey, T(Lm.Q[6]) * Ep.ez));
        V3T<T> ty = fma3(T(Lm.Q[1]), Ep.ex, fma3(T(Lm.Q[4]), Ep.ey, T(Lm.Q[7]) * Ep.ez));
        V3T<T> tz = fma3(T(Lm.Q[2]), Ep.ex, fma3(T(Lm.Q[5]), Ep.ey, T(Lm.Q[8]) * Ep.ez));
        FrT<T> E = {fma3(cs, tx, sn * ty), fma3(cs, ty, (-sn) * tx), tz};
        S[i].a = E.ez;
        S[i].l = cross(p, E.ez);
        SVT<T> v = {fma3(qd[i], S[i].a, vp.a), fma3(qd[i], S[i].l, vp.l)};
        // a = a_parent + (v x S) qd
        SVT<T> a;
        a.a = fma3(qd[i], cross(v.a, S[i].a), ap.a);
        a.l = fma3(qd[i], cross(v.a, S[i].l) + cross(v.l, S[i].a), ap.l);
        // rigid inertia of the link about the FRAME origin, FRAME axes
        RigidT<T> Bi;
        Bi.m = T(L.mass);
        V3T<T> c = p + rot(E, ld3t<T>(L.ipos));
        Bi.h = Bi.m * c;
        {
            V3T<T> ux = fma3(T(L.inertia[0]), E.ex, fma3(T(L.inertia[3]), E.ey, T(L.inertia[4]) * E.ez));
            V3T<T> uy = fma3(T(L.inertia[3]), E.ex, fma3(T(L.inertia[1]), E.ey, T(L.inertia[5]) * E.ez));
            V3T<T> uz = fma3(T(L.inertia[4]), E.ex, fma3(T(L.inertia[5]), E.ey, T(L.inertia[2]) * E.ez));
            T hc = dot(Bi.h, c);
            Bi.I.xx = fma_(ux.x, E.ex.x, fma_(uy.x, E.ey.x, uz.x * E.ez.x)) + hc - Bi.h.x * c.x;
            Bi.I.yy = fma_(ux.y, E.ex.y, fma_(uy.y, E.ey.y, uz.y * E.ez.y)) + hc - Bi.h.y * c.y;
            Bi.I.zz = fma_(ux.z, E.ex.z, fma_(uy.z, E.ey.z, uz.z * E.ez.z)) + hc - Bi.h.z * c.z;
            Bi.I.xy = fma_(ux.x, E.ex.y, fma_(uy.x, E.ey.y, uz.x * E.ez.y)) - Bi.h.x * c.y;
            Bi.I.xz = fma_(ux.x, E.ex.z, fma_(uy.x, E.ey.z, uz.x * E.ez.z)) - Bi.h.x * c.z;
            Bi.I.yz = fma_(ux.y, E.ex.z, fma_(uy.y, E.ey.z, uz.y * E.ez.z)) - Bi.h.y * c.z;
        }
        SVT<T> Iv = mul(Bi, v), Ia = mul(Bi, a);
        f[i].a = Ia.a + cross(v.a, Iv.a) + cross(v.l, Iv.l);
        f[i].l = Ia.l + cross(v.a, Iv.l);
        if constexpr (COMPACT) {
            Bs[i] = Bi;
            Cd[i].mc = zero; Cd[i].w = zero; Cd[i].P = v3<T>(zero, zero, zero);
        } else {
            Ag[i] = sym6_of(Bi);
        }
        T zo = zbase + dot(nb, p);
        bool may = true;
        if (CULL_FEMUR && i == 0) {
            // femur: wave-uniform skip of its contact when no env of the wave can reach the floor with it (bounding sphere
            // of its sample points).  Pays only once several waves share a SIMD (+4.7 % at 262 144 envs, -1.6 % at 4096).
            float b2 = 0.f;
#pragma unroll
            for (int qq = 0; qq < QGK_CP_LINK; ++qq) b2 = fmaxf(b2, L.cp[qq][0] * L.cp[qq][0] + L.cp[qq][1] * L.cp[qq][1] + L.cp[qq][2] * L.cp[qq][2]);
            T reach = zo - T(C.contact_margin);
            may = __any(m_any(m_or(m_not(reach > zero), reach * reach < T(b2 * 1.0002f)))) != 0;
        }
        if (may) {
            SVT<T> fe;
            if constexpr (COMPACT)
                body_contact_compact<T, QGK_CP_LINK>(L.cp, E, p, zo, nb, v, C.contact_k, C.contact_c, C.contact_inv_ramp,
                                                     C.contact_margin, C.contact_mu, h, fe, Cd[i]);
            else
                body_contact<T, QGK_CP_LINK>(L.cp, E, p, zo, nb, v, C.contact_k, C.contact_c, C.contact_inv_ramp,
                                             C.contact_margin, C.contact_mu, h, fe, Ag[i]);
            f[i].a = f[i].a - fe.a;
            f[i].l = f[i].l - fe.l;
        }
        Ep = E; pp = p; vp = v; ap = a;
    }
    // backward pass: composite inertias (mass-matrix columns) and bias torques
    if constexpr (COMPACT) {
        Ic = sym6_of(Bs[2]);
        add_contact_damping(Ic, Cd[2].mc, Cd[2].w, Cd[2].P, nb);
    } else {
        Ic = Ag[2];
    }
    fc = f[2];
    F[2] = mul(Ic, S[2]);
    T H22 = dot(S[2], F[2]), t2 = dot(S[2], fc);
    H12 = dot(S[1], F[2]);
    H02 = dot(S[0], F[2]);
    if constexpr (COMPACT) {
        add_rigid(Ic, Bs[1]);
        add_contact_damping(Ic, Cd[1].mc, Cd[1].w, Cd[1].P, nb);
    } else {
        add(Ic, Ag[1]);
    }
    fc = fc + f[1];
    F[1] = mul(Ic, S[1]);
    T H11 = dot(S[1], F[1]), t1 = dot(S[1], fc);
    H01 = dot(S[0], F[1]);
    if constexpr (COMPACT) {
        add_rigid(Ic, Bs[0]);
        add_contact_damping(Ic, Cd[0].mc, Cd[0].w, Cd[0].P, nb);
    } else {
        add(Ic, Ag[0]);
    }
    fc = fc + f[0];
    F[0] = mul(Ic, S[0]);
    T H00 = dot(S[0], F[0]), t0 = dot(S[0], fc);
    Hd[0] = H00; Hd[1] = H11; Hd[2] = H22;
    T tb[3] = {t0, t1, t2};

    // joint-space terms: damping, armature, servo, soft limits (masks + select: the same source serves float and f2)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const KLink &L = link_of<BAKED>(C, k, i);
        // position servo (quadruped.xml:10-37): force from the PRE-update activation
        T force = L.kp * (act[i] - L.gear * q[i]) - (L.kv * L.gear) * qd[i];
        auto clamped = m_or(force <= T(L.force_lo), force >= T(L.force_hi));
        force = min_(max_(force, T(L.force_lo)), T(L.force_hi));
        T dimp = T(L.damping) + sel(clamped, zero, T(L.kv * L.gear * L.gear));
        T tau = L.gear * force - L.damping * qd[i];
        // soft joint limits: one-sided spring + damper that ramps in with the penetration (continuous torque)
        T below = T(L.lo) - q[i], above = q[i] - T(L.hi);
        T pen = max_(max_(below, above), zero);
        T bl = C.limit_b * min_(pen * C.limit_inv_ramp, T(1.f));
        auto is_below = below > zero;
        auto is_above = m_and(m_not(is_below), above > zero);
        T spring_b = C.limit_k * below, spring_a = C.limit_k * above;
        T tq_b = spring_b - bl * qd[i], tq_a = spring_a + bl * qd[i];
        auto free_b = tq_b < zero, free_a = tq_a < zero;      // leaving the limit fast: no pull-back, secant damping
        T lim_b = sel(free_b, zero, tq_b), be_b = sel(free_b, spring_b * rcp(qd[i]), bl);
        T lim_a = sel(free_a, zero, tq_a), be_a = sel(free_a, -spring_a * rcp(qd[i]), bl);
        tau = tau + sel(is_below, lim_b, zero) - sel(is_above, lim_a, zero);
        dimp = dimp + sel(is_below, be_b, sel(is_above, be_a, zero));
        Hd[i] = Hd[i] + T(L.armature) + h * dimp;
        bj[i] = tau - tb[i];
    }
}

// Block elimination of one leg: LDL^T of the 3x3 joint block, Y = F H^-1 (6x3, stored as three
// 6-vectors), u = H^-1 b; returns the Schur terms  YFt = Y F^T (symmetric 6x6)  and  Fu = F u.
template <class T>
DEV void leg_eliminate(const SVT<T> F[3], const T Hd[3], T H01, T H02, T H12, const T bj[3], T Y0[6], T Y1[6],
                       T Y2[6], T u[3], Sym6T<T> &YFt, SVT<T> &Fu) {
    T d0 = Hd[0], id0 = rcp(d0);
    T l10 = H01 * id0, l20 = H02 * id0;
    T d1 = fma_(-l10, H01, Hd[1]), id1 = rcp(d1);
    T t21 = fma_(-l20, H01, H12);
    T l21 = t21 * id1;
    T d2 = fma_(-l21, t21, fma_(-l20, H02, Hd[2])), id2 = rcp(d2);
    T Fr0[6] = {F[0].a.x, F[0].a.y, F[0].a.z, F[0].l.x, F[0].l.y, F[0].l.z};
    T Fr1[6] = {F[1].a.x, F[1].a.y, F[1].a.z, F[1].l.x, F[1].l.y, F[1].l.z};
    T Fr2[6] = {F[2].a.x, F[2].a.y, F[2].a.z, F[2].l.x, F[2].l.y, F[2].l.z};
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        T z0 = Fr0[r];
        T z1 = fma_(-l10, z0, Fr1[r]);
        T z2 = fma_(-l21, z1, fma_(-l20, z0, Fr2[r]));
        T y2 = z2 * id2;
        T y1 = fma_(-l21, y2, z1 * id1);
        T y0 = fma_(-l20, y2, fma_(-l10, y1, z0 * id0));
        Y0[r] = y0; Y1[r] = y1; Y2[r] = y2;
    }
    {
        T z0 = bj[0];
        T z1 = fma_(-l10, z0, bj[1]);
        T z2 = fma_(-l21, z1, fma_(-l20, z0, bj[2]));
        u[2] = z2 * id2;
        u[1] = fma_(-l21, u[2], z1 * id1);
        u[0] = fma_(-l20, u[2], fma_(-l10, u[1], z0 * id0));
    }
#define YF(r, c) (Y0[r] * Fr0[c] + Y1[r] * Fr1[c] + Y2[r] * Fr2[c])
    YFt.AA.xx = YF(0, 0); YFt.AA.yy = YF(1, 1); YFt.AA.zz = YF(2, 2);
    YFt.AA.xy = YF(0, 1); YFt.AA.xz = YF(0, 2); YFt.AA.yz = YF(1, 2);
    YFt.AL.r0 = v3<T>(YF(0, 3), YF(0, 4), YF(0, 5));
    YFt.AL.r1 = v3<T>(YF(1, 3), YF(1, 4), YF(1, 5));
    YFt.AL.r2 = v3<T>(YF(2, 3), YF(2, 4), YF(2, 5));
    YFt.LL.xx = YF(3, 3); YFt.LL.yy = YF(4, 4); YFt.LL.zz = YF(5, 5);
    YFt.LL.xy = YF(3, 4); YFt.LL.xz = YF(3, 5); YFt.LL.yz = YF(4, 5);
#undef YF
    Fu.a = v3<T>(Fr0[0] * u[0] + Fr1[0] * u[1] + Fr2[0] * u[2], Fr0[1] * u[0] + Fr1[1] * u[1] + Fr2[1] * u[2],
                 Fr0[2] * u[0] + Fr1[2] * u[1] + Fr2[2] * u[2]);
    Fu.l = v3<T>(Fr0[3] * u[0] + Fr1[3] * u[1] + Fr2[3] * u[2], Fr0[4] * u[0] + Fr1[4] * u[1] + Fr2[4] * u[2],
                 Fr0[5] * u[0] + Fr1[5] * u[1] + Fr2[5] * u[2]);
}
template <class T> DEV void sub(Sym3T<T> &a, const Sym3T<T> &b) { a.xx -= b.xx; a.yy -= b.yy; a.zz -= b.zz; a.xy -= b.xy; a.xz -= b.xz; a.yz -= b.yz; }
template <class T> DEV void sub(Sym6T<T> &a, const Sym6T<T> &b) {
    sub(a.AA, b.AA); sub(a.LL, b.LL);
    a.AL.r0 = a.AL.r0 - b.AL.r0; a.AL.r1 = a.AL.r1 - b.AL.r1; a.AL.r2 = a.AL.r2 - b.AL.r2;
}

// 6x6 base solve  Ic0 x = b  by LDL^T, x = [d/dt w_b; classical acceleration of the FRAME origin]
DEV void base_solve(const Sym6 &Ic0, SV rhs, float x6[6]) {
    float A[6][6];
    A[0][0] = Ic0.AA.xx; A[1][1] = Ic0.AA.yy; A[2][2] = Ic0.AA.zz;
    A[1][0] = Ic0.AA.xy; A[2][0] = Ic0.AA.xz; A[2][1] = Ic0.AA.yz;
    A[3][0] = Ic0.AL.r0.x; A[4][0] = Ic0.AL.r0.y; A[5][0] = Ic0.AL.r0.z;
    A[3][1] = Ic0.AL.r1.x; A[4][1] = Ic0.AL.r1.y; A[5][1] = Ic0.AL.r1.z;
    A[3][2] = Ic0.AL.r2.x; A[4][2] = Ic0.AL.r2.y; A[5][2] = Ic0.AL.r2.z;
    A[3][3] = Ic0.LL.xx; A[4][4] = Ic0.LL.yy; A[5][5] = Ic0.LL.zz;
    A[4][3] = Ic0.LL.xy; A[5][3] = Ic0.LL.xz; A[5][4] = Ic0.LL.yz;
    float b[6] = {rhs.a.x, rhs.a.y, rhs.a.z, rhs.l.x, rhs.l.y, rhs.l.z};
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
            float sacc = A[i][j];
#pragma unroll
            for (int t = 0; t < j; ++t) sacc = fmaf(-A[i][t], ld[t], sacc);
            A[i][j] = sacc * idg[j];
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

namespace pk3 {
// The same solve with the ROWS taken in pairs (0,1) (2,3) (4,5) as packed FP32, for the kernel that is alone on its SIMD: right-looking
// LDL^T (column j scaled by 1/d_j, then the trailing columns k > j updated with u_k = the unscaled A[k][j]), forward substitution
// column by column, backward substitution plain -- 64 instructions + 6 reciprocals against 95.  A pair is formed only of two entries that are both
// in range; every entry belongs to one pair, so no copies.  (The sums run in another order than base_solve's left-looking loops: the
// two agree to rounding, not to the bit.)
DEV void base_solve(const Sym6 &Ic0, SV rhs, float x6[6]) {
    float A[6][6];
    A[0][0] = Ic0.AA.xx; A[1][1] = Ic0.AA.yy; A[2][2] = Ic0.AA.zz;
    A[1][0] = Ic0.AA.xy; A[2][0] = Ic0.AA.xz; A[2][1] = Ic0.AA.yz;
    A[3][0] = Ic0.AL.r0.x; A[4][0] = Ic0.AL.r0.y; A[5][0] = Ic0.AL.r0.z;
    A[3][1] = Ic0.AL.r1.x; A[4][1] = Ic0.AL.r1.y; A[5][1] = Ic0.AL.r1.z;
    A[3][2] = Ic0.AL.r2.x; A[4][2] = Ic0.AL.r2.y; A[5][2] = Ic0.AL.r2.z;
    A[3][3] = Ic0.LL.xx; A[4][4] = Ic0.LL.yy; A[5][5] = Ic0.LL.zz;
    A[4][3] = Ic0.LL.xy; A[5][3] = Ic0.LL.xz; A[5][4] = Ic0.LL.yz;
    float b[6] = {rhs.a.x, rhs.a.y, rhs.a.z, rhs.l.x, rhs.l.y, rhs.l.z};
    float idg[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        idg[j] = rcp(A[j][j]);
        float l[6];
        const f2 id2 = {idg[j], idg[j]};
#pragma unroll
        for (int q = 0; q < 3; ++q) {                      // L[i][j] = A[i][j] / d_j, rows i > j
            const int i = 2 * q;
            if (i > j) { const f2 t = f2{A[i][j], A[i + 1][j]} * id2; l[i] = t.x; l[i + 1] = t.y; }
            else if (i + 1 > j) l[i + 1] = A[i + 1][j] * idg[j];
        }
#pragma unroll
        for (int k = j + 1; k < 6; ++k) {                  // A[i][k] -= L[i][j] * A[k][j], rows i >= k
            const f2 nu = {-A[k][j], -A[k][j]};
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int i = 2 * q;
                if (i >= k) { const f2 t = __builtin_elementwise_fma(f2{l[i], l[i + 1]}, nu, f2{A[i][k], A[i + 1][k]}); A[i][k] = t.x; A[i + 1][k] = t.y; }
                else if (i + 1 >= k) A[i + 1][k] = fmaf(l[i + 1], nu.x, A[i + 1][k]);
            }
        }
#pragma unroll
        for (int i = j + 1; i < 6; ++i) A[i][j] = l[i];
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {                          // forward: b[i] -= L[i][j] * b[j], rows i > j
        const f2 nb = {-b[j], -b[j]};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int i = 2 * q;
            if (i > j) { const f2 t = __builtin_elementwise_fma(f2{A[i][j], A[i + 1][j]}, nb, f2{b[i], b[i + 1]}); b[i] = t.x; b[i + 1] = t.y; }
            else if (i + 1 > j) b[i + 1] = fmaf(A[i + 1][j], nb.x, b[i + 1]);
        }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) { const f2 t = f2{b[2 * q], b[2 * q + 1]} * f2{idg[2 * q], idg[2 * q + 1]}; b[2 * q] = t.x; b[2 * q + 1] = t.y; }
#pragma unroll
    for (int i = 4; i >= 0; --i) {                         // backward: plain multiply-adds (row j of L^T is a COLUMN of the pairs above)
#pragma unroll
        for (int t = i + 1; t < 6; ++t) b[i] = fmaf(-A[t][i], b[t], b[i]);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) x6[i] = b[i];
}
}  // namespace pk3

// semi-implicit integration of the base: velocity, then position / quaternion with the NEW velocity
// RSQ: one v_rsq_f32 (1 ulp) for the final normalisation instead of v_sqrt_f32 + v_rcp_f32
template <bool RSQ = false>
DEV void base_integrate(const BaseCtx &c, float h, V3 wdot, V3 acl, BaseState &B) {
    // d/dt v_world = R * (classical acceleration in FRAME axes)
    V3 aw = fma3(acl.x, c.cx, fma3(acl.y, c.cy, acl.z * c.cz));
    B.vw = fma3(h, aw, B.vw);
    B.wb = fma3(h, wdot, B.wb);
    B.pw = fma3(h, B.vw, B.pw);
    // q <- q * exp(h w): half-angle series (|h w| / 2 stays far below 0.5 rad)
    float hh = 0.5f * h;
    float x2 = hh * hh * dot(B.wb, B.wb);
    float sc, cw;
    if constexpr (RSQ) {
        // (h |w| / 2)^2 stays below 1e-3 for any body rate the divergence guard lets through a step of 2 ms at 30 rad/s; the terms
        // dropped here are below 2e-8 of the result even at 100 rad/s, and the quaternion is renormalised right below
        sc = hh * fmaf(x2, fmaf(x2, 1.f / 120.f, -1.f / 6.f), 1.f);
        cw = fmaf(x2, fmaf(x2, 1.f / 24.f, -0.5f), 1.f);
    } else {
        sc = hh * fmaf(x2, fmaf(x2, fmaf(x2, -1.f / 5040.f, 1.f / 120.f), -1.f / 6.f), 1.f);
        cw = fmaf(x2, fmaf(x2, fmaf(x2, -1.f / 720.f, 1.f / 24.f), -0.5f), 1.f);
    }
    V3 dv = sc * B.wb;
    float w = c.w, x = c.x, y = c.y, z = c.z;
    float nw = w * cw - x * dv.x - y * dv.y - z * dv.z;
    float nx = w * dv.x + x * cw + y * dv.z - z * dv.y;
    float ny = w * dv.y - x * dv.z + y * cw + z * dv.x;
    float nz = w * dv.z + x * dv.y - y * dv.x + z * cw;
    const float n2 = nw * nw + nx * nx + ny * ny + nz * nz;
    float inv = RSQ ? __builtin_amdgcn_rsqf(n2) : rcp(__builtin_amdgcn_sqrtf(n2));
    B.qw = nw * inv; B.qx = nx * inv; B.qy = ny * inv; B.qz = nz * inv;
}

namespace pk3 {
// One packed multiply(-add) whose swaps, broadcasts and signs are all operand modifiers (the compiler folds a broadcast into op_sel,
// a swap or a sign only now and then -- v_mov / v_xor otherwise).  Every operand is a pair that exists anyway: the quaternion is held
// as W = (w, z), X = (x, y) from substep to substep, so a broadcast of one component is an op_sel of its pair and nothing rides in a
// pair with a don't-care half (frozen poison costs a v_mov from a zero register per pair: measured, 8 in the loop).
#define QG_PKFMA(dst, a, b, c, MODS) asm("v_pk_fma_f32 %0, %1, %2, %3 " MODS : "=v"(dst) : "v"(a), "v"(b), "v"(c))
#define QG_PKMUL(dst, a, b, MODS) asm("v_pk_mul_f32 %0, %1, %2 " MODS : "=v"(dst) : "v"(a), "v"(b))
// base_prelude<true> with the rotation matrix of the UNIT quaternion in nine instructions: with W2 = 2 W, X2 = 2 X and k = 2ww - 1
//   A = (k, 2wz);  (cx.x, cx.y) = 2x (x, y) + A;  (cy.x, cy.y) = 2y (x, y) + (-A.hi, A.lo);  T = 2w (y, x)
//   (cz.x, cz.y) = (2x, 2y) z + (T.lo, -T.hi);  (cx.z, cy.z) = (2x, 2y) z + (-T.lo, T.hi);  cz.z = 2zz + k
// -- the diagonal as 2(ww + qq) - 1 instead of 1 - 2(rr + ss): the same number for a unit quaternion, other rounding.  The pairs are
// the (x, y) halves of the packed 3-vector operators; (cx.z, cy.z) is the (x, y) half of the world's up axis n.
// a value the optimiser cannot see through (no instruction): keeps the loads of two ADJACENT fields of a struct from being merged
// into one vector load -- here that load would straddle the (qw, qx) pair other code writes as a vector, the struct could no longer
// be split into registers and landed in LDS (.amdhsa_group_segment_fixed_size 2 240 -> 5 312)
DEV float opaque(float v) { asm("" : "+v"(v)); return v; }
DEV BaseCtx base_prelude_unit(const KModel &C, const BaseState &B) {
    BaseCtx c;                                           // (c.w .. c.z stay unset: base_integrate_unit reads the quaternion from B)
    const f2 W = {B.qw, B.qz}, X = {opaque(B.qx), opaque(B.qy)};
    const f2 W2 = W + W, X2 = X + X;
    const f2 m10 = {-1.f, 0.f};
    f2 A, cxp, cyp, T, czp, np;
    QG_PKFMA(A, W2, W, m10, "op_sel_hi:[0,1,1]");                                            // (2ww - 1, 2wz)
    QG_PKFMA(cxp, X2, X, A, "op_sel_hi:[0,1,1]");                                            // 2x (x, y) + A
    QG_PKFMA(cyp, X2, X, A, "op_sel:[1,0,1] op_sel_hi:[1,1,0] neg_lo:[0,0,1]");              // 2y (x, y) + (-A.hi, A.lo)
    QG_PKMUL(T, W2, X, "op_sel:[0,1] op_sel_hi:[0,0]");                                      // 2w (y, x)
    QG_PKFMA(czp, X2, W, T, "op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_hi:[0,0,1]");              // (2x, 2y) z + (T.lo, -T.hi)
    QG_PKFMA(np, X2, W, T, "op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,0,1]");               // (2x, 2y) z + (-T.lo, T.hi)
    c.cx = v3(cxp.x, cxp.y, np.x);
    c.cy = v3(cyp.x, cyp.y, np.y);
    c.cz = v3(czp.x, czp.y, fmaf(W2.y, W.y, A.x));
    c.n = v3(c.cx.z, c.cy.z, c.cz.z);
    V3 gw = ld3(C.g);
    c.gb = v3(dot(c.cx, gw), dot(c.cy, gw), dot(c.cz, gw));
    c.vb = v3(dot(c.cx, B.vw), dot(c.cy, B.vw), dot(c.cz, B.vw));
    c.V0.a = B.wb; c.V0.l = c.vb;
    c.A0.a = v3(0.f, 0.f, 0.f);
    c.A0.l = cross_add(c.vb, B.wb, v3(-c.gb.x, -c.gb.y, -c.gb.z));        // -(w x v) - g
    return c;
}
// base_integrate<true> with the 3-vector updates and the quaternion product q * (cw, dv) as eight packed multiply-adds on
// DW = (cw, dv.z), DX = (dv.x, dv.y) -- dv's own (x, y) pair; cw and dv.z are plain producers that write into their halves:
//   (nw, nz) = w (cw, dz) + x (-dx, dy) + y (-dy, -dx) + z (-dz, cw);   (nx, ny) = w (dx, dy) + x (cw, -dz) + y (dz, cw) + z (-dy, dx)
// (each component sums its four products in the order z, y, x, w)
DEV void base_integrate_unit(const BaseCtx &c, float h, V3 wdot, V3 acl, BaseState &B) {
    V3 aw = fma3(acl.x, c.cx, fma3(acl.y, c.cy, acl.z * c.cz));
    B.vw = fma3(h, aw, B.vw);
    B.wb = fma3(h, wdot, B.wb);
    B.pw = fma3(h, B.vw, B.pw);
    const float hh = 0.5f * h;
    const float x2 = hh * hh * dot(B.wb, B.wb);
    const float sc = hh * fmaf(x2, fmaf(x2, 1.f / 120.f, -1.f / 6.f), 1.f);
    const float cw = fmaf(x2, fmaf(x2, 1.f / 24.f, -0.5f), 1.f);
    const V3 dv = sc * B.wb;
    const f2 W = {B.qw, B.qz}, X = {opaque(B.qx), opaque(B.qy)};
    const f2 DW = {cw, dv.z}, DX = {dv.x, dv.y};
    f2 a, b;
    QG_PKMUL(a, W, DW, "op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]");                                  // z (-dz, cw)
    QG_PKFMA(a, X, DX, a, "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]");          // + y (-dy, -dx)
    QG_PKFMA(a, X, DX, a, "op_sel_hi:[0,1,1] neg_lo:[0,1,0]");                                        // + x (-dx, dy)
    QG_PKFMA(a, W, DW, a, "op_sel_hi:[0,1,1]");                                                       // + w (cw, dz)
    QG_PKMUL(b, W, DX, "op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]");                                  // z (-dy, dx)
    QG_PKFMA(b, X, DW, b, "op_sel:[1,1,0] op_sel_hi:[1,0,1]");                                        // + y (dz, cw)
    QG_PKFMA(b, X, DW, b, "op_sel_hi:[0,1,1] neg_hi:[0,1,0]");                                        // + x (cw, -dz)
    QG_PKFMA(b, W, DX, b, "op_sel_hi:[0,1,1]");                                                       // + w (dx, dy)
    const f2 n2 = __builtin_elementwise_fma(b, b, a * a);
    const float inv = __builtin_amdgcn_rsqf(n2.x + n2.y);
    const f2 i2 = {inv, inv};
    a = a * i2; b = b * i2;
    B.qw = a.x; B.qz = a.y; B.qx = b.x; B.qy = b.y;
}
#undef QG_PKFMA
#undef QG_PKMUL
}  // namespace pk3

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

struct SensorOut { float accel[3]; V3 pw, vw, wb, vb, xaxis, zaxis; };

// one physics substep for the env of this lane (mj_step of quadruped.py:165), one env per lane
template <bool BAKED>
DEV void substep(const KModel *__restrict__ Mp, float *__restrict__ lds, int lane, BaseState &B, bool want_sensors, SensorOut &so,
                 float *__restrict__ jpos_out) {
    const KModel &C = table<BAKED>(Mp);
    const float h = C.h;
    // ---- A. base prelude -----------------------------------------------------------------
    const BaseCtx bc = base_prelude(C, B);
    if (want_sensors) {
        so.pw = B.pw; so.vw = B.vw; so.wb = B.wb; so.vb = bc.vb;
        so.xaxis = bc.cx; so.zaxis = bc.cz;
#pragma unroll
        for (int j = 0; j < 12; ++j) jpos_out[j] = lds[LQ(j) * 64 + lane];
    }
    SV p0;
    Sym6 Ic0;
    frame_body(C, bc, h, p0, Ic0);
    {
        Fr E0 = {v3(1.f, 0.f, 0.f), v3(0.f, 1.f, 0.f), v3(0.f, 0.f, 1.f)};
        SV fe;
        body_contact<float, QGK_CP_FRAME>(C.cp0, E0, v3(0.f, 0.f, 0.f), B.pw.z, bc.n, bc.V0, C.contact_k, C.contact_c, C.contact_inv_ramp,
                                   C.contact_margin, C.contact_mu, h, fe, Ic0);
        p0.a = p0.a - fe.a;
        p0.l = p0.l - fe.l;
    }
    SV rhs0 = {v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 0.f)};               // accumulates -F u of the legs

    // ---- B. legs: dynamics terms and elimination into the base block ---------------------
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        float q[3], qd[3], act[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            q[i] = lds[LQ(3 * k + i) * 64 + lane];
            qd[i] = lds[LQD(3 * k + i) * 64 + lane];
            act[i] = lds[LACT(3 * k + i) * 64 + lane];
        }
        Fr E0 = {v3(1.f, 0.f, 0.f), v3(0.f, 1.f, 0.f), v3(0.f, 0.f, 1.f)};
        Sym6 Ic, YFt;
        SV fc, F[3], Fu;
        float Hd[3], H01, H02, H12, bj[3], Y0[6], Y1[6], Y2[6], u[3];
        leg_pass<float, BAKED, false>(C, k, E0, q, qd, act, bc, B.pw.z, h, Ic, fc, F, Hd, H01, H02, H12, bj);
        leg_eliminate(F, Hd, H01, H02, H12, bj, Y0, Y1, Y2, u, YFt, Fu);
        add(Ic0, Ic);
        sub(Ic0, YFt);
        p0 = p0 + fc;
        rhs0.a = rhs0.a - Fu.a;
        rhs0.l = rhs0.l - Fu.l;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            lds[LY(k, r) * 64 + lane] = Y0[r];
            lds[LY(k, 6 + r) * 64 + lane] = Y1[r];
            lds[LY(k, 12 + r) * 64 + lane] = Y2[r];
        }
        lds[LY(k, 18) * 64 + lane] = u[0];
        lds[LY(k, 19) * 64 + lane] = u[1];
        lds[LY(k, 20) * 64 + lane] = u[2];
    }

    // ---- C. base solve: Ic0 x = rhs0 - p0 ----------------------------------------------------
    float x6[6];
    {
        SV b = {rhs0.a - p0.a, rhs0.l - p0.l};
        base_solve(Ic0, b, x6);
    }
    V3 wdot = v3(x6[0], x6[1], x6[2]);
    V3 acl = v3(x6[3], x6[4], x6[5]);
    if (want_sensors) {
        // accelerometer (quadruped.xml:200): proper acceleration in the site frame = a_c - g_b
        so.accel[0] = acl.x - bc.gb.x; so.accel[1] = acl.y - bc.gb.y; so.accel[2] = acl.z - bc.gb.z;
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
    base_integrate(bc, h, wdot, acl, B);
}

// rows of the output tile: the 33-value sensordata (quadruped.xml:174-217) or the 21-value pack
DEV void write_obs_row(float *r, int od, const float *jpos, const SensorOut &so) {
#pragma unroll
    for (int j = 0; j < 12; ++j) r[j] = jpos[j];
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
}

// ------------------------------------------------------------------------------------------
// env-step kernel, ONE ENV PER LANE: one launch = frame_skip substeps + sensor pack + rewards +
// terminations (+ auto-reset) for every env.  grid = ceil(n / 64) workgroups of one wave.
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
    float jpos[12];
    const int fs = T->frame_skip;
    const bool lag = T->sensor_lag != 0;
#pragma unroll 1
    for (int s = 0; s < fs; ++s) substep<BAKED>(Mp, lds, lane, B, lag && (s == fs - 1), so, jpos);
    nstep += fs;
    if (!lag) {   // un-lagged sensors: one extra forward pass on a scratch copy of the state
        BaseState B2 = B;
        float keep[36];
#pragma unroll
        for (int j = 0; j < 36; ++j) keep[j] = lds[j * 64 + lane];
        substep<BAKED>(Mp, lds, lane, B2, true, so, jpos);
#pragma unroll
        for (int j = 0; j < 36; ++j) lds[j * 64 + lane] = keep[j];
    }

    // ---- rewards and terminations on the post-step state (README.md:64-90) ----------------
    float c_fwd = T->w_forward * B.vw.x;
    float c_ctl = T->w_ctrl * ssq;
    float c_alive = T->alive_bonus;
    float reward = reward_total(c_fwd, c_ctl, c_alive);
    bool done = nstep >= T->limit_substeps;
    if (T->use_fall) done = done || (B.pw.z < T->fall_height);
    if (T->use_flip) done = done || (so.zaxis.z < 0.f);          // walking_quad.py:156-160, on the step's sensordata
    {
        float probe = B.pw.x + B.pw.y + B.pw.z + B.qw + B.vw.x + B.vw.y + B.vw.z + B.wb.x + B.wb.y + B.wb.z;
#pragma unroll
        for (int j = 0; j < 12; ++j) probe += lds[LQ(j) * 64 + lane] + lds[LQD(j) * 64 + lane];
        done = done || state_is_bad(probe);
    }

    // ---- outputs: stage rows in LDS, then store the wave's contiguous chunk coalesced ------
    const int od = T->obs_mode == 1 ? 21 : 33;
    const int row = P.packed ? od + 2 : od;
    {
        float *r = tile + lane * row;
        write_obs_row(r, od, jpos, so);
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
            float a = 6.283185307179586f * uniform24(P.seed, P.env_index_base + (uint64_t)env, (uint64_t)P.st.episode[env]);
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
        if (rst) P.st.episode[env] += 1;
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
// env-step kernel, ONE LEG PER LANE: the four lanes 4e..4e+3 of a quad share env e, lane k owns
// leg k (and a quarter of the FRAME's contact points); the base prelude, the 6x6 base solve and
// the base integration run redundantly in all four lanes; the leg contributions to the base
// block (Schur complements, 33 numbers) are summed over the quad with DPP quad_perm moves --
// no LDS, no run-time indexed arrays.  A wave carries 16 envs, so 4096 envs fill 256 waves
// instead of 64: at batch sizes that cannot fill the chip with one env per lane this cuts the
// instructions per wave (= the time, a lone wave issues one VALU instruction per ~4.5-5 cycles) ~3.5x.
// BAKED variant: the compiled-in robot, whose legs are quarter-turn copies of one another (constants become literals).
// Generic variant: any model numbers; the tables are staged in LDS and each lane reads its own leg's rows.
// ------------------------------------------------------------------------------------------
template <int CTRL> DEV float dpp_quad(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
// sum over the 4 lanes of a quad; every lane gets the bitwise identical result ((a+b)+(c+d), commutative)
// (fp contract off: with the default "fast" contraction a multiply that feeds `x` is folded into the first add -- fma(a, b, dpp(a * b)) --
// which computes the product twice and leaves the DPP move unfused; see env_sum in qg_kernel_link.hip)
DEV float quad_sum(float x) {
#pragma clang fp contract(off)
    x += dpp_quad<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp_quad<0x4E>(x);   // quad_perm [2,3,0,1]
    return x;
}
DEV V3 quad_sum(V3 a) { return v3(quad_sum(a.x), quad_sum(a.y), quad_sum(a.z)); }
DEV void quad_sum(Sym3 &a) { a.xx = quad_sum(a.xx); a.yy = quad_sum(a.yy); a.zz = quad_sum(a.zz); a.xy = quad_sum(a.xy); a.xz = quad_sum(a.xz); a.yz = quad_sum(a.yz); }
DEV void quad_sum(Sym6 &a) { quad_sum(a.AA); quad_sum(a.LL); a.AL.r0 = quad_sum(a.AL.r0); a.AL.r1 = quad_sum(a.AL.r1); a.AL.r2 = quad_sum(a.AL.r2); }

struct LegState {
    float q[3], qd[3], act[3], u[3];
    float sc[6];      // sin, cos of (q[i] - ref_i), advanced with the hinge
};

// Ordered for low register pressure: the leg pass (the widest live set) runs first with only the base context alive;
// the FRAME terms are built afterwards; sensors go straight to the LDS tile; the rotation is rebuilt at integration.
// The one-leg-per-lane kernel of the compiled-in robot at ONE wave per SIMD takes the packed prelude / integration of the
// one-link-per-lane kernel (a lone wave pays per instruction: 2 109 -> 2 072 per substep, 17.74 -> 17.48 us at 8 192 envs, 18.3 -> 18.1
// at 16 384, sequence form 15.4 -> 15.2); at two waves per SIMD they measured 1.2 % slower and with the generic robot's tables 7 %
// slower (profiles/r04/quad_packed_prelude_ab.txt), so those keep the plain forms.
template <bool PKQ> DEV BaseCtx quad_prelude(const KModel &C, const BaseState &B) {
    if constexpr (PKQ) return pk3::base_prelude_unit(C, B);
    else return base_prelude<true>(C, B);
}
template <bool PKQ> DEV void quad_integrate(const BaseCtx &c, float h, V3 wdot, V3 acl, BaseState &B) {
    if constexpr (PKQ) pk3::base_integrate_unit(c, h, wdot, acl, B);
    else base_integrate<true>(c, h, wdot, acl, B);
}
// LOWREG: rebuild the base context (cheap) instead of keeping it alive across the leg pass -- pays off once two waves
// share a SIMD (register cap 256), costs ~2 % when a wave has the register file to itself.
// BAKED: the compiled-in robot, constants are literals and the lane works in its leg's quarter-turn frame.  Otherwise `C`
// is the model table staged in LDS and every lane reads the constants of its own leg (k) from it -- any model numbers.
template <bool BAKED, bool LOWREG>
DEV void substep_quad(const KModel &C, float cm, float sm, BaseState &B, LegState &L, bool want_sensors, float *__restrict__ row, int k, float &zaxis_z) {
    const float h = C.h;
    const BaseCtx bc0 = quad_prelude<BAKED && !LOWREG>(C, B);
    V3 gb_keep;
    Sym6 Ic;
    SV fc, Fu;
    float Y0[6], Y1[6], Y2[6], u[3];
    {
        const BaseCtx &bc = bc0;
        gb_keep = bc.gb;
        if (want_sensors) {          // the step's sensordata describes the state at the start of its last substep
            zaxis_z = bc.cz.z;
            row[3 * k + 0] = L.q[0]; row[3 * k + 1] = L.q[1]; row[3 * k + 2] = L.q[2];
            if (k == 0) {
                row[15] = B.wb.x; row[16] = B.wb.y; row[17] = B.wb.z;
                row[18] = B.pw.x; row[19] = B.pw.y; row[20] = B.pw.z;
                row[21] = B.vw.x; row[22] = B.vw.y; row[23] = B.vw.z;
                row[24] = bc.cx.x; row[25] = bc.cx.y; row[26] = bc.cx.z;
                row[27] = bc.cz.x; row[28] = bc.cz.y; row[29] = bc.cz.z;
                row[30] = bc.vb.x; row[31] = bc.vb.y; row[32] = bc.vb.z;
            }
        }
        Sym6 YFt;
        SV F[3];
        float Hd[3], H01, H02, H12, bj[3];
        if constexpr (BAKED) {
            // this lane's leg, in the frame turned by its quarter turn: there it is leg 0
            Fr Ek = {v3(cm, sm, 0.f), v3(-sm, cm, 0.f), v3(0.f, 0.f, 1.f)};
            leg_pass<float, true, true, LOWREG, false>(C, 0, Ek, L.q, L.qd, L.act, bc, B.pw.z, h, Ic, fc, F, Hd, H01, H02, H12, bj, L.sc);
        } else {
            Fr E0 = {v3(1.f, 0.f, 0.f), v3(0.f, 1.f, 0.f), v3(0.f, 0.f, 1.f)};
            leg_pass<float, false, false, LOWREG, false>(C, k, E0, L.q, L.qd, L.act, bc, B.pw.z, h, Ic, fc, F, Hd, H01, H02, H12, bj, L.sc);
        }
        leg_eliminate(F, Hd, H01, H02, H12, bj, Y0, Y1, Y2, u, YFt, Fu);
        sub(Ic, YFt);                       // this leg's Schur complement
    }
    float x6[6];
    {
        const BaseCtx bc = LOWREG ? quad_prelude<BAKED && !LOWREG>(C, B) : bc0;      // rebuilt (cheap) rather than kept alive across the leg pass
        SV p0;
        Sym6 Ic0;
        frame_body(C, bc, h, p0, Ic0);
        SV b;
        {   // FRAME contact: this lane evaluates its quarter turn of the three base sample points
            float wsum = 0.f;
            V3 s = v3(0.f, 0.f, 0.f);
            float zb = C.contact_margin - B.pw.z;
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                if constexpr (BAKED) {
                    V3 r0 = ld3(C.cp0[4 * o]);
                    contact_point(v3(cm * r0.x - sm * r0.y, sm * r0.x + cm * r0.y, r0.z), bc.n, zb, wsum, s);
                } else {
                    contact_point(ld3(C.cp0[3 * k + o]), bc.n, zb, wsum, s);   // any partition of the 12 points over the 4 lanes
                }
            }
            wsum = quad_sum(wsum);
            // the sums over the quad sit right in front of their uses: the compiler fuses a DPP move into the add that consumes it only
            // when the two end up within a few instructions of each other in one basic block (33 unfused moves per substep otherwise)
            quad_sum(Ic);
            fc.a = quad_sum(fc.a); fc.l = quad_sum(fc.l);
            Fu.a = quad_sum(Fu.a); Fu.l = quad_sum(Fu.l);
            add(Ic0, Ic);
            b.a = v3(0.f, 0.f, 0.f) - Fu.a - p0.a - fc.a;
            b.l = v3(0.f, 0.f, 0.f) - Fu.l - p0.l - fc.l;
            // Wave-uniform skip: under actuation the robot stands on its feet and the FRAME practically never touches the
            // floor; when no env of the wave has a FRAME sample point below the margin every term below is exactly zero.
            // (A/B on one box: +3.4 % at 4096 envs, +6.5 % at 262 144.  The same skip per leg link does NOT pay: a taken
            // branch over a large block stalls the instruction fetch of a wave that is alone on its SIMD.)
            // The branch updates the assembled block IN PLACE (round 3): with the block assembled after it, the usual path -- no
            // contact -- had to materialise the FRAME's constant entries as the other arm of a phi, 17 moves per substep.
            if (__any(wsum > 0.f)) {
                s = quad_sum(s);
                Fr E0 = {v3(1.f, 0.f, 0.f), v3(0.f, 1.f, 0.f), v3(0.f, 0.f, 1.f)};
                SV fe;
                contact_finish(wsum, s, E0, v3(0.f, 0.f, 0.f), bc.n, bc.V0, C.contact_k, C.contact_c, C.contact_inv_ramp, C.contact_mu, h, fe, Ic0);
                b.a = b.a + fe.a;
                b.l = b.l + fe.l;
            }
        }
        if constexpr (BAKED) pk3::base_solve(Ic0, b, x6);      // packed row pairs: fewer instructions for a wave alone on its SIMD, as many
        else base_solve(Ic0, b, x6);                            // issue cycles at two waves per SIMD; the generic robot's register budget has no room for the pairs
    }
    V3 wdot = v3(x6[0], x6[1], x6[2]);
    V3 acl = v3(x6[3], x6[4], x6[5]);
    if (want_sensors && k == 0) {
        row[12] = acl.x - gb_keep.x; row[13] = acl.y - gb_keep.y; row[14] = acl.z - gb_keep.z;   // accelerometer
    }
    // back-substitution and integration of this lane's three hinges
    const float *Y[3] = {Y0, Y1, Y2};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float acc = u[i];
#pragma unroll
        for (int r = 0; r < 6; ++r) acc = fmaf(-Y[i][r], x6[r], acc);
        L.qd[i] = fmaf(h, acc, L.qd[i]);
        L.q[i] = fmaf(h, L.qd[i], L.q[i]);
        hinge_advance(h * L.qd[i], L.sc[2 * i], L.sc[2 * i + 1]);
        L.act[i] = fmaf(L.u[i] - L.act[i], link_of<BAKED>(C, k, i).act_decay, L.act[i]);
    }
    {
        const BaseCtx bc = LOWREG ? quad_prelude<BAKED && !LOWREG>(C, B) : bc0;
        quad_integrate<BAKED && !LOWREG>(bc, h, wdot, acl, B);
    }
}

#define QGK_QUAD_ENVS 16    // envs per wave in the one-leg-per-lane kernel
#define QG_PO_COPY_K 4      // 16-byte groups per lane and substep of the fused observation pack's history copy (po_row_copy_*)
// po_row_copy_load issues its K loads unpredicated: the last batch of a row may read up to (K - 1) * LPE groups + one group past the
// lane's share, i.e. past the end of the LAST env's doubled ring for some (window, rotation) pairs -- windows 8, 13, 18, 40, 45, ...
// (round-3 advisor; the 64 bytes of slack the ring used to have covered the windows the tests ran).  The ring is allocated with
// QG_PO_RING_SLACK bytes behind it; the largest LPE of the wave-level fused forms is 4 (one leg per lane).
static_assert((QG_PO_COPY_K - 1) * 4 * 16 + 16 <= QG_PO_RING_SLACK, "frame-ring slack against po_row_copy_load's over-read");
// Epilogue of the observation pack fused into a kernel whose wave owns ENVS consecutive envs with LPE lanes each (a lane owns NCH =
// 12 / LPE control channels, `aclip` = this step's env-clipped actions of those): the lanes put data.ctrl into the frame, the env's lead
// lane runs the orientation filter on the sensor row `srow` the wave has staged in LDS and builds the frame (po_frame_env: the
// stand-alone kernel's function), then the wave writes the rows (po_wave_emit).  `q` = data.qpos[3:7] as the step leaves it (after an
// auto-reset: the reset pose), what an aliasing estimate shows; RELOAD: the filter state is read here rather than carried through
// the substep loop (two waves per SIMD: the partner covers the latency).
template <int ENVS, int LPE, int WAVES, bool RELOAD>
DEV void po_wave_epilogue(const KPoLaunch &PK, const KWalkLaunch &WK, const KStepArgs &P, int n, int env, int env0, int live_envs, int wave, int lane,
                          int el, int j, bool live, bool lead, PoEnvIn pin, const float *srow, const BaseState &B, const WalkEnvIn &win, bool done,
                          const float *aclip) {
    constexpr int NCH = 12 / LPE;
    __shared__ float s_new_all[WAVES][ENVS][QG_PO_FRAME];     // the frames of this step
    __shared__ float s_rst_all[WAVES][ENVS][QG_PO_FRAME];     // the frames reset() would return (envs that finished)
    __shared__ int s_slot_all[WAVES][ENVS], s_fin_all[WAVES][ENVS];
    if (live) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) s_new_all[wave][el][11 + NCH * j + c] = aclip[c];            // data.ctrl of the frame
    }
    if (lead) {
        if constexpr (RELOAD) pin = po_env_load(PK.S, n, env);
        int slot, fin;
        po_frame_env(PK.P, PK.S, n, env, pin, srow, B.qw, B.qx, B.qy, B.qz, win.cvx, win.cvy, win.hx, win.hy, done, s_new_all[wave][el],
                     s_rst_all[wave][el], slot, fin);
        s_slot_all[wave][el] = slot;
        s_fin_all[wave][el] = fin;
        // random_controls on the device: the new episode's command, drawn only now that both frames show the old one
        if (fin && PK.sample) walk_sample_command(WK.P, WK.S, n, env, P.seed, P.env_index_base, win.episode_key);
    }
    wave_sync();
    QG_MARK(7);                                      // frame built
    po_wave_emit<LPE>(PK.P, PK.S, env0, live_envs, lane, el, j, s_new_all[wave], s_rst_all[wave], s_slot_all[wave], s_fin_all[wave], PK.out, PK.term_out);
    QG_MARK(8);                                      // rows written
}

// WPE = waves per SIMD the register allocation is capped for: 1 (all 512 registers) is fastest while the grid has at
// most one wave per SIMD (n <= 16384); 2 lets a second wave share the SIMD once the grid is larger.
// WALK: the walking task layer fused in (qg_walk_dev.h) -- the whole WalkingQuadrupedEnv.step (walking_quad.py:128-148) is this
// one launch: settling-time action mask and the estimator update of the lane's three control channels in the prologue (they
// need data.ctrl of the PREVIOUS step, which is still in place there), ideal-position integration, the eleven reward terms and
// the episode bookkeeping in the epilogue on the sensor row the wave has just staged in LDS.
// WAVES: waves per workgroup (1, or 4 = one per SIMD of a CU for grids of more than 256 waves: fewer workgroups to dispatch,
// see qg_step_kernel_pair); the waves of a workgroup do not interact.
// PO (with WALK; round 3): the partially observable observation pack fused in as in qg_step_kernel_pair<.., PO> (history copy on the
// substep loop, frame by the env's lead lane, new frames written by the wave); the register-capped development variants (WPE > 2) have none.
// HELP (with WALK, two-waves-per-SIMD register budget): as in qg_step_kernel_link<.., HELP> -- WAVES more waves per workgroup run the
// estimator update of their partner wave's channels while that wave goes straight into the substep loop; for grids of at most one
// physics wave per SIMD (<= 16 384 envs), where the partner's issue slots are otherwise idle.  With the observation pack the helper also
// copies the history rows ring -> out (po_copy_history_now), so that nothing of the copy rides on the physics wave's substep loop: the
// first version, which kept the in-loop copy with the physics wave, lost to the one-role kernel (48.6 against 46.9 us per step at
// 16 384 envs, frame_skip 10: at this register budget the copy cannot park in AGPRs across a substep); this one measures 45.4.
template <int WPE, bool BAKED, bool WALK = false, int WAVES = 1, bool PO = false, bool HELP = false>
__global__ __launch_bounds__(QGK_WAVE * WAVES * (HELP ? 2 : 1), WPE) void qg_step_kernel_quad(const KModel *__restrict__ Mp, const KTask *__restrict__ T, KStepArgs P,
                                                                             const typename WalkArgT<WALK>::type WK,
                                                                             const typename PoArgT<PO>::type PK) {
    static_assert(WALK || !PO, "the observation pack rides on the walking task layer");
    static_assert(WPE == 1 || WPE == 2, "register budget: one wave per SIMD (all 512 registers) or two");
    static_assert(!HELP || (WALK && WPE == 2), "helper waves: the walking forms at the two-waves-per-SIMD register budget");
    constexpr bool PO_COPY = PO && !HELP;          // HELP: the helper wave copies the history rows, nothing of it rides on the substep loop
    __shared__ float tile_all[WAVES][QGK_QUAD_ENVS * 35];
    __shared__ KModel smodel;                       // generic variant: the link / joint tables staged in LDS (3.2 KB)
    constexpr bool RWDH = HELP && !PO;              // walking without the observation pack: the helper wave also evaluates the reward
    constexpr bool POH = HELP && PO;                // with it: the helper wave builds the new frame and writes the rows, the reward stays here
    __shared__ float s_est[HELP ? WAVES : 1][QGK_WAVE][6];      // HELP: (f_est, a_est) of the lane's three channels, helper -> physics wave
    __shared__ float s_done[HELP ? WAVES : 1][QGK_QUAD_ENVS];   // HELP: the step's termination flags, physics -> helper
    __shared__ float s_q[POH ? WAVES : 1][QGK_QUAD_ENVS][4];    // POH: data.qpos[3:7] as the step leaves it (after the auto-reset), physics -> helper
    const int lane = threadIdx.x & (QGK_WAVE - 1);
    const int wave = HELP ? ((threadIdx.x >> 6) % WAVES) : (threadIdx.x >> 6);      // HELP: waves WAVES .. 2 WAVES - 1 shadow waves 0 .. WAVES - 1
    const bool helper = HELP && (int)(threadIdx.x >> 6) >= WAVES;
    float *tile = tile_all[wave];
    QG_MARK(0);
    if constexpr (!BAKED) {
        const float *src = reinterpret_cast<const float *>(Mp);
        float *dst = reinterpret_cast<float *>(&smodel);
        for (int i = threadIdx.x; i < (int)(sizeof(KModel) / sizeof(float)); i += QGK_WAVE * WAVES * (HELP ? 2 : 1)) dst[i] = src[i];
        __syncthreads();
    }
    const KModel &C = BAKED ? QG_BAKED_MODEL : smodel;
    const int k = lane & 3;                         // leg of this lane
    const int el = lane >> 2;                       // env within the wave
    const int env0 = (blockIdx.x * WAVES + wave) * QGK_QUAD_ENVS;
    const int n = P.n;
    const bool live = env0 + el < n;
    const int env = live ? env0 + el : n - 1;       // tail quads shadow the last env; their stores are masked
    // quarter turn of this lane's leg: (cos, sin)(90 deg * k)
    const float cm = (k == 0) ? 1.f : (k == 2) ? -1.f : 0.f;
    const float sm = (k == 1) ? 1.f : (k == 3) ? -1.f : 0.f;

    if constexpr (HELP) {
        if (helper) {
            const int ht[3] = {env * 12 + 3 * k + 0, env * 12 + 3 * k + 1, env * 12 + 3 * k + 2};
            const int hcalls = WK.S.calls[env];
            float hx[3], hf[3] = {0.f, 0.f, 0.f}, ha[3] = {0.f, 0.f, 0.f};
            // RWDH: what the reward needs and does not come out of the physics, loaded ahead of the estimator's stores
            float h_aclip[3] = {0.f, 0.f, 0.f}, h_wprev[3] = {0.f, 0.f, 0.f};
            WalkChanTargets h_wtg[3] = {};
            WalkEnvIn hwin = {};
            if constexpr (HELP) {
                const bool hsettle = P.st.nstep[env] < WK.P.settle_substeps;        // data.time < settling_time (walking_quad.py:142-143)
                if constexpr (RWDH) walk_ldv<3>(WK.S.prev_ctrl + ht[0], h_wprev);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    if constexpr (RWDH) h_wtg[i] = walk_channel_targets(WK.P, 3 * k + i);
                    float a_in = P.actions[(size_t)env * 12 + 3 * k + i];
                    if (hsettle) a_in = WK.P.joint_centers[3 * k + i];
                    h_aclip[i] = fminf(fmaxf(a_in, -1.f), 1.f);                      // quadruped.py:160
                }
                if (k == 0) {
                    hwin = walk_env_load(WK.S, n, env);
                    hwin.episode_key = P.st.episode[env];      // not advanced yet: the physics wave does that behind the barrier
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) hx[i] = P.st.ctrl[(3 * k + i) * n + env];   // data.ctrl of the PREVIOUS step (walking_quad.py:136)
            WalkEstIn<3> hw;
            walk_estimator_load_n<3, true>(WK.P, WK.S, n, ht, hcalls, hw, live);     // block entry in rolled groups: no scratch at 256 registers
            if (live) walk_estimator_finish_n<3>(WK.P, WK.S, n, ht, hx, hcalls, hw, hf, ha);    // math_utils.py:53-131
#pragma unroll
            for (int i = 0; i < 3; ++i) { s_est[wave][lane][i] = hf[i]; s_est[wave][lane][3 + i] = ha[i]; }
            if constexpr (PO) {
                int slot = PK.S.head[env] + 1;                     // the ring slot the NEW frame will take
                if (slot >= PK.P.window) slot = 0;
                if (PK.P.window > 1) po_copy_history_now<4>(PK.P, PK.S, (size_t)env * (PK.P.window * QG_PO_FRAME), slot, k, PK.out, live);
            }
            // (partner: the physics waves' __syncthreads() behind their auto-reset block, "partner of the helper waves' one barrier"
            // below.  Exactly one barrier per wave on every path: a second one in either role, or a return in front of it, deadlocks
            // the workgroup -- and on this pool a hung workgroup is a hung GPU.)
            __syncthreads();      // the one barrier of the workgroup: behind it the physics waves read s_est and write what this wave read
            if constexpr (RWDH) {
                // the reward of the step, on the sensor tile the physics wave has finished (LDS), while that wave goes on with the resets and
                // the state stores
                const bool hdone = s_done[wave][el] != 0.f;
                WalkSums sum = {0.f, 0.f, 0.f, 0.f};
                if (live) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) walk_channel_terms(WK.S, env, 3 * k + i, h_wtg[i], h_aclip[i], h_wprev[i], hf[i], ha[i], sum);
                    walk_stv<3>(WK.S.prev_ctrl + (size_t)env * 12 + 3 * k, h_aclip);                         // previous_ctrl moves on (:260-262)
                }
                sum.cost = quad_sum(sum.cost); sum.posture = quad_sum(sum.posture); sum.amp = quad_sum(sum.amp); sum.frq = quad_sum(sum.frq);
                if (live && k == 0)
                    walk_reward_env(WK.P, WK.S, n, env, tile + el * 35, sum, hwin, hdone, P.reward, WK.comps, WK.sample, P.seed, P.env_index_base);
            }
            if constexpr (POH) {
                // the observation pack's new frame and rows, on the finished sensor tile and the orientation the physics wave handed over,
                // while that wave evaluates the reward and stores the state
                BaseState hb = {};
                hb.qw = s_q[wave][el][0]; hb.qx = s_q[wave][el][1]; hb.qy = s_q[wave][el][2]; hb.qz = s_q[wave][el][3];
                const int h_live_envs = max(0, min(QGK_QUAD_ENVS, n - env0));
                po_wave_epilogue<QGK_QUAD_ENVS, 4, WAVES, true>(PK, WK, P, n, env, env0, h_live_envs, wave, lane, el, k, live, live && k == 0, PoEnvIn{},
                                                                tile + el * 35, hb, hwin, s_done[wave][el] != 0.f, h_aclip);
            }
            return;
        }
    }
    BaseState B;
    B.pw = v3(P.st.qpos[0 * n + env], P.st.qpos[1 * n + env], P.st.qpos[2 * n + env]);
    B.qw = P.st.qpos[3 * n + env]; B.qx = P.st.qpos[4 * n + env]; B.qy = P.st.qpos[5 * n + env]; B.qz = P.st.qpos[6 * n + env];
    B.vw = v3(P.st.qvel[0 * n + env], P.st.qvel[1 * n + env], P.st.qvel[2 * n + env]);
    B.wb = v3(P.st.qvel[3 * n + env], P.st.qvel[4 * n + env], P.st.qvel[5 * n + env]);
    quat_unit(B);   // unit quaternion once per launch (qg_set_state may hand in any length); the substeps keep it normalised (base_prelude<UNIT>)
    LegState L;
    const int nstep0 = P.st.nstep[env];
    float aclip0[3];
    bool settle = false;
    int calls = 0;
    WalkEnvIn win = {};
    if constexpr (WALK) {
        settle = nstep0 < WK.P.settle_substeps;                     // data.time < settling_time (walking_quad.py:142-143)
        if constexpr (!HELP) calls = WK.S.calls[env];
    }
    // WALK: every load of the task layer goes out here, among the state loads, and every store of its prologue part comes after the
    // last load of the kernel's prologue (a load behind a store would wait for that store: vmcnt counts in order)
    const int tt[3] = {env * 12 + 3 * k + 0, env * 12 + 3 * k + 1, env * 12 + 3 * k + 2};     // task state: [n][12]
    float xx[3] = {0.f, 0.f, 0.f}, wprev[3] = {0.f, 0.f, 0.f}, wf[3] = {0.f, 0.f, 0.f}, wa[3] = {0.f, 0.f, 0.f}, a_eff[3] = {0.f, 0.f, 0.f};
    WalkEstIn<3> west;
    WalkChanTargets wtg[3] = {};
    if constexpr (WALK) {
#pragma unroll
        for (int i = 0; i < 3; ++i) if constexpr (!HELP) xx[i] = P.st.ctrl[(3 * k + i) * n + env];   // data.ctrl of the PREVIOUS step: what the estimator takes (walking_quad.py:136)
        walk_ldv<3>(WK.S.prev_ctrl + tt[0], wprev);   // previous_ctrl of the control cost (:260-262)
        if constexpr (!HELP) walk_estimator_load_n<3>(WK.P, WK.S, n, tt, calls, west);
        if constexpr (WPE == 1) {
#pragma unroll
            for (int i = 0; i < 3; ++i) wtg[i] = walk_channel_targets(WK.P, 3 * k + i);
            if (k == 0) {                                   // what the reward epilogue reads: fetched now, behind the physics
                win = walk_env_load(WK.S, n, env);
                win.episode_key = P.st.episode[env];        // not advanced yet: the key of the episode that begins if this one ends
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = 3 * k + i;
        float a_in = P.actions[(size_t)env * 12 + j];
        if constexpr (WALK) {
            if (settle) a_in = WK.P.joint_centers[j];                // the joint centres while the robot settles
            a_eff[i] = a_in;
        }
        float a = fminf(fmaxf(a_in, -1.f), 1.f);    // quadruped.py:160
        aclip0[i] = a;
        L.u[i] = fminf(fmaxf(a, link_of<BAKED>(C, k, i).ctrl_lo), link_of<BAKED>(C, k, i).ctrl_hi);
        L.q[i] = P.st.qpos[(7 + j) * n + env];
        L.qd[i] = P.st.qvel[(6 + j) * n + env];
        L.act[i] = P.st.act[j * n + env];
        sincos_f(L.q[i] - link_of<BAKED>(C, k, i).ref, L.sc[2 * i], L.sc[2 * i + 1]);
    }
    // PO: the env's filter state (lead lane) and ring position, this lane's share of its env's row copy ring -> out -- its loads go
    // out here, among the state loads and before the task layer's stores (see qg_step_kernel_pair)
    PoEnvIn pin = {};
    PoCopyState pcs = {};
    const int live_envs = max(0, min(QGK_QUAD_ENVS, n - env0));
    const int el_c = min(el, max(live_envs - 1, 0));     // row of this lane's env in the wave's block (tail lanes: the last live one)
    unsigned long long po_ring = 0, po_out = 0;          // the wave's block of the frame ring / of the output rows (scalar registers)
    if constexpr (PO) {
        const size_t po_block = (size_t)(live_envs > 0 ? env0 : 0) * (size_t)(PK.P.window * QG_PO_FRAME);
        po_ring = po_uniform_addr(PK.S.stack + 2 * po_block);
        po_out = po_uniform_addr(PK.out + po_block);
        if (k == 0) pin = po_env_load(PK.S, n, env);
        if constexpr (PO_COPY) po_row_copy_init<4>(PK.P, el_c, PK.S.head[env], k, pcs);
    }
    if constexpr (WALK) {
        // every state value is in its register before the first store of the task layer is issued: the waits for those loads
        // would otherwise sit behind the stores (vmcnt is in order) right in front of the substep loop
        asm volatile("" :: "v"(B.pw.x), "v"(B.pw.y), "v"(B.pw.z), "v"(B.qw), "v"(B.qx), "v"(B.qy), "v"(B.qz), "v"(B.vw.x), "v"(B.vw.y), "v"(B.vw.z),
                     "v"(B.wb.x), "v"(B.wb.y), "v"(B.wb.z), "v"(L.q[0]), "v"(L.q[1]), "v"(L.q[2]), "v"(L.qd[0]), "v"(L.qd[1]), "v"(L.qd[2]),
                     "v"(L.act[0]), "v"(L.act[1]), "v"(L.act[2]), "v"(L.u[0]), "v"(L.u[1]), "v"(L.u[2]) : "memory");
        if (live) {
            if constexpr (!HELP) walk_estimator_finish_n<3>(WK.P, WK.S, n, tt, xx, calls, west, wf, wa);    // math_utils.py:53-131
            walk_stv<3>(WK.S.eff_actions + (size_t)env * 12 + 3 * k, a_eff);                        // the action actually applied (the PO pack reads it)
        }
    }

    // the sensor values of the step go straight into this env's row of the output tile (full 33-value layout; the
    // 21-value pack is compacted below)
    float *srow = tile + el * 35;
    float zaxis_z = 1.f;
    const int fs = T->frame_skip;
    const bool lag = T->sensor_lag != 0;
    // Un-lagged sensors (task.sensor_lag = 0) take one extra forward pass whose state changes are discarded: the same loop body
    // runs once more with the state parked in LDS meanwhile -- one copy of the substep code and no extra live registers.
    // (Register caps for three / four waves per SIMD were built and measured in round 2 -- 197 / 255 us against 200 us at 262 144 envs,
    // slower below: profiles/r02/wpe_ab.txt, docs/EXPERIMENTS.md -- and removed in round 4: WPE is 1 or 2.)
    int env_e = env, k_e = k;
    int nstep;
    float aclip[3];
    {
        // Code placement: a wave that is alone on its SIMD is sensitive to where the 14 KB loop body falls relative to the
        // instruction-fetch lines -- the same loop, shifted by one dword through an unrelated edit of the prologue, measured
        // 18.57 instead of 18.36 us per launch at 4096 envs (same-box A/B of eight paddings).  Pinning the loop to a 64-byte
        // boundary makes its layout independent of what precedes it.
        QG_MARK(1);                                  // state in registers, prologue stores issued
        asm volatile(".p2align 6");
        // the history copy's loads at the head of a substep and its stores at the tail -- or both at the head where the registers in
        // between are taken: with two waves per SIMD the partner covers the latency, and the table-driven variant has none to spare
        constexpr bool PO_DEFER = PO && WPE == 1 && BAKED;
#pragma unroll 1
        for (int s = 0; s < fs; ++s) {
            PoCopyRegs<PO ? QG_PO_COPY_K : 1> pcr;
            if constexpr (PO_COPY) po_row_copy_load<QG_PO_COPY_K, 4>(PK.P, po_ring, k, s == 0, pcs, pcr);
            if constexpr (PO_COPY && !PO_DEFER) po_row_copy_store<QG_PO_COPY_K, 4>(PK.P, po_out, live, k, s == 0, pcs, pcr);
            substep_quad<BAKED, (WPE > 1)>(C, cm, sm, B, L, lag && (s == fs - 1), srow, k, zaxis_z);
            if constexpr (PO_DEFER) po_row_copy_store<QG_PO_COPY_K, 4>(PK.P, po_out, live, k, s == 0, pcs, pcr);
        }
        if constexpr (PO_COPY) po_row_copy_rest<QG_PO_COPY_K, 4>(PK.P, po_ring, po_out, live, k, pcs);
        // the epilogue's addresses are derived from (env, leg) AFTER the loop: visible, they are computed before it and carried through
        // it (see the register-capped variants below; with two waves per SIMD that was 60 bytes of scratch in the walking variant)
        if constexpr (WPE > 1 || !BAKED) asm volatile("" : "+v"(env_e), "+v"(k_e));
        QG_MARK(2);                                  // physics done
        if (!lag) {   // un-lagged sensors (task.sensor_lag = 0): one extra forward pass on a scratch copy of the state
            BaseState B2 = B;
            LegState L2 = L;
            substep_quad<BAKED, (WPE > 1)>(C, cm, sm, B2, L2, true, srow, k, zaxis_z);
        }
        nstep = nstep0 + fs;
#pragma unroll
        for (int i = 0; i < 3; ++i) aclip[i] = aclip0[i];
    }
    float ssq = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) ssq = fmaf(aclip[i], aclip[i], ssq);
    ssq = quad_sum(ssq);

    float c_fwd = T->w_forward * B.vw.x;
    float c_ctl = T->w_ctrl * ssq;
    float c_alive = T->alive_bonus;
    float reward = reward_total(c_fwd, c_ctl, c_alive);
    bool done = nstep >= T->limit_substeps;
    if (T->use_fall) done = done || (B.pw.z < T->fall_height);
    {
        float probe = L.q[0] + L.q[1] + L.q[2] + L.qd[0] + L.qd[1] + L.qd[2];
        probe = quad_sum(probe) + B.pw.x + B.pw.y + B.pw.z + B.qw + B.vw.x + B.vw.y + B.vw.z + B.wb.x + B.wb.y + B.wb.z;
        done = done || state_is_bad(probe);
    }

    const int od = T->obs_mode == 1 ? 21 : 33;
    const int row = P.packed ? od + 2 : od;
    if (T->use_flip) done = done || (zaxis_z < 0.f);              // walking_quad.py:156-160, on the step's sensordata
    if (k == 0) {                                                  // lane 0 of the quad wrote these entries itself
        if (od == 21) { srow[18] = srow[30]; srow[19] = srow[31]; srow[20] = srow[32]; }   // IMU pack: velocimeter follows the gyro
        if (P.packed) { srow[od] = reward; srow[od + 1] = done ? 1.f : 0.f; }
    }
    wave_sync();                                                   // the tile is this wave's own
    if constexpr (!PO) {
        const int total = live_envs * row;                               // (a whole wave may lie past the last env: live_envs = 0)
        float *dst = (P.packed ? P.packed : P.obs) + (size_t)env0 * row;
        if (row == 35) {                       // rows were staged with a stride of 35 floats: the packed full layout is a straight copy
            for (int e = lane; e < total; e += QGK_WAVE) dst[e] = tile[e];
        } else {
            // e / row without a division per element: row is 21, 23 or 33 here and e < 2^11, where (e * ceil(2^16 / row)) >> 16 is exact
            const unsigned magic = row == 33 ? 1986u : row == 21 ? 3121u : row == 23 ? 2850u : (65536u + row - 1) / row;
            for (int e = lane; e < total; e += QGK_WAVE) {
                const int er = (int)(((unsigned)e * magic) >> 16), ec = e - er * row;
                dst[e] = tile[er * 35 + ec];
            }
        }
    }
    QG_MARK(3);                                      // obs tile written out
    const bool lead = live && k_e == 0;
    if (lead && !P.packed) {
        if constexpr (!WALK) P.reward[env_e] = reward;
        P.done[env_e] = done ? 1 : 0;
    }
    bool rst_done = false;
    if constexpr (POH) {
        // the auto-reset of the base FIRST (the reward below does not look at B): the helper wave's frame shows data.qpos[3:7] as the step
        // leaves it
        if (done && T->auto_reset) {
            B.pw = v3(C.qpos0[0], C.qpos0[1], C.qpos0[2]);
            B.qw = C.qpos0[3]; B.qx = C.qpos0[4]; B.qy = C.qpos0[5]; B.qz = C.qpos0[6];
            if (T->reset_flags & 1u) {
                float a = 6.283185307179586f * uniform24(P.seed, P.env_index_base + (uint64_t)env_e, (uint64_t)P.st.episode[env_e]);
                float sn, cs;
                sincos_f(0.5f * a, sn, cs);
                B.qw = cs; B.qx = 0.f; B.qy = 0.f; B.qz = sn;
            }
            B.vw = v3(0.f, 0.f, 0.f);
            B.wb = v3(0.f, 0.f, 0.f);
            nstep = 0;
        }
        rst_done = true;
        if (k_e == 0) {
            s_done[wave][lane >> 2] = done ? 1.f : 0.f;
            s_q[wave][lane >> 2][0] = B.qw; s_q[wave][lane >> 2][1] = B.qx; s_q[wave][lane >> 2][2] = B.qy; s_q[wave][lane >> 2][3] = B.qz;
        }
    }
    if constexpr (WALK) {
        WalkSums sum = {0.f, 0.f, 0.f, 0.f};
        if constexpr (WPE > 1) {       // two waves share the SIMD: read the task state again here (the other wave covers the latency)
#pragma unroll                        // instead of carrying 21 values through the substep loop
            for (int i = 0; i < 3; ++i) {
                const int t = env_e * 12 + 3 * k_e + i;
                if constexpr (!RWDH) wprev[i] = WK.S.prev_ctrl[t];
                if constexpr (!HELP) { wf[i] = WK.S.f_est[t]; wa[i] = WK.S.a_est[t]; }
                if constexpr (!RWDH) wtg[i] = walk_channel_targets(WK.P, 3 * k_e + i);
            }
            if ((!RWDH || PO) && lead) {
                win = walk_env_load(WK.S, n, env_e);
                win.episode_key = P.st.episode[env_e];
            }
        }
        if constexpr (HELP) {
            // the helper wave of this SIMD finished its first job long ago (its estimator stores have landed: the barrier's wait covers
            // them); from here on this wave may overwrite what the helper read at entry (data.ctrl, the episode counter), and (RWDH)
            // the helper evaluates the reward on the finished sensor tile
            if constexpr (RWDH) { if (k_e == 0) s_done[wave][lane >> 2] = done ? 1.f : 0.f; }
            __syncthreads();      // partner of the helper waves' one barrier
            if constexpr (!RWDH) {
#pragma unroll
                for (int i = 0; i < 3; ++i) { wf[i] = s_est[wave][lane][i]; wa[i] = s_est[wave][lane][3 + i]; }
            }
        }
        if (!RWDH && live) {
#pragma unroll
            for (int i = 0; i < 3; ++i) walk_channel_terms(WK.S, env_e, 3 * k_e + i, wtg[i], aclip[i], wprev[i], wf[i], wa[i], sum);
            walk_stv<3>(WK.S.prev_ctrl + (size_t)env_e * 12 + 3 * k_e, aclip);                       // previous_ctrl moves on (:260-262)
        }
        if constexpr (!RWDH) { sum.cost = quad_sum(sum.cost); sum.posture = quad_sum(sum.posture); sum.amp = quad_sum(sum.amp); sum.frq = quad_sum(sum.frq); }
        QG_MARK(4);                                  // channel terms + sums
        if (!RWDH && lead) walk_reward_env(WK.P, WK.S, n, env_e, tile + (lane >> 2) * 35, sum, win, done, P.reward, WK.comps, WK.sample, P.seed, P.env_index_base);
    }
    if (lead && P.comps) {
        P.comps[(size_t)env_e * 3 + 0] = c_fwd;
        P.comps[(size_t)env_e * 3 + 1] = c_ctl;
        P.comps[(size_t)env_e * 3 + 2] = c_alive;
    }

    const bool rst = done && T->auto_reset;
    if (rst && !rst_done) {
        B.pw = v3(C.qpos0[0], C.qpos0[1], C.qpos0[2]);
        B.qw = C.qpos0[3]; B.qx = C.qpos0[4]; B.qy = C.qpos0[5]; B.qz = C.qpos0[6];
        if (T->reset_flags & 1u) {
            float a = 6.283185307179586f * uniform24(P.seed, P.env_index_base + (uint64_t)env_e, (uint64_t)P.st.episode[env_e]);
            float sn, cs;
            sincos_f(0.5f * a, sn, cs);
            B.qw = cs; B.qx = 0.f; B.qy = 0.f; B.qz = sn;
        }
        B.vw = v3(0.f, 0.f, 0.f);
        B.wb = v3(0.f, 0.f, 0.f);
        nstep = 0;
    }
    if (lead) {
        P.st.qpos[0 * n + env_e] = B.pw.x; P.st.qpos[1 * n + env_e] = B.pw.y; P.st.qpos[2 * n + env_e] = B.pw.z;
        P.st.qpos[3 * n + env_e] = B.qw; P.st.qpos[4 * n + env_e] = B.qx; P.st.qpos[5 * n + env_e] = B.qy; P.st.qpos[6 * n + env_e] = B.qz;
        P.st.qvel[0 * n + env_e] = B.vw.x; P.st.qvel[1 * n + env_e] = B.vw.y; P.st.qvel[2 * n + env_e] = B.vw.z;
        P.st.qvel[3 * n + env_e] = B.wb.x; P.st.qvel[4 * n + env_e] = B.wb.y; P.st.qvel[5 * n + env_e] = B.wb.z;
        P.st.nstep[env_e] = nstep;
        if (rst) P.st.episode[env_e] += 1;
    }
    if (live) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int j = 3 * k_e + i;
            P.st.qpos[(7 + j) * n + env_e] = rst ? C.qpos0[7 + (BAKED ? i : j)] : L.q[i];
            P.st.qvel[(6 + j) * n + env_e] = rst ? 0.f : L.qd[i];
            P.st.act[j * n + env_e] = rst ? 0.f : L.act[i];
            if (P.track_ctrl) P.st.ctrl[j * n + env_e] = rst ? T->default_ctrl[j] : aclip[i];
        }
    }
    if constexpr (PO && !POH)
        po_wave_epilogue<QGK_QUAD_ENVS, 4, WAVES, (WPE > 1)>(PK, WK, P, n, env, env0, live_envs, wave, lane, el, k, live, lead, pin, srow, B, win, done, aclip);
}

// ------------------------------------------------------------------------------------------
// env-step kernel, TWO LEGS PER LANE (packed FP32): the two lanes 2e, 2e+1 share env e; lane h owns legs 2h and
// 2h+1 as the two components of f2 quantities, so every fma / mul / add of the leg physics is ONE v_pk_* instruction for
// both legs.  32 envs per wave.  Base prelude, FRAME terms, 6x6 solve and base integration are scalar and redundant in the
// two lanes; leg terms are summed over the two components (a scalar add of the two halves of the register pair) and over the
// lane pair (one DPP add).  Per env this issues 0.59x the instructions of one leg per lane.  v_pk_*_f32 is half rate on gfx950
// (tools/ubench/valu_rate.hip), so this pays only where a wave has its SIMD to itself anyway -- a lone wave issues at most
// every ~4.5 cycles, packed or not: grids of 16-32 Ki envs and, by a smaller margin, >= 56 Ki.  Compiled-in robot only.
// ------------------------------------------------------------------------------------------
DEV float pair_sum(float x) {                                     // quad_perm [1,0,3,2]: the other lane of the pair
#pragma clang fp contract(off)
    return x + dpp_quad<0xB1>(x);
}
DEV float hsum(f2 v) {
#pragma clang fp contract(off)
    return v.x + v.y;
}
DEV V3 hsum(V3T<f2> v) { return v3<float>(hsum(v.x), hsum(v.y), hsum(v.z)); }
DEV V3 pair_sum(V3 a) { return v3<float>(pair_sum(a.x), pair_sum(a.y), pair_sum(a.z)); }
DEV Sym3 hpsum(const Sym3T<f2> &a) {
    Sym3 r = {pair_sum(hsum(a.xx)), pair_sum(hsum(a.yy)), pair_sum(hsum(a.zz)), pair_sum(hsum(a.xy)), pair_sum(hsum(a.xz)), pair_sum(hsum(a.yz))};
    return r;
}
DEV Sym6 hpsum(const Sym6T<f2> &a) {
    Sym6 r;
    r.AA = hpsum(a.AA);
    r.LL = hpsum(a.LL);
    r.AL.r0 = pair_sum(hsum(a.AL.r0)); r.AL.r1 = pair_sum(hsum(a.AL.r1)); r.AL.r2 = pair_sum(hsum(a.AL.r2));
    return r;
}
DEV SV hpsum(const SVT<f2> &v) { SV r = {pair_sum(hsum(v.a)), pair_sum(hsum(v.l))}; return r; }

struct LegPair { f2 q[3], qd[3], act[3], u[3], sc[6]; };   // sc: sin, cos of (q[i] - ref_i) of both legs, advanced with the hinges

DEV void substep_pair(const KModel &C, f2 cm, f2 sm, BaseState &B, LegPair &L, bool want_sensors, float *__restrict__ row, int half, float &zaxis_z) {
    const float h = C.h;
    const BaseCtx bc0 = base_prelude<true>(C, B);
    V3 gb_keep;
    Sym6 Ic;
    SV fc, Fu;
    f2 Y0[6], Y1[6], Y2[6], u[3];
    {
        const BaseCtx &bc = bc0;
        gb_keep = bc.gb;
        if (want_sensors) {          // the step's sensordata describes the state at the start of its last substep
            zaxis_z = bc.cz.z;
            float *jr = row + 6 * half;                  // legs 2*half and 2*half + 1
            jr[0] = L.q[0].x; jr[1] = L.q[1].x; jr[2] = L.q[2].x;
            jr[3] = L.q[0].y; jr[4] = L.q[1].y; jr[5] = L.q[2].y;
            if (half == 0) {
                row[15] = B.wb.x; row[16] = B.wb.y; row[17] = B.wb.z;
                row[18] = B.pw.x; row[19] = B.pw.y; row[20] = B.pw.z;
                row[21] = B.vw.x; row[22] = B.vw.y; row[23] = B.vw.z;
                row[24] = bc.cx.x; row[25] = bc.cx.y; row[26] = bc.cx.z;
                row[27] = bc.cz.x; row[28] = bc.cz.y; row[29] = bc.cz.z;
                row[30] = bc.vb.x; row[31] = bc.vb.y; row[32] = bc.vb.z;
            }
        }
        // both legs of this lane, each in the frame turned by its quarter turn: there each of them is leg 0
        const f2 z2 = f2(0.f), o2 = f2(1.f);
        FrT<f2> Ek = {v3<f2>(cm, sm, z2), v3<f2>(-sm, cm, z2), v3<f2>(z2, z2, o2)};
        Sym6T<f2> Ic2, YFt2;
        SVT<f2> fc2, F2[3], Fu2;
        f2 Hd[3], H01, H02, H12, bj[3];
        leg_pass<f2, true, true, false, true>(C, 0, Ek, L.q, L.qd, L.act, bc, B.pw.z, h, Ic2, fc2, F2, Hd, H01, H02, H12, bj, L.sc);
        leg_eliminate<f2>(F2, Hd, H01, H02, H12, bj, Y0, Y1, Y2, u, YFt2, Fu2);
        sub(Ic2, YFt2);                     // the two legs' Schur complements
        // Sum over the two legs of the lane and over the two lanes of the env, HERE, far from the uses: the DPP moves then stay unfused
        // (33 v_mov_b32_dpp per substep: the compiler fuses a move into its add only when the two are close, see substep_quad), but
        // with the sums right in front of the 6x6 solve the lone wave of this kernel ran 2.6 % slower at 32 768 and at 262 144 envs
        // (same-box A/B, round 3: 25.0 -> 25.65 us; the dependent chain sum -> assemble -> solve has nothing to overlap with there)
        Ic = hpsum(Ic2);
        fc = hpsum(fc2);
        Fu = hpsum(Fu2);
    }
    float x6[6];
    {
        const BaseCtx &bc = bc0;
        SV p0;
        Sym6 Ic0;
        frame_body(C, bc, h, p0, Ic0);
        SV b;
        {   // FRAME contact: this lane evaluates two quarter turns of the three base sample points
            f2 wsum2 = f2(0.f);
            V3T<f2> s2 = v3<f2>(f2(0.f), f2(0.f), f2(0.f));
            const f2 zb = f2(C.contact_margin - B.pw.z);
            const V3T<f2> n2 = splat3<f2>(bc.n);
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                V3 r0 = ld3(C.cp0[4 * o]);
                contact_point(v3<f2>(cm * r0.x - sm * r0.y, sm * r0.x + cm * r0.y, f2(r0.z)), n2, zb, wsum2, s2);
            }
            float wsum = pair_sum(hsum(wsum2));
            add(Ic0, Ic);
            b.a = v3<float>(0.f, 0.f, 0.f) - Fu.a - p0.a - fc.a;
            b.l = v3<float>(0.f, 0.f, 0.f) - Fu.l - p0.l - fc.l;
            if (__any(wsum > 0.f)) {        // wave-uniform skip, as in the quad kernel (the block is updated in place)
                V3 s = pair_sum(hsum(s2));
                Fr E0 = {v3<float>(1.f, 0.f, 0.f), v3<float>(0.f, 1.f, 0.f), v3<float>(0.f, 0.f, 1.f)};
                SV fe;
                contact_finish<float>(wsum, s, E0, v3<float>(0.f, 0.f, 0.f), bc.n, bc.V0, C.contact_k, C.contact_c, C.contact_inv_ramp, C.contact_mu, h, fe, Ic0);
                b.a = b.a + fe.a;
                b.l = b.l + fe.l;
            }
        }
        base_solve(Ic0, b, x6);     // (pk3::base_solve: 36 instructions fewer and 2.7 % SLOWER here, 24.96 -> 25.64 us at 32 768 envs: the solve sits on
                                     // this kernel's dependent tail, and a dependent packed instruction costs a lone wave ~1.5 plain ones; the same right-looking
                                     // order in plain FP32, 15 instructions fewer: no difference, 24.9-25.1 us either way)
    }
    V3 wdot = v3<float>(x6[0], x6[1], x6[2]);
    V3 acl = v3<float>(x6[3], x6[4], x6[5]);
    if (want_sensors && half == 0) {
        row[12] = acl.x - gb_keep.x; row[13] = acl.y - gb_keep.y; row[14] = acl.z - gb_keep.z;   // accelerometer
    }
    // back-substitution and integration of this lane's six hinges
    const f2 *Y[3] = {Y0, Y1, Y2};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        f2 acc = u[i];
#pragma unroll
        for (int r = 0; r < 6; ++r) acc = fma_(-Y[i][r], f2(x6[r]), acc);
        L.qd[i] = fma_(f2(h), acc, L.qd[i]);
        L.q[i] = fma_(f2(h), L.qd[i], L.q[i]);
        hinge_advance(f2(h) * L.qd[i], L.sc[2 * i], L.sc[2 * i + 1]);
        L.act[i] = fma_(L.u[i] - L.act[i], f2(C.link[i].act_decay), L.act[i]);
    }
    {
        const BaseCtx &bc = bc0;
        base_integrate<true>(bc, h, wdot, acl, B);
    }
}

#define QGK_PAIR_ENVS 32    // envs per wave in the two-legs-per-lane kernel


// One wave per SIMD by construction (381 registers): capping it to 256 for two resident waves spills 592 B per lane and
// measured slower than this variant at every size (profiles/r01/pair_sweep.txt), so there is only this one.
// WAVES: waves per workgroup (1 or 4, one per SIMD of a CU; they do not interact).  A grid of 1024 one-wave workgroups costs ~1.6 us
// more fixed time per launch than 256 four-wave ones (the dispatch of the workgroups themselves: tools/fs_sweep.sh on the
// one-link-per-lane kernel), so grids of more than 256 waves are launched as four-wave workgroups.
// WALK: the walking task layer fused in as in qg_step_kernel_quad<.., WALK>; the lane owns the six control channels of its two legs
// (6 * half .. 6 * half + 5, contiguous in the env-major task state).
// PO (with WALK; round 3): the partially observable observation pack fused in as well -- POWalkingQuadrupedEnv.step is this one launch
// also at the batch sizes this kernel serves.  The copy of the W - 1 frames the new stack keeps rides on the substep loop
// (po_row_copy_*, qg_po_dev.h: loads at the head of a substep, stores at its tail), the env's lead lane runs the orientation filter
// on the step's sensors in the epilogue, the wave writes the new frames.  The 33 sensors themselves are not written to memory.
template <int WAVES, bool WALK = false, bool PO = false>
__global__ __launch_bounds__(QGK_WAVE * WAVES, 1) void qg_step_kernel_pair(const KTask *__restrict__ T, KStepArgs P,
                                                                           const typename WalkArgT<WALK>::type WK,
                                                                           const typename PoArgT<PO>::type PK) {
    static_assert(WALK || !PO, "the observation pack rides on the walking task layer");
    __shared__ float tile_all[WAVES][QGK_PAIR_ENVS * 35];
    QG_MARK(0);
    const KModel &C = QG_BAKED_MODEL;
    const int lane = threadIdx.x & (QGK_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    float *tile = tile_all[wave];
    const int half = lane & 1;                      // legs 2*half, 2*half + 1
    const int el = lane >> 1;                       // env within the wave
    const int env0 = (blockIdx.x * WAVES + wave) * QGK_PAIR_ENVS;
    const int n = P.n;
    const bool live = env0 + el < n;
    int env = live ? env0 + el : n - 1;             // tail pairs shadow the last env; their stores are masked
    // quarter turns of this lane's two legs: (cos, sin)(90 deg * k), k = 2*half and 2*half + 1
    f2 cm, sm;
    cm.x = half ? -1.f : 1.f; cm.y = 0.f;
    sm.x = 0.f; sm.y = half ? -1.f : 1.f;

    BaseState B;
    B.pw = v3<float>(P.st.qpos[0 * n + env], P.st.qpos[1 * n + env], P.st.qpos[2 * n + env]);
    B.qw = P.st.qpos[3 * n + env]; B.qx = P.st.qpos[4 * n + env]; B.qy = P.st.qpos[5 * n + env]; B.qz = P.st.qpos[6 * n + env];
    B.vw = v3<float>(P.st.qvel[0 * n + env], P.st.qvel[1 * n + env], P.st.qvel[2 * n + env]);
    B.wb = v3<float>(P.st.qvel[3 * n + env], P.st.qvel[4 * n + env], P.st.qvel[5 * n + env]);
    quat_unit(B);   // unit quaternion once per launch (qg_set_state may hand in any length); the substeps keep it normalised (base_prelude<UNIT>)
    int nstep = P.st.nstep[env];
    LegPair L;
    float aclip[6];
    float ssq = 0.f;
    // WALK: every load of the task layer goes out among the state loads, every store of its prologue part after the last of them
    // (see qg_step_kernel_quad)
    bool settle = false;
    int calls = 0;
    WalkEnvIn win = {};
    int tt[6] = {0, 0, 0, 0, 0, 0};
    float xx[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, wprev[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, wf[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f},
          wa[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, a_eff[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    WalkEstIn<6> west;
    WalkChanTargets wtg[6] = {};
    if constexpr (WALK) {
        settle = nstep < WK.P.settle_substeps;                      // data.time < settling_time (walking_quad.py:142-143)
        calls = WK.S.calls[env];
#pragma unroll
        for (int c6 = 0; c6 < 6; ++c6) {
            tt[c6] = env * 12 + 6 * half + c6;
            xx[c6] = P.st.ctrl[(6 * half + c6) * n + env];           // data.ctrl of the PREVIOUS step (walking_quad.py:136)
        }
        {
            float pc[6];
            walk_ldv<6>(WK.S.prev_ctrl + env * 12 + 6 * half, pc);    // previous_ctrl of the control cost (:260-262)
#pragma unroll
            for (int c6 = 0; c6 < 6; ++c6) wprev[c6] = pc[c6];
        }
        walk_estimator_load_n<6>(WK.P, WK.S, n, tt, calls, west);
#pragma unroll
        for (int c6 = 0; c6 < 6; ++c6) wtg[c6] = walk_channel_targets(WK.P, 6 * half + c6);
        if (half == 0) {
            win = walk_env_load(WK.S, n, env);
            win.episode_key = P.st.episode[env];              // not advanced yet: the key of the episode that begins if this one ends
        }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int j = 3 * (2 * half + c) + i;
            float a_in = P.actions[(size_t)env * 12 + j];
            if constexpr (WALK) {
                if (settle) a_in = WK.P.joint_centers[j];            // the joint centres while the robot settles
                a_eff[3 * c + i] = a_in;
            }
            float a = fminf(fmaxf(a_in, -1.f), 1.f);    // quadruped.py:160
            aclip[3 * c + i] = a;
            ssq = fmaf(a, a, ssq);
            float uu = fminf(fmaxf(a, C.link[i].ctrl_lo), C.link[i].ctrl_hi);
            float qq = P.st.qpos[(7 + j) * n + env], qv = P.st.qvel[(6 + j) * n + env], aa = P.st.act[j * n + env];
            if (c == 0) { L.u[i].x = uu; L.q[i].x = qq; L.qd[i].x = qv; L.act[i].x = aa; }
            else { L.u[i].y = uu; L.q[i].y = qq; L.qd[i].y = qv; L.act[i].y = aa; }
        }
    }
    ssq = pair_sum(ssq);
#pragma unroll
    for (int i = 0; i < 3; ++i) sincos_f(L.q[i] - f2(C.link[i].ref), L.sc[2 * i], L.sc[2 * i + 1]);
    // PO: the env's filter state (lead lane) and ring position, this lane's share of its env's row copy ring -> out -- its loads go
    // out here, among the state loads and before the task layer's stores
    PoEnvIn pin = {};
    PoCopyState pcs = {};
    const int live_envs = max(0, min(QGK_PAIR_ENVS, n - env0));
    const int el_c = min(el, max(live_envs - 1, 0));     // row of this lane's env in the wave's block (tail lanes: the last live one)
    unsigned long long po_ring = 0, po_out = 0;          // the wave's block of the frame ring / of the output rows (scalar registers)
    if constexpr (PO) {
        // (a wave that lies wholly past the last env copies nothing, but its loads are unpredicated: they read block 0)
        const size_t po_block = (size_t)(live_envs > 0 ? env0 : 0) * (size_t)(PK.P.window * QG_PO_FRAME);
        po_ring = po_uniform_addr(PK.S.stack + 2 * po_block);
        po_out = po_uniform_addr(PK.out + po_block);
        if (half == 0) pin = po_env_load(PK.S, n, env);
        po_row_copy_init<2>(PK.P, el_c, PK.S.head[env], half, pcs);
    }
    if constexpr (WALK) {
        asm volatile("" :: "v"(B.pw.x), "v"(B.pw.y), "v"(B.pw.z), "v"(B.qw), "v"(B.qx), "v"(B.qy), "v"(B.qz), "v"(B.vw.x), "v"(B.vw.y), "v"(B.vw.z),
                     "v"(B.wb.x), "v"(B.wb.y), "v"(B.wb.z), "v"(L.q[0].x), "v"(L.q[1].x), "v"(L.q[2].x), "v"(L.q[0].y), "v"(L.q[1].y), "v"(L.q[2].y),
                     "v"(L.qd[0].x), "v"(L.qd[1].x), "v"(L.qd[2].x), "v"(L.qd[0].y), "v"(L.qd[1].y), "v"(L.qd[2].y),
                     "v"(L.act[0].x), "v"(L.act[1].x), "v"(L.act[2].x), "v"(L.act[0].y), "v"(L.act[1].y), "v"(L.act[2].y) : "memory");
        if (live) {
            walk_estimator_finish_n<6>(WK.P, WK.S, n, tt, xx, calls, west, wf, wa);    // math_utils.py:53-131
            walk_stv<6>(WK.S.eff_actions + (size_t)env * 12 + 6 * half, a_eff);
        }
    }

    // data.ctrl of this step (quadruped.py:164: the env-clipped action) is known here: it goes out now, behind the loads (the task
    // layer has read the PREVIOUS one above), instead of among the epilogue's stores -- at 32 768 envs all 1024 waves reach their 12.7 MB
    // of epilogue stores at the same moment; an env the step resets gets the default there.  (Same-box A/B, round 3: -0.9 % here, but
    // +0.4 us in the one-link-per-lane kernel at 4096 envs and +0.5 % in the one-leg-per-lane kernel at 16 384: only this kernel has it.)
    if (live && P.track_ctrl) {
#pragma unroll
        for (int c6 = 0; c6 < 6; ++c6) P.st.ctrl[(6 * half + c6) * n + env] = aclip[c6];
    }
    float *srow = tile + el * 35;
    float zaxis_z = 1.f;
    const int fs = T->frame_skip;
    const bool lag = T->sensor_lag != 0;
    // PO: the env's filter state (lead lane) and ring position, this lane's share of its env's row copy ring -> out
    QG_MARK(1);                                      // state in registers, prologue stores issued
#pragma unroll 1
    for (int s = 0; s < fs; ++s) {
        PoCopyRegs<PO ? QG_PO_COPY_K : 1> pcr;
        if constexpr (PO) po_row_copy_load<QG_PO_COPY_K, 2>(PK.P, po_ring, half, s == 0, pcs, pcr);
        substep_pair(C, cm, sm, B, L, lag && (s == fs - 1), srow, half, zaxis_z);
        if constexpr (PO) po_row_copy_store<QG_PO_COPY_K, 2>(PK.P, po_out, live, half, s == 0, pcs, pcr);
    }
    if constexpr (PO) po_row_copy_rest<QG_PO_COPY_K, 2>(PK.P, po_ring, po_out, live, half, pcs);
    nstep += fs;
    // the epilogue's store addresses are derived from `env` AFTER the loop: left visible, the compiler computes two dozen 64-bit
    // addresses before the loop and carries them through it -- ~60 registers of a kernel that already parks values in AGPRs
    // (tools/asm_liveness.py found the same in the one-leg-per-lane kernel in round 2)
    asm volatile("" : "+v"(env));
    QG_MARK(2);                                      // physics done
    if (!lag) {
        BaseState B2 = B;
        LegPair L2 = L;
        substep_pair(C, cm, sm, B2, L2, true, srow, half, zaxis_z);
    }

    float c_fwd = T->w_forward * B.vw.x;
    float c_ctl = T->w_ctrl * ssq;
    float c_alive = T->alive_bonus;
    float reward = reward_total(c_fwd, c_ctl, c_alive);
    bool done = nstep >= T->limit_substeps;
    if (T->use_fall) done = done || (B.pw.z < T->fall_height);
    {
        float probe = hsum(L.q[0]) + hsum(L.q[1]) + hsum(L.q[2]) + hsum(L.qd[0]) + hsum(L.qd[1]) + hsum(L.qd[2]);
        probe = pair_sum(probe) + B.pw.x + B.pw.y + B.pw.z + B.qw + B.vw.x + B.vw.y + B.vw.z + B.wb.x + B.wb.y + B.wb.z;
        done = done || state_is_bad(probe);
    }
    const int od = T->obs_mode == 1 ? 21 : 33;
    const int row = P.packed ? od + 2 : od;
    if (T->use_flip) done = done || (zaxis_z < 0.f);
    if (half == 0) {
        if (od == 21) { srow[18] = srow[30]; srow[19] = srow[31]; srow[20] = srow[32]; }
        if (P.packed) { srow[od] = reward; srow[od + 1] = done ? 1.f : 0.f; }
    }
    wave_sync();                                                   // the tile is this wave's own
    if constexpr (!PO) {
        const int total = live_envs * row;                               // (a whole wave may lie past the last env: live_envs = 0)
        float *dst = (P.packed ? P.packed : P.obs) + (size_t)env0 * row;
        if (row == 35) {
            for (int e = lane; e < total; e += QGK_WAVE) dst[e] = tile[e];
        } else {
            // e / row without a division per element: row is 21, 23 or 33 here and e < 2^11, where (e * ceil(2^16 / row)) >> 16 is exact
            const unsigned magic = row == 33 ? 1986u : row == 21 ? 3121u : row == 23 ? 2850u : (65536u + row - 1) / row;
            for (int e = lane; e < total; e += QGK_WAVE) {
                const int er = (int)(((unsigned)e * magic) >> 16), ec = e - er * row;
                dst[e] = tile[er * 35 + ec];
            }
        }
    }
    QG_MARK(3);                                      // obs tile written out
    const bool lead = live && half == 0;
    if (lead && !P.packed) {
        if constexpr (!WALK) P.reward[env] = reward;
        P.done[env] = done ? 1 : 0;
    }
    if constexpr (WALK) {
        WalkSums sum = {0.f, 0.f, 0.f, 0.f};
        if (live) {
#pragma unroll
            for (int c6 = 0; c6 < 6; ++c6) walk_channel_terms(WK.S, env, 6 * half + c6, wtg[c6], aclip[c6], wprev[c6], wf[c6], wa[c6], sum);
            walk_stv<6>(WK.S.prev_ctrl + (size_t)env * 12 + 6 * half, aclip);                          // previous_ctrl moves on (:260-262)
        }
        sum.cost = pair_sum(sum.cost); sum.posture = pair_sum(sum.posture); sum.amp = pair_sum(sum.amp); sum.frq = pair_sum(sum.frq);
        QG_MARK(4);                                  // channel terms + sums
        if (lead) walk_reward_env(WK.P, WK.S, n, env, tile + el * 35, sum, win, done, P.reward, WK.comps, WK.sample, P.seed, P.env_index_base);
        QG_MARK(5);                                  // reward
    }
    if (lead && P.comps) {
        P.comps[(size_t)env * 3 + 0] = c_fwd;
        P.comps[(size_t)env * 3 + 1] = c_ctl;
        P.comps[(size_t)env * 3 + 2] = c_alive;
    }

    const bool rst = done && T->auto_reset;
    if (rst) {
        B.pw = v3<float>(C.qpos0[0], C.qpos0[1], C.qpos0[2]);
        B.qw = C.qpos0[3]; B.qx = C.qpos0[4]; B.qy = C.qpos0[5]; B.qz = C.qpos0[6];
        if (T->reset_flags & 1u) {
            float a = 6.283185307179586f * uniform24(P.seed, P.env_index_base + (uint64_t)env, (uint64_t)P.st.episode[env]);
            float sn, cs;
            sincos_f(0.5f * a, sn, cs);
            B.qw = cs; B.qx = 0.f; B.qy = 0.f; B.qz = sn;
        }
        B.vw = v3<float>(0.f, 0.f, 0.f);
        B.wb = v3<float>(0.f, 0.f, 0.f);
        nstep = 0;
    }
    if (lead) {
        P.st.qpos[0 * n + env] = B.pw.x; P.st.qpos[1 * n + env] = B.pw.y; P.st.qpos[2 * n + env] = B.pw.z;
        P.st.qpos[3 * n + env] = B.qw; P.st.qpos[4 * n + env] = B.qx; P.st.qpos[5 * n + env] = B.qy; P.st.qpos[6 * n + env] = B.qz;
        P.st.qvel[0 * n + env] = B.vw.x; P.st.qvel[1 * n + env] = B.vw.y; P.st.qvel[2 * n + env] = B.vw.z;
        P.st.qvel[3 * n + env] = B.wb.x; P.st.qvel[4 * n + env] = B.wb.y; P.st.qvel[5 * n + env] = B.wb.z;
        P.st.nstep[env] = nstep;
        if (rst) P.st.episode[env] += 1;
    }
    if (live) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int j = 3 * (2 * half + c) + i;
                P.st.qpos[(7 + j) * n + env] = rst ? C.qpos0[7 + i] : (c == 0 ? L.q[i].x : L.q[i].y);
                P.st.qvel[(6 + j) * n + env] = rst ? 0.f : (c == 0 ? L.qd[i].x : L.qd[i].y);
                P.st.act[j * n + env] = rst ? 0.f : (c == 0 ? L.act[i].x : L.act[i].y);
                if (P.track_ctrl && rst) P.st.ctrl[j * n + env] = T->default_ctrl[j];          // (the step's data.ctrl went out in the prologue)
            }
        }
    }
    QG_MARK(6);                                      // reset block, state stores issued
    if constexpr (PO)
        po_wave_epilogue<QGK_PAIR_ENVS, 2, WAVES, false>(PK, WK, P, n, env, env0, live_envs, wave, lane, el, half, live, lead, pin, srow, B, win, done, aclip);
    QG_MARK(9);
}

// ------------------------------------------------------------------------------------------
// reset (quadruped.py:115-139): mj_resetData, time = 0, ctrl = default; optional random yaw and hinge jitter
// ------------------------------------------------------------------------------------------
__global__ void qg_reset_kernel(const KModel *__restrict__ M, const KTask *__restrict__ T, KState st, int n, const uint8_t *mask,
                                uint64_t seed, uint64_t env_index_base, uint32_t flags, int count_episode) {
    int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= n) return;
    if (mask && !mask[env]) return;
    const int ep = st.episode[env];
    for (int j = 0; j < 19; ++j) st.qpos[j * n + env] = M->qpos0[j];
    if (flags & 2u) {
        for (int j = 0; j < 12; ++j)
            st.qpos[(7 + j) * n + env] = jittered_hinge(M->qpos0[7 + j], M->link[j].lo, M->link[j].hi, T->reset_joint_jitter, seed,
                                                        env_index_base + (uint64_t)env, ep, j);
    }
    if (flags & 1u) {
        float a = 6.283185307179586f * uniform24(seed, env_index_base + (uint64_t)env, (uint64_t)ep);
        float sn, cs;
        sincos_f(0.5f * a, sn, cs);
        st.qpos[3 * n + env] = cs; st.qpos[4 * n + env] = 0.f; st.qpos[5 * n + env] = 0.f; st.qpos[6 * n + env] = sn;
    }
    for (int j = 0; j < 18; ++j) st.qvel[j * n + env] = 0.f;
    for (int j = 0; j < 12; ++j) { st.act[j * n + env] = 0.f; st.ctrl[j * n + env] = T->default_ctrl[j]; }
    st.nstep[env] = 0;
    if (count_episode) st.episode[env] += 1;
}

// QG_RESET_JOINT_JITTER for the envs the step kernel has just auto-reset: their hinges stand at qpos0 and their episode counter
// has advanced, so the key of the episode that begins is episode - 1 (the one its reset yaw used).  A separate launch, issued
// only when the task asks for jitter: inside the step kernels even a never-taken branch of this size measured +0.9 % at 4096
// envs and +2 % at 32 768 (same-box A/B).  One thread per (hinge, env); `done` is the step's done output, either as bytes
// or as the last column of the packed rows.
__global__ void qg_jitter_kernel(const KModel *__restrict__ M, const KTask *__restrict__ T, KState st, int n, const uint8_t *__restrict__ done,
                                 const float *__restrict__ packed, int row, uint64_t seed, uint64_t env_index_base) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 12 * n) return;
    const int j = t / n, env = t - j * n;
    const bool was_reset = done ? (done[env] != 0) : (packed[(size_t)env * row + (row - 1)] > 0.5f);
    if (!was_reset) return;
    st.qpos[(7 + j) * n + env] = jittered_hinge(M->qpos0[7 + j], M->link[j].lo, M->link[j].hi, T->reset_joint_jitter, seed,
                                                env_index_base + (uint64_t)env, st.episode[env] - 1, j);
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
