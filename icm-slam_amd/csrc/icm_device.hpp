// Device-side maths of the ICM pose solve for gfx950 (wave64): angle wrap, the conditional
// energies of SURVEY.md Appendix A.4 and SciPy's Nelder-Mead (Appendix A.5), all FP64.
// One wavefront solves one pose: lanes stride over the pose's observation items and the
// partial energies are combined with a butterfly reduction, so every lane holds the same
// bits and all simplex control flow is wave-uniform.
// Build with -ffp-contract=off: the reference arithmetic is unfused IEEE double.
#pragma once
#include <hip/hip_runtime.h>

namespace icm {

constexpr double kPi = 3.141592653589793;
constexpr double kTwoPi = 6.283185307179586;
constexpr double kHalfPi = 1.5707963267948966;
constexpr int kWave = 64;

// 1 only in eval_probe.hip (tools/count_eval_flops.py): the rarely taken slow paths (fmod in
// wrap_pi, generic sincos beyond |d| = 0.25) are compiled out so that the probe's ISA is the
// straight-line path an energy evaluation executes with the default isotropic Q.
#ifndef ICM_PROBE_FAST_TRIG_ONLY
#define ICM_PROBE_FAST_TRIG_ONLY 0
#endif

// entrepi (reference scripts/ICM_SLAM_tools.py:455-463): numpy mod (sign of the divisor),
// then fold (pi, 2pi) down.  Result in [-pi, pi].
__device__ __forceinline__ double wrap_pi(double a) {
    // fmod(a, b) == a exactly whenever |a| < b: the usual case, without fmod's division loop
    double r = (ICM_PROBE_FAST_TRIG_ONLY || fabs(a) < kTwoPi) ? a : fmod(a, kTwoPi);
    if (r < 0.0) r += kTwoPi;
    if (r > kPi) r -= kTwoPi;
    return r;
}

__device__ __forceinline__ double bcast_first(double v) {
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// DPP move of a double (two dwords); lanes without a source read 0.  With every row enabled the
// zero comes from bound_ctrl (no destination to preset: saves a v_mov per dword); with masked rows
// (row_bcast into rows 1/3 or 2/3) the disabled rows keep the preset 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo, hi;
    if constexpr (ROW_MASK == 0xF) {
        lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
        hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    } else {
        lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    }
    return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes, returned wave-uniform (same bits in every lane).  In-register DPP
// tree: shifts by 1,2,4,8 inside each row of 16 lanes, then row_bcast:15 / row_bcast:31 across
// the rows; the total lands in lane 63 and is read back through scalar registers.
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_mov<0x111, 0xF>(v);  // row_shr:1
    v += dpp_mov<0x112, 0xF>(v);  // row_shr:2
    v += dpp_mov<0x114, 0xF>(v);  // row_shr:4
    v += dpp_mov<0x118, 0xF>(v);  // row_shr:8
    v += dpp_mov<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_mov<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// Everything one pose solve needs besides its observation items.  Wave-uniform.
struct SolveCtx {
    // prev(x; a): a = x_{t-1} (already updated), Appendix A.4
    double gax, gay, gat;  // g(a, u_{t-1})
    double xax, xay, xat;  // a
    double ca, sa;         // cos / sin of a.theta
    double o1x, o1y, o1t;  // Rot(o_{t-1}.th)(o_t.xy - o_{t-1}.xy), o_t.th - o_{t-1}.th
    // next(x; b): b = x_{t+1} (previous sweep); unused when !two_sided
    int two_sided;
    double xpx, xpy, xpt;
    double v, w;           // u_t
    double o2x, o2y, o2t;
    // weights
    double dt, R0, R1, R2, Q0, Q1, cte;
    // moment-form energy only (make_ctx): centre of the previous-pose odometry term, a_xy +
    // Rot(a)^T o1 (|o1 - Rot(a) d|^2 = |d - Rot(a)^T o1|^2: a rotation keeps the norm), and the
    // products dt v, dt w of the unicycle step
    // wn = 1 (two-sided, fun_xn) or 0 (one-sided, fun_x): weight of the next-pose terms;
    // o2tp = o2t - xpt
    double hx, hy, dtv, dtw, wn, o2tp;
};

// Observation items of one pose.  Two forms of the same energy h(x):
//   beams   (kw == nullptr): one item per kept beam, k = 1, b = the beam's body point;
//   entries (kw != nullptr): one item per (pose, landmark) entry -- k beams that share the
//           target y, b = their mean body point.  With r_j = rbar + R(b_j - bbar),
//           sum_j r_j^T Q r_j = k rbar^T Q rbar + sum_j (R d_j)^T Q (R d_j), and the second
//           term summed over the pose's entries is tr(R^T Q R C) with the pose's pooled
//           within-entry scatter C = (cxx, cxy, cyy): exact algebra, x enters it only
//           through theta and only when Q is anisotropic.
struct Items {
    const double* __restrict__ bx;
    const double* __restrict__ by;
    const double* __restrict__ tx;
    const double* __restrict__ ty;
    const int* __restrict__ kw;
    double cxx, cxy, cyy;
    int n;
    const double2* __restrict__ b2;  // entries form: (mean body x, y) and (target x, y) records
    const double2* __restrict__ t2;
};

// One item held in registers (entries form, n <= 64: lane i owns entry i).
struct RegItem {
    double k, bx, by, tx, ty;
};

__device__ __forceinline__ double scatter_term(const SolveCtx& c, const Items& it, double ct, double st) {
    return ((c.Q0 * ct * ct + c.Q1 * st * st) * it.cxx + 2.0 * (ct * st) * (c.Q1 - c.Q0) * it.cxy) +
           (c.Q0 * st * st + c.Q1 * ct * ct) * it.cyy;
}

// h(x) = sum_i k_i (w_i - y_i)^T Q (w_i - y_i) [+ scatter], w_i = p + Rot(th - pi/2) b_i
// (reference scripts/ICM_ROS.py:171-200 with the body-frame points of filtrar_z's columns
// 2:4; d*cos(ang + th - pi/2) is expanded with the rotation tras_rot_z uses,
// scripts/ICM_SLAM_tools.py:476-479).
__device__ __forceinline__ double obs_energy(const SolveCtx& c, const Items& it, double px,
                                             double py, double th, int lane) {
    const double a = th - kHalfPi;
    double ct, st;
    sincos(a, &st, &ct);
    double acc = 0.0;
    if (it.kw) {  // entries form
        for (int j = lane; j < it.n; j += kWave) {
            const double2 b = it.b2[j], t = it.t2[j];
            const double dx = (px + (b.x * ct - b.y * st)) - t.x;
            const double dy = (py + (b.x * st + b.y * ct)) - t.y;
            acc += ((dx * c.Q0) * dx + (dy * c.Q1) * dy) * (double)it.kw[j];
        }
    } else {      // one term per kept beam
        for (int j = lane; j < it.n; j += kWave) {
            const double bx = it.bx[j], by = it.by[j];
            const double dx = (px + (bx * ct - by * st)) - it.tx[j];
            const double dy = (py + (bx * st + by * ct)) - it.ty[j];
            acc += (dx * c.Q0) * dx + (dy * c.Q1) * dy;
        }
    }
    acc = wave_sum(acc);
    return it.kw ? acc + scatter_term(c, it, ct, st) : acc;
}

// Same energy with the (<= 64) entries of the pose held one per lane in registers.
__device__ __forceinline__ double obs_energy_reg(const SolveCtx& c, const Items& it, const RegItem& r,
                                                 double px, double py, double th) {
    const double a = th - kHalfPi;
    double ct, st;
    sincos(a, &st, &ct);
    const double dx = (px + (r.bx * ct - r.by * st)) - r.tx;
    const double dy = (py + (r.bx * st + r.by * ct)) - r.ty;
    const double e = ((dx * c.Q0) * dx + (dy * c.Q1) * dy) * r.k;
    return wave_sum(e) + scatter_term(c, it, ct, st);
}

// ---------------------------------------------------------------------------------------
// Moment form of h(x): the default in the pose solves.
// Expand every entry's residual around the pose's previous-sweep value (p_o, th_o), the pose
// the beams were projected with in phase A.  With w = Rot(th_o - pi/2) bbar (the entry's mean
// offset in the world frame), r0 = p_o + w - y (its residual there), dp = p - p_o,
// d = th - th_o, alpha = cos d - 1, beta = sin d, J = [[0,-1],[1,0]]:
//     r(x) = p + Rot(th - pi/2) bbar - y = r0 + dp + alpha w + beta J w            (exact)
// so h(x) = sum_e k r^T Q r (+ the scatter term) is a quadratic form in (dp, alpha, beta)
// whose 14 coefficients are sums over the pose's entries.  Every term is O(residual): unlike
// the expansion about the origin nothing cancels, so the form is as accurate as the direct
// sum, but an evaluation costs ~40 flops and no memory access -- which lets ONE LANE solve a
// pose (64 poses per wavefront, no cross-lane reduction).
// ---------------------------------------------------------------------------------------
struct PoseMoments {
    double S, Swx, Swy, Srx, Sry, Swxx, Swyy, Swxy, Swxrx, Swyrx, Swxry, Swyry, Srxx, Sryy;
    double cxx, cxy, cyy;      // pooled within-entry scatter (body frame)
    double sc_iso;             // Q0 (cxx + cyy): the scatter term when Q is isotropic
    double pox, poy, tho, co, so;   // expansion point: p_o, th_o, cos/sin th_o
    // isotropic Q (finish_moments): (Swxrx + Swyry) - (Swxx + Swyy), Swxry - Swyrx, Srxx + Sryy
    double AW, Bq, Rr;
};
constexpr int kMomentCount = 14;

// The moment-form energy is the build's own arithmetic -- an exact regrouping of the reference's
// per-beam sum, not its expression tree -- so it may use fused multiply-adds.  They are written
// out (fma_) instead of left to `fp contract`: every kernel form (lane, quad, sequential,
// fast and generic path) then rounds identically by construction.
// The three-operand VOP3 form is spelled out: left to itself hipcc selects the destructive
// two-operand v_fmac_f64 and then has to copy every loop-invariant addend (polynomial
// coefficient, moment sum) into a scratch register first -- a v_mov_b64 per fma inside the
// Nelder-Mead loop, ~20 % of an evaluation's instructions.  fnma_ = fma(-a, b, c), fmas_ = fma(a, b, -c).
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ double fnma_(double a, double b, double c) { return __builtin_fma(-a, b, c); }
__device__ __forceinline__ double fmas_(double a, double b, double c) { return __builtin_fma(a, b, -c); }

// sin d and cos d - 1 for |d| <= 0.25 by their Taylor polynomials (truncation < 1e-18), as
// explicit fused multiply-adds (our own arithmetic: not part of the reference's expression
// tree that -ffp-contract=off protects).
// The thirteen coefficients: literals by default (LitTrigK: the compiler materialises each one in front of its fma,
// a v_mov_b64 per coefficient per evaluation), or held in vector registers for the length of a Nelder-Mead
// (PinnedTrigK: 26 VGPRs the fold-only solve has to spare; a lone wave pays ~5 cycles for every instruction it
// issues, copies included).  Same numbers, same operations, same order.
struct LitTrigK {
    static constexpr double s6 = -1.0 / 6227020800.0, s5 = 1.0 / 39916800.0, s4 = -1.0 / 362880.0, s3 = 1.0 / 5040.0,
                            s2 = -1.0 / 120.0, s1 = 1.0 / 6.0;
    static constexpr double c7 = 1.0 / 87178291200.0, c6 = -1.0 / 479001600.0, c5 = 1.0 / 3628800.0, c4 = -1.0 / 40320.0,
                            c3 = 1.0 / 720.0, c2 = -1.0 / 24.0, c1 = 0.5;
};
struct PinnedTrigK {
    double s6, s5, s4, s3, s2, s1, c7, c6, c5, c4, c3, c2, c1;
    __device__ __forceinline__ PinnedTrigK()
        : s6(LitTrigK::s6), s5(LitTrigK::s5), s4(LitTrigK::s4), s3(LitTrigK::s3), s2(LitTrigK::s2), s1(LitTrigK::s1),
          c7(LitTrigK::c7), c6(LitTrigK::c6), c5(LitTrigK::c5), c4(LitTrigK::c4), c3(LitTrigK::c3), c2(LitTrigK::c2),
          c1(LitTrigK::c1) {
        // (opaque to the optimiser: a value it cannot re-materialise stays in its register)
        asm volatile("" : "+v"(s6), "+v"(s5), "+v"(s4), "+v"(s3), "+v"(s2), "+v"(s1));
        asm volatile("" : "+v"(c7), "+v"(c6), "+v"(c5), "+v"(c4), "+v"(c3), "+v"(c2), "+v"(c1));
    }
};
template <class K = LitTrigK>
__device__ __forceinline__ void small_sincosm1(double d, double& s, double& cm1, const K& k = K()) {
    const double z = d * d;
    double ps = fma_(z, k.s6, k.s5);  // z^6/13!, z^5/11!
    ps = fma_(z, ps, k.s4);
    ps = fma_(z, ps, k.s3);
    ps = fma_(z, ps, k.s2);
    ps = fma_(z, ps, k.s1);
    s = fnma_(d * z, ps, d);          // d - d^3/6 + ...
    double pc = fma_(z, k.c7, k.c6);  // z^7/14!, z^6/12!
    pc = fma_(z, pc, k.c5);
    pc = fma_(z, pc, k.c4);
    pc = fma_(z, pc, k.c3);
    pc = fma_(z, pc, k.c2);
    pc = fma_(z, pc, k.c1);
    cm1 = -(z * pc);                  // -d^2/2 + d^4/24 - ...
}

// Derived sums of the isotropic form (Q0 == Q1): with X, Y the two quadratic forms below,
//   X + Y = S (dx^2 + dy^2) + Rr + 2 [dx (al Swx - be Swy + Srx) + dy (al Swy + be Swx + Sry) + al AW + be Bq]
// because the al be Swxy terms cancel and al^2 + be^2 = (cos d - 1)^2 + sin^2 d = -2 al exactly.
__device__ __forceinline__ void finish_moments(const SolveCtx& c, PoseMoments& m) {
    m.sc_iso = c.Q0 * (m.cxx + m.cyy);
    m.AW = (m.Swxrx + m.Swyry) - (m.Swxx + m.Swyy);
    m.Bq = m.Swxry - m.Swyrx;
    m.Rr = m.Srxx + m.Sryy;
}

// Observation energy in moment form at (px, py) and heading th_o + d, given
// al = cos d - 1 and be = sin d, and (anisotropic Q only) cos/sin of the heading.
__device__ __forceinline__ double moments_energy(const SolveCtx& c, const PoseMoments& m, double px, double py,
                                                 double al, double be, double cth, double sth) {
    const double dx = px - m.pox, dy = py - m.poy;
    if (ICM_PROBE_FAST_TRIG_ONLY || c.Q0 == c.Q1) {   // (wave-uniform: the weights are kernel arguments; the probe counts this path)
        const double g1 = fma_(al, m.Swx, fnma_(be, m.Swy, m.Srx));
        const double g2 = fma_(al, m.Swy, fma_(be, m.Swx, m.Sry));
        const double cross = fma_(dx, g1, fma_(dy, g2, fma_(al, m.AW, be * m.Bq)));
        const double rr = fma_(dx, dx, dy * dy);
        return fma_(c.Q0, fma_(2.0, cross, fma_(m.S, rr, m.Rr)), m.sc_iso);
    }
    const double X = (((m.S * dx) * dx + (al * al) * m.Swxx) + ((be * be) * m.Swyy + m.Srxx)) +
                     2.0 * ((dx * ((al * m.Swx - be * m.Swy) + m.Srx) + al * (m.Swxrx - be * m.Swxy)) - be * m.Swyrx);
    const double Y = (((m.S * dy) * dy + (al * al) * m.Swyy) + ((be * be) * m.Swxx + m.Sryy)) +
                     2.0 * ((dy * ((al * m.Swy + be * m.Swx) + m.Sry) + al * (m.Swyry + be * m.Swxy)) + be * m.Swxry);
    // scatter term tr(R^T Q R C) with cos(th - pi/2) = sin th, sin(th - pi/2) = -cos th
    const double ct = sth, st = -cth;
    const double sc = ((c.Q0 * ct * ct + c.Q1 * st * st) * m.cxx + 2.0 * (ct * st) * (c.Q1 - c.Q0) * m.cxy) +
                      (c.Q0 * st * st + c.Q1 * ct * ct) * m.cyy;
    return (c.Q0 * X + c.Q1 * Y) + sc;
}

// entrepi for |a| < 2 pi without the fmod: the two conditional shifts as selects.
__device__ __forceinline__ double wrap_small(double a) {
    const double r = a < 0.0 ? a + kTwoPi : a;
    return r > kPi ? r - kTwoPi : r;
}

// fun_xn / fun_x with h in moment form (reference scripts/ICM_ROS.py:220-278, Appendix A.4).
// Trigonometry: the NM iterates stay within a fraction of a radian of the pose's previous heading
// th_o, so sin d, cos d - 1 come from short polynomials and cos th, sin th from the rotation of
// (cos th_o, sin th_o) by d.  The angle residuals keep the reference's operation order (the wrap
// adds and subtracts 2 pi, which is not the identity in floating point); the planar terms are
// regrouped:
//   previous-pose odometry  |o1 - Rot(a)(p - a)|^2 = |p - (a + Rot(a)^T o1)|^2   (c.hx, c.hy)
//   unicycle step           g(x).xy - b.xy = (p - b.xy) + (dt v) (cos th, sin th)
// A one-sided energy (fun_x: last pose, c.wn = 0) multiplies the next-pose terms by zero:
// ((0 + prevR) + hh) + prevO is the reference's (prevR + hh) + prevO bit for bit.
// GENERIC = false: straight-line code, valid while |d| <= 0.25 and every wrapped angle is inside
// (-2 pi, 2 pi); GENERIC = true: the same expressions with the rare cases (generic sincos, fmod)
// selected per lane.  pose_energy_moments() takes the generic road for a whole wavefront when any
// of its lanes needs it, so the common case has no divergent branch at all.
template <bool GENERIC>
__device__ __forceinline__ double pose_energy_moments_t(const SolveCtx& c, const PoseMoments& m, double px, double py,
                                                        double th, double dl, double a0, double a1, double a2, double a3) {
    double cth, sth, al, be;
    if (!GENERIC || fabs(dl) <= 0.25) {
        small_sincosm1(dl, be, al);
        cth = m.co + fmas_(m.co, al, m.so * be);
        sth = m.so + fma_(m.so, al, m.co * be);
    } else {
        sth = sin(th);
        cth = cos(th);
        be = sth * m.co - cth * m.so;                 // sin(th - th_o)
        const double cd = cth * m.co + sth * m.so;    // cos(th - th_o)
        const double den = 1.0 + cd;
        al = den > 1e-3 ? -(be * be) / den : cd - 1.0;  // cos d - 1 without cancellation
    }
    const double hh = moments_energy(c, m, px, py, al, be, cth, sth);
    const double r0 = px - c.gax, r1 = py - c.gay, r2 = GENERIC ? wrap_pi(a0) : wrap_small(a0);
    const double prevR = fma_(r0 * c.R0, r0, fma_(r1 * c.R1, r1, (r2 * c.R2) * r2));
    const double q0 = px - c.hx, q1 = py - c.hy, q2 = GENERIC ? wrap_pi(a1) : wrap_small(a1);
    const double prevO = c.cte * fma_(q0, q0, fma_(q1, q1, q2 * q2));
    const double s0 = fma_(c.dtv, cth, px - c.xpx), s1 = fma_(c.dtv, sth, py - c.xpy), s2 = GENERIC ? wrap_pi(a2) : wrap_small(a2);
    const double nextR = fma_(s0 * c.R0, s0, fma_(s1 * c.R1, s1, (s2 * c.R2) * s2));
    const double ex = c.xpx - px, ey = c.xpy - py;
    const double p0 = c.o2x - fma_(cth, ex, sth * ey);
    const double p1 = c.o2y - fmas_(cth, ey, sth * ex);
    const double p2 = GENERIC ? wrap_pi(a3) : wrap_small(a3);
    const double nextO = c.cte * fma_(p0, p0, fma_(p1, p1, p2 * p2));
    return ((c.wn * (nextR + nextO) + prevR) + hh) + prevO;
}

__device__ __noinline__ double pose_energy_moments_generic(const SolveCtx& c, const PoseMoments& m, double px, double py,
                                                           double th, double dl, double a0, double a1, double a2, double a3) {
    return pose_energy_moments_t<true>(c, m, px, py, th, dl, a0, a1, a2, a3);
}

// ---------------------------------------------------------------------------------------
// Folded form: what a Nelder-Mead evaluation actually executes.
// With isotropic weights (Q0 == Q1, R0 == R1) and no angle wrap in play, EVERY term of
// fun_xn / fun_x is a polynomial of degree <= 2 in the planar step d = p - p_o about the pose's
// previous-sweep value, whose coefficients are linear in al = cos dl - 1, be = sin dl (dl = th - th_o),
// plus a quadratic in dl from the four angle residuals:
//   moment-form h          Q [S |d|^2 + 2 d.(g0 + al Sw + be J Sw) + 2 (al AW + be Bq) + Rr] + sc
//   previous pose, model   Rp |d + u|^2 + R2 (dl + k0)^2                      u = p_o - g(a)
//   previous pose, odom.   cte (|d + v|^2 + (k1 - dl)^2)                      v = p_o - (a + Rot(a)^T o1)
//   next pose, model       Rp |d + e0 + dtv (al c_o + be J c_o)|^2 + R2 (dl + k2)^2,   e0 = (p_o - b) + dtv c_o,
//                          c_o = (cos th_o, sin th_o); |al c_o + be J c_o|^2 = al^2 + be^2 = -2 al exactly
//   next pose, odometry    cte (|o2 + Rot(th)(d + w)|^2 + (dl + k3)^2) = cte (|d + w|^2 + |o2|^2 + 2 (d + w).q(th) + ...),
//                          w = p_o - b, q(th) = Rot(th)^T o2 = q0 (1 + al) + be J q0, q0 = Rot(th_o)^T o2
// so that
//   E = A |d|^2 + dx (B0 + B1 al + B2 be) + dy (C0 - B2 al + B1 be) + D1 al + D2 be + dl (V2 + W dl) + D0
// with thirteen per-pose numbers (make_fold, once per solve): ~35 vector instructions per evaluation
// instead of ~145 for the term-by-term form, and a third of its registers.  Every folded coefficient is a
// sum of O(residual) or O(residual^2) quantities (the expansion point is the pose phase A projected with),
// so nothing cancels; the folded value differs from the term-by-term one by a few ulps of E.
// Validity is decided PER LANE from the lane's own data (never from its wave mates): |dl| <= dlim, where
// dlim = min(0.25 (polynomial sin / cos), 3 - max |k_i| (no residual angle reaches +-pi)), dlim < 0 for
// anisotropic weights.  A lane outside it takes the term-by-term form (pose_energy_moments_generic: generic
// sincos, entrepi with its fmod); the wave executes that call only when one of its lanes needs it.
// For |a| < pi entrepi(a) is a up to the rounding of its +2 pi / -2 pi round trip (<= 4.5e-16 absolute), far
// below one ulp of E once squared and weighted.
// ---------------------------------------------------------------------------------------
#ifdef ICM_WAVE_TS
__device__ unsigned long long g_eval_stats[4];   // wave evaluations, of which generic; lane evaluations, of which generic
#endif
struct PoseFold {
    double pox, poy, tho;
    double A, B0, B1, B2, C0, D0, D1, D2, W, V2;
    double dlim;
};

__device__ __forceinline__ void make_fold(const SolveCtx& c, const PoseMoments& m, PoseFold& f) {
    f.pox = m.pox; f.poy = m.poy; f.tho = m.tho;
    const double Q = c.Q0, Rp = c.R0, wn = c.wn, co = m.co, so = m.so;
    const double ux = m.pox - c.gax, uy = m.poy - c.gay;
    const double vx = m.pox - c.hx, vy = m.poy - c.hy;
    const double wx = m.pox - c.xpx, wy = m.poy - c.xpy;        // (times wn = 0 for a one-sided energy)
    const double ex0 = wx + c.dtv * co, ey0 = wy + c.dtv * so;
    const double qx0 = c.o2x * co - c.o2y * so, qy0 = c.o2x * so + c.o2y * co;
    const double zx = wx + qx0, zy = wy + qy0;
    const double nR = wn * Rp, nO = wn * c.cte;
    f.A = (Q * m.S + Rp) + (c.cte + (nR + nO));
    f.B0 = 2.0 * (((Q * m.Srx + Rp * ux) + c.cte * vx) + (nR * ex0 + nO * zx));
    f.C0 = 2.0 * (((Q * m.Sry + Rp * uy) + c.cte * vy) + (nR * ey0 + nO * zy));
    f.B1 = 2.0 * (Q * m.Swx + (nR * (c.dtv * co) + nO * qx0));
    f.B2 = -2.0 * (Q * m.Swy + (nR * (c.dtv * so) + nO * qy0));
    f.D1 = 2.0 * ((Q * m.AW + nR * (c.dtv * ((ex0 * co + ey0 * so) - c.dtv))) + nO * (wx * qx0 + wy * qy0));
    f.D2 = 2.0 * ((Q * m.Bq + nR * (c.dtv * (ey0 * co - ex0 * so))) + nO * (wy * qx0 - wx * qy0));
    const double k0 = m.tho - c.gat, k1 = (c.o1t - m.tho) + c.xat, k2 = (m.tho + c.dtw) - c.xpt, k3 = c.o2tp + m.tho;
    const double nA = wn * c.R2;
    f.W = (c.R2 + c.cte) + (nA + nO);
    f.V2 = 2.0 * ((c.R2 * k0 - c.cte * k1) + (nA * k2 + nO * k3));
    f.D0 = (((Q * m.Rr + m.sc_iso) + Rp * (ux * ux + uy * uy)) + c.cte * (vx * vx + vy * vy)) +
           ((nR * (ex0 * ex0 + ey0 * ey0) + nO * (zx * zx + zy * zy)) +
            ((c.R2 * (k0 * k0) + c.cte * (k1 * k1)) + (nA * (k2 * k2) + nO * (k3 * k3))));
    double kmax = fmax(fabs(k0), fabs(k1));
    if (wn != 0.0) kmax = fmax(kmax, fmax(fabs(k2), fabs(k3)));
    const bool iso = c.Q0 == c.Q1 && c.R0 == c.R1;
    f.dlim = (iso && kmax < 3.0) ? fmin(0.25, 3.0 - kmax) : -1.0;   // (NaN anywhere -> -1: the term-by-term form)
}

template <class K = LitTrigK>
__device__ __forceinline__ double pose_energy_folded(const PoseFold& f, double px, double py, double dl, const K& k = K()) {
    double al, be;
    small_sincosm1(dl, be, al, k);
    const double dx = px - f.pox, dy = py - f.poy;
    const double l1 = fma_(f.B1, al, fma_(f.B2, be, f.B0));
    const double l2 = fma_(f.B1, be, fnma_(f.B2, al, f.C0));
    const double l3 = fma_(f.D1, al, fma_(f.D2, be, fma_(dl, fma_(f.W, dl, f.V2), f.D0)));
    const double rr = fma_(dx, dx, dy * dy);
    return fma_(f.A, rr, fma_(dx, l1, fma_(dy, l2, l3)));
}

// The folded form alone (k_solve_m_fused<., true>): the value is used whatever it is worth, `ok` says whether this
// lane's heading step stayed inside the form's validity range.  A pose with one evaluation outside it is not
// stored by that kernel but marked, and the fix-up launch behind it solves it with pose_energy_moments() below
// (folded where valid, term by term elsewhere): the main kernel then keeps neither the context nor the moment
// sums alive across the Nelder-Mead loop -- thirteen coefficients, no scratch.
template <class K = LitTrigK>
__device__ __forceinline__ double pose_energy_fold_only(const PoseFold& f, double px, double py, double th, bool& ok,
                                                        const K& k = K()) {
    const double dl = th - f.tho;
    ok = fabs(dl) <= f.dlim;   // (NaN -> false)
    return pose_energy_folded(f, px, py, dl, k);
}

__device__ __forceinline__ double pose_energy_moments(const SolveCtx& c, const PoseMoments& m, const PoseFold& f, double px,
                                                      double py, double th) {
    const double dl = th - f.tho;
    double e = pose_energy_folded(f, px, py, dl);
    if (ICM_PROBE_FAST_TRIG_ONLY) return e;
    const bool folded = fabs(dl) <= f.dlim;
#ifdef ICM_WAVE_TS
    {
        const unsigned long long act = __ballot(true), gen = __ballot(!folded);
        if ((unsigned)__builtin_ctzll(act) == (threadIdx.x & 63)) {
            atomicAdd(&g_eval_stats[0], 1ull);
            atomicAdd(&g_eval_stats[1], (unsigned long long)(gen != 0));
            atomicAdd(&g_eval_stats[2], (unsigned long long)__builtin_popcountll(act));
            atomicAdd(&g_eval_stats[3], (unsigned long long)__builtin_popcountll(gen));
        }
    }
#endif
    if (__builtin_expect(__ballot(!folded) != 0ull, 0)) {
        const double a0 = th - c.gat;              // prev: model residual angle
        const double a1 = (c.o1t - th) + c.xat;    // prev: odometry residual angle
        const double a2 = (th + c.dtw) - c.xpt;    // next: model residual angle
        const double a3 = c.o2tp + th;             // next: odometry residual angle, (o2t - xpt) + th
        const double g = pose_energy_moments_generic(c, m, px, py, th, dl, a0, a1, a2, a3);
        e = folded ? e : g;
    }
    return e;
}

// fun_xn (two_sided) / fun_x (reference scripts/ICM_ROS.py:220-278), Appendix A.4, given
// the observation energy hh = h(x).
__device__ __forceinline__ double pose_energy_with(const SolveCtx& c, double hh, double px, double py, double th) {
    // prev(x; a)
    const double r0 = px - c.gax, r1 = py - c.gay, r2 = wrap_pi(th - c.gat);
    const double prevR = ((r0 * c.R0) * r0 + (r1 * c.R1) * r1) + (r2 * c.R2) * r2;
    const double dax = px - c.xax, day = py - c.xay;
    const double q0 = c.o1x - (c.ca * dax + c.sa * day);
    const double q1 = c.o1y - (-c.sa * dax + c.ca * day);
    const double q2 = wrap_pi((c.o1t - th) + c.xat);
    const double prevO = c.cte * ((q0 * q0 + q1 * q1) + q2 * q2);
    if (!c.two_sided) return (prevR + hh) + prevO;
    // next(x; b)
    double cth, sth;
    sincos(th, &sth, &cth);
    const double gx = px + c.dt * (cth * c.v), gy = py + c.dt * (sth * c.v), gt = th + c.dt * c.w;
    const double s0 = gx - c.xpx, s1 = gy - c.xpy, s2 = wrap_pi(gt - c.xpt);
    const double nextR = ((s0 * c.R0) * s0 + (s1 * c.R1) * s1) + (s2 * c.R2) * s2;
    const double ex = c.xpx - px, ey = c.xpy - py;
    const double p0 = c.o2x - (cth * ex + sth * ey);
    const double p1 = c.o2y - (-sth * ex + cth * ey);
    const double p2 = wrap_pi((c.o2t - c.xpt) + th);
    const double nextO = c.cte * ((p0 * p0 + p1 * p1) + p2 * p2);
    return (((nextR + nextO) + prevR) + hh) + prevO;
}

__device__ __forceinline__ double pose_energy(const SolveCtx& c, const Items& it, double px,
                                              double py, double th, int lane) {
    return pose_energy_with(c, obs_energy(c, it, px, py, th, lane), px, py, th);
}

// Fill the x-independent parts of the context for pose t.
//   xa = x[:,t-1], xp = x[:,t+1] (ignored if !two_sided), ua = u[:,t-1], ut = u[:,t],
//   oa/ot/op = odometria[:,t-1], [:,t], [:,t+1]
//   ca, sa = cos / sin of xa[2]; coa, soa of oa[2]; cot, sot of ot[2] (two-sided only): the caller has them -- from the
//   tables whoever wrote the value keeps beside it, or computed on the spot
__device__ __forceinline__ void make_ctx_t(SolveCtx& c, int two_sided, const double xa[3],
                                           const double xp[3], const double ua[2],
                                           const double ut[2], const double oa[3],
                                           const double ot[3], const double op[3], double ca, double sa,
                                           double coa, double soa, double cot, double sot) {
    c.two_sided = two_sided;
    c.xax = xa[0]; c.xay = xa[1]; c.xat = xa[2];
    c.ca = ca; c.sa = sa;
    // g(a, u_{t-1}) (reference scripts/ICM_ROS.py:202-207)
    c.gax = xa[0] + c.dt * (c.ca * ua[0]);
    c.gay = xa[1] + c.dt * (c.sa * ua[0]);
    c.gat = xa[2] + c.dt * ua[1];
    const double d1x = ot[0] - oa[0], d1y = ot[1] - oa[1];
    c.o1x = coa * d1x + soa * d1y;
    c.o1y = -soa * d1x + coa * d1y;
    c.o1t = ot[2] - oa[2];
    c.hx = xa[0] + (c.ca * c.o1x - c.sa * c.o1y);
    c.hy = xa[1] + (c.sa * c.o1x + c.ca * c.o1y);
    c.dtv = c.dtw = c.o2tp = 0.0;
    c.wn = two_sided ? 1.0 : 0.0;
    if (two_sided) {
        c.xpx = xp[0]; c.xpy = xp[1]; c.xpt = xp[2];
        c.v = ut[0]; c.w = ut[1];
        c.dtv = c.dt * ut[0]; c.dtw = c.dt * ut[1];
        const double d2x = op[0] - ot[0], d2y = op[1] - ot[1];
        c.o2x = cot * d2x + sot * d2y;
        c.o2y = -sot * d2x + cot * d2y;
        c.o2t = op[2] - ot[2];
        c.o2tp = c.o2t - c.xpt;
    } else {
        c.xpx = c.xpy = c.xpt = c.v = c.w = c.o2x = c.o2y = c.o2t = 0.0;
    }
}

__device__ __forceinline__ void make_ctx(SolveCtx& c, int two_sided, const double xa[3],
                                         const double xp[3], const double ua[2],
                                         const double ut[2], const double oa[3],
                                         const double ot[3], const double op[3]) {
    const double cot = two_sided ? cos(ot[2]) : 1.0, sot = two_sided ? sin(ot[2]) : 0.0;
    make_ctx_t(c, two_sided, xa, xp, ua, ut, oa, ot, op, cos(xa[2]), sin(xa[2]), cos(oa[2]), sin(oa[2]), cot, sot);
}

struct Vtx {
    double x, y, t, f;
};

// x / 3.0 correctly rounded without the division sequence (v_div_scale, v_rcp, Newton steps,
// v_div_fmas, v_div_fixup): with z = RN(1/3), q = RN(x z) is within an ulp of x/3, the residual
// r = x - 3 q is exact in one fma, and RN(q + r z) is the correctly rounded quotient (Markstein's
// theorem; checked against x / 3.0 on 4e8 random doubles) -- bit-identical to numpy's
// `np.add.reduce(sim[:-1], 0) / N` for every finite x that does not underflow.
__device__ __forceinline__ double div3(double x) {
    const double z = 1.0 / 3.0;
    const double q = x * z;
    const double r = fnma_(q, 3.0, x);
    return fma_(r, z, q);
}

__device__ __forceinline__ void cswap(Vtx& a, Vtx& b) {
    if (b.f < a.f) {
        Vtx tmp = a;
        a = b;
        b = tmp;
    }
}
// Stable ascending order of 4 vertices (= numpy argsort's insertion sort for n <= 16).
__device__ __forceinline__ void sort4(Vtx& v0, Vtx& v1, Vtx& v2, Vtx& v3) {
    cswap(v0, v1);
    cswap(v1, v2);
    cswap(v0, v1);
    cswap(v2, v3);
    cswap(v1, v2);
    cswap(v0, v1);
}

// Sorted insertion of one vertex into the simplex (v0 <= v1 <= v2 by f; v3, the worst, is dropped) for the lanes of
// `ins`; the vertex is t for the lanes of `use_t`, else r.  Spelled out as whole-vertex moves under lane masks -- one
// v_mov_b64 per coordinate, exec set per move -- because the compiler's forms of it (three conditional swaps behind two
// selects, or one `if` per move, each with its own branch) cost half as much again: 32 moves, 3 compares and 10 scalar
// instructions here against 16 selects, 41 moves and the swaps' exec bookkeeping.  Hazards (gfx950 = gfx9 rules): a
// VALU-written SGPR read by the SALU and an SALU-written exec used by the VALU are interlocked by the hardware.
//   g_k = ins & (n.f < v_k.f)   (g0 => g1 => g2 because the simplex is sorted)
//   g2: v3 <- v2;  g1: v2 <- v1;  g0: v1 <- v0, v0 <- n;  g1 & !g0: v1 <- n;  g2 & !g1: v2 <- n;  ins & !g2: v3 <- n
__device__ __forceinline__ void insert_vertex(Vtx& v0, Vtx& v1, Vtx& v2, Vtx& v3, Vtx r, const Vtx& t, bool use_t, bool ins) {
    const unsigned long long m_t = __builtin_amdgcn_ballot_w64(use_t), m_ins = __builtin_amdgcn_ballot_w64(ins);
    unsigned long long sv, m2, m1, m0;
    asm("s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[mt]\n\t"
        "v_mov_b64 %[rx], %[tx]\n\t"
        "v_mov_b64 %[ry], %[ty]\n\t"
        "v_mov_b64 %[rt], %[tt]\n\t"
        "v_mov_b64 %[rf], %[tf]\n\t"
        "s_mov_b64 exec, %[mi]\n\t"
        "v_cmp_lt_f64 %[m2], %[rf], %[f2]\n\t"
        "v_cmp_lt_f64 %[m1], %[rf], %[f1]\n\t"
        "v_cmp_lt_f64 %[m0], %[rf], %[f0]\n\t"
        "s_mov_b64 exec, %[m2]\n\t"
        "v_mov_b64 %[x3], %[x2]\n\t"
        "v_mov_b64 %[y3], %[y2]\n\t"
        "v_mov_b64 %[t3], %[t2]\n\t"
        "v_mov_b64 %[f3], %[f2]\n\t"
        "s_mov_b64 exec, %[m1]\n\t"
        "v_mov_b64 %[x2], %[x1]\n\t"
        "v_mov_b64 %[y2], %[y1]\n\t"
        "v_mov_b64 %[t2], %[t1]\n\t"
        "v_mov_b64 %[f2], %[f1]\n\t"
        "s_mov_b64 exec, %[m0]\n\t"
        "v_mov_b64 %[x1], %[x0]\n\t"
        "v_mov_b64 %[y1], %[y0]\n\t"
        "v_mov_b64 %[t1], %[t0]\n\t"
        "v_mov_b64 %[f1], %[f0]\n\t"
        "v_mov_b64 %[x0], %[rx]\n\t"
        "v_mov_b64 %[y0], %[ry]\n\t"
        "v_mov_b64 %[t0], %[rt]\n\t"
        "v_mov_b64 %[f0], %[rf]\n\t"
        "s_andn2_b64 exec, %[m1], %[m0]\n\t"
        "v_mov_b64 %[x1], %[rx]\n\t"
        "v_mov_b64 %[y1], %[ry]\n\t"
        "v_mov_b64 %[t1], %[rt]\n\t"
        "v_mov_b64 %[f1], %[rf]\n\t"
        "s_andn2_b64 exec, %[m2], %[m1]\n\t"
        "v_mov_b64 %[x2], %[rx]\n\t"
        "v_mov_b64 %[y2], %[ry]\n\t"
        "v_mov_b64 %[t2], %[rt]\n\t"
        "v_mov_b64 %[f2], %[rf]\n\t"
        "s_andn2_b64 exec, %[mi], %[m2]\n\t"
        "v_mov_b64 %[x3], %[rx]\n\t"
        "v_mov_b64 %[y3], %[ry]\n\t"
        "v_mov_b64 %[t3], %[rt]\n\t"
        "v_mov_b64 %[f3], %[rf]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [sv] "=&s"(sv), [m2] "=&s"(m2), [m1] "=&s"(m1), [m0] "=&s"(m0),
          [rx] "+v"(r.x), [ry] "+v"(r.y), [rt] "+v"(r.t), [rf] "+v"(r.f),
          [x0] "+v"(v0.x), [y0] "+v"(v0.y), [t0] "+v"(v0.t), [f0] "+v"(v0.f),
          [x1] "+v"(v1.x), [y1] "+v"(v1.y), [t1] "+v"(v1.t), [f1] "+v"(v1.f),
          [x2] "+v"(v2.x), [y2] "+v"(v2.y), [t2] "+v"(v2.t), [f2] "+v"(v2.f),
          [x3] "+v"(v3.x), [y3] "+v"(v3.y), [t3] "+v"(v3.t), [f3] "+v"(v3.f)
        : [tx] "v"(t.x), [ty] "v"(t.y), [tt] "v"(t.t), [tf] "v"(t.f), [mt] "s"(m_t), [mi] "s"(m_ins));
}

__device__ __forceinline__ double amax3(const Vtx& a, const Vtx& b) {
    return fmax(fmax(fabs(a.x - b.x), fabs(a.y - b.y)), fabs(a.t - b.t));
}

// scipy.optimize.fmin(f, x0, xtol=1e-3, disp=0) for N = 3:
// rho=1 chi=2 psi=sigma=0.5, xatol=1e-3 AND fatol=1e-4, maxiter=maxfun=600, initial simplex
// x0 with one coordinate *1.05 (0.00025 if it is exactly 0), one stable sort per iteration.
// A function call beyond maxfun aborts the iteration like SciPy's _MaxFuncCallError.
//
// Shaped for SIMT: lanes that solve different poses stay in lockstep per ITERATION -- one
// shared call site for the reflection, one for the second point of the iteration (expansion
// or either contraction; lanes that accepted the reflection idle through it), a rarely
// entered shrink loop, and one for the four initial vertices.  Four inlined copies of the
// energy in total, and no per-evaluation state dispatch.
// out = {x, y, theta, f, nit, nfev}.
// `stop` (optional): a per-lane predicate looked at once in front of the loop and once at the end of every iteration,
// beside SciPy's own termination test: "did anything since the last look disqualify this solve?".  A lane for which it
// holds leaves the loop and the function returns true for it (its result is discarded by the caller: fold-only
// solves).  The predicate is asked about the time since the last look so that nothing of it is carried around the
// loop (a carried flag is a byte in a vector register and five instructions an iteration).  (The second point of an
// iteration is computed by every lane and used by some: a lane may stop on a point it would have discarded.  Masking
// the predicate with `can2` was tried in round 4: it makes the flag a carried byte again, +7 instructions an iteration
// for every pose, to spare the rare pose outside the folded range a second solve.)
struct NeverStop {
    __device__ __forceinline__ bool operator()() const { return false; }
};
template <class F, class S = NeverStop>
__device__ __forceinline__ bool nelder_mead3(F f, double sx, double sy, double st, double out[6], S stop = S()) {
    constexpr int maxfun = 600, maxiter = 600;
    const double xatol = 1e-3, fatol = 1e-4;
    const double grow = 1 + 0.05;
    Vtx v0{sx, sy, st, 0.0};
    Vtx v1{sx != 0.0 ? grow * sx : 0.00025, sy, st, 0.0};
    Vtx v2{sx, sy != 0.0 ? grow * sy : 0.00025, st, 0.0};
    Vtx v3{sx, sy, st != 0.0 ? grow * st : 0.00025, 0.0};
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
        Vtx c = v0;
        if (i == 1) c = v1;
        if (i == 2) c = v2;
        if (i == 3) c = v3;
        const double fv = f(c.x, c.y, c.t);
        if (i == 0) v0.f = fv;
        if (i == 1) v1.f = fv;
        if (i == 2) v2.f = fv;
        if (i == 3) v3.f = fv;
    }
    int nfev = 4, it = 1;
    sort4(v0, v1, v2, v3);
    // The iteration is written as straight-line code with selects: every lane computes the reflection AND a
    // second point (lanes that accept the reflection as it is compute a point they discard), decisions are
    // predicates, and only the rare events -- a shrink, the evaluation budget running out -- are branches.
    // A wavefront executes both evaluations of an iteration anyway as soon as one of its 64 poses needs the
    // second one, so the predicated form costs nothing extra and drops the exec-mask bookkeeping of nested
    // branches (a third of the instructions of an iteration).  Same arithmetic, same decisions:
    //   2 xbar - x_w           = fma(2, xbar, -x_w)        (2 xbar is exact)
    //   ca xbar + cb x_w       = fma(cb, x_w, ca xbar)     (cb = -2, -0.5, 0.5: cb x_w is exact)
    // ONE loop condition, formed at the bottom of the iteration (and once in front of the loop) from everything that
    // ends the solve -- SciPy's termination test, the budgets, an aborted iteration, `stop` -- as lane-mask algebra:
    // the loop costs one exec update per iteration instead of a nest of break blocks (a lone wave pays ~5 cycles for
    // every scalar instruction too, `tools/ubench_issue.hip`).
    auto settled = [&]() {
        const double dx = fmax(fmax(amax3(v1, v0), amax3(v2, v0)), amax3(v3, v0));
        const double df = fmax(fmax(fabs(v0.f - v1.f), fabs(v0.f - v2.f)), fabs(v0.f - v3.f));
        return (dx <= xatol) & (df <= fatol);
    };
    bool stopped = stop();
    bool go = !(settled() | stopped);   // (nfev = 4 < maxfun, it = 1 < maxiter)
    while (go) {
        const double bx = div3((v0.x + v1.x) + v2.x);
        const double by = div3((v0.y + v1.y) + v2.y);
        const double bt = div3((v0.t + v1.t) + v2.t);
        Vtx r{fmas_(2.0, bx, v3.x), fmas_(2.0, by, v3.y), fmas_(2.0, bt, v3.t), 0.0};
        r.f = f(r.x, r.y, r.t);
        ++nfev;
        // second point of the iteration: expansion (fr < f0), outside (f2 <= fr < f3) or inside (fr >= f3)
        // contraction; none when f0 <= fr < f2
        // (bitwise operators on the predicates: lane-mask algebra, no short-circuit branches)
        const bool lt0 = r.f < v0.f, lt2 = r.f < v2.f, lt3 = r.f < v3.f;
        const bool need2 = lt0 | !lt2;
        const bool can2 = need2 & (nfev < maxfun);      // (a call beyond maxfun aborts the iteration, SciPy's _MaxFuncCallError)
        // ca = 3, 1.5, 0.5 and cb = -2, -0.5, 0.5: only the high words differ
        const int ca_hi = lt0 ? 0x40080000 : (lt3 ? 0x3FF80000 : 0x3FE00000);
        const int cb_hi = lt0 ? (int)0xC0000000 : (lt3 ? (int)0xBFE00000 : 0x3FE00000);
        const double ca = __hiloint2double(ca_hi, 0), cb = __hiloint2double(cb_hi, 0);
        Vtx t{fma_(cb, v3.x, ca * bx), fma_(cb, v3.y, ca * by), fma_(cb, v3.t, ca * bt), 0.0};
        t.f = f(t.x, t.y, t.t);
        nfev += can2 ? 1 : 0;
        const bool t_lt_r = t.f < r.f, t_le_r = t.f <= r.f, t_lt_w = t.f < v3.f;
        const bool take_t = (lt0 & t_lt_r) | (!lt0 & lt3 & t_le_r) | (!lt3 & t_lt_w);   // (!lt3 implies !lt0: f0 <= f3)
        const bool shrink = can2 & !lt0 & !take_t;
        const bool aborted0 = need2 & !can2;
        const bool use_t = can2 & take_t;
        const bool keep = shrink | aborted0;
        // The vertex that enters (the second point where it was taken, else the reflection) goes straight to its place:
        // numpy's insertion argsort moves it left past the strictly larger ones, i.e. the vertices above it shift up by
        // one (`keep`: a shrink or an aborted iteration leaves the simplex to the block below).
        insert_vertex(v0, v1, v2, v3, r, t, use_t, !keep);
        bool aborted = aborted0;
        if (__builtin_expect(shrink, 0)) {
#pragma unroll 1
            for (int j = 1; j < 4; ++j) {
                // sim[j] = sim[0] + sigma (sim[j] - sim[0]) is stored before the call that may abort
                Vtx c = j == 1 ? v1 : (j == 2 ? v2 : v3);
                c.x = v0.x + 0.5 * (c.x - v0.x); c.y = v0.y + 0.5 * (c.y - v0.y); c.t = v0.t + 0.5 * (c.t - v0.t);
                const bool go = nfev < maxfun;
                if (go) {
                    c.f = f(c.x, c.y, c.t);
                    ++nfev;
                }
                if (j == 1) v1 = c; else if (j == 2) v2 = c; else v3 = c;
                if (!go) { aborted = true; break; }
            }
            sort4(v0, v1, v2, v3);
        }
        it += aborted ? 0 : 1;
        stopped = stop();
        // (it < maxiter needs no test of its own: every iteration costs at least one evaluation, so nfev >= it + 3, and
        // the two budgets are the same number)
        static_assert(maxfun <= maxiter + 3, "the evaluation budget implies the iteration budget");
        go = !aborted & (nfev < maxfun) & !(settled() | stopped);
    }
    out[0] = v0.x; out[1] = v0.y; out[2] = v0.t; out[3] = v0.f;
    out[4] = (double)it; out[5] = (double)nfev;
    return stopped;
}

// ---------------------------------------------------------------------------------------
// Latency form of the same Nelder-Mead: FOUR lanes (one DPP quad) per pose.  The reflection,
// expansion, outside and inside contraction points of an iteration depend only on the centroid
// and the worst vertex, so the quad evaluates all four at once (lane r evaluates point r) and
// then takes SciPy's decision from the four values; the three shrink vertices and the four
// initial vertices are evaluated in parallel the same way.  An iteration then costs ONE energy
// evaluation of latency instead of up to two (shrink: one instead of three), at four times the
// arithmetic -- the right trade when there are fewer poses than lanes to fill (small shards,
// the sequential schedule).  Same arithmetic per point, same decisions, same nfev accounting
// (only logically evaluated points are counted, maxfun aborts as in the scalar form), so the
// result is bit-identical to nelder_mead3.
// ---------------------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ double quad_bcast(double v) {  // value of lane K of the quad, in all four lanes
    constexpr int ctrl = K | (K << 2) | (K << 4) | (K << 6);  // quad_perm:[K,K,K,K]
    int sl = __double2loint(v), sh = __double2hiint(v);
    asm("" : "+v"(sl), "+v"(sh));   // (a DPP source is a vector register, also where the value happens to be wave-uniform: k_init_pass)
    const int lo = __builtin_amdgcn_update_dpp(0, sl, ctrl, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, sh, ctrl, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// true in all four lanes of a DPP quad when it holds in one of them
__device__ __forceinline__ bool quad_any(bool b) {
    int v = b ? 1 : 0;
    asm("" : "+v"(v));
    v |= __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);   // quad_perm:[1,0,3,2]
    v |= __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);   // quad_perm:[2,3,0,1]
    return v != 0;
}

template <class F, class S = NeverStop>
__device__ __forceinline__ bool nelder_mead3_quad(F f, double sx, double sy, double st, int role, double out[6], S stop = S()) {
    // Written like nelder_mead3: one straight-line iteration with predicates, ONE evaluation call site in the loop (lane
    // `role` of the quad evaluates point `role`: reflection, expansion, outside, inside contraction), the shrink a rarely
    // entered block, one loop condition at the bottom.  `stop` must be quad-uniform (the caller folds it over the quad).
    constexpr int maxfun = 600, maxiter = 600;
    const double xatol = 1e-3, fatol = 1e-4;
    const double grow = 1 + 0.05;
    Vtx v0{sx, sy, st, 0.0};
    Vtx v1{sx != 0.0 ? grow * sx : 0.00025, sy, st, 0.0};
    Vtx v2{sx, sy != 0.0 ? grow * sy : 0.00025, st, 0.0};
    Vtx v3{sx, sy, st != 0.0 ? grow * st : 0.00025, 0.0};
    {
        Vtx c = v0;
        if (role == 1) c = v1;
        if (role == 2) c = v2;
        if (role == 3) c = v3;
        const double fv = f(c.x, c.y, c.t);
        v0.f = quad_bcast<0>(fv);
        v1.f = quad_bcast<1>(fv);
        v2.f = quad_bcast<2>(fv);
        v3.f = quad_bcast<3>(fv);
    }
    int nfev = 4, it = 1;
    sort4(v0, v1, v2, v3);
    // point `role` of an iteration = ca xbar + cb sim[-1]: (2, -1), (3, -2), (1.5, -0.5), (0.5, 0.5); cb sim[-1] is exact,
    // so fma(cb, x_w, ca xbar) rounds like the scalar form's reflection 2 xbar - x_w and its second point
    const double ca = role == 0 ? 2.0 : (role == 1 ? 3.0 : (role == 2 ? 1.5 : 0.5));
    const double cb = role == 0 ? -1.0 : (role == 1 ? -2.0 : (role == 2 ? -0.5 : 0.5));
    auto settled = [&]() {
        const double dx = fmax(fmax(amax3(v1, v0), amax3(v2, v0)), amax3(v3, v0));
        const double df = fmax(fmax(fabs(v0.f - v1.f), fabs(v0.f - v2.f)), fabs(v0.f - v3.f));
        return (dx <= xatol) & (df <= fatol);
    };
    bool stopped = stop();
    bool go = !(settled() | stopped);
    while (go) {
        const double bx = div3((v0.x + v1.x) + v2.x);
        const double by = div3((v0.y + v1.y) + v2.y);
        const double bt = div3((v0.t + v1.t) + v2.t);
        const double mx = fma_(cb, v3.x, ca * bx), my = fma_(cb, v3.y, ca * by), mt = fma_(cb, v3.t, ca * bt);
        const double fm = f(mx, my, mt);
        Vtx r{quad_bcast<0>(mx), quad_bcast<0>(my), quad_bcast<0>(mt), quad_bcast<0>(fm)};
        ++nfev;
        const bool lt0 = r.f < v0.f, lt2 = r.f < v2.f, lt3 = r.f < v3.f;
        const bool need2 = lt0 | !lt2;
        const bool can2 = need2 & (nfev < maxfun);
        // the second point: expansion (lane 1), outside (lane 2) or inside (lane 3) contraction
        const double e1x = quad_bcast<1>(mx), e1y = quad_bcast<1>(my), e1t = quad_bcast<1>(mt), e1f = quad_bcast<1>(fm);
        const double e2x = quad_bcast<2>(mx), e2y = quad_bcast<2>(my), e2t = quad_bcast<2>(mt), e2f = quad_bcast<2>(fm);
        const double e3x = quad_bcast<3>(mx), e3y = quad_bcast<3>(my), e3t = quad_bcast<3>(mt), e3f = quad_bcast<3>(fm);
        Vtx t{lt0 ? e1x : (lt3 ? e2x : e3x), lt0 ? e1y : (lt3 ? e2y : e3y), lt0 ? e1t : (lt3 ? e2t : e3t), lt0 ? e1f : (lt3 ? e2f : e3f)};
        nfev += can2 ? 1 : 0;
        const bool t_lt_r = t.f < r.f, t_le_r = t.f <= r.f, t_lt_w = t.f < v3.f;
        const bool take_t = (lt0 & t_lt_r) | (!lt0 & lt3 & t_le_r) | (!lt3 & t_lt_w);
        const bool shrink = can2 & !lt0 & !take_t;
        const bool aborted0 = need2 & !can2;
        const bool use_t = can2 & take_t;
        const bool keep = shrink | aborted0;
        insert_vertex(v0, v1, v2, v3, r, t, use_t, !keep);
        bool aborted = aborted0;
        if (__builtin_expect(shrink, 0)) {
            // sim[j] = sim[0] + sigma (sim[j] - sim[0]), j = 1..3, each followed by its evaluation; a call beyond maxfun
            // aborts after the vertex was moved (SciPy's order)
            const int room = maxfun - nfev;
            Vtx n1 = v1, n2 = v2, n3 = v3;
            n1.x = v0.x + 0.5 * (v1.x - v0.x); n1.y = v0.y + 0.5 * (v1.y - v0.y); n1.t = v0.t + 0.5 * (v1.t - v0.t);
            n2.x = v0.x + 0.5 * (v2.x - v0.x); n2.y = v0.y + 0.5 * (v2.y - v0.y); n2.t = v0.t + 0.5 * (v2.t - v0.t);
            n3.x = v0.x + 0.5 * (v3.x - v0.x); n3.y = v0.y + 0.5 * (v3.y - v0.y); n3.t = v0.t + 0.5 * (v3.t - v0.t);
            Vtx c = n1;
            if (role == 2) c = n2;
            if (role == 3) c = n3;
            const double fs = f(c.x, c.y, c.t);
            n1.f = quad_bcast<1>(fs);
            n2.f = quad_bcast<2>(fs);
            n3.f = quad_bcast<3>(fs);
            if (room >= 3) {
                v1 = n1; v2 = n2; v3 = n3;
                nfev += 3;
            } else {  // vertex `room + 1` is moved but not evaluated, later ones are untouched
                aborted = true;
                if (room >= 1) v1 = n1; else { v1.x = n1.x; v1.y = n1.y; v1.t = n1.t; }
                if (room >= 2) v2 = n2; else if (room == 1) { v2.x = n2.x; v2.y = n2.y; v2.t = n2.t; }
                if (room == 2) { v3.x = n3.x; v3.y = n3.y; v3.t = n3.t; }
                nfev += room > 0 ? room : 0;
            }
            sort4(v0, v1, v2, v3);
        }
        it += aborted ? 0 : 1;
        stopped = stop();
        static_assert(maxfun <= maxiter + 3, "the evaluation budget implies the iteration budget");
        go = !aborted & (nfev < maxfun) & !(settled() | stopped);
    }
    out[0] = v0.x; out[1] = v0.y; out[2] = v0.t; out[3] = v0.f;
    out[4] = (double)it; out[5] = (double)nfev;
    return stopped;
}

}  // namespace icm
