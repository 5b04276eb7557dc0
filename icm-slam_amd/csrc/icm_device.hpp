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

// entrepi (reference scripts/ICM_SLAM_tools.py:455-463): numpy mod (sign of the divisor),
// then fold (pi, 2pi) down.  Result in [-pi, pi].
__device__ __forceinline__ double wrap_pi(double a) {
    double r = fmod(a, kTwoPi);
    if (r < 0.0) r += kTwoPi;
    if (r > kPi) r -= kTwoPi;
    return r;
}

__device__ __forceinline__ double bcast_first(double v) {
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Butterfly sum over the 64 lanes: every lane ends with the same bits.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
    return bcast_first(v);
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
};

// Observation items of one pose: weight k, body-frame point b, world target y.
// Plain beams have k = 1 (kw == nullptr).
struct Items {
    const double* __restrict__ bx;
    const double* __restrict__ by;
    const double* __restrict__ tx;
    const double* __restrict__ ty;
    const double* __restrict__ kw;
    double cst;  // x-independent remainder of h when items are aggregated moments
    int n;
};

// h(x) = sum_i k_i (w_i - y_i)^T Q (w_i - y_i), w_i = p + Rot(th - pi/2) b_i
// (reference scripts/ICM_ROS.py:171-200 with the body-frame points of filtrar_z's columns
// 2:4; d*cos(ang + th - pi/2) is expanded with the rotation tras_rot_z uses,
// scripts/ICM_SLAM_tools.py:476-479).
__device__ __forceinline__ double obs_energy(const SolveCtx& c, const Items& it, double px,
                                             double py, double th, int lane) {
    const double a = th - kHalfPi;
    const double ct = cos(a), st = sin(a);
    double acc = 0.0;
    for (int j = lane; j < it.n; j += kWave) {
        const double bx = it.bx[j], by = it.by[j];
        const double dx = (px + (bx * ct - by * st)) - it.tx[j];
        const double dy = (py + (bx * st + by * ct)) - it.ty[j];
        double e = (dx * c.Q0) * dx + (dy * c.Q1) * dy;
        if (it.kw) e *= it.kw[j];
        acc += e;
    }
    return wave_sum(acc) + it.cst;
}

// fun_xn (two_sided) / fun_x (reference scripts/ICM_ROS.py:220-278), Appendix A.4.
__device__ __forceinline__ double pose_energy(const SolveCtx& c, const Items& it, double px,
                                              double py, double th, int lane) {
    const double hh = obs_energy(c, it, px, py, th, lane);
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
    const double cth = cos(th), sth = sin(th);
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

// Fill the x-independent parts of the context for pose t.
//   xa = x[:,t-1], xp = x[:,t+1] (ignored if !two_sided), ua = u[:,t-1], ut = u[:,t],
//   oa/ot/op = odometria[:,t-1], [:,t], [:,t+1]
__device__ __forceinline__ void make_ctx(SolveCtx& c, int two_sided, const double xa[3],
                                         const double xp[3], const double ua[2],
                                         const double ut[2], const double oa[3],
                                         const double ot[3], const double op[3]) {
    c.two_sided = two_sided;
    c.xax = xa[0]; c.xay = xa[1]; c.xat = xa[2];
    c.ca = cos(xa[2]); c.sa = sin(xa[2]);
    // g(a, u_{t-1}) (reference scripts/ICM_ROS.py:202-207)
    c.gax = xa[0] + c.dt * (c.ca * ua[0]);
    c.gay = xa[1] + c.dt * (c.sa * ua[0]);
    c.gat = xa[2] + c.dt * ua[1];
    const double coa = cos(oa[2]), soa = sin(oa[2]);
    const double d1x = ot[0] - oa[0], d1y = ot[1] - oa[1];
    c.o1x = coa * d1x + soa * d1y;
    c.o1y = -soa * d1x + coa * d1y;
    c.o1t = ot[2] - oa[2];
    if (two_sided) {
        c.xpx = xp[0]; c.xpy = xp[1]; c.xpt = xp[2];
        c.v = ut[0]; c.w = ut[1];
        const double cot = cos(ot[2]), sot = sin(ot[2]);
        const double d2x = op[0] - ot[0], d2y = op[1] - ot[1];
        c.o2x = cot * d2x + sot * d2y;
        c.o2y = -sot * d2x + cot * d2y;
        c.o2t = op[2] - ot[2];
    } else {
        c.xpx = c.xpy = c.xpt = c.v = c.w = c.o2x = c.o2y = c.o2t = 0.0;
    }
}

struct Vtx {
    double x, y, t, f;
};

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

__device__ __forceinline__ double amax3(const Vtx& a, const Vtx& b) {
    return fmax(fmax(fabs(a.x - b.x), fabs(a.y - b.y)), fabs(a.t - b.t));
}

// scipy.optimize.fmin(f, x0, xtol=1e-3, disp=0) for N = 3:
// rho=1 chi=2 psi=sigma=0.5, xatol=1e-3 AND fatol=1e-4, maxiter=maxfun=600, initial simplex
// x0 with one coordinate *1.05 (0.00025 if it is exactly 0), one stable sort per iteration.
// A function call beyond maxfun aborts the iteration like SciPy's _MaxFuncCallError.
// out = {x, y, theta, f, nit, nfev}.
template <class F>
__device__ __forceinline__ void nelder_mead3(F f, double sx, double sy, double st, double out[6]) {
    const int maxfun = 600, maxiter = 600;
    const double xatol = 1e-3, fatol = 1e-4;
    const double grow = 1 + 0.05;
    Vtx v0{sx, sy, st, 0.0};
    Vtx v1{sx != 0.0 ? grow * sx : 0.00025, sy, st, 0.0};
    Vtx v2{sx, sy != 0.0 ? grow * sy : 0.00025, st, 0.0};
    Vtx v3{sx, sy, st != 0.0 ? grow * st : 0.00025, 0.0};
    v0.f = f(v0.x, v0.y, v0.t);
    v1.f = f(v1.x, v1.y, v1.t);
    v2.f = f(v2.x, v2.y, v2.t);
    v3.f = f(v3.x, v3.y, v3.t);
    int nfev = 4, it = 1;
    sort4(v0, v1, v2, v3);
    while (nfev < maxfun && it < maxiter) {
        const double dx = fmax(fmax(amax3(v1, v0), amax3(v2, v0)), amax3(v3, v0));
        const double df = fmax(fmax(fabs(v0.f - v1.f), fabs(v0.f - v2.f)), fabs(v0.f - v3.f));
        if (dx <= xatol && df <= fatol) break;
        const double bx = ((v0.x + v1.x) + v2.x) / 3.0;
        const double by = ((v0.y + v1.y) + v2.y) / 3.0;
        const double bt = ((v0.t + v1.t) + v2.t) / 3.0;
        do {
            Vtx r{2 * bx - v3.x, 2 * by - v3.y, 2 * bt - v3.t, 0.0};
            if (nfev >= maxfun) break;
            r.f = f(r.x, r.y, r.t);
            ++nfev;
            bool shrink = false;
            if (r.f < v0.f) {
                Vtx e{3 * bx - 2 * v3.x, 3 * by - 2 * v3.y, 3 * bt - 2 * v3.t, 0.0};
                if (nfev >= maxfun) break;
                e.f = f(e.x, e.y, e.t);
                ++nfev;
                v3 = (e.f < r.f) ? e : r;
            } else if (r.f < v2.f) {
                v3 = r;
            } else if (r.f < v3.f) {
                Vtx c{1.5 * bx - 0.5 * v3.x, 1.5 * by - 0.5 * v3.y, 1.5 * bt - 0.5 * v3.t, 0.0};
                if (nfev >= maxfun) break;
                c.f = f(c.x, c.y, c.t);
                ++nfev;
                if (c.f <= r.f) v3 = c; else shrink = true;
            } else {
                Vtx c{0.5 * bx + 0.5 * v3.x, 0.5 * by + 0.5 * v3.y, 0.5 * bt + 0.5 * v3.t, 0.0};
                if (nfev >= maxfun) break;
                c.f = f(c.x, c.y, c.t);
                ++nfev;
                if (c.f < v3.f) v3 = c; else shrink = true;
            }
            if (shrink) {
                v1.x = v0.x + 0.5 * (v1.x - v0.x); v1.y = v0.y + 0.5 * (v1.y - v0.y); v1.t = v0.t + 0.5 * (v1.t - v0.t);
                if (nfev >= maxfun) break;
                v1.f = f(v1.x, v1.y, v1.t); ++nfev;
                v2.x = v0.x + 0.5 * (v2.x - v0.x); v2.y = v0.y + 0.5 * (v2.y - v0.y); v2.t = v0.t + 0.5 * (v2.t - v0.t);
                if (nfev >= maxfun) break;
                v2.f = f(v2.x, v2.y, v2.t); ++nfev;
                v3.x = v0.x + 0.5 * (v3.x - v0.x); v3.y = v0.y + 0.5 * (v3.y - v0.y); v3.t = v0.t + 0.5 * (v3.t - v0.t);
                if (nfev >= maxfun) break;
                v3.f = f(v3.x, v3.y, v3.t); ++nfev;
            }
            ++it;
        } while (0);
        sort4(v0, v1, v2, v3);
    }
    out[0] = v0.x; out[1] = v0.y; out[2] = v0.t; out[3] = v0.f;
    out[4] = (double)it; out[5] = (double)nfev;
}

}  // namespace icm
