/*
 * CPU oracle in plain C -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The same restatement of the reference's hot path as oracle/icm_oracle.py (Seba-san/icm-slam,
 * scripts/ICM_ROS.py:121-278 and scripts/ICM_SLAM_tools.py), written literally -- brute-force
 * cdist/argmin association, the running-mean recurrence of Mapa.actualizar, the per-beam
 * energy sum, SciPy's Nelder-Mead -- so that sweeps of 1e4 poses check in seconds and the
 * benchmark has a compiled single-core CPU baseline.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it (through oracle/c_oracle.py).
 *
 * Pin: tests/test_oracle_golden.py compares it with the golden vectors of the real reference
 * (poses/maps after sweeps 1 and 2 within 1e-9; it is not bit-identical to NumPy because
 * libm's sin/cos and the summation order of np.sum / BLAS differ in the last ulp).
 * Build: gcc -O2 -ffp-contract=off (unfused IEEE double like the reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PI 3.141592653589793
#define TWO_PI 6.283185307179586
#define HALF_PI 1.5707963267948966

typedef struct {
    double deltat, Q[2], R[3], cte_odom, cota, dist_thr, rango_laser_max;
    int64_t L;
} oc_config;

/* entrepi, scripts/ICM_SLAM_tools.py:455-463 */
static double wrap_pi(double a) {
    double r = fmod(a, TWO_PI);
    if (r < 0.0) r += TWO_PI;
    if (r > PI) r -= TWO_PI;
    return r;
}

static double med3(double a, double b, double c) {
    double lo = a < b ? a : b, hi = a < b ? b : a;
    double m = hi < c ? hi : c;
    return lo > m ? lo : m;
}

/* filtrar_z, scripts/ICM_SLAM_tools.py:22-58.  scan[B]; out rows [k, d, bx, by]; returns n. */
int64_t oc_filtrar_z(const oc_config* cfg, const double* scan, const double* cosb, const double* sinb,
                     int64_t B, int32_t* out_k, double* out_d, double* out_bx, double* out_by) {
    double* m = (double*)malloc(sizeof(double) * (size_t)B * 3);
    double *px = m + B, *py = m + 2 * B;
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)B);
    int64_t n = 0, kept = 0;
    for (int64_t k = 0; k < B; ++k) {
        double a = k > 0 ? scan[k - 1] : 0.0, c = k + 1 < B ? scan[k + 1] : 0.0;
        double v = med3(a, scan[k], c);
        if (v < cfg->rango_laser_max) {
            idx[n] = (int32_t)k;
            m[n] = v;
            px[n] = cosb[k] * v;
            py[n] = sinb[k] * v;
            ++n;
        }
    }
    if (n > 1) {
        for (int64_t i = 0; i < n; ++i) {
            double best = 100.0;
            for (int64_t j = 0; j < n; ++j) {
                double dx = px[i] - px[j], dy = py[i] - py[j];
                double d = sqrt(dx * dx + dy * dy);
                if (d == 0.0) d = 100.0;
                if (d < best) best = d;
            }
            if (best <= cfg->dist_thr) {
                out_k[kept] = idx[i];
                out_d[kept] = m[i];
                out_bx[kept] = px[i];
                out_by[kept] = py[i];
                ++kept;
            }
        }
    }
    free(m);
    free(idx);
    return kept;
}

/* ---- energies (scripts/ICM_ROS.py:171-278) ---- */
typedef struct {
    const oc_config* cfg;
    int two_sided;
    double xa[3], xp[3], ua[2], ut[2], oa[3], ot[3], op[3];
    const double *d, *ang, *tx, *ty;
    int64_t n;
} pose_problem;

static double obs_h(const pose_problem* p, const double x[3]) {
    double acc = 0.0;
    for (int64_t i = 0; i < p->n; ++i) {
        double alfa = p->ang[i] + x[2] - HALF_PI;
        double rx = (x[0] + p->d[i] * cos(alfa)) - p->tx[i];
        double ry = (x[1] + p->d[i] * sin(alfa)) - p->ty[i];
        acc += (rx * p->cfg->Q[0]) * rx;
        acc += (ry * p->cfg->Q[1]) * ry;
    }
    return acc;
}

static void g_step(const oc_config* c, const double x[3], const double u[2], double out[3]) {
    out[0] = x[0] + c->deltat * (cos(x[2]) * u[0]);
    out[1] = x[1] + c->deltat * (sin(x[2]) * u[0]);
    out[2] = x[2] + c->deltat * u[1];
}

static double odom_term(const oc_config* c, const double pa[3], const double pb[3], const double oa[3], const double ob[3]) {
    double co = cos(oa[2]), so = sin(oa[2]), dx = ob[0] - oa[0], dy = ob[1] - oa[1];
    double cp = cos(pa[2]), sp = sin(pa[2]), ex = pb[0] - pa[0], ey = pb[1] - pa[1];
    double q0 = (co * dx + so * dy) - (cp * ex + sp * ey);
    double q1 = (-so * dx + co * dy) - (-sp * ex + cp * ey);
    double q2 = wrap_pi(((ob[2] - oa[2]) - pb[2]) + pa[2]);
    return c->cte_odom * ((q0 * q0 + q1 * q1) + q2 * q2);
}

static double energy(const pose_problem* p, const double x[3]) {
    const oc_config* c = p->cfg;
    double ga[3], r[3];
    g_step(c, p->xa, p->ua, ga);
    r[0] = x[0] - ga[0]; r[1] = x[1] - ga[1]; r[2] = wrap_pi(x[2] - ga[2]);
    double prevR = ((r[0] * c->R[0]) * r[0] + (r[1] * c->R[1]) * r[1]) + (r[2] * c->R[2]) * r[2];
    double prevO = odom_term(c, p->xa, x, p->oa, p->ot);
    double hh = obs_h(p, x);
    if (!p->two_sided) return (prevR + hh) + prevO;
    double gx[3], s[3];
    g_step(c, x, p->ut, gx);
    s[0] = gx[0] - p->xp[0]; s[1] = gx[1] - p->xp[1]; s[2] = wrap_pi(gx[2] - p->xp[2]);
    double nextR = ((s[0] * c->R[0]) * s[0] + (s[1] * c->R[1]) * s[1]) + (s[2] * c->R[2]) * s[2];
    double nextO = odom_term(c, x, p->xp, p->ot, p->op);
    return (((nextR + nextO) + prevR) + hh) + prevO;
}

/* ---- scipy.optimize.fmin(f, x0, xtol=1e-3, disp=0) for N = 3 ---- */
typedef struct { double x[3], f; } vtx;

static void sort4(vtx v[4]) { /* stable insertion sort = numpy argsort for n < 16 */
    for (int i = 1; i < 4; ++i) {
        vtx t = v[i];
        int j = i - 1;
        while (j >= 0 && t.f < v[j].f) { v[j + 1] = v[j]; --j; }
        v[j + 1] = t;
    }
}

static void nelder_mead(const pose_problem* p, const double x0[3], double out[3], int* nit_out, int* nfev_out) {
    const int maxfun = 600, maxiter = 600;
    vtx v[4];
    int nfev = 0, it = 1;
    for (int k = 0; k < 4; ++k) memcpy(v[k].x, x0, sizeof(double) * 3);
    for (int k = 0; k < 3; ++k) v[k + 1].x[k] = x0[k] != 0.0 ? (1 + 0.05) * x0[k] : 0.00025;
    for (int k = 0; k < 4; ++k) { v[k].f = energy(p, v[k].x); ++nfev; }
    sort4(v);
    while (nfev < maxfun && it < maxiter) {
        double dx = 0.0, df = 0.0;
        for (int k = 1; k < 4; ++k) {
            for (int q = 0; q < 3; ++q) { double a = fabs(v[k].x[q] - v[0].x[q]); if (a > dx) dx = a; }
            double a = fabs(v[0].f - v[k].f); if (a > df) df = a;
        }
        if (dx <= 1e-3 && df <= 1e-4) break;
        double b[3];
        for (int q = 0; q < 3; ++q) b[q] = ((v[0].x[q] + v[1].x[q]) + v[2].x[q]) / 3.0;
        vtx r, t;
        int aborted = 0, shrink = 0;
        for (int q = 0; q < 3; ++q) r.x[q] = 2 * b[q] - v[3].x[q];
        r.f = energy(p, r.x); ++nfev;
        if (r.f < v[0].f) {
            for (int q = 0; q < 3; ++q) t.x[q] = 3 * b[q] - 2 * v[3].x[q];
            if (nfev >= maxfun) aborted = 1; else { t.f = energy(p, t.x); ++nfev; v[3] = t.f < r.f ? t : r; }
        } else if (r.f < v[2].f) {
            v[3] = r;
        } else if (r.f < v[3].f) {
            for (int q = 0; q < 3; ++q) t.x[q] = 1.5 * b[q] - 0.5 * v[3].x[q];
            if (nfev >= maxfun) aborted = 1; else { t.f = energy(p, t.x); ++nfev; if (t.f <= r.f) v[3] = t; else shrink = 1; }
        } else {
            for (int q = 0; q < 3; ++q) t.x[q] = 0.5 * b[q] + 0.5 * v[3].x[q];
            if (nfev >= maxfun) aborted = 1; else { t.f = energy(p, t.x); ++nfev; if (t.f < v[3].f) v[3] = t; else shrink = 1; }
        }
        if (shrink)
            for (int j = 1; j < 4 && !aborted; ++j) {
                for (int q = 0; q < 3; ++q) v[j].x[q] = v[0].x[q] + 0.5 * (v[j].x[q] - v[0].x[q]);
                if (nfev >= maxfun) aborted = 1; else { v[j].f = energy(p, v[j].x); ++nfev; }
            }
        if (!aborted) ++it;
        sort4(v);
        if (aborted) break;
    }
    memcpy(out, v[0].x, sizeof(double) * 3);
    if (nit_out) *nit_out = it;
    if (nfev_out) *nfev_out = nfev;
}

/* One solve with explicit inputs (the SURVEY Appendix C unit pin).  u (2,2)/(2,1), odo (3,3)/(3,2)
 * row-major like the numpy slices; beams as (d, ang). out = x,y,theta,f,nit,nfev */
void oc_solve_one(const oc_config* cfg, int two_sided, const double* x_ant, const double* x_pos, const double* u,
                  const double* odo, const double* d, const double* ang, const double* tx, const double* ty, int64_t n,
                  double* out) {
    pose_problem p;
    int uc = two_sided ? 2 : 1, oc = two_sided ? 3 : 2, nit, nfev;
    p.cfg = cfg; p.two_sided = two_sided; p.d = d; p.ang = ang; p.tx = tx; p.ty = ty; p.n = n;
    memcpy(p.xa, x_ant, 24);
    if (two_sided) memcpy(p.xp, x_pos, 24);
    p.ua[0] = u[0]; p.ua[1] = u[uc];
    if (two_sided) { p.ut[0] = u[1]; p.ut[1] = u[uc + 1]; }
    for (int r = 0; r < 3; ++r) { p.oa[r] = odo[r * oc]; p.ot[r] = odo[r * oc + 1]; if (two_sided) p.op[r] = odo[r * oc + 2]; }
    double st[3];
    if (two_sided) for (int q = 0; q < 3; ++q) st[q] = (p.xa[q] + p.xp[q]) / 2.0; else g_step(cfg, p.xa, p.ua, st);
    nelder_mead(&p, st, out, &nit, &nfev);
    out[3] = energy(&p, out); out[4] = nit; out[5] = nfev;
}

/* ---- Mapa.filtrar, scripts/ICM_SLAM_tools.py:204-265, literal O(n^2) form ---- */
int oc_filtrar(const oc_config* cfg, const double* y, const double* cnt, int64_t lact, double* y_out, double* cnt_out,
               int64_t* lact_out) {
    const int64_t L = cfg->L;
    int64_t n = 0;
    double *px = (double*)malloc(sizeof(double) * (size_t)(3 * lact + 3)), *py = px + lact + 1, *pc = py + lact + 1;
    for (int64_t i = 0; i < lact; ++i)
        if (cnt[i] >= cfg->cota) { px[n] = y[i]; py[n] = y[L + i]; pc[n] = cnt[i]; ++n; }
    if (n == 0) { free(px); return -4; }
    double amax = 0.0;
#pragma omp parallel for reduction(max : amax) if (n > 512)
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j) {
            double dx = px[i] - px[j], dy = py[i] - py[j], d = sqrt(dx * dx + dy * dy);
            if (d > amax) amax = d;
        }
    int64_t* c = (int64_t*)malloc(sizeof(int64_t) * (size_t)n * 3);
    int64_t *nn = c + n, *rank = c + 2 * n;
    double* nd = (double*)malloc(sizeof(double) * (size_t)n);
#pragma omp parallel for if (n > 512)
    for (int64_t i = 0; i < n; ++i) {
        double best = INFINITY; int64_t bj = 0;
        for (int64_t j = 0; j < n; ++j) {
            double dx = px[j] - px[i], dy = py[j] - py[i], d = sqrt(dx * dx + dy * dy);
            if (d == 0.0) d = amax;
            if (d < best) { best = d; bj = j; }
        }
        nd[i] = best; nn[i] = bj; c[i] = i;
    }
    for (int64_t i = 0; i < n; ++i) {
        if (!(nd[i] < cfg->dist_thr)) continue;
        int64_t from = c[nn[i]], to = c[i];
        if (from == to) continue;
        for (int64_t k = 0; k < n; ++k) if (c[k] == from) c[k] = to;
    }
    for (int64_t i = 0; i < n; ++i) rank[i] = 0;
    for (int64_t i = 0; i < n; ++i) rank[c[i]] = 1;
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) { int64_t u = rank[i]; rank[i] = m; m += u; }
    for (int64_t i = 0; i < 2 * L; ++i) y_out[i] = 0.0;
    for (int64_t i = 0; i < L; ++i) cnt_out[i] = 0.0;
    double* acc = (double*)calloc((size_t)m * 3, sizeof(double));
    for (int64_t i = 0; i < n; ++i) {
        int64_t r = rank[c[i]];
        acc[3 * r] += pc[i]; acc[3 * r + 1] += px[i] * pc[i]; acc[3 * r + 2] += py[i] * pc[i];
    }
    for (int64_t r = 0; r < m; ++r) { cnt_out[r] = acc[3 * r]; y_out[r] = acc[3 * r + 1] / acc[3 * r]; y_out[L + r] = acc[3 * r + 2] / acc[3 * r]; }
    *lact_out = m;
    free(acc); free(nd); free(c); free(px);
    return 0;
}

/* ---- association of one world point, scripts/ICM_SLAM_tools.py:169-172 ----
 * cdist + argmin (first index on ties) + gate d > dist_thr -> -1.  Brute force over the km
 * matchable columns, or -- same answer, O(1) instead of O(km) -- through a uniform grid with
 * cell edge >= dist_thr: the winner of the brute-force argmin either lies within dist_thr of
 * the point (then it is in the 3x3 cells around it, and so is every landmark that ties with it)
 * or it is gated to -1 whatever it was.  oc_set_grid(0) forces the literal scan;
 * tests/test_oracle_golden.py checks the two against each other. */
typedef struct {
    double x0, y0, inv;
    int64_t nx, ny;
    int64_t* start; /* [nx*ny+1] */
    int64_t* idx;   /* landmark ids, cell-sorted, ascending inside a cell */
} oc_grid;

static int g_use_grid = 1;
static int g_threads = 0; /* 0 = OpenMP default */
void oc_set_grid(int on) { g_use_grid = on; }
void oc_set_threads(int n) { g_threads = n; }
int oc_get_threads(void) {
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}

static int64_t cell_of(double v, double v0, double inv, int64_t n) {
    double c = floor((v - v0) * inv);
    if (!(c > 0.0)) return 0;
    if (c > (double)(n - 1)) return n - 1;
    return (int64_t)c;
}

static int grid_build(oc_grid* g, const double* mx, const double* my, int64_t km, double cell) {
    g->start = 0; g->idx = 0;
    if (km <= 0) return 0;
    double x0 = mx[0], x1 = mx[0], y0 = my[0], y1 = my[0];
    for (int64_t i = 1; i < km; ++i) {
        if (mx[i] < x0) x0 = mx[i];
        if (mx[i] > x1) x1 = mx[i];
        if (my[i] < y0) y0 = my[i];
        if (my[i] > y1) y1 = my[i];
    }
    if (!(x1 - x0 < 1e300) || !(y1 - y0 < 1e300) || !(cell > 0.0)) return 0; /* non-finite map: literal scan */
    while (((x1 - x0) / cell + 1.0) * ((y1 - y0) / cell + 1.0) > 16.0 * (double)km + 4096.0) cell *= 2.0;
    g->x0 = x0; g->y0 = y0; g->inv = 1.0 / cell;
    g->nx = (int64_t)floor((x1 - x0) / cell) + 1; g->ny = (int64_t)floor((y1 - y0) / cell) + 1;
    int64_t nc = g->nx * g->ny;
    g->start = (int64_t*)calloc((size_t)nc + 1, sizeof(int64_t));
    g->idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)km);
    int64_t* fill = (int64_t*)calloc((size_t)nc + 1, sizeof(int64_t));
    for (int64_t i = 0; i < km; ++i) g->start[cell_of(my[i], y0, g->inv, g->ny) * g->nx + cell_of(mx[i], x0, g->inv, g->nx) + 1]++;
    for (int64_t c = 0; c < nc; ++c) g->start[c + 1] += g->start[c];
    for (int64_t i = 0; i < km; ++i) { /* ascending i inside every cell */
        int64_t c = cell_of(my[i], y0, g->inv, g->ny) * g->nx + cell_of(mx[i], x0, g->inv, g->nx);
        g->idx[g->start[c] + fill[c]++] = i;
    }
    free(fill);
    return 1;
}

static int64_t associate_point(const oc_grid* g, const double* mx, const double* my, int64_t km, double wx, double wy, double thr) {
    double best = INFINITY; int64_t bid = -1;
    if (g && g->start) {
        int64_t cx = cell_of(wx, g->x0, g->inv, g->nx), cy = cell_of(wy, g->y0, g->inv, g->ny);
        for (int64_t ry = cy > 0 ? cy - 1 : 0; ry <= (cy + 1 < g->ny ? cy + 1 : g->ny - 1); ++ry)
            for (int64_t rx = cx > 0 ? cx - 1 : 0; rx <= (cx + 1 < g->nx ? cx + 1 : g->nx - 1); ++rx)
                for (int64_t p = g->start[ry * g->nx + rx]; p < g->start[ry * g->nx + rx + 1]; ++p) {
                    int64_t i = g->idx[p];
                    double dx = mx[i] - wx, dy = my[i] - wy, d = sqrt(dx * dx + dy * dy);
                    if (d < best || (d == best && i < bid)) { best = d; bid = i; }
                }
    } else {
        for (int64_t i = 0; i < km; ++i) {
            double dx = mx[i] - wx, dy = my[i] - wy, d = sqrt(dx * dx + dy * dy);
            if (d < best) { best = d; bid = i; }
        }
    }
    return (bid >= 0 && !(best > thr)) ? bid : -1;
}

/* ---- one sweep, scripts/ICM_ROS.py:121-164 in the three-phase form ----
 * Kept beams as CSR by pose (off[T+1]; d, ang, bx, by).  schedule 0 = reference order,
 * 1 = red-black.  x (3,T) in place.  Returns 0, -3 (IndexError cases), -4 (empty map), 1 (scan 0 empty).
 * labels_out / tgt_out (optional, nnz / 2*nnz): label of every kept beam and its target y[:, c].
 * OpenMP (when built with -fopenmp) spreads the per-pose association and the solves of one
 * colour over threads; every pose is still computed by one thread with the same arithmetic, and
 * the running means are folded in pose order by one thread, so results do not depend on the
 * thread count. */
int oc_sweep2(const oc_config* cfg, int64_t T, const int64_t* off, const double* bd, const double* bang,
              const double* bx, const double* by, const double* odo, const double* u, const double* x0,
              const double* map_in, int64_t K, int64_t lact_in, int schedule, double* x, double* map_out,
              double* cnt_out, int64_t* K_out, double* y_raw_out, double* cnt_raw_out, int64_t* lact_raw_out,
              int64_t* labels_out, double* tgt_out) {
    const int64_t L = cfg->L, nnz = off[T];
    if (off[1] == off[0]) return 1;
    if (off[T] == off[T - 1]) return -3;
    double* y = (double*)calloc((size_t)(3 * L), sizeof(double));
    double* cnt = y + 2 * L;
    double* tx = (double*)malloc(sizeof(double) * (size_t)(2 * nnz + 2));
    double* ty = tx + nnz + 1;
    int64_t* lab = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nnz + 1));
    int64_t lact = lact_in, km = K < lact_in ? K : lact_in;
    int rc = 0;
    oc_grid grid; grid.start = 0; grid.idx = 0;
    const int have_grid = g_use_grid && km > 64 && grid_build(&grid, map_in, map_in + K, km, cfg->dist_thr > 0.0 ? cfg->dist_thr : 1.0);
#ifdef _OPENMP
    const int nthr = g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    const int nthr = 1;
#endif
    (void)nthr;
    /* phase A: project with the previous-sweep pose, associate against the fixed mapa_viejo */
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthr)
    for (int64_t t = 0; t < T; ++t) {
        int64_t j0 = off[t], n = off[t + 1] - j0;
        if (n == 0) continue;
        const double px = t == 0 ? x0[0] : x[t], py = t == 0 ? x0[1] : x[T + t], th = t == 0 ? x0[2] : x[2 * T + t];
        double ct = cos(th - HALF_PI), st = sin(th - HALF_PI);
        for (int64_t j = j0; j < j0 + n; ++j) {
            double wx = (bx[j] * ct + by[j] * (-st)) + px, wy = (bx[j] * st + by[j] * ct) + py;
            lab[j] = associate_point(have_grid ? &grid : 0, map_in, map_in + K, km, wx, wy, cfg->dist_thr);
            tx[j] = wx; ty[j] = wy; /* world point for now */
        }
    }
    /* phase B in pose order: new-landmark ids, running means, record y[:, c] */
    for (int64_t t = 0; t < T && rc == 0; ++t) {
        int64_t j0 = off[t], n = off[t + 1] - j0;
        if (n == 0) continue;
        int anynew = 0;
        for (int64_t j = j0; j < j0 + n; ++j) if (lab[j] < 0) anynew = 1;
        if (anynew) {
            if (lact >= L) { rc = -3; break; }
            for (int64_t j = j0; j < j0 + n; ++j) if (lab[j] < 0) lab[j] = lact;
            ++lact;
        }
        for (int64_t j = j0; j < j0 + n; ++j) { /* first beam of each label folds its group */
            int leader = 1;
            for (int64_t q = j0; q < j; ++q) if (lab[q] == lab[j]) { leader = 0; break; }
            if (!leader) continue;
            double sx = 0.0, sy = 0.0; int64_t k = 0;
            for (int64_t q = j; q < j0 + n; ++q) if (lab[q] == lab[j]) { sx += tx[q]; sy += ty[q]; ++k; }
            int64_t i = lab[j];
            double nn = cnt[i], tot = nn + (double)k;
            y[i] = sx / tot + y[i] * nn / tot;
            y[L + i] = sy / tot + y[L + i] * nn / tot;
            cnt[i] = tot;
        }
        for (int64_t j = j0; j < j0 + n; ++j) { tx[j] = y[lab[j]]; ty[j] = y[L + lab[j]]; }
    }
    if (rc == 0 && labels_out) memcpy(labels_out, lab, sizeof(int64_t) * (size_t)nnz);
    if (rc == 0 && tgt_out) { memcpy(tgt_out, tx, sizeof(double) * (size_t)nnz); memcpy(tgt_out + nnz, ty, sizeof(double) * (size_t)nnz); }
    /* phase C */
    if (rc == 0) {
        int passes = schedule == 0 ? 1 : 2;
        for (int pass = 0; pass < passes; ++pass) {
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthr) if (schedule == 1)
            for (int64_t t = 1; t < T; ++t) {
                if (schedule == 1 && (t & 1) != (pass == 0 ? 1 : 0)) continue;
                int64_t j0 = off[t], n = off[t + 1] - j0;
                if (n == 0) {
                    const double p0 = t == 1 ? x0[0] : x[t - 1], p1 = t == 1 ? x0[1] : x[T + t - 1], p2 = t == 1 ? x0[2] : x[2 * T + t - 1];
                    x[t] = (p0 + x[t + 1]) / 2.0; x[T + t] = (p1 + x[T + t + 1]) / 2.0; x[2 * T + t] = (p2 + x[2 * T + t + 1]) / 2.0;
                    continue;
                }
                pose_problem p;
                p.cfg = cfg; p.two_sided = t + 1 < T; p.n = n;
                p.d = bd + j0; p.ang = bang + j0; p.tx = tx + j0; p.ty = ty + j0;
                for (int r = 0; r < 3; ++r) {
                    p.xa[r] = x[r * T + t - 1]; p.oa[r] = odo[r * T + t - 1]; p.ot[r] = odo[r * T + t];
                    if (p.two_sided) { p.xp[r] = x[r * T + t + 1]; p.op[r] = odo[r * T + t + 1]; }
                }
                p.ua[0] = u[t - 1]; p.ua[1] = u[T + t - 1];
                if (p.two_sided) { p.ut[0] = u[t]; p.ut[1] = u[T + t]; }
                double st[3], res[3];
                if (p.two_sided) for (int q = 0; q < 3; ++q) st[q] = (p.xa[q] + p.xp[q]) / 2.0; else g_step(cfg, p.xa, p.ua, st);
                nelder_mead(&p, st, res, 0, 0);
                x[t] = res[0]; x[T + t] = res[1]; x[2 * T + t] = res[2];
            }
        }
        if (y_raw_out) memcpy(y_raw_out, y, sizeof(double) * (size_t)(2 * L));
        if (cnt_raw_out) memcpy(cnt_raw_out, cnt, sizeof(double) * (size_t)L);
        if (lact_raw_out) *lact_raw_out = lact;
        rc = oc_filtrar(cfg, y, cnt, lact, map_out, cnt_out, K_out);
    }
    free(grid.start); free(grid.idx);
    free(lab); free(tx); free(y);
    return rc;
}

/* ---- the causal initialisation pass, scripts/ICM_ROS.py:102-119 (inicializar_online_process) on recorded data ----
 * The map y (2,L) / cnt (L) / *lact arrive seeded by the first scan's clusters (Mapa.actualizar's Lact == 0 branch,
 * scripts/ICM_SLAM_tools.py:160-165: SciPy's linkage, done by the caller); then for t = 1 .. T-1: predict with g,
 * project the kept beams with the prediction, associate against the RUNNING map (:168-172), fold the scan into it
 * (:184-195), one-sided solve from g(x_{t-1}, u_{t-1}) (minimizar_x, scripts/ICM_ROS.py:253-262).  x (3,T) out; x[:,0]
 * = odo[:,0].  Returns 0 or -3 (label capacity). */
int oc_init_pass(const oc_config* cfg, int64_t T, const int64_t* off, const double* bd, const double* bang,
                 const double* bx, const double* by, const double* odo, const double* u, double* y, double* cnt,
                 int64_t* lact_io, double* x) {
    const int64_t L = cfg->L;
    int64_t lact = *lact_io, maxn = 0;
    for (int64_t t = 0; t < T; ++t) if (off[t + 1] - off[t] > maxn) maxn = off[t + 1] - off[t];
    double* wx = (double*)malloc(sizeof(double) * (size_t)(4 * maxn + 4));
    double *wy = wx + maxn + 1, *tx = wy + maxn + 1, *ty = tx + maxn + 1;
    int64_t* lab = (int64_t*)malloc(sizeof(int64_t) * (size_t)(maxn + 1));
    double xt[3] = {odo[0], odo[T], odo[2 * T]};
    x[0] = xt[0]; x[T] = xt[1]; x[2 * T] = xt[2];
    int rc = 0;
    for (int64_t t = 1; t < T && rc == 0; ++t) {
        const double ua[2] = {u[t - 1], u[T + t - 1]};
        double xc[3];
        g_step(cfg, xt, ua, xc);
        const int64_t j0 = off[t], n = off[t + 1] - j0;
        if (n > 0) {
            const double ct = cos(xc[2] - HALF_PI), st = sin(xc[2] - HALF_PI);
            int anynew = 0;
            for (int64_t j = 0; j < n; ++j) {
                wx[j] = (bx[j0 + j] * ct + by[j0 + j] * (-st)) + xc[0];
                wy[j] = (bx[j0 + j] * st + by[j0 + j] * ct) + xc[1];
                lab[j] = associate_point(0, y, y + L, lact, wx[j], wy[j], cfg->dist_thr);
                if (lab[j] < 0) anynew = 1;
            }
            if (anynew) {
                if (lact >= L) { rc = -3; break; }
                for (int64_t j = 0; j < n; ++j) if (lab[j] < 0) lab[j] = lact;
                ++lact;
            }
            for (int64_t j = 0; j < n; ++j) {
                int leader = 1;
                for (int64_t q = 0; q < j; ++q) if (lab[q] == lab[j]) { leader = 0; break; }
                if (!leader) continue;
                double sx = 0.0, sy = 0.0; int64_t k = 0;
                for (int64_t q = j; q < n; ++q) if (lab[q] == lab[j]) { sx += wx[q]; sy += wy[q]; ++k; }
                const int64_t i = lab[j];
                const double nn = cnt[i], tot = nn + (double)k;
                y[i] = sx / tot + y[i] * nn / tot;
                y[L + i] = sy / tot + y[L + i] * nn / tot;
                cnt[i] = tot;
            }
            for (int64_t j = 0; j < n; ++j) { tx[j] = y[lab[j]]; ty[j] = y[L + lab[j]]; }
            pose_problem p;
            p.cfg = cfg; p.two_sided = 0; p.n = n;
            p.d = bd + j0; p.ang = bang + j0; p.tx = tx; p.ty = ty;
            for (int r = 0; r < 3; ++r) { p.xa[r] = xt[r]; p.oa[r] = odo[r * T + t - 1]; p.ot[r] = odo[r * T + t]; p.xp[r] = 0.0; p.op[r] = 0.0; }
            p.ua[0] = ua[0]; p.ua[1] = ua[1]; p.ut[0] = p.ut[1] = 0.0;
            double st0[3];
            g_step(cfg, p.xa, p.ua, st0);
            nelder_mead(&p, st0, xt, 0, 0);
        } else {
            xt[0] = xc[0]; xt[1] = xc[1]; xt[2] = xc[2];
        }
        x[t] = xt[0]; x[T + t] = xt[1]; x[2 * T + t] = xt[2];
    }
    *lact_io = lact;
    free(lab); free(wx);
    return rc;
}

int oc_sweep(const oc_config* cfg, int64_t T, const int64_t* off, const double* bd, const double* bang,
             const double* bx, const double* by, const double* odo, const double* u, const double* x0,
             const double* map_in, int64_t K, int64_t lact_in, int schedule, double* x, double* map_out,
             double* cnt_out, int64_t* K_out, double* y_raw_out, double* cnt_raw_out, int64_t* lact_raw_out) {
    return oc_sweep2(cfg, T, off, bd, bang, bx, by, odo, u, x0, map_in, K, lact_in, schedule, x, map_out, cnt_out, K_out,
                     y_raw_out, cnt_raw_out, lact_raw_out, 0, 0);
}
