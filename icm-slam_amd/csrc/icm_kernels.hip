// HIP kernels of the offline ICM sweep for gfx950 (MI355X), FP64, wave64.
//
// Phase map:
//   once      k_prefilter      filtrar_z for every scan            -> kept beams, CSR by pose
//   phase A   k_associate      project + gated nearest landmark     -> label per kept beam
//             k_group          per pose: distinct labels, ordered sums of world points
//             k_compact        entries to pose-major compact arrays, fresh ids for new landmarks
//             (radix sort by label, rocPRIM)                        -> CSR by landmark
//   phase B/D k_lm_local       per-landmark local sufficient statistics (sum x, sum y, n)
//             k_stats_prefix   totals + exclusive prefix over lower ranks (after all-gather)
//             k_lm_chain       per-landmark time-ordered prefix -> running-mean targets
//             k_beam_targets   target per kept beam
//   phase C   k_solve          one wavefront per pose: Nelder-Mead on the conditional energy
//
// Mapping rule everywhere: one wavefront (64 lanes) per pose, lanes stride over the pose's
// kept beams / entries; 256-thread workgroups = 4 poses.  No MFMA: there is no dense
// contraction on this path.
#include <hip/hip_runtime.h>

#include "icm_device.hpp"

namespace icm {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kGroupCap = 256;  // distinct landmarks one scan may touch

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ int wave_in_block() { return threadIdx.x >> 6; }
__device__ __forceinline__ int prefix_count(unsigned long long mask, int lane) {
    return __popcll(mask & ((1ull << lane) - 1ull));
}

// ---------------------------------------------------------------------------------------
// Single-workgroup exclusive scan for the small (<= ~1e5) offset arrays; out[n] = total.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_exscan_i32(const int* __restrict__ in, int* __restrict__ out, int n) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int chunk = (n + 1023) / 1024;
    const int lo = min(tid * chunk, n), hi = min(lo + chunk, n);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += in[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;
    for (int i = lo; i < hi; ++i) {
        int v = in[i];
        out[i] = run;
        run += v;
    }
    if (tid == 1023) out[n] = part[1023];
}

// ---------------------------------------------------------------------------------------
// filtrar_z (reference scripts/ICM_SLAM_tools.py:22-58).
// One wave per scan.  WRITE=false counts the kept beams, WRITE=true stores them at boff[t].
// LDS per wave: the in-range beams (index, range, x, y), B entries each.
// ---------------------------------------------------------------------------------------
template <bool WRITE>
__global__ __launch_bounds__(kBlock) void k_prefilter(const double* __restrict__ ranges,
                                                      const double* __restrict__ cosb,
                                                      const double* __restrict__ sinb, int nloc, int B,
                                                      double rmax, double thr, int* __restrict__ nkept,
                                                      const int* __restrict__ boff, int* __restrict__ bk,
                                                      double* __restrict__ bd, double* __restrict__ bx,
                                                      double* __restrict__ by) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int w = wave_in_block(), lane = lane_id();
    const int t = blockIdx.x * kWavesPerBlock + w;
    if (t >= nloc) return;
    double* lpx = reinterpret_cast<double*>(smem) + (size_t)w * 3 * B;
    double* lpy = lpx + B;
    double* lm = lpy + B;
    int* lk = reinterpret_cast<int*>(reinterpret_cast<double*>(smem) + (size_t)kWavesPerBlock * 3 * B) + (size_t)w * B;
    const double* r = ranges + (size_t)t * B;
    // median of 3 with zero padding, range cut, compaction of the in-range beams
    int cnt = 0;
    for (int base = 0; base < B; base += kWave) {
        const int k = base + lane;
        bool in = false;
        double m = 0.0;
        if (k < B) {
            const double a = k > 0 ? r[k - 1] : 0.0, b = r[k], c = k + 1 < B ? r[k + 1] : 0.0;
            m = fmax(fmin(a, b), fmin(fmax(a, b), c));
            in = m < rmax;
        }
        const unsigned long long mask = __ballot(in);
        if (in) {
            const int p = cnt + prefix_count(mask, lane);
            lk[p] = k;
            lm[p] = m;
            lpx[p] = cosb[k] * m;
            lpy[p] = sinb[k] * m;
        }
        cnt += __popcll(mask);
    }
    __builtin_amdgcn_wave_barrier();
    if (cnt <= 1) {
        if (!WRITE && lane == 0) nkept[t] = 0;
        return;
    }
    // isolated-beam rejection: nearest other in-range beam, exact zeros count as 100
    int kept = 0;
    for (int base = 0; base < cnt; base += kWave) {
        const int i = base + lane;
        bool keep = false;
        if (i < cnt) {
            const double xi = lpx[i], yi = lpy[i];
            double smin = __builtin_huge_val();
            for (int j = 0; j < cnt; ++j) {
                const double dx = xi - lpx[j], dy = yi - lpy[j];
                const double s = dx * dx + dy * dy;
                if (s != 0.0 && s < smin) smin = s;
            }
            const double nn = fmin(100.0, sqrt(smin));
            keep = nn <= thr;
        }
        const unsigned long long mask = __ballot(keep);
        if (WRITE && keep) {
            const int p = boff[t] + kept + prefix_count(mask, lane);
            bk[p] = lk[i];
            bd[p] = lm[i];
            bx[p] = lpx[i];
            by[p] = lpy[i];
        }
        kept += __popcll(mask);
    }
    if (!WRITE && lane == 0) nkept[t] = kept;
}

// ---------------------------------------------------------------------------------------
// Phase A: project the kept beams with the previous-sweep pose (tras_rot_z, reference
// scripts/ICM_SLAM_tools.py:465-480) and associate each to the nearest landmark of
// mapa_viejo under the distance gate (Mapa.actualizar, scripts/ICM_SLAM_tools.py:168-172).
// The landmark table is searched through a uniform grid (cell >= dist_thr): 3 rows of 3
// cells, each row one contiguous range of the cell-sorted table.  label = column index of
// the nearest landmark (first index on ties, like np.argmin), -1 if farther than dist_thr.
// ---------------------------------------------------------------------------------------
struct GridView {
    double gx0, gy0, inv;
    int nx, ny;
    const int* __restrict__ cell_start;
    const double* __restrict__ lx;
    const double* __restrict__ ly;
    const int* __restrict__ id;
};

__device__ __forceinline__ int grid_cell(double v, double g0, double inv, int n) {
    double f = floor((v - g0) * inv);
    f = fmin(fmax(f, 0.0), (double)(n - 1));  // NaN -> 0
    return (int)f;
}

__device__ __forceinline__ void pose_of(const double* __restrict__ x, const double* __restrict__ x0, int tg,
                                        double& px, double& py, double& th) {
    if (tg == 0) {  // scan 0 is projected with self.x0 (scripts/ICM_ROS.py:125,137)
        px = x0[0]; py = x0[1]; th = x0[2];
    } else {
        px = x[3 * (size_t)tg]; py = x[3 * (size_t)tg + 1]; th = x[3 * (size_t)tg + 2];
    }
}

__global__ __launch_bounds__(kBlock) void k_associate(const double* __restrict__ x, const double* __restrict__ x0,
                                                      int t_begin, int nloc, const int* __restrict__ boff,
                                                      const double* __restrict__ bx, const double* __restrict__ by,
                                                      GridView g, double thr, int* __restrict__ label) {
    const int lane = lane_id();
    const int tl = blockIdx.x * kWavesPerBlock + wave_in_block();
    if (tl >= nloc) return;
    const int j0 = boff[tl], j1 = boff[tl + 1];
    if (j0 == j1) return;
    double px, py, th;
    pose_of(x, x0, t_begin + tl, px, py, th);
    const double ct = cos(th - kHalfPi), st = sin(th - kHalfPi);
    for (int j = j0 + lane; j < j1; j += kWave) {
        const double bxx = bx[j], byy = by[j];
        const double wx = (bxx * ct - byy * st) + px;
        const double wy = (bxx * st + byy * ct) + py;
        const int cx = grid_cell(wx, g.gx0, g.inv, g.nx), cy = grid_cell(wy, g.gy0, g.inv, g.ny);
        const int c0 = max(cx - 1, 0), c1 = min(cx + 1, g.nx - 1);
        double best = __builtin_huge_val();
        int bid = -1;
        for (int ry = max(cy - 1, 0); ry <= min(cy + 1, g.ny - 1); ++ry) {
            const int p0 = g.cell_start[ry * g.nx + c0], p1 = g.cell_start[ry * g.nx + c1 + 1];
            for (int p = p0; p < p1; ++p) {
                const double dx = g.lx[p] - wx, dy = g.ly[p] - wy;
                const double d = sqrt(dx * dx + dy * dy);
                const int id = g.id[p];
                if (d < best || (d == best && id < bid)) {
                    best = d;
                    bid = id;
                }
            }
        }
        label[j] = (bid >= 0 && !(best > thr)) ? bid : -1;
    }
}

// Brute-force form of the same association (all K landmarks, LDS-tiled table): the literal
// cdist/argmin of the reference.  Used to cross-check the grid search on the GPU.
__global__ __launch_bounds__(kBlock) void k_associate_brute(const double* __restrict__ x, const double* __restrict__ x0,
                                                            int t_begin, int nloc, const int* __restrict__ boff,
                                                            const double* __restrict__ bx, const double* __restrict__ by,
                                                            const double* __restrict__ mapx, const double* __restrict__ mapy,
                                                            int K, double thr, int* __restrict__ label) {
    constexpr int TILE = 1024;
    __shared__ double sx[TILE], sy[TILE];
    const int lane = lane_id();
    const int tl = blockIdx.x * kWavesPerBlock + wave_in_block();
    const bool live = tl < nloc;
    int j0 = 0, j1 = 0;
    double px = 0, py = 0, th = 0;
    if (live) {
        j0 = boff[tl];
        j1 = boff[tl + 1];
        pose_of(x, x0, t_begin + tl, px, py, th);
    }
    const double ct = cos(th - kHalfPi), st = sin(th - kHalfPi);
    const int iters = (j1 - j0 + kWave - 1) / kWave;
    int maxit = iters;  // block-uniform trip count so every wave reaches the barriers
    __shared__ int s_maxit;
    if (threadIdx.x == 0) s_maxit = 0;
    __syncthreads();
    atomicMax(&s_maxit, iters);
    __syncthreads();
    maxit = s_maxit;
    for (int itn = 0; itn < maxit; ++itn) {
        const int j = j0 + itn * kWave + lane;
        const bool on = live && j < j1;
        double wx = 0, wy = 0;
        if (on) {
            wx = (bx[j] * ct - by[j] * st) + px;
            wy = (bx[j] * st + by[j] * ct) + py;
        }
        double best = __builtin_huge_val();
        int bid = -1;
        for (int k0 = 0; k0 < K; k0 += TILE) {
            __syncthreads();
            for (int k = threadIdx.x; k < TILE && k0 + k < K; k += kBlock) {
                sx[k] = mapx[k0 + k];
                sy[k] = mapy[k0 + k];
            }
            __syncthreads();
            const int kn = min(TILE, K - k0);
            if (on)
                for (int k = 0; k < kn; ++k) {
                    const double dx = sx[k] - wx, dy = sy[k] - wy;
                    const double d = sqrt(dx * dx + dy * dy);
                    if (d < best) {
                        best = d;
                        bid = k0 + k;
                    }
                }
        }
        if (on) label[j] = (bid >= 0 && !(best > thr)) ? bid : -1;
    }
}

// ---------------------------------------------------------------------------------------
// Per pose: distinct labels of the scan (label -1 = the scan's gated-out beams, which the
// reference folds into ONE new landmark) with the ordered sum of
// their world points and the count -- the per-scan terms of the running mean
// (scripts/ICM_SLAM_tools.py:184-195).  Entries are staged at the front of the pose's beam
// range; bloc[j] = entry of beam j within its pose.
// ---------------------------------------------------------------------------------------
struct GroupScratch {
    int tlab[kGroupCap];
    int tk[kGroupCap];
    double tsx[kGroupCap];
    double tsy[kGroupCap];
    double cwx[kWave];
    double cwy[kWave];
    int ce[kWave];
};

__global__ __launch_bounds__(kBlock) void k_group(const double* __restrict__ x, const double* __restrict__ x0,
                                                  int t_begin, int nloc, const int* __restrict__ boff,
                                                  const double* __restrict__ bx, const double* __restrict__ by,
                                                  const int* __restrict__ label, int* __restrict__ bloc,
                                                  int* __restrict__ st_label, int* __restrict__ st_k,
                                                  double* __restrict__ st_sx, double* __restrict__ st_sy,
                                                  int* __restrict__ nent_out, int* __restrict__ isnew_out,
                                                  int* __restrict__ flags) {
    __shared__ GroupScratch scratch[kWavesPerBlock];
    const int lane = lane_id();
    const int tl = blockIdx.x * kWavesPerBlock + wave_in_block();
    if (tl >= nloc) return;
    GroupScratch& s = scratch[wave_in_block()];
    const int j0 = boff[tl], j1 = boff[tl + 1];
    if (j0 == j1) {
        if (lane == 0) {
            nent_out[tl] = 0;
            isnew_out[tl] = 0;
        }
        return;
    }
    double px, py, th;
    pose_of(x, x0, t_begin + tl, px, py, th);
    const double ct = cos(th - kHalfPi), st = sin(th - kHalfPi);
    int nent = 0;
    bool overflow = false;
    for (int base = j0; base < j1; base += kWave) {
        const int j = base + lane;
        const bool valid = j < j1;
        const int cn = min(kWave, j1 - base);
        int lab = -2;
        double wx = 0, wy = 0;
        if (valid) {
            lab = label[j];
            wx = (bx[j] * ct - by[j] * st) + px;
            wy = (bx[j] * st + by[j] * ct) + py;
        }
        // look the label up among the entries found so far
        int e = -1;
        for (int q = 0; q < nent; ++q)
            if (s.tlab[q] == lab) e = q;
        // append the labels that are new in this chunk, in beam order
        bool pend = valid && e < 0;
        unsigned long long m;
        while ((m = __ballot(pend)) != 0ull) {
            const int l0 = __ffsll((long long)m) - 1;
            const int lab0 = __shfl(lab, l0, kWave);
            if (nent >= kGroupCap) {
                overflow = true;
                pend = false;
                continue;
            }
            if (lane == l0) {
                s.tlab[nent] = lab0;
                s.tk[nent] = 0;
                s.tsx[nent] = 0.0;
                s.tsy[nent] = 0.0;
            }
            if (pend && lab == lab0) {
                e = nent;
                pend = false;
            }
            ++nent;
        }
        s.cwx[lane] = wx;
        s.cwy[lane] = wy;
        s.ce[lane] = valid ? e : -3;
        __builtin_amdgcn_wave_barrier();
        // ordered accumulation: every member walks the chunk in beam order, the first
        // member of each entry writes the result back
        if (valid && e >= 0) {
            double sx = s.tsx[e], sy = s.tsy[e];
            int kk = s.tk[e];
            bool leader = true;
            for (int q = 0; q < cn; ++q) {
                if (s.ce[q] == e) {
                    sx += s.cwx[q];
                    sy += s.cwy[q];
                    ++kk;
                    if (q < lane) leader = false;
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (leader) {
                s.tsx[e] = sx;
                s.tsy[e] = sy;
                s.tk[e] = kk;
            }
            bloc[j] = e;
        } else if (valid) {
            bloc[j] = 0;
        }
        __builtin_amdgcn_wave_barrier();
    }
    bool isnew = false;
    for (int q = lane; q < nent; q += kWave) {
        st_label[j0 + q] = s.tlab[q];
        st_k[j0 + q] = s.tk[q];
        st_sx[j0 + q] = s.tsx[q];
        st_sy[j0 + q] = s.tsy[q];
        isnew |= s.tlab[q] == -1;
    }
    const unsigned long long anynew = __ballot(isnew);
    if (lane == 0) {
        nent_out[tl] = nent;
        isnew_out[tl] = anynew != 0ull;
        if (overflow) flags[0] = 1;
    }
}

// Entries -> pose-major compact arrays; the gated-out group of a pose gets the fresh id
// lact0 + (number of earlier poses that created a landmark) (phase B).
__global__ __launch_bounds__(kBlock) void k_compact(int nloc, const int* __restrict__ boff,
                                                    const int* __restrict__ ent_off, const int* __restrict__ new_rank,
                                                    int lact0, const int* __restrict__ st_label,
                                                    const int* __restrict__ st_k, const double* __restrict__ st_sx,
                                                    const double* __restrict__ st_sy, unsigned* __restrict__ e_key,
                                                    int* __restrict__ e_val, int* __restrict__ e_k,
                                                    double* __restrict__ e_sx, double* __restrict__ e_sy) {
    const int lane = lane_id();
    const int tl = blockIdx.x * kWavesPerBlock + wave_in_block();
    if (tl >= nloc) return;
    const int j0 = boff[tl], e0 = ent_off[tl], n = ent_off[tl + 1] - e0;
    for (int q = lane; q < n; q += kWave) {
        int lab = st_label[j0 + q];
        if (lab < 0) lab = lact0 + new_rank[tl];
        e_key[e0 + q] = (unsigned)lab;
        e_val[e0 + q] = e0 + q;
        e_k[e0 + q] = st_k[j0 + q];
        e_sx[e0 + q] = st_sx[j0 + q];
        e_sy[e0 + q] = st_sy[j0 + q];
    }
}

// CSR by landmark from the label-sorted entry keys: lm_off[i] = first position with key >= i.
__global__ __launch_bounds__(kBlock) void k_lm_bounds(const unsigned* __restrict__ skey, int E, int nlab,
                                                      int* __restrict__ lm_off) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i > nlab) return;
    int lo = 0, hi = E;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (skey[mid] < (unsigned)i) lo = mid + 1; else hi = mid;
    }
    lm_off[i] = lo;
}

// Local sufficient statistics of every label: (sum x, sum y, n) over this rank's poses in
// time order.  stats layout: [sx(L) | sy(L) | n(L) | header(8)].
__global__ __launch_bounds__(kBlock) void k_lm_local(int nlab, int L, const int* __restrict__ lm_off,
                                                     const int* __restrict__ sval, const int* __restrict__ e_k,
                                                     const double* __restrict__ e_sx, const double* __restrict__ e_sy,
                                                     double* __restrict__ stats) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= L) return;
    double sx = 0.0, sy = 0.0, n = 0.0;
    if (i < nlab)
        for (int p = lm_off[i]; p < lm_off[i + 1]; ++p) {
            const int e = sval[p];
            sx += e_sx[e];
            sy += e_sy[e];
            n += (double)e_k[e];
        }
    stats[i] = sx;
    stats[L + i] = sy;
    stats[2 * L + i] = n;
}

// After the all-gather of the per-rank statistics: for existing landmarks (i < lact0) the
// exclusive prefix over lower ranks (the state of the running mean when this rank's first
// pose is folded in) and the total over all ranks; landmarks created during the sweep are
// singletons and are laid out rank after rank behind lact0.
// header per rank: [0] number of new landmarks, [1] error flags.
__global__ __launch_bounds__(kBlock) void k_stats_prefix(const double* __restrict__ stats_all, int stride, int rank,
                                                         int world, int L, int lact0, double* __restrict__ off_sx,
                                                         double* __restrict__ off_sy, double* __restrict__ off_n,
                                                         double* __restrict__ y_raw, double* __restrict__ cnt_raw) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= L) return;
    if (i < lact0) {
        double sx = 0.0, sy = 0.0, n = 0.0;
        for (int r = 0; r < world; ++r) {
            if (r == rank) {
                off_sx[i] = sx;
                off_sy[i] = sy;
                off_n[i] = n;
            }
            const double* s = stats_all + (size_t)r * stride;
            sx += s[i];
            sy += s[L + i];
            n += s[2 * L + i];
        }
        cnt_raw[i] = n;
        y_raw[i] = n > 0.0 ? sx / n : 0.0;
        y_raw[L + i] = n > 0.0 ? sy / n : 0.0;
    } else {
        off_sx[i] = off_sy[i] = off_n[i] = 0.0;
        // which rank's new landmark lands in column i?
        int q = i - lact0;
        double sx = 0.0, sy = 0.0, n = 0.0;
        for (int r = 0; r < world; ++r) {
            const double* s = stats_all + (size_t)r * stride;
            const int nn = (int)s[3 * L];
            if (q < nn) {
                sx = s[lact0 + q];
                sy = s[L + lact0 + q];
                n = s[2 * L + lact0 + q];
                break;
            }
            q -= nn;
        }
        cnt_raw[i] = n;
        y_raw[i] = n > 0.0 ? sx / n : 0.0;
        y_raw[L + i] = n > 0.0 ? sy / n : 0.0;
    }
}

// Running-mean target of every (pose, landmark) entry: the mean of all observations of the
// landmark through that pose inclusive (SURVEY Appendix A.3/A.6 phase B), as prefix sums of
// the sufficient statistics in time order, seeded with the lower ranks' totals.
__global__ __launch_bounds__(kBlock) void k_lm_chain(int nlab, const int* __restrict__ lm_off,
                                                     const int* __restrict__ sval, const int* __restrict__ e_k,
                                                     const double* __restrict__ e_sx, const double* __restrict__ e_sy,
                                                     const double* __restrict__ off_sx, const double* __restrict__ off_sy,
                                                     const double* __restrict__ off_n, double* __restrict__ tgt_x,
                                                     double* __restrict__ tgt_y) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nlab) return;
    double sx = off_sx[i], sy = off_sy[i], n = off_n[i];
    for (int p = lm_off[i]; p < lm_off[i + 1]; ++p) {
        const int e = sval[p];
        sx += e_sx[e];
        sy += e_sy[e];
        n += (double)e_k[e];
        tgt_x[e] = sx / n;
        tgt_y[e] = sy / n;
    }
}

// Target per kept beam: y[:, c] of scripts/ICM_ROS.py:152.
__global__ __launch_bounds__(kBlock) void k_beam_targets(int nloc, const int* __restrict__ boff,
                                                         const int* __restrict__ ent_off, const int* __restrict__ bloc,
                                                         const double* __restrict__ tgt_x, const double* __restrict__ tgt_y,
                                                         double* __restrict__ btx, double* __restrict__ bty) {
    const int lane = lane_id();
    const int tl = blockIdx.x * kWavesPerBlock + wave_in_block();
    if (tl >= nloc) return;
    const int e0 = ent_off[tl];
    for (int j = boff[tl] + lane; j < boff[tl + 1]; j += kWave) {
        const int e = e0 + bloc[j];
        btx[j] = tgt_x[e];
        bty[j] = tgt_y[e];
    }
}

// ---------------------------------------------------------------------------------------
// Phase C: pose solves.
// ---------------------------------------------------------------------------------------
struct SolveArgs {
    double* x;            // (T,3) poses, updated in place
    const double* x0;     // self.x0
    const double* odo;    // (3,T)
    const double* u;      // (2,T)
    int T, t_begin, nloc;
    const int* boff;
    const double *bx, *by, *btx, *bty;
    double dt, R0, R1, R2, Q0, Q1, cte;
    double* diag;         // optional (T,3): f, nit, nfev per pose
};

__device__ __forceinline__ void load3(const double* __restrict__ a, int T, int t, double o[3]) {
    o[0] = a[t]; o[1] = a[(size_t)T + t]; o[2] = a[2 * (size_t)T + t];
}

// Solve pose tg (global index); `prev` = x[:,tg-1] as it stands now.
__device__ __forceinline__ void solve_pose(const SolveArgs& a, int tg, const double prev[3], double res[3], int lane) {
    const int tl = tg - a.t_begin;
    const int j0 = a.boff[tl], n = a.boff[tl + 1] - j0;
    const bool last = tg + 1 >= a.T;
    if (n == 0) {
        // no beams: midpoint of the last solved pose and the old next pose
        // (scripts/ICM_ROS.py:143-147); before the first solve the former is self.x0
        const double* nx = a.x + 3 * (size_t)(tg + 1);
        const double p0 = tg == 1 ? a.x0[0] : prev[0], p1 = tg == 1 ? a.x0[1] : prev[1], p2 = tg == 1 ? a.x0[2] : prev[2];
        res[0] = (p0 + nx[0]) / 2.0;
        res[1] = (p1 + nx[1]) / 2.0;
        res[2] = (p2 + nx[2]) / 2.0;
        return;
    }
    SolveCtx c;
    c.dt = a.dt; c.R0 = a.R0; c.R1 = a.R1; c.R2 = a.R2; c.Q0 = a.Q0; c.Q1 = a.Q1; c.cte = a.cte;
    double xp[3] = {0, 0, 0}, ua[2], ut[2] = {0, 0}, oa[3], ot[3], op[3] = {0, 0, 0};
    ua[0] = a.u[tg - 1]; ua[1] = a.u[(size_t)a.T + tg - 1];
    load3(a.odo, a.T, tg - 1, oa);
    load3(a.odo, a.T, tg, ot);
    if (!last) {
        xp[0] = a.x[3 * (size_t)(tg + 1)]; xp[1] = a.x[3 * (size_t)(tg + 1) + 1]; xp[2] = a.x[3 * (size_t)(tg + 1) + 2];
        ut[0] = a.u[tg]; ut[1] = a.u[(size_t)a.T + tg];
        load3(a.odo, a.T, tg + 1, op);
    }
    make_ctx(c, !last, prev, xp, ua, ut, oa, ot, op);
    Items it{a.bx + j0, a.by + j0, a.btx + j0, a.bty + j0, nullptr, 0.0, n};
    double sx, sy, st;
    if (!last) {  // minimizar_xn start (scripts/ICM_ROS.py:217)
        sx = (prev[0] + xp[0]) / 2.0; sy = (prev[1] + xp[1]) / 2.0; st = (prev[2] + xp[2]) / 2.0;
    } else {      // minimizar_x start g(x_{t-1}, u_{t-1}) (scripts/ICM_ROS.py:258)
        sx = c.gax; sy = c.gay; st = c.gat;
    }
    double out[6];
    nelder_mead3([&](double px, double py, double th) { return pose_energy(c, it, px, py, th, lane); }, sx, sy, st, out);
    res[0] = out[0]; res[1] = out[1]; res[2] = out[2];
    if (a.diag && lane == 0) {
        a.diag[3 * (size_t)tg] = out[3];
        a.diag[3 * (size_t)tg + 1] = out[4];
        a.diag[3 * (size_t)tg + 2] = out[5];
    }
}

// Red-black half sweep: all poses of one parity (colour = tg & 1) of this shard, one wave
// each.  Neighbours have the other parity, so nothing read here is written by this launch.
__global__ __launch_bounds__(kBlock) void k_solve_colour(SolveArgs a, int colour) {
    const int lane = lane_id();
    const int w = blockIdx.x * kWavesPerBlock + wave_in_block();
    int first = a.t_begin > 1 ? a.t_begin : 1;
    if ((first & 1) != colour) ++first;
    const int tg = first + 2 * w;
    if (tg >= a.t_begin + a.nloc) return;
    double prev[3] = {a.x[3 * (size_t)(tg - 1)], a.x[3 * (size_t)(tg - 1) + 1], a.x[3 * (size_t)(tg - 1) + 2]};
    double res[3];
    solve_pose(a, tg, prev, res, lane);
    if (lane == 0) {
        a.x[3 * (size_t)tg] = res[0];
        a.x[3 * (size_t)tg + 1] = res[1];
        a.x[3 * (size_t)tg + 2] = res[2];
    }
}

// Reference order: one wave walks the chain t = 1..T-1, each solve conditioned on the pose
// it has just written (Gauss-Seidel, scripts/ICM_ROS.py:141-158).
__global__ __launch_bounds__(kWave) void k_solve_sequential(SolveArgs a) {
    const int lane = lane_id();
    double prev[3] = {a.x[0], a.x[1], a.x[2]};
    for (int tg = 1; tg < a.T; ++tg) {
        double res[3];
        solve_pose(a, tg, prev, res, lane);
        if (lane == 0) {
            a.x[3 * (size_t)tg] = res[0];
            a.x[3 * (size_t)tg + 1] = res[1];
            a.x[3 * (size_t)tg + 2] = res[2];
        }
        prev[0] = res[0]; prev[1] = res[1]; prev[2] = res[2];
    }
}

// One explicit solve / energy evaluation (parity tests).  io: see icm_solve_one.
struct OneArgs {
    int two_sided, energy_only, n;
    const double* p;  // packed: x(3) x_ant(3) x_pos(3) ua(2) ut(2) oa(3) ot(3) op(3)
    const double *bx, *by, *tx, *ty;
    double dt, R0, R1, R2, Q0, Q1, cte;
    double* out;  // 6
};
__global__ __launch_bounds__(kWave) void k_solve_one(OneArgs a) {
    const int lane = lane_id();
    SolveCtx c;
    c.dt = a.dt; c.R0 = a.R0; c.R1 = a.R1; c.R2 = a.R2; c.Q0 = a.Q0; c.Q1 = a.Q1; c.cte = a.cte;
    const double* p = a.p;
    make_ctx(c, a.two_sided, p + 3, p + 6, p + 9, p + 11, p + 13, p + 16, p + 19);
    Items it{a.bx, a.by, a.tx, a.ty, nullptr, 0.0, a.n};
    double out[6] = {0, 0, 0, 0, 0, 0};
    if (a.energy_only == 2) {
        out[3] = obs_energy(c, it, p[0], p[1], p[2], lane);
    } else if (a.energy_only) {
        out[3] = pose_energy(c, it, p[0], p[1], p[2], lane);
    } else {
        double sx, sy, st;
        if (a.two_sided) {
            sx = (p[3] + p[6]) / 2.0; sy = (p[4] + p[7]) / 2.0; st = (p[5] + p[8]) / 2.0;
        } else {
            sx = c.gax; sy = c.gay; st = c.gat;
        }
        nelder_mead3([&](double px, double py, double th) { return pose_energy(c, it, px, py, th, lane); }, sx, sy, st, out);
    }
    if (lane == 0)
        for (int i = 0; i < 6; ++i) a.out[i] = out[i];
}

// (3,T) <-> (T,3) pose layout change between the reference's host layout and HBM.
__global__ void k_fill_i32(int* p, int v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace icm
