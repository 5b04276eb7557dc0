// HIP kernels of the offline ICM sweep for gfx950 (MI355X), FP64, wave64.
//
// Phase map:
//   once      k_prefilter      filtrar_z for every scan            -> kept beams, CSR by pose
//             k_run_build      geometric runs of every scan's kept beams (bounding circle, count, sum)
//   phase A   k_assoc_runs     project + gated nearest landmark BY RUNS (exact bounding-circle test, beam by beam
//                              where it does not settle) + per-pose grouping: one (pose, landmark) ENTRY per
//                              distinct label of the scan with the count and the sum of the body points
//             k_assoc_group    the same beam by beam (until round 4 the hot kernel; now the cross-check form and
//                              the grouping behind the brute-force search)
//             k_neigh_table    per-cell 3x3 neighbourhood records of the search grid
//             k_scan_*         entry offsets, ranks of poses that create a landmark
//   phase B/D (default) hierarchical running sums, no sort:
//             k_chunk_l1       one wave per 64-pose chunk, poses in time order, LDS table by landmark
//             k_chunk_l2       one workgroup per superchunk of chunks -> dense [superchunk x L] totals
//             k_lm_l3          one thread per landmark: column prefix, landmark totals (raw map)
//             k_rec_push       per (chunk, landmark) record: sums before the chunk
//   phase B/D (fallback for very dense maps; per-beam / per-entry cross-check forms; debug dump):
//             k_compact        entries -> pose-major compact records, fresh ids for new landmarks
//             (radix sort of the entry ids by label, rocPRIM)      -> CSR by landmark
//             k_lm_scan        one wave per landmark: time-ordered prefix of its entries
//                              -> running-mean target of every entry, landmark totals
//             k_stats_prefix   totals + exclusive prefix over lower ranks (after all-gather)
//             k_fl_*           Mapa.filtrar + search grid of the refined map (multi-workgroup chain,
//                              side stream, under the solves; merges on the device)
//   phase C   k_pose_moments[_h]  14 moment sums of the pose's observation energy
//             k_solve_m_*      Nelder-Mead on the conditional energy in moment form:
//                              ONE LANE per pose, everything in registers
//             k_solve_* (wave per pose, per-beam / per-entry energy): cross-checks
//   init      k_init_pass      the causal initialisation pass (one wave walks the sequence)
//
// Mapping: phase A one wavefront per pose (lanes over its runs / its kept beams), entry kernels one DPP
// row (16 lanes) per pose, landmark kernels one wavefront per landmark, solves one lane per
// pose; 256-thread workgroups.  No MFMA: there is no dense contraction on this path.
#include <hip/hip_runtime.h>

#include "icm_device.hpp"

namespace icm {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kEmpty = (int)0x80000000;

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ int wave_in_block() { return threadIdx.x >> 6; }
__device__ __forceinline__ int prefix_count(unsigned long long mask, int lane) {
    return __popcll(mask & ((1ull << lane) - 1ull));
}
__device__ __forceinline__ double row_sum16(double v) {  // sum over the 16 lanes of a DPP row, in every lane
    v += dpp_mov<0x121, 0xF>(v);  // row_ror:1
    v += dpp_mov<0x122, 0xF>(v);  // row_ror:2
    v += dpp_mov<0x124, 0xF>(v);  // row_ror:4
    v += dpp_mov<0x128, 0xF>(v);  // row_ror:8
    return v;
}

// ---------------------------------------------------------------------------------------
// Two-level exclusive scan of two int arrays at once (entry counts, new-landmark flags).
// k_scan_tiles: 1024-element tiles, local exclusive scan + tile totals;
// k_scan_fix:   adds the totals of the preceding tiles; out[n] = grand total.
// ---------------------------------------------------------------------------------------
constexpr int kScanTile = 1024;
// abort (nullable): a device word; non-zero = leave the outputs alone (a sweep queued without a host
// check in between must not touch the map state after a table overflow).
__global__ __launch_bounds__(kBlock) void k_scan_tiles(const int* __restrict__ a, const int* __restrict__ b,
                                                       int* __restrict__ oa, int* __restrict__ ob,
                                                       int* __restrict__ tot, int n, const int* __restrict__ abort = nullptr) {
    __shared__ int wa[kWavesPerBlock], wb[kWavesPerBlock];
    if (abort && *abort) return;
    const int base = blockIdx.x * kScanTile + threadIdx.x * 4;
    int va[4], vb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        va[i] = base + i < n ? a[base + i] : 0;
        vb[i] = base + i < n ? b[base + i] : 0;
    }
    int sa = va[0] + va[1] + va[2] + va[3], sb = vb[0] + vb[1] + vb[2] + vb[3];
    int ia = sa, ib = sb;  // inclusive scan over the wave
    const int lane = lane_id(), w = wave_in_block();
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int ua = __shfl_up(ia, d, kWave), ub = __shfl_up(ib, d, kWave);
        if (lane >= d) {
            ia += ua;
            ib += ub;
        }
    }
    if (lane == kWave - 1) {
        wa[w] = ia;
        wb[w] = ib;
    }
    __syncthreads();
    int pa = 0, pb = 0;
    for (int q = 0; q < w; ++q) {
        pa += wa[q];
        pb += wb[q];
    }
    int ea = pa + ia - sa, eb = pb + ib - sb;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (base + i < n) {
            oa[base + i] = ea;
            ob[base + i] = eb;
        }
        ea += va[i];
        eb += vb[i];
    }
    if (threadIdx.x == kBlock - 1) {
        tot[2 * blockIdx.x] = pa + ia;
        tot[2 * blockIdx.x + 1] = pb + ib;
    }
}

// carry_in (nullable): totals of everything scanned before this range (a sequence scanned in
// consecutive ranges); carry_out (nullable): grand total including the carry.
__global__ __launch_bounds__(kBlock) void k_scan_fix(int* __restrict__ oa, int* __restrict__ ob,
                                                     const int* __restrict__ tot, int n, int ntiles,
                                                     const int* __restrict__ carry_in = nullptr, int* __restrict__ carry_out = nullptr,
                                                     const int* __restrict__ abort = nullptr) {
    __shared__ int ra[kBlock], rb[kBlock];
    if (abort && *abort) return;
    // sum of the totals of all tiles before this one (and, for the last block, the grand total)
    const int mine = blockIdx.x;
    int sa = 0, sb = 0, ga = 0, gb = 0;
    for (int q = threadIdx.x; q < ntiles; q += kBlock) {
        const int ta = tot[2 * q], tb = tot[2 * q + 1];
        if (q < mine) {
            sa += ta;
            sb += tb;
        }
        ga += ta;
        gb += tb;
    }
    ra[threadIdx.x] = sa;
    rb[threadIdx.x] = sb;
    __syncthreads();
    for (int d = kBlock / 2; d > 0; d >>= 1) {
        if (threadIdx.x < d) {
            ra[threadIdx.x] += ra[threadIdx.x + d];
            rb[threadIdx.x] += rb[threadIdx.x + d];
        }
        __syncthreads();
    }
    const int ca = carry_in ? carry_in[0] : 0, cb = carry_in ? carry_in[1] : 0;
    const int offa = ra[0] + ca, offb = rb[0] + cb;
    const int base = blockIdx.x * kScanTile + threadIdx.x * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (base + i < n) {
            oa[base + i] += offa;
            ob[base + i] += offb;
        }
    if (mine == ntiles - 1) {
        __syncthreads();
        ra[threadIdx.x] = ga;
        rb[threadIdx.x] = gb;
        __syncthreads();
        for (int d = kBlock / 2; d > 0; d >>= 1) {
            if (threadIdx.x < d) {
                ra[threadIdx.x] += ra[threadIdx.x + d];
                rb[threadIdx.x] += rb[threadIdx.x + d];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            oa[n] = ra[0] + ca;
            ob[n] = rb[0] + cb;
            if (carry_out) {
                carry_out[0] = ra[0] + ca;
                carry_out[1] = rb[0] + cb;
            }
        }
    }
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

// Value of lane `l` (wave-uniform index) in every lane: v_readlane, no LDS round trip.
__device__ __forceinline__ int lane_bcast(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ double lane_bcast(double v, int l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), l);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// ---------------------------------------------------------------------------------------
// filtrar_z (reference scripts/ICM_SLAM_tools.py:22-58).
// One wave per scan.  WRITE=false decides which beams are kept (one 64-bit mask per 64 in-range beams, kmask) and
// counts them, WRITE=true stores them at boff[t] from the masks.
// LDS per wave: the in-range beams (index, range, x, y), B entries each.
// The isolation test `min(100, nearest other in-range beam) <= thr` is an EXISTENCE test: some other beam at a
// non-zero squared distance s with sqrt(s) <= thr, i.e. s <= thr2 (the largest double whose rounded sqrt is <= thr:
// the association's gate) -- or thr >= 100.  No minimum, no square root, any order.  Round 3: the beams next to a beam
// in the list (next in bearing) are tried first, sixteen of them, which settles nearly every beam that stays; the few
// left over -- the isolated ones, which need every other beam to be ruled out -- are then taken one at a time with the
// other beams spread over the wave's lanes (64 candidates per step instead of one).  ~1500 -> ~430 steps of the pair
// test per scan, and the second pass repeats none of it: 5.9 -> 1.x ms per 100 000 scans of 720 beams.
// ---------------------------------------------------------------------------------------
template <bool WRITE>
__global__ __launch_bounds__(kBlock) void k_prefilter(const double* __restrict__ ranges,
                                                      const double* __restrict__ cosb,
                                                      const double* __restrict__ sinb, int nloc, int B,
                                                      double rmax, double thr, int* __restrict__ nkept,
                                                      const int* __restrict__ boff, int* __restrict__ bk,
                                                      double* __restrict__ bd, double* __restrict__ bx,
                                                      double* __restrict__ by, double* __restrict__ pose_s2,
                                                      double2* __restrict__ bxy, double thr2,
                                                      unsigned long long* __restrict__ kmask) {
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
        if (WRITE && lane == 0) pose_s2[3 * (size_t)t] = pose_s2[3 * (size_t)t + 1] = pose_s2[3 * (size_t)t + 2] = 0.0;
        return;
    }
    // isolated-beam rejection
    int kept = 0;
    double sxx = 0.0, sxy = 0.0, syy = 0.0;  // sum of b b^T over the kept beams (pose constant)
    const int nchunk = (B + kWave - 1) / kWave;
    const bool all = 100.0 <= thr;           // (the reference caps the distance at 100)
    for (int base = 0; base < cnt; base += kWave) {
        const int i = base + lane;
        unsigned long long mask;
        if (!WRITE) {
            const bool live = i < cnt;
            const double xi = live ? lpx[i] : 0.0, yi = live ? lpy[i] : 0.0;
            bool found = all;
            // 1. the beams next to it in the list
            constexpr int kNear = 8;
#pragma unroll
            for (int k = 1; k <= kNear; ++k) {
                const int ja = i - k, jb = i + k;
                if (live && ja >= 0) {
                    const double dx = xi - lpx[ja], dy = yi - lpy[ja];
                    const double q = dx * dx + dy * dy;
                    found |= (q != 0.0) & (q <= thr2);
                }
                if (live && jb < cnt) {
                    const double dx = xi - lpx[jb], dy = yi - lpy[jb];
                    const double q = dx * dx + dy * dy;
                    found |= (q != 0.0) & (q <= thr2);
                }
            }
            mask = __ballot(live && found);
            // 2. the beams still without a neighbour, one at a time against all the others, 64 per step
            unsigned long long open = __ballot(live && !found);
            while (open != 0ull) {
                const int l = (int)__builtin_ctzll(open);
                open &= open - 1ull;
                const double xl = lane_bcast(xi, l), yl = lane_bcast(yi, l);
                bool hit = false;
                for (int jb = 0; jb < cnt && !hit; jb += kWave) {
                    const int j = jb + lane;
                    bool h = false;
                    if (j < cnt) {
                        const double dx = xl - lpx[j], dy = yl - lpy[j];
                        const double q = dx * dx + dy * dy;
                        h = (q != 0.0) & (q <= thr2);
                    }
                    hit = __ballot(h) != 0ull;
                }
                if (hit) mask |= 1ull << l;
            }
            if (lane == 0) kmask[(size_t)t * nchunk + (base >> 6)] = mask;
        } else {
            mask = kmask[(size_t)t * nchunk + (base >> 6)];
        }
        const bool keep = (mask >> lane) & 1ull;
        if (WRITE && keep) {
            const int p = boff[t] + kept + prefix_count(mask, lane);
            bk[p] = lk[i];
            bd[p] = lm[i];
            bx[p] = lpx[i];
            by[p] = lpy[i];
            if (bxy) bxy[p] = make_double2(lpx[i], lpy[i]);   // (phase A's copy: one 16-byte load per beam, see k_assoc_group)
            sxx += lpx[i] * lpx[i];
            sxy += lpx[i] * lpy[i];
            syy += lpy[i] * lpy[i];
        }
        kept += __popcll(mask);
    }
    if (!WRITE && lane == 0) nkept[t] = kept;
    if (WRITE) {
        sxx = wave_sum(sxx);
        sxy = wave_sum(sxy);
        syy = wave_sum(syy);
        if (lane == 0) {
            pose_s2[3 * (size_t)t] = sxx;
            pose_s2[3 * (size_t)t + 1] = sxy;
            pose_s2[3 * (size_t)t + 2] = syy;
        }
    }
}

// ---------------------------------------------------------------------------------------
// Phase A: project the kept beams with the previous-sweep pose (tras_rot_z, reference
// scripts/ICM_SLAM_tools.py:465-480) and associate each to the nearest landmark of
// mapa_viejo under the distance gate (Mapa.actualizar, scripts/ICM_SLAM_tools.py:168-172).
// The landmark table is searched through a uniform grid (cell >= dist_thr): 3 rows of 3
// cells, each row one contiguous range of the cell-sorted table.  label = column index of
// the nearest landmark (first index on ties, like np.argmin), -1 if farther than dist_thr.
// ---------------------------------------------------------------------------------------
struct LmRec {      // one landmark of the cell-sorted table (32 B: one aligned gather)
    double x, y;
    int id, pad0, pad1, pad2;
};

struct GridParams {  // device resident: written by the host (icm_set_state) or by the k_fl_* chain
    double gx0, gy0, inv;
    int nx, ny;
};

// Everything a beam landing in one cell can be matched to: the landmarks of the cell's 3x3
// neighbourhood, inline, in the order the range walk below visits them.  One aligned 128-byte
// record = ONE memory round trip per beam instead of the dependent chain cell_start ->
// ranges -> records.  Unused slots hold x = +inf (squared distance +inf, never the nearest), y = 0, id = -1;
// n > kNeighCap sends the beam down the range walk.
// Laid out for the common case (round 3: k_assoc_group's time follows the NUMBER of vector loads a beam issues -- four
// instead of six: -13 % -- far more than their bytes or the arithmetic behind them): the first two candidates, both ids
// and the count are three 16-byte loads out of the first 48 bytes; a neighbourhood holds one or two landmarks in
// 93 % of the cells of S2's map, three in 5 %, four in 1 %.
#ifndef ICM_NEIGH_FAST
#define ICM_NEIGH_FAST 2
#endif
constexpr int kNeighCap = 4;
constexpr int kNeighFast = ICM_NEIGH_FAST;   // candidates every beam reads (2 or 3); the rest of a record only where a neighbourhood has more
struct __attribute__((aligned(128))) NeighRec {
    double x0, y0;         //  0
    double x1, y1;         // 16
    int id0, id1, n, id2;  // 32
    double x2, y2;         // 48
    double x3, y3;         // 64
    int id3;               // 80
    int pad[11];
};
static_assert(sizeof(NeighRec) == 128, "NeighRec is one 128-byte line");

struct GridView {
    const GridParams* __restrict__ par;
    const int* __restrict__ cell_start;
    const LmRec* __restrict__ lm;
    const NeighRec* __restrict__ nb;
};

__device__ __forceinline__ int grid_cell(double v, double g0, double inv, int n) {
    // floor + clamp to [0, n-1] in integer arithmetic: the conversion truncates toward zero (so
    // (-1, 0) -> 0 like the clamp would), saturates for huge values and maps NaN to 0
    const int c = (int)((v - g0) * inv);
    return min(max(c, 0), n - 1);
}

__device__ __forceinline__ void pose_of(const double* __restrict__ x, const double* __restrict__ x0, int tg,
                                        double& px, double& py, double& th) {
    if (tg == 0) {  // scan 0 is projected with self.x0 (scripts/ICM_ROS.py:125,137)
        px = x0[0]; py = x0[1]; th = x0[2];
    } else {
        px = x[3 * (size_t)tg]; py = x[3 * (size_t)tg + 1]; th = x[3 * (size_t)tg + 2];
    }
}

// cos / sin of (theta - pi/2): the rotation tras_rot_z applies (scripts/ICM_SLAM_tools.py:476-479).
// One argument reduction for both (the same one every kernel uses, so all of them rotate a
// pose's beams with identical coefficients).
__device__ __forceinline__ void pose_rot(double th, double& ct, double& st) {
    double s_, c_;
    sincos(th - kHalfPi, &s_, &c_);
    ct = c_;
    st = s_;
}

// (cos, sin)(theta - pi/2) of every pose of the shard, once per sweep: phase A and the moment kernel
// read them instead of every wavefront re-deriving its pose's pair (a generic FP64 sincos is ~150
// vector instructions, 15 % of a k_assoc_group wave).  rot[2 tl], rot[2 tl + 1].
// cs (nullable, indexed by the GLOBAL pose): (cos, sin)(theta) of x[:, t] itself -- what the pose solves need of a
// neighbour's and of the pose's own previous heading (make_ctx, make_fold); kept by whoever writes a pose, like rot.
__global__ __launch_bounds__(kBlock) void k_pose_rot(const double* __restrict__ x, const double* __restrict__ x0, int t_begin,
                                                     int nloc, double* __restrict__ rot, double* __restrict__ cs = nullptr) {
    const int tl = blockIdx.x * kBlock + threadIdx.x;
    if (tl >= nloc) return;
    double px, py, th, ct, st;
    pose_of(x, x0, t_begin + tl, px, py, th);
    pose_rot(th, ct, st);
    rot[2 * (size_t)tl] = ct;
    rot[2 * (size_t)tl + 1] = st;
    if (cs) {
        const double t0 = x[3 * (size_t)(t_begin + tl) + 2];
        cs[2 * (size_t)(t_begin + tl)] = cos(t0);
        cs[2 * (size_t)(t_begin + tl) + 1] = sin(t0);
    }
}

// (cos, sin) of the odometry headings, once per sequence (they are the sequence's constants; every pose solve used to
// form four of them with the generic cos / sin: ~500 vector instructions at the head of every lane's serial chain)
__global__ __launch_bounds__(kBlock) void k_odo_trig(const double* __restrict__ odo, int T, double* __restrict__ out) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= T) return;
    const double th = odo[2 * (size_t)T + t];
    out[2 * (size_t)t] = cos(th);
    out[2 * (size_t)t + 1] = sin(th);
}

// v = mask bit of the lane ? if_set : if_clear, the mask in a scalar register pair (one v_cndmask, no per-lane boolean)
__device__ __forceinline__ int mask_select(unsigned long long m, int if_set, int if_clear) {
    int r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
    return r;
}
__device__ __forceinline__ int mask_select_or_minus1(unsigned long long m, int if_set) {
    int r;
    asm("v_cndmask_b32_e64 %0, -1, %1, %2" : "=v"(r) : "v"(if_set), "s"(m));
    return r;
}

// Gated nearest landmark of the world point (wx, wy).  The three cell rows around the point
// are three contiguous ranges of the cell-sorted table; they are walked as ONE loop so the
// wave iterates max-over-lanes of the candidate COUNT (about 1.5 on average), not three
// times max-over-lanes per row.  Candidates are ranked by squared distance s.  The reference
// ranks d = sqrt(s) (cdist) with the first index winning ties and gates on d > dist_thr:
//   * sqrt is monotone, so the ranking can only differ when the two smallest s agree to a few
//     ulps; that case (detected from the runner-up) is re-ranked exactly on sqrt;
//   * d > thr  <=>  s > thr2, with thr2 = the largest double whose correctly rounded sqrt is
//     <= thr (computed on the host), so the gate needs no sqrt either.
__device__ __noinline__ int assoc_grid_walk(const GridView& g, const GridParams& gp, int cx, int cy, double wx, double wy,
                                            double thr, double thr2) {
    const int c0 = max(cx - 1, 0), c1 = min(cx + 1, gp.nx - 1);
    const int r0 = max(cy - 1, 0), r2 = min(cy + 1, gp.ny - 1);
    // rows r0, cy, r2 (clamped rows may coincide: count each distinct row once)
    const int pa = g.cell_start[r0 * gp.nx + c0], na = g.cell_start[r0 * gp.nx + c1 + 1] - pa;
    const int pb = g.cell_start[cy * gp.nx + c0], nb = cy != r0 ? g.cell_start[cy * gp.nx + c1 + 1] - pb : 0;
    const int pc = g.cell_start[r2 * gp.nx + c0], nc = r2 != cy ? g.cell_start[r2 * gp.nx + c1 + 1] - pc : 0;
    const int n = na + nb + nc;
    double best = __builtin_huge_val(), second = __builtin_huge_val();
    int bid = -1;
    for (int i = 0; i < n; ++i) {
        const int p = i < na ? pa + i : (i < na + nb ? pb + (i - na) : pc + (i - na - nb));
        const LmRec c = g.lm[p];
        const double dx = c.x - wx, dy = c.y - wy;
        const double s = dx * dx + dy * dy;
        if (s < best) {
            second = best;
            best = s;
            bid = c.id;
        } else {
            second = fmin(second, s);
        }
    }
    if (second <= best * (1.0 + 1e-15)) {  // (near) tie: the reference's exact rule on sqrt
        double db = __builtin_huge_val();
        bid = -1;
        for (int i = 0; i < n; ++i) {
            const int p = i < na ? pa + i : (i < na + nb ? pb + (i - na) : pc + (i - na - nb));
            const LmRec c = g.lm[p];
            const double dx = c.x - wx, dy = c.y - wy;
            const double d = sqrt(dx * dx + dy * dy);
            if (d < db || (d == db && c.id < bid)) {
                db = d;
                bid = c.id;
            }
        }
        return (bid >= 0 && !(db > thr)) ? bid : -1;
    }
    return (bid >= 0 && !(best > thr2)) ? bid : -1;
}

// The same search through the cell's inline neighbourhood record (the common case: at most
// kNeighCap candidates).  Same candidates in the same order as the range walk, so the same
// winner; near ties and crowded cells take the walk.
__device__ __forceinline__ int assoc_grid(const GridView& g, const GridParams& gp, double wx, double wy, double thr,
                                          double thr2) {
    const int cx = grid_cell(wx, gp.gx0, gp.inv, gp.nx), cy = grid_cell(wy, gp.gy0, gp.inv, gp.ny);
    // (byte offset in 32 bits: the table has at most 2^25 cells, checked where it is sized -- a scalar base plus a
    // 32-bit lane offset instead of 64-bit address arithmetic per beam)
    const unsigned off = ((unsigned)cy * (unsigned)gp.nx + (unsigned)cx) << 7;
    const char* __restrict__ r = reinterpret_cast<const char*>(g.nb) + off;
    // the part of the record every beam reads: candidates 0 .. kNeighFast-1, the ids 0..2 and the count
    const double2 p0 = *reinterpret_cast<const double2*>(r), p1 = *reinterpret_cast<const double2*>(r + 16);
    const int4 ic = *reinterpret_cast<const int4*>(r + 32);   // id0, id1, n, id2
    double2 p2 = {__builtin_huge_val(), 0.0};
    if (kNeighFast >= 3) p2 = *reinterpret_cast<const double2*>(r + 48);
    const int n = ic.z;
    // squared distances to the candidates; empty slots hold x = +inf
    double dx = p0.x - wx, dy = p0.y - wy;
    const double s0 = dx * dx + dy * dy;
    dx = p1.x - wx; dy = p1.y - wy;
    const double s1 = dx * dx + dy * dy;
    // (kNeighFast == 2: what an empty slot gives a lane that does not read its slots 2 / 3 below -- (inf - wx)^2 + wy^2 is
    // +inf for every finite point, and a non-finite point ends with no label whichever of inf / NaN stands here)
    double s2 = __builtin_huge_val();
    if (kNeighFast >= 3) {
        dx = p2.x - wx; dy = p2.y - wy;
        s2 = dx * dx + dy * dy;
    }
    double s3 = s2;
    int id2 = kNeighFast >= 3 ? ic.w : -1, id3 = -1;
    // The rest of the record, for the lanes whose neighbourhood holds more candidates than that -- and only when the wave
    // has such a lane at all.  A lane that does not read its slots 2 / 3 uses what an empty slot holds (x = +inf, y = 0,
    // id = -1: its record's slots ARE empty), so every lane decides on exactly the four values the whole record gives.
    const bool more = n > kNeighFast;
    bool four = kNeighFast >= 3;   // s2 / s3 enter the decision below (always, when slot 2 is read by everybody)
    if (__builtin_expect(__ballot(more) != 0ull, 0)) {
        four = true;
        if (kNeighFast >= 3) {   // empty slot 3 of the lanes that stay out
            dx = __builtin_huge_val() - wx; dy = 0.0 - wy;
            s3 = dx * dx + dy * dy;
        }
        if (more) {
            if (kNeighFast < 3) {
                const double2 q2 = *reinterpret_cast<const double2*>(r + 48);
                dx = q2.x - wx; dy = q2.y - wy;
                s2 = dx * dx + dy * dy;
                id2 = ic.w;
            }
            const double2 q3 = *reinterpret_cast<const double2*>(r + 64);
            id3 = *reinterpret_cast<const int*>(r + 80);
            dx = q3.x - wx; dy = q3.y - wy;
            s3 = dx * dx + dy * dy;
        }
    } else if (kNeighFast >= 3) {
        dx = __builtin_huge_val() - wx; dy = 0.0 - wy;   // (slot 3 of a record with at most three candidates)
        s3 = dx * dx + dy * dy;
    }
    // the nearest one and whether a second candidate is within a relative 1e-15 of it (then the reference's
    // rule on the rounded sqrt decides: the range walk).  With no near tie exactly one c_i is set.
    // Two-slot form (no lane of the wave has a third candidate): what the four-slot form below gives when slots 2 and 3
    // are empty -- fmin skips a NaN, an infinite s2 is "within the limit" only when every candidate's distance is infinite
    // too, and then slot 0 or 1 is picked first either way.
    // The decision is lane-mask algebra in scalar registers (round 4: as per-lane booleans the compiler kept c0..c3 as 0 / 1
    // words in vector registers -- fifteen more vector instructions per 64 beams in a kernel whose vector pipe is the
    // busiest unit): one compare per slot writes a mask, the masks combine on the scalar unit, and the selects read them.
    typedef unsigned long long u64;
    double best = fmin(s0, s1);
    if (four) best = fmin(best, fmin(s2, s3));
    const double lim = best * (1.0 + 1e-15);
    const u64 m0 = __ballot(s0 <= lim), m1 = __ballot(s1 <= lim);
    u64 m2 = 0ull, m3 = 0ull;
    if (four) {
        m2 = __ballot(s2 <= lim);
        m3 = __ballot(s3 <= lim);
    }
    const u64 tie = (m0 & (m1 | m2 | m3)) | (m1 & (m2 | m3)) | (m2 & m3);
    int bid = mask_select(m2, id2, id3);
    bid = mask_select(m1, ic.y, bid);
    bid = mask_select(m0, ic.x, bid);
    const u64 ok = (m0 | m1 | m2 | m3) & __ballot(bid >= 0) & ~__ballot(best > thr2);   // (no m_i: a non-finite point)
    int lab = mask_select_or_minus1(ok, bid);
    const u64 walk = __ballot(n > kNeighCap) | (__ballot(n != 0) & tie);
    if (__builtin_expect(walk != 0ull, 0)) {
        if ((walk >> lane_id()) & 1ull) lab = assoc_grid_walk(g, gp, cx, cy, wx, wy, thr, thr2);
    }
    return n == 0 ? -1 : lab;
}

// Fills the neighbourhood records of every cell of the current grid (one thread per cell;
// launched with the cell CAPACITY because the grid's size may only be known on the device).
__global__ __launch_bounds__(kBlock) void k_neigh_table(GridView g, NeighRec* __restrict__ out, int max_cells,
                                                        const int* __restrict__ abort = nullptr) {
    if (abort && *abort) return;
    const GridParams gp = *g.par;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= gp.nx * gp.ny || c >= max_cells) return;
    const int cy = c / gp.nx, cx = c - cy * gp.nx;
    const int c0 = max(cx - 1, 0), c1 = min(cx + 1, gp.nx - 1);
    const int r0 = max(cy - 1, 0), r2 = min(cy + 1, gp.ny - 1);
    const int pa = g.cell_start[r0 * gp.nx + c0], na = g.cell_start[r0 * gp.nx + c1 + 1] - pa;
    const int pb = g.cell_start[cy * gp.nx + c0], nb = cy != r0 ? g.cell_start[cy * gp.nx + c1 + 1] - pb : 0;
    const int pc = g.cell_start[r2 * gp.nx + c0], nc = r2 != cy ? g.cell_start[r2 * gp.nx + c1 + 1] - pc : 0;
    const int n = na + nb + nc;
    double x[kNeighCap], y[kNeighCap];
    int id[kNeighCap];
#pragma unroll
    for (int i = 0; i < kNeighCap; ++i) {
        x[i] = __builtin_huge_val();
        y[i] = 0.0;
        id[i] = -1;
    }
#pragma unroll
    for (int i = 0; i < kNeighCap; ++i) {  // (a fixed-trip loop: the record stays in registers)
        if (i < n) {
            const int p = i < na ? pa + i : (i < na + nb ? pb + (i - na) : pc + (i - na - nb));
            const LmRec q = g.lm[p];
            x[i] = q.x;
            y[i] = q.y;
            id[i] = q.id;
        }
    }
    double4* o = reinterpret_cast<double4*>(out + c);   // (the NeighRec layout, in 32-byte stores)
    o[0] = make_double4(x[0], y[0], x[1], y[1]);
    int4* oi = reinterpret_cast<int4*>(&out[c].id0);
    oi[0] = make_int4(id[0], id[1], n, id[2]);
    double2* o2 = reinterpret_cast<double2*>(&out[c].x2);
    o2[0] = make_double2(x[2], y[2]);
    o2[1] = make_double2(x[3], y[3]);
    out[c].id3 = id[3];
}

// Brute-force form of the same association (all K landmarks, table tiled through LDS): the
// literal cdist/argmin of the reference.  Writes labels only; used to cross-check the grid
// search on the GPU (k_assoc_group<PRELABEL> then consumes the labels).
__global__ __launch_bounds__(kBlock) void k_associate_brute(const double* __restrict__ x, const double* __restrict__ x0,
                                                            int t_begin, int nloc, const int* __restrict__ boff,
                                                            const double* __restrict__ bx, const double* __restrict__ by,
                                                            const double* __restrict__ mapx, const double* __restrict__ mapy,
                                                            int K, double thr, int* __restrict__ label) {
    constexpr int TILE = 1024;
    __shared__ double sx[TILE], sy[TILE];
    __shared__ int s_maxit;
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
    double ct, st;
    pose_rot(th, ct, st);
    const int iters = (j1 - j0 + kWave - 1) / kWave;
    if (threadIdx.x == 0) s_maxit = 0;
    __syncthreads();
    atomicMax(&s_maxit, iters);  // block-uniform trip count: every wave reaches the barriers
    __syncthreads();
    const int maxit = s_maxit;
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

// One scan against a reference map, on its own (Mapa.actualizar's association step,
// scripts/ICM_SLAM_tools.py:169-172, outside a sweep): obs (n,2) row-major world points, the
// first K columns of the reference map; the literal cdist / argmin (first index on ties) / gate.
// One thread per observation, landmark table tiled through LDS.
__global__ __launch_bounds__(kBlock) void k_scan_labels(const double* __restrict__ obs, int n,
                                                        const double* __restrict__ mapx, const double* __restrict__ mapy,
                                                        int K, double thr, int* __restrict__ label) {
    constexpr int TILE = 1024;
    __shared__ double sx[TILE], sy[TILE];
    const int j = blockIdx.x * kBlock + threadIdx.x;
    const bool on = j < n;
    double wx = 0.0, wy = 0.0;
    if (on) {
        wx = obs[2 * (size_t)j];
        wy = obs[2 * (size_t)j + 1];
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

// ---------------------------------------------------------------------------------------
// Fused phase A: association + per-pose grouping.  For every distinct label of the scan
// (label -1 = the scan's gated-out beams, which the reference folds into ONE new landmark,
// SURVEY Appendix B.1) one ENTRY: the beam count k and the sums of the body-frame points --
// the sufficient statistics of both the running-mean update (scripts/ICM_SLAM_tools.py:184-195:
// sum of world points = k p + R sum b) and of the pose energy (icm_device.hpp, Items).
//
// Beams arrive sorted by bearing, so equal labels form runs: a wave-level segmented scan
// reduces each run, and the run tails fold their totals into an LDS hash table keyed
// by label (linear probing, claimed with ds_cmpst).  Deterministic: runs are folded chunk
// by chunk, and two runs of one label inside a chunk are folded in lane order.
// Entries are staged at the front of the pose's beam range, in slot order.
//   PRELABEL: labels come from `label` (brute-force cross-check) instead of the grid search
//   DEBUG:    also write label[] and the beam -> entry map bloc[]
// ---------------------------------------------------------------------------------------
// ds_add_f64 without a return value (gfx90a and later): an LDS double accumulates in place
__device__ __forceinline__ void lds_add_f64(double* p, double v) {
    typedef __attribute__((address_space(3))) double lds_double;
    (void)__builtin_amdgcn_ds_atomic_fadd_f64((lds_double*)p, v);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, false);
}
// One step of the segmented inclusive scan of two doubles: fold in the partial sums of the
// source lane when `take`.  Lanes without a source (and masked rows) read zeros.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void seg_step(bool take, double& ax, double& ay) {
    const double xu = dpp_mov<CTRL, ROW_MASK>(ax), yu = dpp_mov<CTRL, ROW_MASK>(ay);
    if (take) {
        ax += xu;
        ay += yu;
    }
}

// The same step across rows (row_bcast): `take` is false in every lane the move does not write, whose operand is
// then whatever the register held.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void seg_step_rows(bool take, double& ax, double& ay) {
    const int xl = __builtin_amdgcn_mov_dpp(__double2loint(ax), CTRL, ROW_MASK, 0xF, false);
    const int xh = __builtin_amdgcn_mov_dpp(__double2hiint(ax), CTRL, ROW_MASK, 0xF, false);
    const int yl = __builtin_amdgcn_mov_dpp(__double2loint(ay), CTRL, ROW_MASK, 0xF, false);
    const int yh = __builtin_amdgcn_mov_dpp(__double2hiint(ay), CTRL, ROW_MASK, 0xF, false);
    if (take) {
        ax += __hiloint2double(xh, xl);
        ay += __hiloint2double(yh, yl);
    }
}

// Where a pose's staged entries go.  Dense, without a scan in front and without an atomic: the PREVIOUS sweep's
// exclusive scan of the entry counts (still in ent_off when phase A runs) is this sweep's reservation plan -- pose t owns
// [E_prev[t] + kStageSlack t, E_prev[t+1] + kStageSlack (t+1)) of the packed area, room for its last count plus a slack.
// The plan is disjoint whatever the array holds (stale after a restore, zero before the first sweep: any non-decreasing
// array works), so a pose whose entries fit takes its reserved place and one whose do not takes the front of its own beam
// range in the sparse area behind the packed one -- always correct, dense once the counts have settled (on S2 every pose
// from the second sweep on).  The readers stream a pose's ~37 entries from consecutive memory instead of from the front of
// every 230-beam range.
// (kStageSlack and the sizes of the staging area: staging_layout, icm_host.hpp)

// HS = hash slots per pose; at most 3/4 of them may be used (distinct landmarks of one scan).
// HS = 128 keeps the kernel at 14 KB of LDS and under 64 VGPRs; a scan that overflows it makes the host relaunch the
// sweep's phase A with HS = 256.  Its waves-per-EU attribute is (7, 8), not (8, 8): the compiler's scalar-register
// budget at eight waves is 80 (800 per SIMD less the trap handler's 16 per wave), which spilled 44 scalars into vector
// lanes, each read back by a vector instruction in the batch loop; at seven it is 96 -- no spill reads to speak of, and
// the hardware then keeps seven waves per SIMD resident, which costs nothing: the kernel saturates at about seven
// (profiles/r04_assoc_occupancy_scaling.txt, r04_assoc_wave_life_experiments.txt).

template <int HS>
struct PoseTable {
    int key[HS];
    int cnt[HS];
    double sx[HS];
    double sy[HS];
    int owner[HS];
};

// One wave per pose, four poses per 256-thread workgroup.  (Measured and dropped, DESIGN.md appendix: several consecutive
// poses per wave with the next pose's header and first beams in flight -- 0.192 / 0.202 / 0.217 ms at 1 / 2 / 4 poses per
// wave; persistent waves striding over the poses -- +19 % .. +51 %; one-wave workgroups -- no difference.)
#ifdef ICM_ASSOC_TS   // measurement builds only (tools/assoc_timeline.py): shader-clock stamps of every 16th pose's wave
__device__ unsigned long long g_assoc_ts[8 * 8192];
#define ASSOC_TS(slot) do { if (lane == 0 && (tl & 15) == 0 && (tl >> 4) < 8192) g_assoc_ts[8 * (tl >> 4) + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define ASSOC_TS_VAL(slot, v) do { if (lane == 0 && (tl & 15) == 0 && (tl >> 4) < 8192) g_assoc_ts[8 * (tl >> 4) + (slot)] = (unsigned long long)(v); } while (0)
#else
#define ASSOC_TS(slot) do { } while (0)
#define ASSOC_TS_VAL(slot, v) do { } while (0)
#endif
template <bool PRELABEL, bool DEBUG, int HS>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(HS == 128 ? 7 : 4, HS == 128 ? 8 : 5)))
void k_assoc_group(const double* __restrict__ x, const int* __restrict__ boff, const double2* __restrict__ bxy,
                                                        const double* __restrict__ rot, const GridParams* __restrict__ gpar,
                                                        const int* __restrict__ plan, int nloc, int t_begin,
                                                        // ^ the fourteen dwords every wave needs at once: they arrive in scalar registers
                                                        //   with the wave (kernel-argument preload, Makefile), not by a load of its own
                                                        const double* __restrict__ x0,
                                                        GridView g, double thr, double thr2, int* __restrict__ label,
                                                        int* __restrict__ bloc, int* __restrict__ st_label,
                                                        unsigned short* __restrict__ st_k, double* __restrict__ st_sbx,
                                                        double* __restrict__ st_sby, int* __restrict__ nent_out,
                                                        int* __restrict__ isnew_out, int* __restrict__ flags,
                                                        int nnz_total = 0, int* __restrict__ st_off = nullptr,
                                                        int pose0 = 0, int sparse0 = 0) {
    constexpr int kHash = HS, kGroupCap = HS * 3 / 4;
    constexpr int kHashShift = HS == 128 ? 25 : 24;
    __shared__ PoseTable<HS> tables[kWavesPerBlock];
    const int lane = lane_id();
    const int tl = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + wave_in_block());
    if (tl >= nloc) return;
    ASSOC_TS(0);
    PoseTable<HS>& T = tables[wave_in_block()];
    const GridParams gp = *gpar;   // (= *g.par)
    // (explicit 32-bit byte offsets from a scalar base: a pose has far fewer than 2^29 beams)
    // (a beam's body-frame point is ONE 16-byte load from the interleaved copy the pre-filter leaves beside its two
    // arrays: this kernel's time follows the number of vector loads it issues)
    auto beam_at = [](const double2* __restrict__ base, unsigned idx) {
        return *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(base) + (idx << 4));
    };
    // (wave-uniform values through scalar registers: the pose's beam range becomes a scalar base pointer plus a
    // 32-bit lane offset)
    const int j0 = __builtin_amdgcn_readfirstlane(boff[tl]), j1 = __builtin_amdgcn_readfirstlane(boff[tl + 1]);
    // this pose's reserved place (scalar loads, in flight while the beams are grouped)
    const int plan0 = __builtin_amdgcn_readfirstlane(plan[tl]), plan1 = __builtin_amdgcn_readfirstlane(plan[tl + 1]);
    // (cos, sin)(theta - pi/2) from the sweep's table (k_pose_rot / whoever wrote the pose): computed once per pose and sweep
    const double ct = rot[2 * (size_t)tl], st = rot[2 * (size_t)tl + 1];
    double px, py, th;
    pose_of(x, x0, t_begin + tl, px, py, th);
    if (j0 == j1) {
        if (lane == 0) {
            nent_out[tl] = 0;
            isnew_out[tl] = 0;
            st_off[tl] = 0;
        }
        return;
    }
    double nbx, nby;   // the pose's first 64 beams
    {
        const unsigned i0 = min((unsigned)lane, (unsigned)(j1 - j0) - 1u);
        const double2 f = beam_at(bxy + j0, i0);
        nbx = f.x;
        nby = f.y;
    }
    const double2* __restrict__ bxyp = bxy + j0;
    const unsigned nbeam = (unsigned)(j1 - j0);
    for (int s = lane; s < kHash; s += kWave) {
        T.key[s] = kEmpty;
        T.cnt[s] = 0;
        T.sx[s] = 0.0;
        T.sy[s] = 0.0;
    }
    int nent = 0;
    bool overflow = false;
    __builtin_amdgcn_wave_barrier();
#ifdef ICM_ASSOC_TS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ASSOC_TS(1);   // header scalars are in
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ASSOC_TS(2);   // first beams are in
    int nbatch = 0;
#endif
    // body points of the next 64 beams are requested one iteration ahead (unconditional loads at a clamped
    // index, so that the wait in front of their use counts exactly them): a pose's batches no longer pay the
    // latency of this load and of the grid record's one after the other
    for (int base = j0; base < j1 && !overflow; base += kWave) {
        const int j = base + lane;
        const bool valid = j < j1;
        const int cn = min(kWave, j1 - base);
        int lab = -2;
        const double bxx = nbx, byy = nby;   // (lanes beyond the pose's last beam hold a copy of it: never a head or a tail, read by nobody)
        {
            const unsigned on = min((unsigned)(j - j0) + (unsigned)kWave, nbeam - 1u);
            const double2 nb = beam_at(bxyp, on);
            nbx = nb.x;
            nby = nb.y;
        }
        if (valid) {
            if (PRELABEL) {
                lab = label[j];
            } else {
                const double wx = (bxx * ct - byy * st) + px;
                const double wy = (bxx * st + byy * ct) + py;
                lab = assoc_grid(g, gp, wx, wy, thr, thr2);
                if (DEBUG) label[j] = lab;
            }
        }
        // runs of equal labels.  dist = distance of the lane to the head of its run (from
        // the ballot of the head flags); the segmented inclusive scan of (bx, by) then adds
        // the partial of lane - d exactly when dist >= d (DPP row shifts inside each 16-lane
        // row, then row_bcast:15 / :31 across rows); the run's beam count is dist + 1.
        const int prev = dpp_mov_i<0x138, 0xF>(lab);  // wave_shr:1
        const bool head = valid && (lane == 0 || prev != lab);
        const unsigned long long hm = __ballot(head) & ((2ull << lane) - 1ull);
        const int dist = lane - (63 - (int)__builtin_clzll(hm | 1ull));
        const int c = dist + 1;
        const int nexthead = dpp_mov_i<0x130, 0xF>(head ? 1 : 0);   // wave_shl:1 (lane 63 reads 0): no LDS round trip
        bool tail = valid && (lane == cn - 1 || nexthead);
        // Run tails claim / find the slot of their label in the pose's LDS table.  The compare-and-swap on the label's
        // home slot goes out NOW, in front of the segmented scan (vector / DPP work only), whose instructions cover its
        // round trip; it either claims the free slot or reports who holds it.
        int slot = (int)(((unsigned)lab * 2654435761u) >> kHashShift);
        int held = kEmpty;
        if (tail) held = atomicCAS(&T.key[slot], kEmpty, lab);
        // Segmented inclusive scan of (bx, by) over the runs, over the distance to the run head: at step d a lane at least
        // d beams into its run adds the partial of lane - d; the partials move through DPP row shifts / row broadcasts.
        // (Round 3 measured the same scan with the partials travelling through LDS -- one 16-byte write and read per lane
        // and step, 18 % fewer vector instructions per wave: 0.195 against 0.188 ms, SLOWER)
        double ax = bxx, ay = byy;
        // (DPP row shifts inside each 16-lane row, then row_bcast:15 / :31 across rows; the run's beam count is dist + 1)
        seg_step<0x111, 0xF>(dist >= 1, ax, ay);
        seg_step<0x112, 0xF>(dist >= 2, ax, ay);
        seg_step<0x114, 0xF>(dist >= 4, ax, ay);
        if (__ballot(dist >= 8) != 0ull) seg_step<0x118, 0xF>(dist >= 8, ax, ay);   // (wave-uniform skip)
        // across rows: rows 1 and 3 take lane 15 / 47 (row_bcast:15), rows 2 and 3 take lane 31 (row_bcast:31); the
        // predicate names the receiving rows itself, so the lanes the DPP move leaves alone need no preset zero
        seg_step_rows<0x142, 0xA>(((lane & 16) != 0) & (dist > (lane & 15)), ax, ay);   // run began in an earlier row
        seg_step_rows<0x143, 0xC>((lane >= 32) & (dist > (lane & 31)), ax, ay);          // run began in rows 0-1
        // ... the slot search goes on only where the home slot holds another label (linear probing; a probe IS the
        // compare-and-swap: one LDS round trip per step instead of a read and then a swap)
        bool inserted = false, found = false;
        if (tail) {
            // bounded: the table is checked against its 3/4 budget only between chunks, and one
            // chunk can bring up to 64 new labels -- a full table must end the probe, not spin
            for (int probes = 0; probes < kHash; ++probes) {
                if (held == kEmpty || held == lab) {
                    inserted = held == kEmpty;
                    found = true;
                    break;
                }
                slot = (slot + 1) & (kHash - 1);
                held = atomicCAS(&T.key[slot], kEmpty, lab);
            }
        }
        if (__ballot(tail && !found) != 0ull) overflow = true;   // more distinct landmarks than slots
        tail = tail && found;
        nent += __popcll(__ballot(inserted));
        // The run's totals are ADDED to the slot by LDS atomics (ds_add_u32 / ds_add_f64, no return value): no
        // read-modify-write round trip, and two runs of one label inside a batch (a landmark seen left and right of an
        // occluder) need no arbitration -- the LDS serialises the two additions itself, in one fixed order.
        if (tail) {
            atomicAdd(&T.cnt[slot], c);
            lds_add_f64(&T.sx[slot], ax);
            lds_add_f64(&T.sy[slot], ay);
        }
        if (DEBUG) {  // every beam learns the slot of its run (from the run's tail)
            const unsigned long long tm = __ballot(tail);
            const int mytail = lane + (int)__builtin_ctzll((tm >> lane) | (1ull << 63));
            const int sl = __shfl(slot, mytail & (kWave - 1), kWave);
            if (valid) bloc[j] = sl;
        }
        if (nent > kGroupCap) overflow = true;
        __builtin_amdgcn_wave_barrier();
#ifdef ICM_ASSOC_TS
        ++nbatch;
        if (nbatch == 1) ASSOC_TS(3);
        if (nbatch == 2) ASSOC_TS(4);
        ASSOC_TS(5);
#endif
    }
    // compact the used slots into the pose's place, slot order: its reserved one if the entries fit, else the front
    // of its own beam range in the sparse area
    const int room = (plan1 - plan0) + kStageSlack;
    const bool fits = plan0 >= 0 && plan1 >= plan0 && plan1 <= nnz_total && nent <= room;   // (a stale plan is still a plan; a wild one is not)
    const int sbase = fits ? plan0 + kStageSlack * (pose0 + tl) : sparse0 + j0;
    int written = 0;
    bool isnew = false;
    for (int s0 = 0; s0 < kHash; s0 += kWave) {
        const int s = s0 + lane;
        const int k = T.key[s];
        const bool occ = k != kEmpty;
        const unsigned long long mask = __ballot(occ);
        if (occ) {
            const unsigned q = (unsigned)(written + prefix_count(mask, lane));   // (scalar bases + 32-bit offsets, like the beam loads)
            *reinterpret_cast<int*>(reinterpret_cast<char*>(st_label + sbase) + (q << 2)) = k;
            *reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(st_k + sbase) + (q << 1)) = (unsigned short)T.cnt[s];   // (beams of one scan: <= B <= 8192)
            *reinterpret_cast<double*>(reinterpret_cast<char*>(st_sbx + sbase) + (q << 3)) = T.sx[s];
            *reinterpret_cast<double*>(reinterpret_cast<char*>(st_sby + sbase) + (q << 3)) = T.sy[s];
            isnew |= k == -1;
            if (DEBUG) T.owner[s] = q;
        }
        written += __popcll(mask);
    }
    const unsigned long long anynew = __ballot(isnew);
    if (lane == 0) {
        nent_out[tl] = written;
        st_off[tl] = sbase;
        isnew_out[tl] = anynew != 0ull;
        // poses outside their reserved place: the host refreshes the plan when many (not counted while there is no plan
        // at all -- the first sweep of a sequence: 100 000 atomics on one word)
        if (!fits && plan1 > 0) atomicAdd(&flags[3], 1);

        if (overflow) flags[0] = 1;
    }
#ifdef ICM_ASSOC_TS
    ASSOC_TS(6);
    ASSOC_TS_VAL(7, nbatch);
#endif
    if (DEBUG) {  // beam -> entry index within the pose
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        for (int j = j0 + lane; j < j1; j += kWave) bloc[j] = T.owner[bloc[j]];
    }
}

// ---------------------------------------------------------------------------------------
// Phase A by RUNS (round 5).  The cdist / argmin / gate of Mapa.actualizar (reference scripts/ICM_SLAM_tools.py:168-172)
// is a per-beam rule, but its outcome is almost always the same for every beam of a small cluster of neighbouring
// returns (one trunk): 6.3 beams share a label on S2.  A geometric RUN is a stretch of consecutive kept beams of one
// scan whose body-frame points lie close together -- a function of the scan alone, so it is cut ONCE per sequence, beside
// filtrar_z (k_run_build), with its sum of body points, beam count k and bounding circle (centre c = sum / k, radius r).  Per
// sweep ONE lane per run projects the centre with the pose's previous-sweep value (tras_rot_z, :465-480, is a rigid
// motion: every beam's world point stays within r of the centre's), reads ONE grid record and takes the nearest
// candidate i1 at distance d1 and the second nearest at d2.  With cell >= dist_thr the edge of the search grid:
//     d1 + r <= dist_thr            every beam is inside the gate of i1            (|w_j - y_i1| <= d1 + r)
//     d2 - r >  d1 + r              every other candidate of the record is farther from every beam than i1
//     d1 + 2r < cell                every landmark outside the record's 3x3 cells is farther than cell - r > d1 + r
// (each with a margin of 1e-4 dist_thr, five orders of magnitude above the rounding of a projected point) => argmin and
// gate of EVERY beam of the run are i1, exactly as the reference decides beam by beam, ties impossible.  A run the test
// does not settle (crowded or distant landmarks, gated-out beams, more than four candidates) goes beam by beam through
// assoc_grid, the rule of k_assoc_group.  A settled run then IS a partial entry: its cached (k, sum b) are added to the
// pose's label table -- no beam is read in the common case.  S2: 3.7 M lanes and grid records instead of 23.0 M.
// ---------------------------------------------------------------------------------------
constexpr int kRunCap = 64;        // beams per run at most (an unsettled run is one batch of the beam-by-beam path)
constexpr int kRunUndecided = -3;

// One thread per pose cuts its kept beams into runs: a new run starts where the next body point is farther than `gap`
// from the last one, farther than `ext` from the run's first point, or after kRunCap beams.  FILL = false counts the
// runs (nrun), FILL = true writes them at roff[t]: sum of the body points in beam order (the circle's centre is their mean,
// sum / k: formed again by k_assoc_runs with the same division, not stored -- 24 instead of 40 bytes per run through a
// kernel bound by its memory path, -3.5 %), radius = largest distance to the centre rounded UP into a float, k | first
// beam's offset within the pose << 16.
template <bool FILL>
__global__ __launch_bounds__(kBlock) void k_run_build(const int* __restrict__ boff, const double2* __restrict__ bxy, int nloc,
                                                      double gap2, double ext2, int* __restrict__ nrun,
                                                      const int* __restrict__ roff, double2* __restrict__ r_s,
                                                      uint2* __restrict__ r_m) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= nloc) return;
    const int j0 = boff[t], j1 = boff[t + 1];
    int n = 0, o = FILL ? roff[t] : 0;
    int js = j0;   // first beam of the open run
    double fx = 0.0, fy = 0.0, lx = 0.0, ly = 0.0, sx = 0.0, sy = 0.0;
    for (int j = j0; j <= j1; ++j) {
        bool cut = j == j1;
        double bx = 0.0, by = 0.0;
        if (!cut) {
            const double2 b = bxy[j];
            bx = b.x; by = b.y;
            if (j > js) {
                const double gx = bx - lx, gy = by - ly, ex = bx - fx, ey = by - fy;
                cut = gx * gx + gy * gy > gap2 || ex * ex + ey * ey > ext2 || j - js >= kRunCap;
            }
        }
        if (cut && j > js) {   // close [js, j)
            if (FILL) {
                const int k = j - js;
                const double cx = sx / (double)k, cy = sy / (double)k;
                double r2 = 0.0;
                for (int i = js; i < j; ++i) {
                    const double2 q = bxy[i];
                    const double dx = q.x - cx, dy = q.y - cy;
                    r2 = fmax(r2, dx * dx + dy * dy);
                }
                const float rf = __double2float_ru(sqrt(r2) * 1.000001 + 1e-12);
                r_s[o] = make_double2(sx, sy);   // (the centre is not stored: k_assoc_runs forms it by the same division, sx / k)
                r_m[o] = make_uint2(__float_as_uint(rf), (unsigned)k | ((unsigned)(js - j0) << 16));
                ++o;
            }
            ++n;
            js = j;
        }
        if (j < j1) {
            if (j == js) {
                fx = bx; fy = by;
                sx = bx; sy = by;
            } else {
                sx += bx; sy += by;
            }
            lx = bx; ly = by;
        }
    }
    if (!FILL) nrun[t] = n;
}

// The run's decision from the record of its centre's cell (see the header above).  Returns the label every beam of the
// run takes, or kRunUndecided.  thr_m = dist_thr - margin, eps = the margin, all in single precision: the squared
// distances are formed in double from the double world point (a map may lie kilometres from its origin), only their
// square roots and the three comparisons are single (relative error 1e-7 of a value of a few dist_thr << margin).
__device__ __forceinline__ int assoc_run(const GridView& g, const GridParams& gp, double wx, double wy, float r, float thr_m,
                                         float eps, float invf) {
    const int cx = grid_cell(wx, gp.gx0, gp.inv, gp.nx), cy = grid_cell(wy, gp.gy0, gp.inv, gp.ny);
    const unsigned off = ((unsigned)cy * (unsigned)gp.nx + (unsigned)cx) << 7;
    const char* __restrict__ rec = reinterpret_cast<const char*>(g.nb) + off;
    const double2 p0 = *reinterpret_cast<const double2*>(rec), p1 = *reinterpret_cast<const double2*>(rec + 16);
    const int4 ic = *reinterpret_cast<const int4*>(rec + 32);   // id0, id1, n, id2
    const int n = ic.z;
    double dx = p0.x - wx, dy = p0.y - wy;
    const double s0 = dx * dx + dy * dy;
    dx = p1.x - wx; dy = p1.y - wy;
    const double s1 = dx * dx + dy * dy;
    // nearest and second nearest of the (up to four) candidates; empty slots hold x = +inf
    double best = fmin(s0, s1), second = fmax(s0, s1);
    int bid = s1 < s0 ? ic.y : ic.x;
    if (__builtin_expect(__ballot(n > 2) != 0ull, 0)) {
        if (n > 2) {
            const double2 q2 = *reinterpret_cast<const double2*>(rec + 48), q3 = *reinterpret_cast<const double2*>(rec + 64);
            const int id3 = *reinterpret_cast<const int*>(rec + 80);
            dx = q2.x - wx; dy = q2.y - wy;
            const double s2 = dx * dx + dy * dy;
            dx = q3.x - wx; dy = q3.y - wy;
            const double s3 = dx * dx + dy * dy;
            const double lo = fmin(s2, s3), hi = fmax(s2, s3);
            const int lid = s3 < s2 ? id3 : ic.w;
            second = fmin(fmax(best, lo), fmin(second, hi));
            if (lo < best) bid = lid;
            best = fmin(best, lo);
        }
    }
    const float d1 = __builtin_amdgcn_sqrtf((float)best), d2 = __builtin_amdgcn_sqrtf((float)second);
    const float r2 = r + r;
    const bool settled = (n >= 1) & (n <= kNeighCap) & (d1 + r <= thr_m) & ((d1 + r2) * invf <= 1.0f - 1e-4f) & (d2 - d1 >= r2 + eps) & (bid >= 0);
    return settled ? bid : kRunUndecided;
}

// One wave per pose, lanes over its RUNS.  Same outputs as k_assoc_group (the pose's entries, one per distinct label, in
// the order of the label table's slots, staged at the pose's reserved place): everything behind phase A is unchanged.
template <bool DEBUG, int HS>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(HS == 128 ? 7 : 4, HS == 128 ? 8 : 5)))
void k_assoc_runs(const double* __restrict__ x, const int* __restrict__ roff, const double2* __restrict__ r_s,
                  const double* __restrict__ rot, const GridParams* __restrict__ gpar, const int* __restrict__ plan, int nloc, int t_begin,
                  // ^ the fourteen dwords that arrive in scalar registers with the wave (kernel-argument preload)
                  const double* __restrict__ x0, const uint2* __restrict__ r_m,
                  const int* __restrict__ boff, const double2* __restrict__ bxy, GridView g, double thr, double thr2,
                  float thr_m, float eps, int* __restrict__ label, int* __restrict__ bloc, int* __restrict__ st_label,
                  unsigned short* __restrict__ st_k, double* __restrict__ st_sbx, double* __restrict__ st_sby,
                  int* __restrict__ nent_out, int* __restrict__ isnew_out, int* __restrict__ flags, int nnz_total,
                  int* __restrict__ st_off, int pose0, int sparse0, unsigned long long* __restrict__ run_counts) {
    constexpr int kHash = HS, kGroupCap = HS * 3 / 4;
    constexpr int kHashShift = HS == 128 ? 25 : 24;
    __shared__ PoseTable<HS> tables[kWavesPerBlock];
    const int lane = lane_id();
    // (one wave per pose, consecutive poses on consecutive workgroups: an XCD-contiguous order, eight resident waves and
    // several poses per wave were all measured -- profiles/r05_assoc_runs_experiments.txt -- and lost or changed nothing)
    const int tl = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + wave_in_block());
    if (tl >= nloc) return;
    ASSOC_TS(0);
    PoseTable<HS>& T = tables[wave_in_block()];
    const GridParams gp = *gpar;
    const int R0 = __builtin_amdgcn_readfirstlane(roff[tl]), R1 = __builtin_amdgcn_readfirstlane(roff[tl + 1]);
    const int plan0 = __builtin_amdgcn_readfirstlane(plan[tl]), plan1 = __builtin_amdgcn_readfirstlane(plan[tl + 1]);
    const int j0 = __builtin_amdgcn_readfirstlane(boff[tl]);   // (first kept beam: the sparse staging place, the beam-by-beam path)
    const double ct = rot[2 * (size_t)tl], st = rot[2 * (size_t)tl + 1];
    double px, py, th;
    pose_of(x, x0, t_begin + tl, px, py, th);
    if (R0 == R1) {   // no kept beams
        if (lane == 0) {
            nent_out[tl] = 0;
            isnew_out[tl] = 0;
            st_off[tl] = 0;
        }
        return;
    }
    const unsigned nrun = (unsigned)(R1 - R0);
    auto at16 = [](const double2* __restrict__ base, unsigned idx) {
        return *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(base) + (idx << 4));
    };
    const double2* __restrict__ rsp = r_s + R0;
    const uint2* __restrict__ rmp = r_m + R0;
    // the pose's first 64 runs (lanes beyond the last run hold a copy of it and take no part)
    double2 c, sb;
    uint2 m;
    {
        const unsigned i0 = min((unsigned)lane, nrun - 1u);
        sb = at16(rsp, i0);
        m = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(rmp) + (i0 << 3));
        const double kd = (double)(m.y & 0xffffu);
        c = make_double2(sb.x / kd, sb.y / kd);   // the circle's centre: the division k_run_build measured the radius against
    }
    for (int s = lane; s < kHash; s += kWave) {
        T.key[s] = kEmpty;
        T.cnt[s] = 0;
        T.sx[s] = 0.0;
        T.sy[s] = 0.0;
    }
    const float invf = (float)gp.inv;
    int nent = 0;
    bool overflow = false;
    unsigned n_und = 0;
    __builtin_amdgcn_wave_barrier();
#ifdef ICM_ASSOC_TS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ASSOC_TS(1);   // header scalars are in
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ASSOC_TS(2);   // the first runs' records are in
    int nbatch = 0;
#endif
    for (unsigned base = 0; base < nrun && !overflow; base += kWave) {
        const bool valid = base + (unsigned)lane < nrun;
        const int k = (int)(m.y & 0xffffu);
        const int jr = j0 + (int)(m.y >> 16);
        int lab = kRunUndecided;
        if (valid) {
            const double wx = (c.x * ct - c.y * st) + px;
            const double wy = (c.x * st + c.y * ct) + py;
            lab = assoc_run(g, gp, wx, wy, __uint_as_float(m.x), thr_m, eps, invf);
        }
        const bool settled = valid && lab != kRunUndecided;
#ifdef ICM_ASSOC_TS
        if (nbatch == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ASSOC_TS(3); }   // grid records in, decisions made
#endif
        // settled runs claim / find the slot of their label (linear probing; a probe IS the compare-and-swap) and add
        // their cached totals by LDS atomics -- two runs of one landmark (seen left and right of an occluder, or
        // across the scan's wrap-around) need no arbitration
        int slot = (int)(((unsigned)lab * 2654435761u) >> kHashShift);
        bool inserted = false, found = false;
        if (settled) {
            int held = atomicCAS(&T.key[slot], kEmpty, lab);
            for (int probes = 0; probes < kHash; ++probes) {
                if (held == kEmpty || held == lab) {
                    inserted = held == kEmpty;
                    found = true;
                    break;
                }
                slot = (slot + 1) & (kHash - 1);
                held = atomicCAS(&T.key[slot], kEmpty, lab);
            }
        }
        if (__ballot(settled && !found) != 0ull) overflow = true;
        nent += __popcll(__ballot(inserted));
        if (settled && found) {
            atomicAdd(&T.cnt[slot], k);
            lds_add_f64(&T.sx[slot], sb.x);
            lds_add_f64(&T.sy[slot], sb.y);
        }
        if (DEBUG && settled) {
            for (int i = 0; i < k; ++i) {
                label[jr + i] = lab;
                bloc[jr + i] = slot;
            }
        }
        // the runs the test did not settle: beam by beam, the reference's rule literally (assoc_grid), one run at a time
        unsigned long long und = __ballot(valid && !settled);
        n_und += (unsigned)__popcll(und);
        while (und != 0ull && !overflow) {
            const int l = (int)__builtin_ctzll(und);
            und &= und - 1ull;
            const int kk = lane_bcast(k, l), jj = lane_bcast(jr, l);
            const bool on = lane < kk;
            int bl = -2, bs = 0;
            bool bins = false, bfound = false;
            if (on) {
                const double2 b = bxy[jj + lane];
                const double wx = (b.x * ct - b.y * st) + px;
                const double wy = (b.x * st + b.y * ct) + py;
                bl = assoc_grid(g, gp, wx, wy, thr, thr2);
                bs = (int)(((unsigned)bl * 2654435761u) >> kHashShift);
                int held = atomicCAS(&T.key[bs], kEmpty, bl);
                for (int probes = 0; probes < kHash; ++probes) {
                    if (held == kEmpty || held == bl) {
                        bins = held == kEmpty;
                        bfound = true;
                        break;
                    }
                    bs = (bs + 1) & (kHash - 1);
                    held = atomicCAS(&T.key[bs], kEmpty, bl);
                }
                if (bfound) {
                    atomicAdd(&T.cnt[bs], 1);
                    lds_add_f64(&T.sx[bs], b.x);
                    lds_add_f64(&T.sy[bs], b.y);
                }
                if (DEBUG) {
                    label[jj + lane] = bl;
                    bloc[jj + lane] = bs;
                }
            }
            if (__ballot(on && !bfound) != 0ull) overflow = true;
            nent += __popcll(__ballot(bins));
            if (nent > kGroupCap) overflow = true;
        }
        if (nent > kGroupCap) overflow = true;
        if (base + kWave < nrun) {   // (a scan with more than 64 runs)
            const unsigned in = min(base + (unsigned)kWave + (unsigned)lane, nrun - 1u);
            sb = at16(rsp, in);
            m = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(rmp) + (in << 3));
            const double kd = (double)(m.y & 0xffffu);
            c = make_double2(sb.x / kd, sb.y / kd);
        }
        __builtin_amdgcn_wave_barrier();
#ifdef ICM_ASSOC_TS
        ++nbatch;
        if (nbatch == 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ASSOC_TS(4); }   // table updated
        ASSOC_TS(5);
#endif
    }
    // compact the used slots into the pose's place, slot order (as k_assoc_group does)
    const int room = (plan1 - plan0) + kStageSlack;
    const bool fits = plan0 >= 0 && plan1 >= plan0 && plan1 <= nnz_total && nent <= room;
    const int sbase = fits ? plan0 + kStageSlack * (pose0 + tl) : sparse0 + j0;
    int written = 0;
    bool isnew = false;
    for (int s0 = 0; s0 < kHash; s0 += kWave) {
        const int s = s0 + lane;
        const int key = T.key[s];
        const bool occ = key != kEmpty;
        const unsigned long long mask = __ballot(occ);
        if (occ) {
            const unsigned q = (unsigned)(written + prefix_count(mask, lane));
            *reinterpret_cast<int*>(reinterpret_cast<char*>(st_label + sbase) + (q << 2)) = key;
            *reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(st_k + sbase) + (q << 1)) = (unsigned short)T.cnt[s];
            *reinterpret_cast<double*>(reinterpret_cast<char*>(st_sbx + sbase) + (q << 3)) = T.sx[s];
            *reinterpret_cast<double*>(reinterpret_cast<char*>(st_sby + sbase) + (q << 3)) = T.sy[s];
            isnew |= key == -1;
            if (DEBUG) T.owner[s] = q;
        }
        written += __popcll(mask);
    }
    const unsigned long long anynew = __ballot(isnew);
    if (lane == 0) {
        nent_out[tl] = written;
        st_off[tl] = sbase;
        isnew_out[tl] = anynew != 0ull;
        if (!fits && plan1 > 0) atomicAdd(&flags[3], 1);
        if (overflow) flags[0] = 1;
        if (run_counts && n_und) atomicAdd(&run_counts[1], (unsigned long long)n_und);   // (runs that went beam by beam: rare, counted over the handle's life)
    }
#ifdef ICM_ASSOC_TS
    ASSOC_TS(6);
    ASSOC_TS_VAL(7, nbatch);
#endif
    if (DEBUG) {  // beam -> entry index within the pose
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        const int j1 = boff[tl + 1];
        for (int j = j0 + lane; j < j1; j += kWave) bloc[j] = T.owner[bloc[j]];
    }
}

// Running-mean term of one entry: sum of its beams' world points and their count.  One
// 32-byte record so that the per-landmark gather touches one line per entry.
struct EntW {
    double wx, wy, k, pad;
};

// Entries -> pose-major compact arrays.  The gated-out group of a pose gets the fresh id
// lact0 + (number of earlier poses that created a landmark) (phase B).
// Per entry: count, sum of world points k p + R sum b (running-mean term), mean offset
// R bbar (moment-form energy) [, mean body point for the per-entry energy form].  Per pose: the pooled within-entry scatter C = sum_j b b^T - sum_e k
// bbar bbar^T of the energy's scatter term.
__global__ __launch_bounds__(kBlock) void k_compact(const double* __restrict__ x, const double* __restrict__ x0,
                                                    int t_begin, int nloc, const int* __restrict__ boff,
                                                    const int* __restrict__ ent_off, const int* __restrict__ new_rank,
                                                    int lact0, const int* __restrict__ st_label,
                                                    const unsigned short* __restrict__ st_k, const double* __restrict__ st_sbx,
                                                    const double* __restrict__ st_sby, const double* __restrict__ pose_s2,
                                                    unsigned* __restrict__ e_key, int* __restrict__ e_val,
                                                    int* __restrict__ e_k, double2* __restrict__ e_b,
                                                    EntW* __restrict__ e_w, double2* __restrict__ e_wr,
                                                    double* __restrict__ pose_c) {
    // one DPP row (16 lanes) per pose, four poses per wavefront (a pose has ~40 entries)
    const int sub = threadIdx.x & 15;
    const int tl = (blockIdx.x * kBlock + threadIdx.x) >> 4;
    const bool live = tl < nloc;
    int j0 = 0, e0 = 0, n = 0, nrank = 0;
    double px = 0.0, py = 0.0, th = 0.0;
    if (live) {
        j0 = boff[tl];
        e0 = ent_off[tl];
        n = ent_off[tl + 1] - e0;
        nrank = new_rank[tl];
        pose_of(x, x0, t_begin + tl, px, py, th);
    }
    double ct, st;
    pose_rot(th, ct, st);
    double mxx = 0.0, mxy = 0.0, myy = 0.0;
    for (int q = sub; q < n; q += 16) {
        int lab = st_label[j0 + q];
        if (lab < 0) lab = lact0 + nrank;
        const int k = st_k[j0 + q];
        const double kd = (double)k, sbx = st_sbx[j0 + q], sby = st_sby[j0 + q];
        e_key[e0 + q] = (unsigned)lab;
        e_val[e0 + q] = e0 + q;
        e_k[e0 + q] = k;
        if (e_b) e_b[e0 + q] = make_double2(sbx / kd, sby / kd);  // per-entry energy form only
        const double rx = ct * sbx - st * sby, ry = st * sbx + ct * sby;  // R sum b
        e_w[e0 + q] = EntW{kd * px + rx, kd * py + ry, kd, 0.0};
        e_wr[e0 + q] = make_double2(rx / kd, ry / kd);
        mxx += sbx * sbx / kd;
        mxy += sbx * sby / kd;
        myy += sby * sby / kd;
    }
    mxx = row_sum16(mxx);
    mxy = row_sum16(mxy);
    myy = row_sum16(myy);
    if (live && sub == 0) {
        pose_c[3 * (size_t)tl] = pose_s2[3 * (size_t)tl] - mxx;
        pose_c[3 * (size_t)tl + 1] = pose_s2[3 * (size_t)tl + 1] - mxy;
        pose_c[3 * (size_t)tl + 2] = pose_s2[3 * (size_t)tl + 2] - myy;
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

// ---------------------------------------------------------------------------------------
// Phase B/D, the per-landmark gather: one wave per label walks the label's entries in time
// order (they are contiguous in the label-sorted entry ids) and forms the inclusive prefix
// of the sufficient statistics (n, sum x, sum y), seeded with the lower ranks' totals.
//   TOTALS: only the landmark's local totals are wanted (sharded run, before the exchange):
//           stats layout [sx(L) | sy(L) | n(L) | header(8)];
//   else:   target of every entry = running mean through that pose inclusive (SURVEY
//           Appendix A.3/A.6 phase B), and (single rank) the raw map y = S/n, cant_obs_i = n.
// ---------------------------------------------------------------------------------------
template <bool TOTALS>
__global__ __launch_bounds__(kBlock) void k_lm_scan(int nlab, int L, const int* __restrict__ lm_off,
                                                    const int* __restrict__ sval, const EntW* __restrict__ e_w,
                                                    const double* __restrict__ off_sx, const double* __restrict__ off_sy,
                                                    const double* __restrict__ off_n, double2* __restrict__ tgt,
                                                    double* __restrict__ stats,
                                                    double* __restrict__ y_raw, double* __restrict__ cnt_raw) {
    const int lane = lane_id();
    const int i = blockIdx.x * kWavesPerBlock + wave_in_block();
    if (i >= L) return;
    double cx = 0.0, cy = 0.0, cn = 0.0;
    if (i < nlab) {
        if (!TOTALS && off_n) {
            cx = off_sx[i];
            cy = off_sy[i];
            cn = off_n[i];
        }
        const int p0 = lm_off[i], p1 = lm_off[i + 1];
        for (int base = p0; base < p1; base += kWave) {
            const int p = base + lane;
            const bool valid = p < p1;
            const int last = min(kWave, p1 - base) - 1;
            int e = 0;
            double vx = 0.0, vy = 0.0, vn = 0.0;
            if (valid) {
                e = sval[p];
                const EntW w = e_w[e];
                vx = w.wx;
                vy = w.wy;
                vn = w.k;
            }
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) {
                const double ux = __shfl_up(vx, d, kWave), uy = __shfl_up(vy, d, kWave), un = __shfl_up(vn, d, kWave);
                if (lane >= d) {
                    vx += ux;
                    vy += uy;
                    vn += un;
                }
            }
            vx += cx;
            vy += cy;
            vn += cn;
            if (!TOTALS && valid) tgt[e] = make_double2(vx / vn, vy / vn);
            cx = __shfl(vx, last, kWave);
            cy = __shfl(vy, last, kWave);
            cn = __shfl(vn, last, kWave);
        }
    }
    if (lane == 0) {
        if (TOTALS) {
            stats[i] = cx;
            stats[L + i] = cy;
            stats[2 * L + i] = cn;
        } else if (y_raw) {
            cnt_raw[i] = cn;
            y_raw[i] = cn > 0.0 ? cx / cn : 0.0;
            y_raw[L + i] = cn > 0.0 ? cy / cn : 0.0;
        }
    }
}

// After the all-gather of the per-rank statistics: for existing landmarks (i < lact0) the
// exclusive prefix over lower ranks (the state of the running mean when this rank's first
// pose is folded in) and the total over all ranks; landmarks created during the sweep are
// singletons and are laid out rank after rank behind lact0.
// header per rank: [0] number of new landmarks, [1] error flags.
__global__ __launch_bounds__(kBlock) void k_stats_prefix(const double* __restrict__ stats_all, int stride, int rank,
                                                         int world, int L, int lact0, double* __restrict__ off_sx,
                                                         double* __restrict__ off_sy, double* __restrict__ off_n,
                                                         double* __restrict__ y_raw, double* __restrict__ cnt_raw,
                                                         int* __restrict__ flags = nullptr) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i == 0 && flags) {   // a sweep queued without a host look: what any rank flagged, every rank honours
        int any = 0;
        double total_new = 0.0;
        for (int r = 0; r < world; ++r) {
            const double* hd = stats_all + (size_t)r * stride + 3 * (size_t)L;
            total_new += hd[0];
            any |= hd[1] != 0.0;
        }
        if (any) flags[0] = 1;
        if ((double)lact0 + total_new > (double)L) flags[2] = 1;   // labels beyond the map capacity, over all ranks
    }
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
        int q = i - lact0;  // which rank's new landmark lands in column i?
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

// ---------------------------------------------------------------------------------------
// Phase B/D WITHOUT a sort: hierarchical running sums over time.
//
// The target of entry (pose t, landmark l) is the mean of l's observations through pose t
// (SURVEY Appendix A.3/A.6 phase B): a prefix, in time, of the per-landmark sums.  Consecutive
// poses see nearly the same landmarks, so the prefix is formed level by level instead of
// sorting all entries by landmark:
//   k_chunk_l1   one WAVE per chunk of 64 (32, 16) consecutive poses walks its poses in time order;
//                a small LDS table keyed by landmark holds the running sums of the chunk:
//                every entry gets its prefix inside the chunk, every (chunk, landmark) RECORD
//                the chunk's total.  Record r = chunk * kT1 + table slot (unused slots empty).
//   k_chunk_l2   one workgroup per SUPERCHUNK of G consecutive chunks walks its chunks in
//                order with a larger LDS table: every record gets its prefix inside the
//                superchunk; the superchunk's per-landmark totals go to one row of a dense
//                [superchunks x L] matrix (at most kMaxSuper rows).
//   k_lm_l3      one thread per landmark: exclusive prefix down its matrix column, landmark
//                totals (= the raw map, or the rank's statistics for the exchange).
//   k_rec_push   record prefix := (lower ranks) + matrix prefix + superchunk prefix.
//   k_pose_moments_h then forms  target = (record prefix + entry prefix) / n  on the fly.
// Everything streams or stays in LDS/L2: no global sort, no per-landmark gather/scatter.
// A chunk with more than ~190 distinct landmarks, or a superchunk with more than kT2Cap,
// raises flags[1]; the host then runs the sort-based pipeline (k_compact .. k_lm_scan), which
// has no such limits.
// ---------------------------------------------------------------------------------------
constexpr int kCHMax = 64;       // poses per chunk: 64, 32 or 16 (lane p holds pose p's header); short
                                 // sequences use short chunks -- a chunk is ONE wave's serial work
constexpr int kT1 = 256;         // slots of a chunk table (a slot index is one byte: k_chunk_l1 packs it beside the entry's count)
constexpr int kT2 = 2048;        // slots of a superchunk table
constexpr int kT2Cap = 1536;
constexpr int kMaxSuper = 64;    // rows of the dense matrix
constexpr int kRawBuffer = 0x00020000;   // dword 3 of a raw (stride 0) buffer descriptor on gfx9 / CDNA: 32-bit data format

struct ChunkTable {
    int key[kT1];
    double sx[kT1 + kWave];   // (+ one scratch slot per lane: idle lanes run the same code branch-free)
    double sy[kT1 + kWave];
    double sn[kT1 + kWave];
    int used;
};

__device__ __forceinline__ int table_slot(int* key, int mask, int shift, int lab, bool& inserted) {
    int slot = (int)(((unsigned)lab * 2654435761u) >> shift);
    inserted = false;
    for (;;) {
        const int k = key[slot];
        if (k == lab) break;
        if (k == kEmpty) {
            const int old = atomicCAS(&key[slot], kEmpty, lab);
            if (old == kEmpty) {
                inserted = true;
                break;
            }
            if (old == lab) break;
        }
        slot = (slot + 1) & mask;
    }
    return slot;
}

// First 64 staged entries of kGroup consecutive poses (lane = entry), requested together.
constexpr int kGroup = 8;
struct EntryGroup {
    int lab[kGroup], k[kGroup];
    double bx[kGroup], by[kGroup];
};

__device__ __forceinline__ void load_group(EntryGroup& g, int p0, int lane, int n, int j0, const int* __restrict__ st_label,
                                           const unsigned short* __restrict__ st_k, const double* __restrict__ st_sbx,
                                           const double* __restrict__ st_sby) {
#pragma unroll
    for (int i = 0; i < kGroup; ++i) {
        const int np = lane_bcast(n, p0 + i), jp = lane_bcast(j0, p0 + i);
        // every lane loads (idle lanes re-read the pose's last entry -- a line the active lanes fetch anyway, not one
        // address that every wave of the launch would share): no branch around the loads, so the compiler can count
        // them and wait for exactly the group it is about to use
        const int j = jp + min(lane, max(np - 1, 0));
        const int lab = st_label[j];
        g.lab[i] = lane < np ? lab : kEmpty;
        g.k[i] = st_k[j];
        g.bx[i] = st_sbx[j];
        g.by[i] = st_sby[j];
    }
}

template <int kCH>
__global__ __launch_bounds__(kBlock) void k_chunk_l1(const double* __restrict__ x, const double* __restrict__ x0,
                                                     int t_begin, int nloc, int nchunks, const int* __restrict__ boff,
                                                     const int* __restrict__ nent, const int* __restrict__ ent_off,
                                                     const int* __restrict__ new_rank, int lact0,
                                                     const int* __restrict__ st_label, const unsigned short* __restrict__ st_k,
                                                     const double* __restrict__ st_sbx, const double* __restrict__ st_sby,
                                                     double* __restrict__ pre_x, double* __restrict__ pre_y,
                                                     unsigned* __restrict__ pre_n,
                                                     int* __restrict__ rec_label, double* __restrict__ rec_sx,
                                                     double* __restrict__ rec_sy, double* __restrict__ rec_n,
                                                     int* __restrict__ flags, int c_begin = 0,
                                                     int* __restrict__ totals = nullptr,   /* non-null = no scan kernels this sweep: [1] += landmark-creating poses */
                                                     const int* __restrict__ isnew = nullptr,
                                                     unsigned long long* __restrict__ pub = nullptr, unsigned epoch = 0u,
                                                     int spin_limit = 0) {
    __shared__ ChunkTable tables[kWavesPerBlock];
    const int lane = lane_id();
    const int c = c_begin + blockIdx.x * kWavesPerBlock + wave_in_block();   // chunks [c_begin, nchunks)
    if (c >= nchunks) return;
    ChunkTable& T = tables[wave_in_block()];
    for (int s = lane; s < kT1; s += kWave) {
        T.key[s] = kEmpty;
        T.sx[s] = 0.0;
        T.sy[s] = 0.0;
        T.sn[s] = 0.0;
    }
    if (lane == 0) T.used = 0;
    // lane p: header of pose p of the chunk
    const int tl = c * kCH + lane;
    int j0 = 0, n = 0, e0 = 0, nr = 0;
    double px = 0.0, py = 0.0, th = 0.0;
    if (lane < kCH && tl < nloc) {
        j0 = boff[tl];      // (the pose's place in the staging area: st_off)
        n = nent[tl];
        e0 = j0;            // the per-entry prefixes live at the entries' own places: no entry-offset scan needed for them
        nr = totals ? 0 : new_rank[tl];
        pose_of(x, x0, t_begin + tl, px, py, th);
    }
    if (totals) {
        // No scan kernels this sweep (the host keeps reporting the last scanned sweep's entry count: an atomic per chunk
        // on one word to add it up here cost 9 us).  The ranks of the poses that create a landmark (their fresh labels: lact0 + number of such poses before) are the only thing
        // the scan would still be needed for, and such poses are rare: every chunk PUBLISHES how many it has (one tagged
        // 8-byte word, write-through; tag = this launch's epoch, so nothing is reset between launches), and only a chunk
        // that has one adds up the words of the chunks before it.  Bounded polls: a word that does not arrive (a dispatch
        // order that starts later chunks first on a chip too full to hold them all) raises flags[1]'s neighbour flags[0]
        // -- nothing is then replaced and the host repeats the sweep with the scan kernels.
        const int fl = (lane < kCH && tl < nloc) ? isnew[tl] : 0;
        const unsigned long long fm = __ballot(fl != 0);
        const int cnt = __popcll(fm);
        if (lane == 0) {
            if (cnt) atomicAdd(totals + 1, cnt);   // (the sweep's new-landmark count; rare)
            __hip_atomic_store(&pub[c], ((unsigned long long)epoch << 32) | (unsigned)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (cnt) {   // (wave-uniform, rare)
            int before = 0;
            bool gave_up = false;
            for (int q0 = 0; q0 < c; q0 += kWave * 8) {   // eight words per lane in flight
                unsigned long long w[8];
#pragma unroll
                for (int b = 0; b < 8; ++b) w[b] = __hip_atomic_load(&pub[min(q0 + b * kWave + lane, c - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const int q = q0 + b * kWave + lane;
                    if (q < c) {
                        int spins = 0;
                        while ((unsigned)(w[b] >> 32) != epoch) {
                            if (++spins > spin_limit) {
                                gave_up = true;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(4);
                            w[b] = __hip_atomic_load(&pub[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        before += (int)(unsigned)w[b];
                    }
                }
            }
            if (__ballot(gave_up) != 0ull) {
                for (int s = lane; s < kT1; s += kWave) rec_label[c * kT1 + s] = kEmpty;
                if (lane == 0) flags[0] = 1;
                return;
            }
#pragma unroll
            for (int d = kWave / 2; d > 0; d >>= 1) before += __shfl_xor(before, d, kWave);
            nr = before + prefix_count(fm, lane);
        }
    }
    if (__ballot(n > kWave) != 0ull) {  // a scan with more than 64 distinct landmarks: the lanes of this
        for (int s = lane; s < kT1; s += kWave) rec_label[c * kT1 + s] = kEmpty;  // kernel map to entries
        if (lane == 0) flags[1] = 1;                                               // -> sort-based path
        return;
    }
    EntryGroup ga, gb;
    load_group(ga, 0, lane, n, j0, st_label, st_k, st_sbx, st_sby);
    double ct, st;
    pose_rot(th, ct, st);
    __builtin_amdgcn_wave_barrier();
    bool overflow = false;
    // one group of poses: `cur` is folded in while `nxt` fills (the two buffers swap roles from group to group, no copy)
    auto fold_group = [&](EntryGroup& cur, EntryGroup& nxt, int p0) {
        // the next group's entries are in flight while this one is folded in
        if (p0 + kGroup < kCH) load_group(nxt, p0 + kGroup, lane, n, j0, st_label, st_k, st_sbx, st_sby);
        // 1. table slots of the group's entries: independent of the running sums, all at once
        int slot[kGroup];
        unsigned pending = 0u;   // bit i: this lane's entry of pose i has no slot yet
#pragma unroll
        for (int i = 0; i < kGroup; ++i) {
            const int fresh = lact0 + lane_bcast(nr, p0 + i);
            if (cur.lab[i] != kEmpty) {
                if (cur.lab[i] < 0) cur.lab[i] = fresh;            // gated-out beams: one fresh landmark
                pending |= 1u << i;
            }
            slot[i] = (int)(((unsigned)cur.lab[i] * 2654435761u) >> 24);
        }
        // the kGroup probe sequences advance together: one LDS round trip per step for all of them
        for (int probes = 0; probes < kT1 && __ballot(pending != 0u) != 0ull; ++probes) {
            int kv[kGroup];
#pragma unroll
            for (int i = 0; i < kGroup; ++i) kv[i] = T.key[slot[i]];
#pragma unroll
            for (int i = 0; i < kGroup; ++i) {
                if (pending & (1u << i)) {
                    int k = kv[i];
                    if (k == kEmpty) {
                        k = atomicCAS(&T.key[slot[i]], kEmpty, cur.lab[i]);
                        if (k == kEmpty) {
                            atomicAdd(&T.used, 1);
                            k = cur.lab[i];
                        }
                    }
                    if (k == cur.lab[i]) pending &= ~(1u << i);
                    else slot[i] = (slot[i] + 1) & (kT1 - 1);
                }
            }
        }
        if (pending) slot[0] = -1;   // table full
        __builtin_amdgcn_wave_barrier();
        {
            const bool bad = T.used > kT1 - 32 || slot[0] < 0;
            if (__ballot(bad) != 0ull) return true;   // too many distinct landmarks for the table: sort-based path
        }
        // 2. the poses in time order: running sums of their landmarks
#pragma unroll
        for (int i = 0; i < kGroup; ++i) {
            const int p = p0 + i;
            const int np = lane_bcast(n, p);
            if (np == 0) continue;
            const int ep = lane_bcast(e0, p);
            const double pxp = lane_bcast(px, p), pyp = lane_bcast(py, p);
            const double ctp = lane_bcast(ct, p), stp = lane_bcast(st, p);
            {
                const bool act = lane < np;
                const double kd = (double)cur.k[i], sbx = cur.bx[i], sby = cur.by[i];
                const double rx = ctp * sbx - stp * sby, ry = stp * sbx + ctp * sby;  // R sum b
                const double wx = kd * pxp + rx, wy = kd * pyp + ry;                  // sum of world points
                const int sl = act ? slot[i] : kT1 + lane;   // idle lanes: scratch slots behind the table
                // labels are distinct within a pose: no two lanes meet in a slot
                const double ax = T.sx[sl] + wx, ay = T.sy[sl] + wy, an = T.sn[sl] + kd;
                T.sx[sl] = ax;
                T.sy[sl] = ay;
                T.sn[sl] = an;
                // One buffer descriptor per array spans exactly this pose's np entries: the hardware drops the stores of the
                // lanes beyond it (raw buffer addressing, offset >= range) -- no branch around the stores, so the compiler
                // still counts them exactly, and no idle-lane traffic at all.
                // (beams of the landmark inside the chunk through this pose -- exact; at most 64 poses x 8192 beams < 2^24,
                // both bounds checked at upload -- and, in the low byte, the slot of the entry's record: one word)
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                const int off8 = lane << 3;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ax), __builtin_amdgcn_make_buffer_rsrc(pre_x + ep, 0, np << 3, kRawBuffer), off8, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ay), __builtin_amdgcn_make_buffer_rsrc(pre_y + ep, 0, np << 3, kRawBuffer), off8, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(((unsigned)an << 8) | ((unsigned)sl & 255u), __builtin_amdgcn_make_buffer_rsrc(pre_n + ep, 0, np << 2, kRawBuffer), lane << 2, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
        return false;
    };
    static_assert((kCH / kGroup) % 2 == 0, "the groups of a chunk come in pairs");
    for (int p0 = 0; p0 < kCH && !overflow; p0 += 2 * kGroup) {
        overflow = fold_group(ga, gb, p0);
        if (!overflow) overflow = fold_group(gb, ga, p0 + kGroup);
    }
    __builtin_amdgcn_wave_barrier();
    for (int s = lane; s < kT1; s += kWave) {
        const int r = c * kT1 + s;
        const int k = overflow ? kEmpty : T.key[s];
        rec_label[r] = k;
        rec_sx[r] = T.sx[s];   // (unused slots hold zeros: whole lines go out, and k_chunk_l2 reads every slot anyway)
        rec_sy[r] = T.sy[s];
        rec_n[r] = T.sn[s];
    }
    if (overflow && lane == 0) flags[1] = 1;
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Level 2: thread s of the workgroup <-> slot s of each chunk's record block.
__global__ __launch_bounds__(kT1) void k_chunk_l2(int nchunks, int G, int L, const int* __restrict__ rec_label,
                                                  const double* __restrict__ rec_sx, const double* __restrict__ rec_sy,
                                                  const double* __restrict__ rec_n, double* __restrict__ off_x,
                                                  double* __restrict__ off_y, double* __restrict__ off_n,
                                                  double* __restrict__ ms_x, double* __restrict__ ms_y,
                                                  double* __restrict__ ms_n, int* __restrict__ flags, int sc_begin = 0,
                                                  int zero_row = 0) {
    __shared__ int key[kT2];
    __shared__ double sx[kT2], sy[kT2], sn[kT2];
    __shared__ int used, full[2];
    const int tid = threadIdx.x, sc = sc_begin + blockIdx.x;
    // this superchunk's row of the dense matrix takes its landmarks' totals at the end; zero_row: clear
    // it first (otherwise the host has cleared the whole matrix, on a side stream under the solves)
    if (zero_row)
        for (int k = tid; k < L; k += kT1) {
            ms_x[(size_t)sc * L + k] = 0.0;
            ms_y[(size_t)sc * L + k] = 0.0;
            ms_n[(size_t)sc * L + k] = 0.0;
        }
    for (int s = tid; s < kT2; s += kT1) {
        key[s] = kEmpty;
        sx[s] = 0.0;
        sy[s] = 0.0;
        sn[s] = 0.0;
    }
    if (tid == 0) used = full[0] = full[1] = 0;
    __syncthreads();
    const int c0 = sc * G, c1 = min(c0 + G, nchunks);
    // The records of the next kAhead chunks are in flight while one chunk is folded in.  Loads and
    // stores are issued by every thread in every step (clamped index / the record's own, unused
    // slot), never under a branch, so that the compiler can count them and wait (vmcnt(N)) for
    // exactly the chunk it is about to use instead of draining the queue.
    constexpr int kAhead = 3;
    int lab_q[kAhead];
    double vx_q[kAhead], vy_q[kAhead], vn_q[kAhead];
    auto fetch = [&](int q, int cn) {   // records of chunk cn (clamped; empty beyond the superchunk) into queue slot q
        const size_t r = (size_t)min(cn, max(c1 - 1, c0)) * kT1 + tid;
        const int l = rec_label[r];
        lab_q[q] = cn < c1 ? l : kEmpty;
        vx_q[q] = rec_sx[r];
        vy_q[q] = rec_sy[r];
        vn_q[q] = rec_n[r];
    };
#pragma unroll
    for (int q = 0; q < kAhead; ++q) fetch(q, c0 + q);
    bool overflow = false;
    for (int cb = c0; cb < c1 && !overflow; cb += kAhead) {
#pragma unroll
        for (int q = 0; q < kAhead; ++q) {   // slot q holds chunk cb + q; it is refilled in place (no register shuffling)
            const int c = cb + q;
            if (c >= c1) break;
            const int lab = lab_q[q];
            const double vx = vx_q[q], vy = vy_q[q], vn = vn_q[q];
            fetch(q, c + kAhead);
            double ox = 0.0, oy = 0.0, on = 0.0;
            if (lab != kEmpty) {
                bool inserted;
                const int slot = table_slot(key, kT2 - 1, 21, lab, inserted);
                if (inserted && atomicAdd(&used, 1) >= kT2Cap) full[c & 1] = 1;   // (a chunk adds < 256 landmarks: the 2048 slots cannot run out before the look below)
                ox = sx[slot];   // a landmark has one record per chunk: no two threads meet in a slot
                oy = sy[slot];
                on = sn[slot];
                sx[slot] = ox + vx;
                sy[slot] = oy + vy;
                sn[slot] = on + vn;
            }
            const size_t r = (size_t)c * kT1 + tid;   // (an empty record's slot is never read)
            off_x[r] = ox;
            off_y[r] = oy;
            off_n[r] = on;
            // ONE workgroup barrier per chunk (round 3 had two, around a look at the fill count: the overflow mark now
            // alternates between two words, so a thread already in the next chunk cannot change the one being looked at),
            // and one that waits for this wave's LDS traffic only: __syncthreads() would also drain the global loads in
            // flight for the next chunks
            lds_barrier();
            if (full[c & 1]) {   // the same value in every thread: more than kT2Cap distinct landmarks in the superchunk
                overflow = true;
                break;
            }
        }
    }
    if (overflow) {
        if (tid == 0) flags[1] = 1;
        return;
    }
    for (int s = tid; s < kT2; s += kT1) {
        const int k = key[s];
        if (k != kEmpty && k < L) {
            ms_x[(size_t)sc * L + k] = sx[s];
            ms_y[(size_t)sc * L + k] = sy[s];
            ms_n[(size_t)sc * L + k] = sn[s];
        }
    }
}

// Level 3: exclusive prefix down each landmark's column of the matrix, rows [row_begin, row_end).
// A sweep may take the rows in consecutive ranges (time segments): carry_in = the column sums of the
// rows before (null: none), carry_out = those through row_end (null: not needed).  final: the totals
// go out -- stats != nullptr: the rank's totals for the exchange [sx(L) | sy(L) | n(L)]; else the raw map.
// Mapping: a workgroup takes 64 columns; its four waves take sixteen rows each -- every thread has its
// (at most kMaxSuper / 4) rows in flight at once, prefixes them locally, and the groups' totals meet in LDS.  (One
// thread per column walking all rows, eight loads at a time, took 20 us for 30 MB: 157 waves cannot keep HBM busy.)
constexpr int kL3Groups = kBlock / kWave;                                   // row groups = waves of a workgroup
constexpr int kL3Rows = (kMaxSuper + kL3Groups - 1) / kL3Groups;            // rows per group (upper bound)
__global__ __launch_bounds__(kBlock) void k_lm_l3(int nsuper, int L, int lact0, const int* __restrict__ n_new_dev,
                                                  double* __restrict__ ms_x, double* __restrict__ ms_y,
                                                  double* __restrict__ ms_n, double* __restrict__ stats,
                                                  double* __restrict__ y_raw, double* __restrict__ cnt_raw,
                                                  const int* __restrict__ n_ent_dev, int* __restrict__ flags,
                                                  int row_begin = 0, int row_end = -1, const double* __restrict__ carry_in = nullptr,
                                                  double* __restrict__ carry_out = nullptr, int final = 1,
                                                  int* __restrict__ flags_next = nullptr, int* __restrict__ host_words = nullptr,
                                                  int host_flags = 0, int* __restrict__ done = nullptr, int done_epoch = 0) {
    // done (nullable): [0] workgroups of this launch that are through (the last one resets it), [1] := done_epoch by the
    // last one -- the side stream's k_wait_word polls it: Mapa.filtrar starts the moment the raw map is out, without a
    // stop event on this queue (5-6 us of nothing behind this launch).  The raw map and the flags go out write-through.
    __shared__ double tot[3][kL3Groups][kWave];
    const int col = threadIdx.x & (kWave - 1), grp = threadIdx.x >> 6;
    const int i = blockIdx.x * kWave + col;
    if (row_end < 0) row_end = nsuper;
    if (blockIdx.x == 0 && threadIdx.x == 0 && final) {  // what the host reads back at its one synchronisation of the sweep, in one 16-byte copy
        flags[4] = *n_ent_dev;
        flags[5] = *n_new_dev;
        flags[6] = flags[0];
        flags[7] = flags[1];
        flags[2] = lact0 + *n_new_dev > L ? 1 : 0;   // labels beyond the map capacity (the reference's IndexError)
        if (done) __hip_atomic_store(&flags[2], flags[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (read by a kernel that does not wait for this launch's end)
        // the NEXT sweep's flag block (the sweeps alternate between two): nobody reads it any more -- the sweep before this
        // one is over, host included -- and clearing it here saves the next sweep a memset launch at its head
        if (flags_next)
            for (int q = 0; q < 16; ++q) flags_next[q] = 0;
        // the same words straight into the host's (mapped, pinned) block: no 16-byte copy launch -- a 5 us blit kernel -- on
        // any stream; host_flags: the sweep's flags as well (single rank: nothing behind this launch changes them)
        if (host_words) {
            for (int q = 0; q < 4; ++q) host_words[q] = flags[4 + q];
            if (host_flags)
                for (int q = 0; q < 4; ++q) host_words[12 + q] = flags[q];
        }
    }
    // groups are ABSOLUTE row ranges [16 g, 16 g + 16): a sweep that takes the rows in two ranges (cut at a multiple of
    // kL3Rows: pipeline_split_super) adds every column up in the same association as one that takes them at once
    const int r0 = max(grp * kL3Rows, row_begin), r1 = max(min((grp + 1) * kL3Rows, row_end), r0);
    const bool live = i < L && i < lact0 + *n_new_dev;          // columns of labels that do not exist are all zero
    double vx[kL3Rows], vy[kL3Rows], vn[kL3Rows];
    double tx = 0.0, ty = 0.0, tn = 0.0;
    if (live) {
#pragma unroll
        for (int b = 0; b < kL3Rows; ++b) {   // all loads first (clamped row: unused values are never added)
            const size_t q = (size_t)min(r0 + b, max(row_end - 1, row_begin)) * L + i;
            vx[b] = ms_x[q];
            vy[b] = ms_y[q];
            vn[b] = ms_n[q];
        }
#pragma unroll
        for (int b = 0; b < kL3Rows; ++b) {   // exclusive prefix inside the group, in place
            const bool on = r0 + b < r1;
            const double x_ = vx[b], y_ = vy[b], n_ = vn[b];
            vx[b] = tx; vy[b] = ty; vn[b] = tn;
            if (on) { tx += x_; ty += y_; tn += n_; }
        }
    }
    tot[0][grp][col] = tx;
    tot[1][grp][col] = ty;
    tot[2][grp][col] = tn;
    __syncthreads();
    double ax = 0.0, ay = 0.0, an = 0.0;   // sums before this group's rows: the carry and the groups before
    if (carry_in && i < L) {
        ax = carry_in[i];
        ay = carry_in[L + i];
        an = carry_in[2 * (size_t)L + i];
    }
    for (int g2 = 0; g2 < grp; ++g2) {
        ax += tot[0][g2][col];
        ay += tot[1][g2][col];
        an += tot[2][g2][col];
    }
    if (live) {
#pragma unroll
        for (int b = 0; b < kL3Rows; ++b) {
            if (r0 + b < r1) {
                const size_t q = (size_t)(r0 + b) * L + i;
                ms_x[q] = ax + vx[b];
                ms_y[q] = ay + vy[b];
                ms_n[q] = an + vn[b];
            }
        }
    }
    if (grp == kL3Groups - 1 && i < L) {   // the last group holds the column totals
        ax += tx; ay += ty; an += tn;
        if (carry_out) {
            carry_out[i] = ax;
            carry_out[L + i] = ay;
            carry_out[2 * (size_t)L + i] = an;
        }
        if (final && stats) {
            stats[i] = ax;
            stats[L + i] = ay;
            stats[2 * L + i] = an;
        } else if (final) {
            const double mx = an > 0.0 ? ax / an : 0.0, my = an > 0.0 ? ay / an : 0.0;
            if (done) {
                __hip_atomic_store(&cnt_raw[i], an, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&y_raw[i], mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&y_raw[L + i], my, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                cnt_raw[i] = an;
                y_raw[i] = mx;
                y_raw[L + i] = my;
            }
        }
    }
    if (done) {   // (block-uniform)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores have been acknowledged
        __syncthreads();
        if (threadIdx.x == 0) {
            const int old = __hip_atomic_fetch_add(&done[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == (int)gridDim.x - 1) {
                __hip_atomic_store(&done[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&done[1], done_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// One wave on the side stream waits for a word to take a value (k_lm_l3's last workgroup: the raw map is out): what
// is queued behind it starts then, a kernel boundary later.  Bounded (the launch that would set the word may have
// failed, or the streams may be serialised -- counter collection, HIP_LAUNCH_BLOCKING, two streams on one hardware queue --
// so that the launch that sets the word cannot start while this one runs): on giving up it raises *give_up, Mapa.filtrar's
// kernels, queued behind, leave the map alone, and the host runs Mapa.filtrar itself for this sweep (icm_sweep_finish:
// correct, slower), counts the event (icm_get_wait_giveups) and starts the side stream by the stop event from then on.
__global__ __launch_bounds__(kWave) void k_wait_word(const int* __restrict__ word, int value, int polls, int* __restrict__ give_up) {
    if (threadIdx.x != 0) return;
    for (int p = 0; p < polls; ++p) {
        if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == value) return;
        __builtin_amdgcn_s_sleep(32);
    }
    *give_up = value;
}

// Sums before each record's chunk: lower ranks (rank_*; null on a single rank) + the
// superchunks before + the chunks before inside the superchunk.  In place on off_*.
__global__ __launch_bounds__(kBlock) void k_rec_push(int nrec, int G, int L, const int* __restrict__ rec_label,
                                                     const double* __restrict__ ms_x, const double* __restrict__ ms_y,
                                                     const double* __restrict__ ms_n, const double* __restrict__ rank_x,
                                                     const double* __restrict__ rank_y, const double* __restrict__ rank_n,
                                                     double* __restrict__ off_x, double* __restrict__ off_y,
                                                     double* __restrict__ off_n, int r_begin = 0) {
    const int r = r_begin + blockIdx.x * kBlock + threadIdx.x;   // records [r_begin, nrec)
    if (r >= nrec) return;
    const int lab = rec_label[r];
    if (lab == kEmpty || lab >= L) return;
    const size_t q = (size_t)((r / kT1) / G) * L + lab;
    double bx = ms_x[q], by = ms_y[q], bn = ms_n[q];
    if (rank_n) {
        bx = rank_x[lab] + bx;
        by = rank_y[lab] + by;
        bn = rank_n[lab] + bn;
    }
    off_x[r] = bx + off_x[r];
    off_y[r] = by + off_y[r];
    off_n[r] = bn + off_n[r];
}

// ---------------------------------------------------------------------------------------
// Phase D on the GPU: Mapa.filtrar (reference scripts/ICM_SLAM_tools.py:204-265) and the search
// grid of the refined map, on a side stream under the pose solves.  A chain of small
// multi-workgroup kernels (the map is ~1e4 landmarks: every step is one or two items per thread;
// a single workgroup spent 0.16 ms here on dependent-load latency alone):
//   k_fl_count     per block of the landmark range: survivors (seen >= cota times) and their extent
//   k_fl_scatter   order-preserving compaction of the survivors; grid parameters; cell counters := 0
//   k_fl_cell_count / k_scan_tiles+fix / k_fl_fill   counting-sort grid (cell >= dist_thr), written
//                  straight into the search structures of the next sweep
//   k_fl_pairs     nearest other survivor of each survivor (3x3 cells): close / coincident counts
//   k_fl_finalize  no pair closer than dist_thr (the usual case): the refined map is the
//                  survivors, each as the reference's count-weighted mean of one term, (y*n)/n
// Pairs closer than dist_thr (info[1] = 1) are merged on the device too (host-triggered, rare):
//   k_fl_components  every component of the nearest-neighbour graph (<= kCompMax members) replays
//                    the reference's sequential label propagation (:246-249) on its own members
//   k_fl_label_flags / k_exscan_i32 / k_fl_gather   gap-closing renumbering (:251-253), count-weighted
//                    means (:256-260); then the grid chain again over the refined map.
// Coincident survivors (whose zero distance the reference replaces by the global maximum), an empty
// map and components larger than kCompMax (numpy's pairwise summation kicks in at 8 terms) set
// info[1] = 2: the exact host routine takes over.
// info: [0] landmarks_actuales after the filter, [1] 0 done / 1 merge pending / 2 host, [2] close pairs.
// ---------------------------------------------------------------------------------------
constexpr int kFB = 1024;          // threads per block of the filter kernels
constexpr int kFlMaxBlocks = 64;   // blocks of the order-preserving prune
constexpr int kCompMax = 7;        // largest component merged on the device
constexpr int kCompStride = 8;

struct FlState {
    int n;         // survivors of the prune
    int close;     // survivors with another survivor closer than dist_thr
    int same;      // survivors coincident with another one
    int host;      // the host routine must take over
    int n_ref;     // landmarks after merging
    int abort;     // the sweep's tables overflowed / its labels exceed L: leave the map state alone
    int pad[2];
    GridParams gp;
    int part_keep[kFlMaxBlocks];
    double part_mm[4 * kFlMaxBlocks];   // per block: min x, max x, min y, max y of its points
};

struct FiltrarArgs {
    const double* y_raw;    // (2,L)
    const double* cnt_raw;  // (L)
    const double* stats_all;  // sharded: per-rank headers hold the new-landmark counts
    const int* n_new_dev;     // single rank: landmarks created this sweep (device word)
    const int* sweep_flags;   // nullable: the sweep's flags ([0], [1] table overflows, [2] labels beyond L)
    int L, lact0, world, stride;
    double cota, thr;
    int max_cells;
    // scratch
    FlState* st;
    double *px, *py, *pc, *nd;
    int *cid, *cell_cnt, *cell_fill, *nn, *lab, *comp, *csize, *isl, *rank;
    // outputs
    double *mapx, *mapy, *counts_new;
    GridParams* gpar;
    int* g_cell;
    LmRec* g_lm;
    int* info;
    int* info_host = nullptr;   // nullable: the host's mapped copy of info[0..2] (k_fl_finalize writes both)
    double* host_map = nullptr; // nullable: [x (L) | y (L) | counters (L)] in the host's mapped memory: k_fl_finalize writes the refined map there as well
    const int* stale = nullptr; // nullable: *stale == stale_epoch = this sweep ran from poses that were not the caller's (SolveSeg::stale)
    int stale_epoch = 0;
    const int* gave_up = nullptr;   // nullable: *gave_up == gave_up_epoch = the wait for the raw map in front of this chain gave up (k_wait_word)
    int gave_up_epoch = 0;
};

__device__ __forceinline__ int block_exscan_1024(int v, int* wsum, int& total) {
    // exclusive scan of one int per thread over the 1024-thread block
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int u = __shfl_up(inc, d, kWave);
        if (lane >= d) inc += u;
    }
    __syncthreads();
    if (lane == kWave - 1) wsum[w] = inc;
    __syncthreads();
    int pre = 0, tot = 0;
    for (int q = 0; q < kFB / kWave; ++q) {
        const int s = wsum[q];
        if (q < w) pre += s;
        tot += s;
    }
    total = tot;
    return pre + inc - v;
}

__device__ __forceinline__ int fl_lact(const FiltrarArgs& a) {   // landmarks in use before the filter
    int lact = a.lact0;
    if (a.world > 1) {
        for (int r = 0; r < a.world; ++r) lact += (int)a.stats_all[(size_t)r * a.stride + 3 * (size_t)a.L];
    } else {
        lact += *a.n_new_dev;
    }
    return min(lact, a.L);
}

// min / max of (x, y) over the block -> part_mm[4 b ..]
__device__ __forceinline__ void fl_block_extent(double x0, double x1, double y0, double y1, double* red, double* out4) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        x0 = fmin(x0, __shfl_xor(x0, d, kWave));
        x1 = fmax(x1, __shfl_xor(x1, d, kWave));
        y0 = fmin(y0, __shfl_xor(y0, d, kWave));
        y1 = fmax(y1, __shfl_xor(y1, d, kWave));
    }
    __syncthreads();
    if (lane == 0) {
        red[4 * w] = x0; red[4 * w + 1] = x1; red[4 * w + 2] = y0; red[4 * w + 3] = y1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < kFB / kWave; ++q) {
            x0 = fmin(x0, red[4 * q]); x1 = fmax(x1, red[4 * q + 1]);
            y0 = fmin(y0, red[4 * q + 2]); y1 = fmax(y1, red[4 * q + 3]);
        }
        out4[0] = x0; out4[1] = x1; out4[2] = y0; out4[3] = y1;
    }
}

// Grid over n points of the given extent: cell edge >= cell_min, at most max_cells cells.
__device__ __forceinline__ GridParams fl_grid_params(double x0, double x1, double y0, double y1, int n, double cell_min, int max_cells) {
    double cell = cell_min > 0.0 ? cell_min : 1.0;
    int nx = 1, ny = 1;
    if (n > 0) {
        // (bounded: with a non-finite coordinate in the map the extent never fits; everything
        // then lands in one cell instead of the kernel spinning)
        for (int tries = 0; tries < 128; ++tries) {
            const double nxd = floor((x1 - x0) / cell) + 1.0, nyd = floor((y1 - y0) / cell) + 1.0;
            if (nxd * nyd <= (double)max_cells) {
                nx = (int)nxd;
                ny = (int)nyd;
                break;
            }
            cell *= 2.0;
        }
        if (!(x1 - x0 < __builtin_huge_val()) || !(y1 - y0 < __builtin_huge_val())) {
            x0 = y0 = 0.0;
            cell = 1.0;
        }
    } else {
        x0 = y0 = 0.0;
    }
    return GridParams{x0, y0, 1.0 / cell, nx, ny};
}

// Extent of all points from the per-block partials, grid parameters, cell counters := 0.
__device__ __forceinline__ void fl_grid_setup(const FiltrarArgs& a, int nb, int n) {
    double x0 = __builtin_huge_val(), x1 = -__builtin_huge_val(), y0 = x0, y1 = x1;
    for (int q = 0; q < nb; ++q) {
        x0 = fmin(x0, a.st->part_mm[4 * q]); x1 = fmax(x1, a.st->part_mm[4 * q + 1]);
        y0 = fmin(y0, a.st->part_mm[4 * q + 2]); y1 = fmax(y1, a.st->part_mm[4 * q + 3]);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) a.st->gp = fl_grid_params(x0, x1, y0, y1, n, a.thr * (1.0 + 1e-9), a.max_cells);
    for (int c = blockIdx.x * kFB + threadIdx.x; c <= a.max_cells; c += gridDim.x * kFB) a.cell_cnt[c] = 0;
}

__global__ __launch_bounds__(kFB) void k_fl_count(FiltrarArgs a, int chunk) {
    __shared__ double red[4 * (kFB / kWave)];
    __shared__ int wsum[kFB / kWave];
    const int lact = fl_lact(a), b = blockIdx.x;
    const int lo = min(b * chunk, lact), hi = min(lo + chunk, lact);
    int keep = 0;
    double x0 = __builtin_huge_val(), x1 = -__builtin_huge_val(), y0 = x0, y1 = x1;
    for (int i = lo + threadIdx.x; i < hi; i += kFB)
        if (a.cnt_raw[i] >= a.cota) {
            ++keep;
            const double x = a.y_raw[i], y = a.y_raw[a.L + i];
            x0 = fmin(x0, x); x1 = fmax(x1, x);
            y0 = fmin(y0, y); y1 = fmax(y1, y);
        }
    int tot;
    (void)block_exscan_1024(keep, wsum, tot);
    fl_block_extent(x0, x1, y0, y1, red, a.st->part_mm + 4 * b);
    if (threadIdx.x == 0) {
        a.st->part_keep[b] = tot;
        if (b == 0) {
            a.st->close = a.st->same = a.st->host = 0;
            a.st->abort = (a.sweep_flags ? (a.sweep_flags[0] | a.sweep_flags[1] | a.sweep_flags[2]) : 0) | ((a.stale && *a.stale == a.stale_epoch) ? 1 : 0) |
                          ((a.gave_up && *a.gave_up == a.gave_up_epoch) ? 1 : 0);
        }
    }
}

__global__ __launch_bounds__(kFB) void k_fl_scatter(FiltrarArgs a, int chunk) {
    __shared__ int wsum[kFB / kWave];
    const int lact = fl_lact(a), b = blockIdx.x, nb = gridDim.x;
    int off = 0, n = 0;
    for (int q = 0; q < nb; ++q) {
        const int v = a.st->part_keep[q];
        if (q < b) off += v;
        n += v;
    }
    const int lo = min(b * chunk, lact), hi = min(lo + chunk, lact);
    for (int base = lo; base < hi; base += kFB) {   // keeps the order (block-uniform trip count)
        const int i = base + threadIdx.x;
        const int keep = (i < hi && a.cnt_raw[i] >= a.cota) ? 1 : 0;
        int tot;
        const int pos = off + block_exscan_1024(keep, wsum, tot);
        if (keep) {
            a.px[pos] = a.y_raw[i];
            a.py[pos] = a.y_raw[a.L + i];
            a.pc[pos] = a.cnt_raw[i];
        }
        off += tot;
        __syncthreads();
    }
    if (b == 0 && threadIdx.x == 0) a.st->n = n;
    fl_grid_setup(a, nb, n);
}

// extent of n points (the refined map after merging) -> per-block partials
__global__ __launch_bounds__(kFB) void k_fl_extent(FiltrarArgs a, const double* __restrict__ x, const double* __restrict__ y,
                                                   const int* __restrict__ n_dev) {
    __shared__ double red[4 * (kFB / kWave)];
    const int n = *n_dev;
    double x0 = __builtin_huge_val(), x1 = -__builtin_huge_val(), y0 = x0, y1 = x1;
    for (int i = blockIdx.x * kFB + threadIdx.x; i < n; i += gridDim.x * kFB) {
        x0 = fmin(x0, x[i]); x1 = fmax(x1, x[i]);
        y0 = fmin(y0, y[i]); y1 = fmax(y1, y[i]);
    }
    fl_block_extent(x0, x1, y0, y1, red, a.st->part_mm + 4 * blockIdx.x);
}
__global__ __launch_bounds__(kFB) void k_fl_setup(FiltrarArgs a, const int* __restrict__ n_dev) { fl_grid_setup(a, gridDim.x, *n_dev); }

__global__ __launch_bounds__(kBlock) void k_fl_cell_count(FiltrarArgs a, const double* __restrict__ x, const double* __restrict__ y,
                                                          const int* __restrict__ n_dev) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= *n_dev) return;
    const GridParams gp = a.st->gp;
    const int c = grid_cell(y[i], gp.gy0, gp.inv, gp.ny) * gp.nx + grid_cell(x[i], gp.gx0, gp.inv, gp.nx);
    a.cid[i] = c;
    atomicAdd(&a.cell_cnt[c], 1);
}

// order inside a cell is arbitrary: every consumer breaks ties on the landmark id explicitly
__global__ __launch_bounds__(kBlock) void k_fl_fill(FiltrarArgs a, const double* __restrict__ x, const double* __restrict__ y,
                                                    const int* __restrict__ n_dev) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= *n_dev || a.st->abort) return;
    const int p = atomicAdd(&a.cell_fill[a.cid[i]], 1);
    a.g_lm[p] = LmRec{x[i], y[i], i, 0, 0, 0};
}

// nearest other survivor (smallest index on ties, like np.argmin over the column)
__global__ __launch_bounds__(kBlock) void k_fl_pairs(FiltrarArgs a) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.st->n || a.st->abort) return;
    const GridParams gp = a.st->gp;
    const double xi = a.px[i], yi = a.py[i];
    const int cx = grid_cell(xi, gp.gx0, gp.inv, gp.nx), cy = grid_cell(yi, gp.gy0, gp.inv, gp.ny);
    const int c0 = max(cx - 1, 0), c1 = min(cx + 1, gp.nx - 1);
    double best = __builtin_huge_val();
    int bid = -1;
    bool same = false;
    for (int ry = max(cy - 1, 0); ry <= min(cy + 1, gp.ny - 1); ++ry)
        for (int p = a.g_cell[ry * gp.nx + c0]; p < a.g_cell[ry * gp.nx + c1 + 1]; ++p) {
            const LmRec c = a.g_lm[p];
            if (c.id == i) continue;
            const double dx = xi - c.x, dy = yi - c.y;
            const double d = sqrt(dx * dx + dy * dy);
            same |= d == 0.0;
            if (d < best || (d == best && c.id < bid)) {
                best = d;
                bid = c.id;
            }
        }
    a.nn[i] = bid;
    a.nd[i] = best;
    if (best < a.thr) atomicAdd(&a.st->close, 1);
    if (same) atomicAdd(&a.st->same, 1);
}

__global__ __launch_bounds__(kFB) void k_fl_finalize(FiltrarArgs a) {
    // (info_host: the host's mapped copy of the three words, written here instead of copied behind the chain)
    if (a.st->abort) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            a.info[0] = 0;
            a.info[1] = 3;   // nothing touched: the host re-runs the sweep
            a.info[2] = 0;
            if (a.info_host) { a.info_host[0] = 0; a.info_host[1] = 3; a.info_host[2] = 0; }
        }
        return;
    }
    const int n = a.st->n;
    const bool host = n == 0 || a.st->same > 0;
    const bool merge = a.st->close > 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.info[0] = n;
        a.info[1] = host ? 2 : (merge ? 1 : 0);
        a.info[2] = a.st->close;
        if (a.info_host) { a.info_host[0] = n; a.info_host[1] = host ? 2 : (merge ? 1 : 0); a.info_host[2] = a.st->close; }
    }
    if (host || merge) {
        // the search structures hold the raw survivors (a consistent grid: k_neigh_table, queued behind,
        // must never read stale parameters); the merge chain / the host path rebuild them
        if (blockIdx.x == 0 && threadIdx.x == 0) *a.gpar = a.st->gp;
        return;
    }
    // refined map = survivors; the count-weighted mean of a single term is (y*n)/n (:258-260).
    // The grid cells were assigned from the raw coordinates, which differ from the refined
    // ones by an ulp at most -- far inside the 1e-9 slack of the cell edge over dist_thr --
    // so only the table's coordinates are rewritten.
    const int stride = gridDim.x * kFB, t0 = blockIdx.x * kFB + threadIdx.x;
    for (int i = t0; i < a.L; i += stride) {
        double mx = 0.0, my = 0.0, c = 0.0;
        if (i < n) {
            c = a.pc[i];
            mx = (a.px[i] * c) / c;
            my = (a.py[i] * c) / c;
            a.mapx[i] = mx;
            a.mapy[i] = my;
        }
        a.counts_new[i] = c;
        if (a.host_map) {   // (the drop-in call: what the caller gets back, without a copy launch behind the chain)
            a.host_map[i] = mx;
            a.host_map[(size_t)a.L + i] = my;
            a.host_map[2 * (size_t)a.L + i] = c;
        }
    }
    for (int p = t0; p < n; p += stride) {
        const int id = a.g_lm[p].id;
        const double c = a.pc[id];
        a.g_lm[p].x = (a.px[id] * c) / c;
        a.g_lm[p].y = (a.py[id] * c) / c;
    }
    if (t0 == 0) *a.gpar = a.st->gp;
}

// ---- merging (rare) ----
// One thread per survivor.  A survivor with a neighbour closer than dist_thr collects its component
// of the nearest-neighbour graph (edges i -> nn[i], followed both ways; the members are within
// dist_thr of each other along the chain, so the reverse edges are found in the 3x3 cells around a
// member).  The smallest member replays the reference's loop over its component:
//   for i in close (ascending): c[c == c[b[i]]] = c[i]          (scripts/ICM_SLAM_tools.py:246-249)
// -- components never interact, so the global sequential loop factorises over them.
__global__ __launch_bounds__(kBlock) void k_fl_components(FiltrarArgs a) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int n = a.st->n;
    if (i >= n) return;
    if (!(a.nd[i] < a.thr)) {   // nobody close: its own landmark
        a.lab[i] = i;
        a.csize[i] = 1;
        a.comp[(size_t)i * kCompStride] = i;
        return;
    }
    const GridParams gp = a.st->gp;
    int mem[kCompMax + 1];
    int cnt = 1;
    mem[0] = i;
    bool over = false;
    for (int k = 0; k < cnt && !over; ++k) {
        const int u = mem[k];
        auto add = [&](int v) {
            for (int q = 0; q < cnt; ++q)
                if (mem[q] == v) return;
            if (cnt > kCompMax - 1) {
                over = true;
                return;
            }
            mem[cnt++] = v;
        };
        add(a.nn[u]);
        const double xu = a.px[u], yu = a.py[u];
        const int cx = grid_cell(xu, gp.gx0, gp.inv, gp.nx), cy = grid_cell(yu, gp.gy0, gp.inv, gp.ny);
        const int c0 = max(cx - 1, 0), c1 = min(cx + 1, gp.nx - 1);
        for (int ry = max(cy - 1, 0); ry <= min(cy + 1, gp.ny - 1) && !over; ++ry)
            for (int p = a.g_cell[ry * gp.nx + c0]; p < a.g_cell[ry * gp.nx + c1 + 1] && !over; ++p) {
                const int v = a.g_lm[p].id;
                if (v != u && a.nd[v] < a.thr && a.nn[v] == u) add(v);
            }
    }
    if (over) {
        a.st->host = 1;
        return;
    }
    for (int k = 1; k < cnt; ++k) {   // ascending
        const int v = mem[k];
        int q = k - 1;
        while (q >= 0 && mem[q] > v) {
            mem[q + 1] = mem[q];
            --q;
        }
        mem[q + 1] = v;
    }
    if (mem[0] != i) return;   // the smallest member leads
    int c[kCompMax + 1];
    for (int k = 0; k < cnt; ++k) c[k] = mem[k];
    for (int k = 0; k < cnt; ++k) {
        int kb = 0;
        for (int q = 0; q < cnt; ++q)
            if (mem[q] == a.nn[mem[k]]) kb = q;
        const int from = c[kb], to = c[k];
        if (from != to)
            for (int q = 0; q < cnt; ++q)
                if (c[q] == from) c[q] = to;
    }
    const int l = c[0];
    for (int k = 0; k < cnt; ++k) {
        a.lab[mem[k]] = l;
        a.csize[mem[k]] = 0;
        a.comp[(size_t)l * kCompStride + k] = mem[k];
    }
    a.csize[l] = cnt;
}

__global__ __launch_bounds__(kBlock) void k_fl_label_flags(FiltrarArgs a) {   // isl[l] = label l survives
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < a.st->n) a.isl[i] = a.csize[i] > 0 ? 1 : 0;
}

// renumbered, count-weighted means (:251-260); rank = exclusive scan of isl
__global__ __launch_bounds__(kBlock) void k_fl_gather(FiltrarArgs a) {
    const int l = blockIdx.x * kBlock + threadIdx.x;
    const int n = a.st->n;
    if (a.st->host) {   // a component beyond kCompMax: labels are incomplete, the host routine takes over;
        if (l == 0) a.st->n_ref = 0;   // the grid chain queued behind finds nothing to do
        return;
    }
    const int n_ref = a.rank[n];
    if (l == 0) a.st->n_ref = n_ref;
    if (l < a.L && l >= n_ref) a.counts_new[l] = 0.0;
    if (l >= n || a.csize[l] == 0) return;
    const int r = a.rank[l], k = min(a.csize[l], kCompMax);
    double cs = 0.0, sx = 0.0, sy = 0.0;
    for (int q = 0; q < k; ++q) {   // members in ascending order, like the boolean mask
        const int m = a.comp[(size_t)l * kCompStride + q];
        const double c = a.pc[m];
        cs += c;
        sx += a.px[m] * c;
        sy += a.py[m] * c;
    }
    a.mapx[r] = sx / cs;
    a.mapy[r] = sy / cs;
    a.counts_new[r] = cs;
}

__global__ void k_fl_finalize_merged(FiltrarArgs a) {
    a.info[0] = a.st->n_ref;
    a.info[1] = a.st->host ? 2 : 0;
    *a.gpar = a.st->gp;
}

// Target per kept beam: y[:, c] of scripts/ICM_ROS.py:152 (parity tests, per-beam energy).
__global__ __launch_bounds__(kBlock) void k_beam_targets(int nloc, const int* __restrict__ boff,
                                                         const int* __restrict__ ent_off, const int* __restrict__ bloc,
                                                         const double2* __restrict__ tgt,
                                                         double* __restrict__ btx, double* __restrict__ bty) {
    const int lane = lane_id();
    const int tl = blockIdx.x * kWavesPerBlock + wave_in_block();
    if (tl >= nloc) return;
    const int e0 = ent_off[tl];
    for (int j = boff[tl] + lane; j < boff[tl + 1]; j += kWave) {
        const int e = e0 + bloc[j];
        const double2 t = tgt[e];
        btx[j] = t.x;
        bty[j] = t.y;
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
    int per_beam;         // 1: energy summed beam by beam (needs btx/bty); 0: entry form
    const int* boff;
    const double *bx, *by, *btx, *bty;
    const int *ent_off, *e_k;
    const double2 *e_b, *tgt;
    const double *pose_c, *pose_m;
    double dt, R0, R1, R2, Q0, Q1, cte;
    double* diag;         // optional (T,3): f, nit, nfev per pose
    double* rot;          // optional (nloc,2): (cos, sin)(theta - pi/2) of the solved pose, the table phase A and the
                          // moment kernel of the NEXT sweep read (k_pose_rot's values: whoever writes a pose writes its pair)
    const double* odo_cs; // (T,2): (cos, sin) of the odometry headings (k_odo_trig)
    double* cs;           // (T,2): (cos, sin)(theta) of every pose as it stands (k_pose_rot / store_pose_tables / the headers)
    int epoch;            // of the one-launch solve (k_solve_m_fused): its flags and deferral stamps hold the epoch that set them
    double* xh;           // nullable (k_solve_m_fused only): the caller's own (3,T) pose array, registered host memory
                          // (icm_pin_host): the even waves write every pose there as well, over PCIe, while the launch's
                          // chains run -- the drop-in call then needs no download behind the sweep
    // ghost pose of a shard (rank > 0): the lower neighbour's last pose t_begin - 1, solved redundantly so that the
    // shard's first even pose needs nothing from another rank in the middle of the sweep (local index tl = -1)
    int ghost_n;          // its kept beams (0: none)
    const double* ghost_m;   // its 17 moment sums, contiguous (k_ghost_moments)
};

// Result of one pose solve -> x (write-through when other waves of the same launch wait for it) and the rotation table.
__device__ __forceinline__ void store_pose_xyz(const SolveArgs& a, int tg, const double res[3], bool publish) {
    if (publish) {
        __hip_atomic_store(&a.x[3 * (size_t)tg], res[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a.x[3 * (size_t)tg + 1], res[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a.x[3 * (size_t)tg + 2], res[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        a.x[3 * (size_t)tg] = res[0];
        a.x[3 * (size_t)tg + 1] = res[1];
        a.x[3 * (size_t)tg + 2] = res[2];
    }
}
// The caller's own (3,T) array (registered host memory): an EVEN pose's lane writes the pair (tg - 1, tg) of each row --
// its odd neighbour is final by then -- so that a wave's stores are whole lines over PCIe (one 8-byte store per pose and
// row took the solve launch from 0.08 to 0.3 ms); the last pose, when it is odd, goes with its even neighbour too.
__device__ __forceinline__ void mirror_pose_pair(const SolveArgs& a, int tg, double r0, double r1, double r2) {
    struct __attribute__((packed, aligned(8))) Pair { double lo, hi; };   // (tg - 1 is odd: 8-byte aligned; one 16-byte store)
    const double* lo = a.x + 3 * (size_t)(tg - 1);
    const double r[3] = {r0, r1, r2};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double* row = a.xh + (size_t)i * a.T;
        *reinterpret_cast<Pair*>(row + tg - 1) = Pair{lo[i], r[i]};
        if (tg + 2 == a.T) row[tg + 1] = lo[6 + i];
    }
}
// The pairs kept beside a pose: (cos, sin)(theta - pi/2) for phase A / the moment kernel of the next sweep, (cos, sin)(theta)
// for the next solves.  Two generic sincos, ~260 vector instructions: in the one-launch solve they come AFTER the wave has
// published its poses (nobody in this launch reads them), off the odd -> even hand-off.
__device__ __forceinline__ void store_pose_tables(const SolveArgs& a, int tg, double th) {
    if (a.rot && tg >= a.t_begin) {   // (the ghost's pair is recomputed from its owner's value: k_halo_from_headers)
        double ct, st;
        pose_rot(th, ct, st);
        a.rot[2 * (size_t)(tg - a.t_begin)] = ct;
        a.rot[2 * (size_t)(tg - a.t_begin) + 1] = st;
    }
    if (a.cs) {
        a.cs[2 * (size_t)tg] = cos(th);
        a.cs[2 * (size_t)tg + 1] = sin(th);
    }
}
__device__ __forceinline__ void store_pose(const SolveArgs& a, int tg, const double res[3], bool publish) {
    store_pose_xyz(a, tg, res, publish);
    store_pose_tables(a, tg, res[2]);
}

__device__ __forceinline__ void load3(const double* __restrict__ a, int T, int t, double o[3]) {
    o[0] = a[t]; o[1] = a[(size_t)T + t]; o[2] = a[2 * (size_t)T + t];
}

// Solve pose tg (global index); `prev` = x[:,tg-1] as it stands now.
template <bool PER_BEAM>
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
    double sx, sy, st;
    if (!last) {  // minimizar_xn start (scripts/ICM_ROS.py:217)
        sx = (prev[0] + xp[0]) / 2.0; sy = (prev[1] + xp[1]) / 2.0; st = (prev[2] + xp[2]) / 2.0;
    } else {      // minimizar_x start g(x_{t-1}, u_{t-1}) (scripts/ICM_ROS.py:258)
        sx = c.gax; sy = c.gay; st = c.gat;
    }
    double out[6];
    if (PER_BEAM) {
        Items it{a.bx + j0, a.by + j0, a.btx + j0, a.bty + j0, nullptr, 0.0, 0.0, 0.0, n, nullptr, nullptr};
        nelder_mead3([&](double px, double py, double th) { return pose_energy(c, it, px, py, th, lane); }, sx, sy, st, out);
    } else {
        const int e0 = a.ent_off[tl], ne = a.ent_off[tl + 1] - e0;
        Items it{nullptr, nullptr, nullptr, nullptr, a.e_k + e0,
                 a.pose_c[3 * (size_t)tl], a.pose_c[3 * (size_t)tl + 1], a.pose_c[3 * (size_t)tl + 2], ne,
                 a.e_b + e0, a.tgt + e0};
        // the usual case: at most one entry per lane, held in registers for the whole solve
        RegItem r{0.0, 0.0, 0.0, 0.0, 0.0};
        const bool inreg = ne <= kWave;
        if (inreg && lane < ne) {
            r.k = (double)it.kw[lane];
            r.bx = it.b2[lane].x; r.by = it.b2[lane].y; r.tx = it.t2[lane].x; r.ty = it.t2[lane].y;
        }
        nelder_mead3([&](double px, double py, double th) {
            const double hh = inreg ? obs_energy_reg(c, it, r, px, py, th) : obs_energy(c, it, px, py, th, lane);
            return pose_energy_with(c, hh, px, py, th);
        }, sx, sy, st, out);
    }
    res[0] = out[0]; res[1] = out[1]; res[2] = out[2];
    if (a.diag && lane == 0) {
        a.diag[3 * (size_t)tg] = out[3];
        a.diag[3 * (size_t)tg + 1] = out[4];
        a.diag[3 * (size_t)tg + 2] = out[5];
    }
}

// Moments of the moment-form energy (icm_device.hpp, PoseMoments).  A pose has ~40 entries, so
// one DPP row (16 lanes) per pose: four poses per wavefront, lanes stride over the pose's
// entries, 14 row reductions (4 DPP steps each, no cross-row traffic).  Stored [17][nloc] so
// that the lane-per-pose solver reads them coalesced.
__global__ __launch_bounds__(kBlock) void k_pose_moments(const double* __restrict__ x, const double* __restrict__ x0,
                                                         int t_begin, int nloc, const int* __restrict__ ent_off,
                                                         const int* __restrict__ e_k, const double2* __restrict__ e_wr,
                                                         const double2* __restrict__ tgt, const double* __restrict__ pose_c,
                                                         double* __restrict__ pose_m) {
    const int sub = threadIdx.x & 15;
    const int tl = (blockIdx.x * kBlock + threadIdx.x) >> 4;
    const bool live = tl < nloc;
    double px = 0.0, py = 0.0, th = 0.0;
    int e0 = 0, e1 = 0;
    if (live) {
        pose_of(x, x0, t_begin + tl, px, py, th);
        e0 = ent_off[tl];
        e1 = ent_off[tl + 1];
    }
    double m[kMomentCount];
#pragma unroll
    for (int q = 0; q < kMomentCount; ++q) m[q] = 0.0;
    for (int e = e0 + sub; e < e1; e += 16) {
        const double2 w = e_wr[e], tg = tgt[e];
        const double k = (double)e_k[e], wx = w.x, wy = w.y;
        const double rx = (px + wx) - tg.x, ry = (py + wy) - tg.y;
        m[0] += k; m[1] += k * wx; m[2] += k * wy; m[3] += k * rx; m[4] += k * ry;
        m[5] += k * wx * wx; m[6] += k * wy * wy; m[7] += k * wx * wy;
        m[8] += k * wx * rx; m[9] += k * wy * rx; m[10] += k * wx * ry; m[11] += k * wy * ry;
        m[12] += k * rx * rx; m[13] += k * ry * ry;
    }
#pragma unroll
    for (int q = 0; q < kMomentCount; ++q) m[q] = row_sum16(m[q]);
    if (live && sub == 0) {
#pragma unroll
        for (int q = 0; q < kMomentCount; ++q) pose_m[(size_t)q * nloc + tl] = m[q];
        pose_m[(size_t)14 * nloc + tl] = pose_c[3 * (size_t)tl];
        pose_m[(size_t)15 * nloc + tl] = pose_c[3 * (size_t)tl + 1];
        pose_m[(size_t)16 * nloc + tl] = pose_c[3 * (size_t)tl + 2];
    }
}

// The same moments straight from the staged entries and the hierarchical prefixes (k_chunk_l1 ..
// k_rec_push): target = (sums before the chunk + sums inside the chunk through this pose) / n.
// Also folds in what k_compact does for the sort-based pipeline (rotated mean body point,
// scatter term), so the staged entries are read exactly once more.
__global__ __launch_bounds__(kBlock) void k_pose_moments_h(const double* __restrict__ x, const double* __restrict__ x0,
                                                           int t_begin, int nloc, const int* __restrict__ boff,
                                                           const int* __restrict__ nent, const int* __restrict__ ent_off,
                                                           const unsigned short* __restrict__ st_k, const double* __restrict__ st_sbx,
                                                           const double* __restrict__ st_sby,
                                                           const double* __restrict__ pose_s2,
                                                           const double* __restrict__ pre_x, const double* __restrict__ pre_y,
                                                           const unsigned* __restrict__ pre_n, int chunk_poses,
                                                           const double* __restrict__ off_x, const double* __restrict__ off_y,
                                                           const double* __restrict__ off_n, double* __restrict__ pose_m,
                                                           double2* __restrict__ tgt_out, int tl_begin = 0, int tl_end = -1,
                                                           const double* __restrict__ rot = nullptr) {
    const int sub = threadIdx.x & 15;
    const int tl = tl_begin + ((blockIdx.x * kBlock + threadIdx.x) >> 4);   // poses [tl_begin, tl_end)
    const bool live = tl < (tl_end < 0 ? nloc : tl_end);
    double px = 0.0, py = 0.0, th = 0.0;
    int j0 = 0, e0 = 0, n = 0;
    if (live) {
        pose_of(x, x0, t_begin + tl, px, py, th);
        j0 = boff[tl];      // (the pose's place in the staging area: st_off; the per-entry prefixes are at the same places)
        e0 = j0;
        n = nent[tl];
    }
    const int et = (live && tgt_out) ? ent_off[tl] : 0;   // the association dump keeps the scan's (pose-major, gap-free) numbering
    double ct, st;
    if (rot) {
        ct = live ? rot[2 * (size_t)tl] : 1.0;
        st = live ? rot[2 * (size_t)tl + 1] : 0.0;
    } else {
        pose_rot(th, ct, st);
    }
    double m[kMomentCount];
#pragma unroll
    for (int q = 0; q < kMomentCount; ++q) m[q] = 0.0;
    double mxx = 0.0, mxy = 0.0, myy = 0.0;
    for (int q = sub; q < n; q += 16) {
        const double k = (double)st_k[j0 + q], sbx = st_sbx[j0 + q], sby = st_sby[j0 + q];
        const unsigned nw = pre_n[e0 + q];   // (beams inside the chunk through this pose) << 8 | record slot
        const int r = (tl / chunk_poses) * kT1 + (int)(nw & 255u);
        const double sx = off_x[r] + pre_x[e0 + q], sy = off_y[r] + pre_y[e0 + q], sn = off_n[r] + (double)(nw >> 8);
        const double tx = sx / sn, ty = sy / sn;
        if (tgt_out) tgt_out[et + q] = make_double2(tx, ty);   // association dump (icm_set_debug)
        const double wx = (ct * sbx - st * sby) / k, wy = (st * sbx + ct * sby) / k;
        const double rx = (px + wx) - tx, ry = (py + wy) - ty;
        m[0] += k; m[1] += k * wx; m[2] += k * wy; m[3] += k * rx; m[4] += k * ry;
        m[5] += k * wx * wx; m[6] += k * wy * wy; m[7] += k * wx * wy;
        m[8] += k * wx * rx; m[9] += k * wy * rx; m[10] += k * wx * ry; m[11] += k * wy * ry;
        m[12] += k * rx * rx; m[13] += k * ry * ry;
        mxx += sbx * sbx / k;
        mxy += sbx * sby / k;
        myy += sby * sby / k;
    }
#pragma unroll
    for (int q = 0; q < kMomentCount; ++q) m[q] = row_sum16(m[q]);
    mxx = row_sum16(mxx);
    mxy = row_sum16(mxy);
    myy = row_sum16(myy);
    if (live && sub == 0) {
#pragma unroll
        for (int q = 0; q < kMomentCount; ++q) pose_m[(size_t)q * nloc + tl] = m[q];
        pose_m[(size_t)14 * nloc + tl] = pose_s2[3 * (size_t)tl] - mxx;
        pose_m[(size_t)15 * nloc + tl] = pose_s2[3 * (size_t)tl + 1] - mxy;
        pose_m[(size_t)16 * nloc + tl] = pose_s2[3 * (size_t)tl + 2] - myy;
    }
}

// Moments of a shard's GHOST pose (the lower neighbour's last pose t_begin - 1, which this rank solves as well:
// SolveSeg).  Its kept beams were associated and grouped by a one-pose launch of k_assoc_group into the ghost's own
// staging arrays; its targets are the running means through the ghost pose INCLUSIVE, i.e. the totals over all lower
// ranks -- exactly what k_stats_prefix left in off_* for the landmarks of mapa_viejo.  A landmark the ghost pose itself
// created is the last new landmark of the rank below: its one-pose statistics sit in that rank's slot of the exchange
// buffer at column lact0 + n_new - 1.  Same sums, same formulas as k_pose_moments_h; one wave.
__global__ __launch_bounds__(kWave) void k_ghost_moments(const double* __restrict__ x, int tg, const int* __restrict__ nent,
                                                        const int* __restrict__ st_off, const int* __restrict__ st_label,
                                                        const unsigned short* __restrict__ st_k, const double* __restrict__ st_sbx,
                                                        const double* __restrict__ st_sby, const double* __restrict__ s2,
                                                        const double* __restrict__ rot, const double* __restrict__ off_x,
                                                        const double* __restrict__ off_y, const double* __restrict__ off_n,
                                                        const double* __restrict__ stats_below, int L, int lact0,
                                                        double* __restrict__ gm, const int* __restrict__ gflags, int* __restrict__ flags) {
    const int lane = lane_id();
    const int n = nent[0], j0 = st_off[0];
    const double px = x[3 * (size_t)tg], py = x[3 * (size_t)tg + 1];
    const double ct = rot[0], st = rot[1];
    double m[kMomentCount];
#pragma unroll
    for (int q = 0; q < kMomentCount; ++q) m[q] = 0.0;
    double mxx = 0.0, mxy = 0.0, myy = 0.0;
    bool bad = false;
    // (the owner's order of additions, k_pose_moments_h: sixteen lanes stride over the entries, then a DPP row sum -- the other
    // 48 lanes idle; what still differs from the owner is the last bits of the TARGETS, whose running sums the two ranks add
    // up in different associations)
    for (int q = lane; q < n && lane < 16; q += 16) {
        const int lab = st_label[j0 + q];
        const double k = (double)st_k[j0 + q], sbx = st_sbx[j0 + q], sby = st_sby[j0 + q];
        double sx, sy, sn;
        if (lab >= 0) {
            sx = off_x[lab]; sy = off_y[lab]; sn = off_n[lab];
        } else {   // the landmark this very pose created: the last new one of the rank below
            const int c = lact0 + (int)stats_below[3 * (size_t)L] - 1;
            const bool okc = c >= lact0 && c < L;
            bad |= !okc;
            sx = okc ? stats_below[c] : 0.0; sy = okc ? stats_below[(size_t)L + c] : 0.0; sn = okc ? stats_below[2 * (size_t)L + c] : 1.0;
        }
        const double tx = sx / sn, ty = sy / sn;
        const double wx = (ct * sbx - st * sby) / k, wy = (st * sbx + ct * sby) / k;
        const double rx = (px + wx) - tx, ry = (py + wy) - ty;
        m[0] += k; m[1] += k * wx; m[2] += k * wy; m[3] += k * rx; m[4] += k * ry;
        m[5] += k * wx * wx; m[6] += k * wy * wy; m[7] += k * wx * wy;
        m[8] += k * wx * rx; m[9] += k * wy * rx; m[10] += k * wx * ry; m[11] += k * wy * ry;
        m[12] += k * rx * rx; m[13] += k * ry * ry;
        mxx += sbx * sbx / k;
        mxy += sbx * sby / k;
        myy += sby * sby / k;
    }
#pragma unroll
    for (int q = 0; q < kMomentCount; ++q) m[q] = row_sum16(m[q]);
    mxx = row_sum16(mxx);
    mxy = row_sum16(mxy);
    myy = row_sum16(myy);
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < kMomentCount; ++q) gm[q] = m[q];
        gm[14] = s2[0] - mxx;
        gm[15] = s2[1] - mxy;
        gm[16] = s2[2] - myy;
    }
    // (the ghost's tables can only overflow where its owner's did, and the owner's flag travelled with its statistics;
    // a label column out of range likewise means the labels exceed L on every rank)
    if (__ballot(bad) != 0ull || gflags[0]) {
        if (lane == 0) flags[0] = 1;
    }
}

// One LANE (QUAD: one DPP quad, role = lane & 3) solves pose tg with the moment-form energy.
// `prev` = x[:,tg-1] as it stands now.
// Everything a pose solve reads that the launch it runs in does not write: loaded BEFORE an even wave of the one-launch
// solve starts to wait for its odd neighbours, so that what is left behind the wait is the two neighbour poses, one
// sincos and the folding.
struct PoseIn {
    int n;                     // kept beams of the pose (0: the midpoint rule, nothing else is loaded)
    double ua0, ua1, ut0, ut1; // u[:, t-1], u[:, t]
    double oa[3], ot[3], op[3];
    double coa, soa, cot, sot; // (cos, sin) of the odometry headings t-1 and t (k_odo_trig)
    double pm[17];             // the moment sums
    double pox, poy, tho, co, so;   // the pose's previous-sweep value and its (cos, sin) (the cs table)
};

__device__ __forceinline__ void load_pose_in(const SolveArgs& a, int tg, PoseIn& in) {
    const int tl = tg - a.t_begin;   // (-1: the shard's ghost pose)
    in.n = tl < 0 ? a.ghost_n : a.boff[tl + 1] - a.boff[tl];
    if (in.n == 0) return;
    const bool last = tg + 1 >= a.T;
    in.ua0 = a.u[tg - 1]; in.ua1 = a.u[(size_t)a.T + tg - 1];
    load3(a.odo, a.T, tg - 1, in.oa);
    load3(a.odo, a.T, tg, in.ot);
    in.coa = a.odo_cs[2 * (size_t)(tg - 1)]; in.soa = a.odo_cs[2 * (size_t)(tg - 1) + 1];
    in.ut0 = in.ut1 = 0.0; in.op[0] = in.op[1] = in.op[2] = 0.0; in.cot = 1.0; in.sot = 0.0;
    if (!last) {
        in.ut0 = a.u[tg]; in.ut1 = a.u[(size_t)a.T + tg];
        load3(a.odo, a.T, tg + 1, in.op);
        in.cot = a.odo_cs[2 * (size_t)tg]; in.sot = a.odo_cs[2 * (size_t)tg + 1];
    }
    const double* pm = tl < 0 ? a.ghost_m : a.pose_m + tl;
    const size_t st_ = tl < 0 ? (size_t)1 : (size_t)a.nloc;
#pragma unroll
    for (int q = 0; q < 17; ++q) in.pm[q] = pm[q * st_];
    // expansion point = this pose's previous-sweep value (still in x: nobody else writes it)
    in.pox = a.x[3 * (size_t)tg]; in.poy = a.x[3 * (size_t)tg + 1]; in.tho = a.x[3 * (size_t)tg + 2];
    in.co = a.cs[2 * (size_t)tg]; in.so = a.cs[2 * (size_t)tg + 1];
}

// FOLD: the Nelder-Mead evaluates the folded form only (pose_energy_fold_only) and the function returns false -- res
// then means nothing -- as soon as one evaluation it uses left the form's validity range: the caller repeats the
// solve with FOLD = false (folded where valid, term by term elsewhere).
// prev = x[:, tg-1] as it stands now; prev_in_table: its (cos, sin) are in the cs table (it was written before this
// launch began) -- else they are formed here.
template <bool QUAD, bool FOLD = false>
__device__ __forceinline__ bool solve_pose_in(const SolveArgs& a, int tg, const PoseIn& in, const double prev[3], bool prev_in_table,
                                              double& r0, double& r1, double& r2, int role = 0) {
    const bool last = tg + 1 >= a.T;
    if (in.n == 0) {  // no beams (scripts/ICM_ROS.py:143-147)
        const double* nx = a.x + 3 * (size_t)(tg + 1);
        const double p0 = tg == 1 ? a.x0[0] : prev[0], p1 = tg == 1 ? a.x0[1] : prev[1], p2 = tg == 1 ? a.x0[2] : prev[2];
        r0 = (p0 + nx[0]) / 2.0;
        r1 = (p1 + nx[1]) / 2.0;
        r2 = (p2 + nx[2]) / 2.0;
        return true;
    }
    SolveCtx c;
    c.dt = a.dt; c.R0 = a.R0; c.R1 = a.R1; c.R2 = a.R2; c.Q0 = a.Q0; c.Q1 = a.Q1; c.cte = a.cte;
    double xp[3] = {0, 0, 0};
    if (!last) {
        xp[0] = a.x[3 * (size_t)(tg + 1)]; xp[1] = a.x[3 * (size_t)(tg + 1) + 1]; xp[2] = a.x[3 * (size_t)(tg + 1) + 2];
    }
    double ca, sa;
    if (prev_in_table) {
        ca = a.cs[2 * (size_t)(tg - 1)];
        sa = a.cs[2 * (size_t)(tg - 1) + 1];
    } else {
        ca = cos(prev[2]);
        sa = sin(prev[2]);
    }
    const double ua[2] = {in.ua0, in.ua1}, ut[2] = {in.ut0, in.ut1};
    make_ctx_t(c, !last, prev, xp, ua, ut, in.oa, in.ot, in.op, ca, sa, in.coa, in.soa, in.cot, in.sot);
    PoseMoments m;
    m.S = in.pm[0]; m.Swx = in.pm[1]; m.Swy = in.pm[2]; m.Srx = in.pm[3]; m.Sry = in.pm[4];
    m.Swxx = in.pm[5]; m.Swyy = in.pm[6]; m.Swxy = in.pm[7]; m.Swxrx = in.pm[8]; m.Swyrx = in.pm[9];
    m.Swxry = in.pm[10]; m.Swyry = in.pm[11]; m.Srxx = in.pm[12]; m.Sryy = in.pm[13];
    m.cxx = in.pm[14]; m.cxy = in.pm[15]; m.cyy = in.pm[16];
    finish_moments(c, m);
    m.pox = in.pox; m.poy = in.poy; m.tho = in.tho; m.co = in.co; m.so = in.so;
    PoseFold f;
    make_fold(c, m, f);
    double sx, sy, st;
    if (!last) {
        sx = (prev[0] + xp[0]) / 2.0; sy = (prev[1] + xp[1]) / 2.0; st = (prev[2] + xp[2]) / 2.0;
    } else {
        sx = c.gax; sy = c.gay; st = c.gat;
    }
    double out[6];
    if (FOLD) {
        bool left = false;    // an evaluation since the Nelder-Mead last asked left the folded form's range
        const PinnedTrigK trig;
        auto ef = [&](double px, double py, double th) {
            bool ok;
            const double e = pose_energy_fold_only(f, px, py, th, ok, trig);
            left |= !ok;
            return e;
        };
        if (QUAD) {   // (any of the quad's four points: a point the iteration discards may ask for the second solve too -- same result)
            auto stopq = [&]() { const bool l = quad_any(left); left = false; return l; };
            if (nelder_mead3_quad(ef, sx, sy, st, role, out, stopq)) return false;
        } else {
            auto stop = [&]() { const bool l = left; left = false; return l; };
            if (nelder_mead3(ef, sx, sy, st, out, stop)) return false;
        }
    } else if (QUAD) {
        nelder_mead3_quad([&](double px, double py, double th) { return pose_energy_moments(c, m, f, px, py, th); }, sx, sy, st, role, out);
    } else {
        nelder_mead3([&](double px, double py, double th) { return pose_energy_moments(c, m, f, px, py, th); }, sx, sy, st, out);
    }
    r0 = out[0]; r1 = out[1]; r2 = out[2];
    if (a.diag && role == 0) {
        a.diag[3 * (size_t)tg] = out[3];
        a.diag[3 * (size_t)tg + 1] = out[4];
        a.diag[3 * (size_t)tg + 2] = out[5];
    }
    return true;
}

// load + solve in one go (the launches in which nothing waits between the two)
template <bool QUAD, bool FOLD = false>
__device__ __forceinline__ bool solve_pose_moments(const SolveArgs& a, int tg, const double prev[3], bool prev_in_table, double res[3],
                                                   int role = 0) {
    PoseIn in;
    load_pose_in(a, tg, in);
    return solve_pose_in<QUAD, FOLD>(a, tg, in, prev, prev_in_table, res[0], res[1], res[2], role);
}

// A time segment of the sequence solved by one launch: poses [t0, t1).  shift = 0 for the segment
// that starts at pose 0 (which is never solved), 1 for a segment that starts at an even pose > 0:
//   odd poses   o_j = t0 + 1 + 2 j
//   even poses  e_j = t0 + 2 j + 2 (1 - shift)      -> e_j reads o_{j - shift} and o_{j - shift + 1}
// (the first even pose of a shift-1 segment reads the last pose of the segment before, which the
// stream has finished by then).  Splitting the sequence at an even pose keeps the red-black order:
// every segment's odd poses read only old even poses, its even poses only finished odd ones.
// A SHARD [a, b) of a multi-rank job (a even, a > 0) is the segment t0 = a - 2, shift = 0: its first odd pose
// o_0 = a - 1 is the GHOST -- the lower neighbour's last pose, solved here as well from replicated inputs (its
// kept beams, its moments from k_ghost_moments, its neighbours' previous-sweep values out of the statistics
// header) -- so that the shard's first even pose a finds its odd neighbour in this launch and the sweep needs
// no exchange between its two colours (scripts/ICM_ROS.py:141-158: pose t reads t - 1 and t + 1 only).
struct SolveSeg {
    int t0, t1, shift;
    const int* abort;   // nullable: the sweep's flags; any of [0..2] set = leave the poses alone
    const int* stale = nullptr;   // nullable: *stale == stale_epoch = the device poses this sweep started from were not the
    int stale_epoch = 0;          // caller's (k_x_compare, the drop-in call without an upload): leave the poses alone
};
__device__ __forceinline__ bool seg_aborted(const SolveSeg& g) {   // (final before the launch began: uniform over the grid)
    return (g.abort && (g.abort[0] | g.abort[1] | g.abort[2])) || (g.stale && *g.stale == g.stale_epoch);
}

__device__ __forceinline__ int seg_pose(const SolveSeg& g, bool even, int j) {
    return g.t0 + 2 * j + (even ? 2 * (1 - g.shift) : 1);
}

// Red-black half sweep over a segment, moment form: one LANE per pose of the colour (64 poses per wave).
__global__ __launch_bounds__(kBlock) void k_solve_m_colour(SolveArgs a, SolveSeg g, int colour) {
    if (seg_aborted(g)) return;
    const int lane = lane_id();
    const int j = (blockIdx.x * kWavesPerBlock + wave_in_block()) * kWave + lane;
    const int tg = seg_pose(g, colour == 0, j);
    if (tg >= g.t1) return;
    double prev[3] = {a.x[3 * (size_t)(tg - 1)], a.x[3 * (size_t)(tg - 1) + 1], a.x[3 * (size_t)(tg - 1) + 2]};
    double res[3];
    solve_pose_moments<false>(a, tg, prev, true, res);   // (prev was written before this launch began: its pair is in the table)
    store_pose(a, tg, res, false);
}

// Both half sweeps of a red-black sweep in ONE launch.  An even pose reads only its
// two odd neighbours, so an even wave need not wait for the slowest odd pose of the whole
// sequence (which is what a kernel boundary between the colours does) but only for the two odd
// waves that hold its poses' neighbours.  Waves [0, nw) solve the odd poses and publish a
// per-wave flag; waves [nw, 2 nw) poll the two flags they depend on, then solve the even poses.
//
// Forward progress does NOT rest on dispatch order: an even wave polls at most `spin_limit`
// times; if its flags have not arrived by then it stamps `deferred[]`, touches nothing and is done
// (freeing its slot for whatever odd waves are still waiting to be dispatched).  Every wave ends
// with one atomic increment of sync[0]; the wave that finds itself LAST -- every odd pose is final
// then -- solves the waves that deferred, one after the other, inside this launch (round 3 queued
// two fix-up launches behind every solve launch for this and for the poses outside the folded
// form: 2 x 5 us of empty launches per sweep).  With the observed dispatch (lower workgroup ids
// first) no wave ever defers; under any other order the sweep is slower, never wrong, never stuck.
// Hand-off per MI355X_MICROARCH.md (valid forms): producer = write-through (sc1, agent-scope) stores of the
// poses, vmcnt(0), relaxed agent flag store -- no cache-wide release (782 odd waves each writing back their
// XCD's whole L2 queued behind one another: the even waves started 19 us after their flags' poses were
// final); consumer = relaxed polls, ONE agent acquire fence, vmcnt(0), then plain loads.  The odd waves that READ an even wave's poses (as the old values of
// their neighbours) are exactly the two it waits for, so nothing is overwritten while in use.
//
// FOLD: the lanes evaluate the folded form of the energy ONLY (thirteen coefficients per pose, no context, no
// moment sums live across the Nelder-Mead loop).  A pose one of whose evaluations leaves the form's validity range
// is solved once more, on the spot, with the complete energy (folded where valid, term by term elsewhere: what the
// FOLD = false kernel runs) -- a cold block behind the loop that reloads everything from memory, so the loop's own
// registers are what they were; the wave publishes its poses after it.  Which road a pose takes is decided from the
// pose's own data and both evaluate identical arithmetic: the sweep's result does not depend on it.
//   sync[0]: waves of this launch that are done (the last one resets it), sync[1]: waves that deferred in this launch
//   counts[0] += even waves that deferred, counts[1] += poses solved with the complete energy behind the folded loop
#ifdef ICM_WAVE_TS   // measurement builds only (tools/wave_timeline.py): per-wave start / go / end times
__device__ unsigned long long g_wave_ts[4 * 16384];
#define WAVE_TS(slot) do { if (lane == 0 && gw < 16384) g_wave_ts[4 * gw + (slot)] = wall_clock64(); } while (0)
#else
#define WAVE_TS(slot) do { } while (0)
#endif
template <bool FOLD>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2))) void k_solve_m_fused(SolveArgs a, SolveSeg g, int nw, int* __restrict__ flags,
                                                          int spin_limit, int* __restrict__ deferred, int* __restrict__ sync,
                                                          unsigned long long* __restrict__ counts,
                                                          double* __restrict__ zero_out = nullptr, unsigned zero_n = 0, int ppw = kWave) {
    // ppw: poses per wave, 64 or (short colours: fewer waves than the chip has SIMDs either way) 32 -- a wave lasts as
    // long as its slowest lane and an even wave waits for the slowest of its two odd waves' lanes: half-filled waves
    // shorten both maxima
    const int lane = lane_id();
    const int gw = blockIdx.x * kWavesPerBlock + wave_in_block();
    if (gw >= 2 * nw) return;
    // zero_out: phase B's [superchunk x L] matrix, whose last reader ran before this launch: the even waves clear it for
    // the next sweep WHILE they wait for their odd neighbours, one store per lane and poll -- a few dozen stores per lane
    // into a memory system this launch leaves idle, instead of a memset launch (or 15 MB more for one of phase B's
    // bandwidth-bound kernels).  (Not all at once at the head of the launch: 15 MB of stores in front of the odd waves'
    // first loads delayed the whole chain by 4-5 us.)
    if (seg_aborted(g)) {
        if (zero_out && gw >= nw)   // (the matrix is cleared whatever becomes of the sweep: the host counts on it)
            for (unsigned q = (unsigned)(gw - nw) * kWave + lane; q < zero_n; q += (unsigned)nw * kWave) zero_out[q] = 0.0;
        return;
    }
    const int epoch = a.epoch;
    const bool even = gw >= nw;
    const int wv = even ? gw - nw : gw;
    const int tg = seg_pose(g, even, wv * ppw + lane);
    const bool mine = lane < ppw && tg < g.t1;
    WAVE_TS(0);
    // everything this launch does not write -- moment sums, odometry, controls, the pose's own previous value and the
    // trigonometry kept beside them -- is requested BEFORE an even wave starts to wait for its odd neighbours
    PoseIn in;
    in.n = 0;
    if (FOLD && mine) load_pose_in(a, tg, in);
    int ready = 1;
    if (even) {
        unsigned zq = (unsigned)wv * kWave + lane;   // this lane's next word of zero_out
        const unsigned zstride = (unsigned)nw * kWave;
        for (int d = 0; d < 2 && ready; ++d) {   // the (at most) two odd waves that hold this wave's neighbours
            const int ow = wv + d - g.shift;     // (wave-uniform loop: lane 0 polls, every lane clears)
            if (ow < 0 || ow >= nw) continue;
            int spins = 0;
            for (;;) {
                int f = 0;
                if (lane == 0) f = __hip_atomic_load(&flags[ow], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                f = __builtin_amdgcn_readfirstlane(f);
                if (f == epoch) break;
                if (++spins > spin_limit) {
                    ready = 0;
                    break;
                }
                if (zero_out && zq < zero_n) {
                    zero_out[zq] = 0.0;
                    zq += zstride;
                }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        if (zero_out)
            for (; zq < zero_n; zq += zstride) zero_out[zq] = 0.0;   // (what the wait left over)
        if (!ready) {   // wave-uniform: x is left untouched
            if (lane == 0) {
                __hip_atomic_store(&deferred[wv], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(&sync[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicAdd(&counts[0], 1ull);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (stamp and count are out before this wave counts as done)
        } else {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    WAVE_TS(1);
    double r0 = 0.0, r1 = 0.0, r2 = 0.0;   // (three scalars, not an array: an array that lives across the branches goes to scratch)
    bool redo = ready && mine;             // lanes whose pose the block below solves with the complete energy
    if (FOLD && redo) {
        double prev[3] = {a.x[3 * (size_t)(tg - 1)], a.x[3 * (size_t)(tg - 1) + 1], a.x[3 * (size_t)(tg - 1) + 2]};
        // an odd pose's lower neighbour is an OLD even pose (its pair is in the table); an even pose's was written a
        // moment ago by an odd wave of this launch, which publishes the pose, not the pair
        redo = !solve_pose_in<false, true>(a, tg, in, prev, !even, r0, r1, r2);   // (false: an evaluation left the folded form's range)
    }
    // The complete-energy solve, everything reloaded from memory (tgc is opaque to the compiler, so nothing of it is
    // shared with -- kept alive across -- the folded loop above).  First pass: this wave's own poses (FOLD: the rare ones
    // outside the folded range; else all of them), then the hand-off.  Further passes: only in the wave that finishes
    // the launch last, one per even wave that deferred.
    int tgc = tg;
    bool first = true, prev_tab = !even;
    int scan_from = 0;
    for (;;) {
        const unsigned long long mredo = __ballot(redo);
        if (mredo != 0ull) {
            asm volatile("" : "+v"(tgc));
            if (redo) {
                double prev[3] = {a.x[3 * (size_t)(tgc - 1)], a.x[3 * (size_t)(tgc - 1) + 1], a.x[3 * (size_t)(tgc - 1) + 2]};
                double res[3];
                solve_pose_moments<false, false>(a, tgc, prev, prev_tab, res);
                r0 = res[0]; r1 = res[1]; r2 = res[2];
            }
            if (FOLD && first && lane == (int)__builtin_ctzll(mredo)) atomicAdd(&counts[1], (unsigned long long)__popcll(mredo));
        }
        if (!first) {   // a deferred wave's poses: nobody in this launch waits for them
            if (redo) {
                const double res[3] = {r0, r1, r2};
                store_pose(a, tgc, res, false);
                if (a.xh) mirror_pose_pair(a, tgc, r0, r1, r2);
            }
        } else {
            // (odd poses are handed to the even waves of this launch: write-through (sc1) stores, no cache-wide release needed)
            if (ready && mine) {
                const double res[3] = {r0, r1, r2};
                store_pose_xyz(a, tg, res, !even);
                if (a.xh && even) mirror_pose_pair(a, tg, r0, r1, r2);
            }
            WAVE_TS(2);
            if (!even) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every write-through store of this wave has been acknowledged
                if (lane == 0) __hip_atomic_store(&flags[wv], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                WAVE_TS(3);
            }
            // done: one increment per wave, issued here -- everything a later reader INSIDE this launch needs of this wave
            // (an odd wave's poses, a deferral stamp) went out write-through and is acknowledged -- and looked at behind
            // the rotation pairs, whose ~260 instructions hide its round trip
            int old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(&sync[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the rotation pairs kept beside the poses are next sweep's business: behind the hand-off
            if (ready && mine) store_pose_tables(a, tg, r2);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old != 2 * nw - 1) return;
            int nd = 0;
            if (lane == 0) {   // the last wave of the launch: the counters are the next launch's again
                nd = __hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&sync[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&sync[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            nd = __builtin_amdgcn_readfirstlane(nd);
            if (nd == 0) return;
            // waves deferred (never seen under the in-order dispatch of this chip): every odd pose is final and visible
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            first = false;
            prev_tab = false;   // (an even pose's lower neighbour was written in this launch: its pair may not be out yet)
        }
        // the next wave that deferred in this launch
        int found = -1;
        for (int q0 = scan_from; q0 < nw && found < 0; q0 += kWave) {
            const int q = q0 + lane;
            const int d = q < nw ? __hip_atomic_load(&deferred[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            const unsigned long long m = __ballot(d == epoch);
            if (m) found = q0 + (int)__builtin_ctzll(m);
        }
        if (found < 0) return;
        scan_from = found + 1;
        tgc = seg_pose(g, true, found * ppw + lane);
        redo = lane < ppw && tgc < g.t1;
    }
}

// The same half sweep in latency form: one DPP quad (4 lanes) per pose, 16 poses per wave
// (nelder_mead3_quad).
__global__ __launch_bounds__(kBlock) void k_solve_mq_colour(SolveArgs a, SolveSeg g, int colour) {
    if (seg_aborted(g)) return;
    const int gid = blockIdx.x * kBlock + threadIdx.x;
    const int j = gid >> 2, role = gid & 3;
    const int tg = seg_pose(g, colour == 0, j);
    if (tg >= g.t1) return;  // whole quads leave together
    double prev[3] = {a.x[3 * (size_t)(tg - 1)], a.x[3 * (size_t)(tg - 1) + 1], a.x[3 * (size_t)(tg - 1) + 2]};
    double res[3];
    solve_pose_moments<true>(a, tg, prev, true, res, role);
    if (role == 0) store_pose(a, tg, res, false);
}

// Reference order (scripts/ICM_ROS.py:141: t = 1 .. T-1, every pose from its just-solved predecessor), moment form.
// Round 4: ONE lane walks the chain (the quad form's broadcasts no longer pay since the folded energy: an evaluation is
// a sixth of an iteration).  The inputs of pose t + 1 that do not depend on pose t's result (moment sums, odometry,
// controls, its own previous value) are requested before pose t is solved; the Nelder-Mead loop evaluates the folded
// energy only (FOLD: isotropic weights) and a pose that leaves its range is solved once more with the complete energy;
// the rotation pairs kept beside the poses are NOT written here -- two generic sincos per pose on a chain of T - 1
// solves -- but by k_pose_rot at the head of the next sweep, in parallel (the host clears rot_valid).  data_IJAC2018:
// 53 -> ?? ms per sweep; the same arithmetic per pose as every other solve form (bit-identical).
// Round 5 measured the chain walked by ONE DPP QUAD (QUAD = true: nelder_mead3_quad, the four candidate points of an
// iteration evaluated at once, lane r point r, the folded energy in the loop -- one evaluation of latency per iteration
// instead of two): 45.2 against 43.1 ms on data_IJAC2018, and 55.3 against 50.6 ms for k_init_pass.  The folded energy
// is ~40 of an iteration's ~210 instructions; what the chain costs is the Nelder-Mead's bookkeeping, which the quad does
// not shorten (it adds the exchanges), at one dependent instruction per ~4.6 cycles of a lone wave.  Kept as a
// cross-check (bit-identical), the lane form is what runs.
template <bool FOLD, bool QUAD = false>
__global__ __launch_bounds__(kWave) void k_solve_m_sequential(SolveArgs a) {
    if (threadIdx.x >= (QUAD ? 4 : 1)) return;
    const int role = threadIdx.x;
    double prev[3] = {a.x[0], a.x[1], a.x[2]};
    if constexpr (!FOLD) {
        // the complete energy in the loop (anisotropic weights; a cross-check otherwise): load and solve, no prefetch -- with
        // the next pose's inputs carried around this larger loop the compiler keeps them in scratch behind a flat pointer
        // and its null check does not assemble on gfx950
        for (int tg = 1; tg < a.T; ++tg) {
            double res[3];
            solve_pose_moments<QUAD, false>(a, tg, prev, false, res, role);
            if (role == 0) store_pose_xyz(a, tg, res, false);
            prev[0] = res[0]; prev[1] = res[1]; prev[2] = res[2];
        }
        return;
    }
    PoseIn cur;
    cur.n = 0;
    if (a.T > 1) load_pose_in(a, 1, cur);
    for (int tg = 1; tg < a.T; ++tg) {
        PoseIn nxt;
        nxt.n = 0;
        if (tg + 1 < a.T) load_pose_in(a, tg + 1, nxt);
        double r0 = 0.0, r1 = 0.0, r2 = 0.0;
        bool ok = solve_pose_in<QUAD, FOLD>(a, tg, cur, prev, false, r0, r1, r2, role);   // (prev is the pose this lane / quad has just solved: its pair is formed here)
        if (FOLD && !ok) {
            int tgc = tg;
            asm volatile("" : "+v"(tgc));
            double res[3];
            solve_pose_moments<QUAD, false>(a, tgc, prev, false, res, role);
            r0 = res[0]; r1 = res[1]; r2 = res[2];
        }
        const double res[3] = {r0, r1, r2};
        if (role == 0) store_pose_xyz(a, tg, res, false);
        prev[0] = r0; prev[1] = r1; prev[2] = r2;
        cur = nxt;
    }
}

// Red-black half sweep: all poses of one parity (colour = tg & 1) of this shard, one wave
// each.  Neighbours have the other parity, so nothing read here is written by this launch.
template <bool PER_BEAM>
__global__ __launch_bounds__(kBlock) void k_solve_colour(SolveArgs a, int colour) {
    const int lane = lane_id();
    const int w = blockIdx.x * kWavesPerBlock + wave_in_block();
    int first = a.t_begin > 1 ? a.t_begin : 1;
    if ((first & 1) != colour) ++first;
    const int tg = first + 2 * w;
    if (tg >= a.t_begin + a.nloc) return;
    double prev[3] = {a.x[3 * (size_t)(tg - 1)], a.x[3 * (size_t)(tg - 1) + 1], a.x[3 * (size_t)(tg - 1) + 2]};
    double res[3];
    solve_pose<PER_BEAM>(a, tg, prev, res, lane);
    if (lane == 0) {
        a.x[3 * (size_t)tg] = res[0];
        a.x[3 * (size_t)tg + 1] = res[1];
        a.x[3 * (size_t)tg + 2] = res[2];
    }
}

// Reference order: one wave walks the chain t = 1..T-1, each solve conditioned on the pose
// it has just written (Gauss-Seidel, scripts/ICM_ROS.py:141-158).
template <bool PER_BEAM>
__global__ __launch_bounds__(kWave) void k_solve_sequential(SolveArgs a) {
    const int lane = lane_id();
    double prev[3] = {a.x[0], a.x[1], a.x[2]};
    for (int tg = 1; tg < a.T; ++tg) {
        double res[3];
        solve_pose<PER_BEAM>(a, tg, prev, res, lane);
        if (lane == 0) {
            a.x[3 * (size_t)tg] = res[0];
            a.x[3 * (size_t)tg + 1] = res[1];
            a.x[3 * (size_t)tg + 2] = res[2];
        }
        prev[0] = res[0]; prev[1] = res[1]; prev[2] = res[2];
    }
}

// ---------------------------------------------------------------------------------------
// The causal initialisation pass (reference inicializar_online_process, scripts/ICM_ROS.py:102-119,
// driven over a recorded sequence): for t = 1..T-1 predict with the unicycle model, project
// the scan with the PREDICTED pose, associate against the RUNNING map (every landmark seen so
// far, brute force like the reference's cdist/argmin), fold the scan into the running means
// with the reference's recurrence y <- S/(n+k) + y n/(n+k) (scripts/ICM_SLAM_tools.py:194), then
// solve the one-sided energy fun_x.  Each step depends on the previous one (pose AND map), so
// this is one wavefront walking the sequence; lanes parallelise the beams of the current scan.
// LDS per scan: label, world point and target of every kept beam.
// ---------------------------------------------------------------------------------------
struct InitArgs {
    double* x;  // (T,3); x[0] = x0 on entry
    const double* odo;
    const double* u;
    int T;
    const int* boff;
    const double *bx, *by;
    double* y;    // (2,L) running map, in/out
    double* cnt;  // (L) observation counts, in/out
    int* lact;    // landmarks in use, in/out
    int L, maxb;
    double thr, dt, R0, R1, R2, Q0, Q1, cte;
    int* flags;   // [0] = a new landmark would not fit in L (the reference raises IndexError)
};

__global__ __launch_bounds__(kWave) void k_init_pass(InitArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    double* lwx = reinterpret_cast<double*>(smem);
    double* lwy = lwx + a.maxb;
    double* ltx = lwy + a.maxb;
    double* lty = ltx + a.maxb;
    int* llab = reinterpret_cast<int*>(lty + a.maxb);
    const int lane = lane_id();
    int lact = *a.lact;
    double xt[3] = {a.x[0], a.x[1], a.x[2]};
    for (int t = 1; t < a.T; ++t) {
        const double v = a.u[t - 1], w = a.u[(size_t)a.T + t - 1];
        const double xc0 = xt[0] + a.dt * (cos(xt[2]) * v), xc1 = xt[1] + a.dt * (sin(xt[2]) * v), xc2 = xt[2] + a.dt * w;
        const int j0 = a.boff[t], n = a.boff[t + 1] - j0;
        if (n == 0) {  // no observation: keep the prediction (scripts/ICM_ROS.py:110-113)
            xt[0] = xc0; xt[1] = xc1; xt[2] = xc2;
        } else {
            double ct, st;
            pose_rot(xc2, ct, st);
            bool isnew = false;
            for (int j = lane; j < n; j += kWave) {
                const double bxx = a.bx[j0 + j], byy = a.by[j0 + j];
                const double wx = (bxx * ct - byy * st) + xc0, wy = (bxx * st + byy * ct) + xc1;
                double best = __builtin_huge_val();
                int bid = -1;
                for (int i = 0; i < lact; ++i) {
                    const double dx = a.y[i] - wx, dy = a.y[a.L + i] - wy;
                    const double d = sqrt(dx * dx + dy * dy);
                    if (d < best) {
                        best = d;
                        bid = i;
                    }
                }
                const int lab = (bid >= 0 && !(best > a.thr)) ? bid : -1;
                isnew |= lab < 0;
                lwx[j] = wx;
                lwy[j] = wy;
                llab[j] = lab;
            }
            if (__ballot(isnew) != 0ull) {  // all gated-out beams of the scan share ONE new label
                if (lact >= a.L) {
                    if (lane == 0) a.flags[0] = 1;
                    break;
                }
                for (int j = lane; j < n; j += kWave)
                    if (llab[j] < 0) llab[j] = lact;
                ++lact;
            }
            __builtin_amdgcn_wave_barrier();
            // fold the scan into the running means: the first beam of each label sums its group
            for (int j = lane; j < n; j += kWave) {
                const int lab = llab[j];
                bool leader = true;
                for (int q = 0; q < j; ++q) leader &= llab[q] != lab;
                if (leader) {
                    double sx = 0.0, sy = 0.0;
                    int k = 0;
                    for (int q = j; q < n; ++q)
                        if (llab[q] == lab) {
                            sx += lwx[q];
                            sy += lwy[q];
                            ++k;
                        }
                    const double nn = a.cnt[lab], tot = nn + (double)k;
                    a.y[lab] = sx / tot + a.y[lab] * nn / tot;
                    a.y[a.L + lab] = sy / tot + a.y[a.L + lab] * nn / tot;
                    a.cnt[lab] = tot;
                }
            }
            __threadfence_block();
            for (int j = lane; j < n; j += kWave) {  // y[:, c] after the update (scripts/ICM_ROS.py:117-118)
                ltx[j] = a.y[llab[j]];
                lty[j] = a.y[a.L + llab[j]];
            }
            __builtin_amdgcn_wave_barrier();
            // The one-sided solve (fun_x / minimizar_x, scripts/ICM_ROS.py:254-278) in the moment form the sweeps use (round 4;
            // the per-beam sum with a wave reduction per evaluation took 52 us per pose): the wave forms the pose's 14 sums
            // once, about the prediction (the Nelder-Mead's start), one term per beam (an entry of one beam: no scatter
            // term); then the folded energy, every lane running the same chain on the same numbers.
            PoseIn in;
            in.n = n;
            in.ua0 = v; in.ua1 = w; in.ut0 = in.ut1 = 0.0;
            load3(a.odo, a.T, t - 1, in.oa);
            load3(a.odo, a.T, t, in.ot);
            in.op[0] = in.op[1] = in.op[2] = 0.0;
            in.coa = cos(in.oa[2]); in.soa = sin(in.oa[2]); in.cot = 1.0; in.sot = 0.0;
            in.pox = xc0; in.poy = xc1; in.tho = xc2; in.co = cos(xc2); in.so = sin(xc2);
            {
                double m[kMomentCount];
#pragma unroll
                for (int q = 0; q < kMomentCount; ++q) m[q] = 0.0;
                for (int j = lane; j < n; j += kWave) {
                    const double bxx = a.bx[j0 + j], byy = a.by[j0 + j];
                    const double wx = ct * bxx - st * byy, wy = st * bxx + ct * byy;
                    const double rx = (xc0 + wx) - ltx[j], ry = (xc1 + wy) - lty[j];
                    m[0] += 1.0; m[1] += wx; m[2] += wy; m[3] += rx; m[4] += ry;
                    m[5] += wx * wx; m[6] += wy * wy; m[7] += wx * wy;
                    m[8] += wx * rx; m[9] += wy * rx; m[10] += wx * ry; m[11] += wy * ry;
                    m[12] += rx * rx; m[13] += ry * ry;
                }
#pragma unroll
                for (int q = 0; q < kMomentCount; ++q) in.pm[q] = wave_sum(m[q]);
                in.pm[14] = in.pm[15] = in.pm[16] = 0.0;
            }
            SolveArgs sa;
            sa.x = a.x; sa.x0 = nullptr; sa.odo = a.odo; sa.u = a.u;
            sa.T = t + 1;   // (one-sided: pose t is the last one there is)
            sa.t_begin = 0; sa.nloc = a.T;
            sa.dt = a.dt; sa.R0 = a.R0; sa.R1 = a.R1; sa.R2 = a.R2; sa.Q0 = a.Q0; sa.Q1 = a.Q1; sa.cte = a.cte;
            sa.diag = nullptr; sa.rot = nullptr; sa.cs = nullptr; sa.odo_cs = nullptr; sa.epoch = 0; sa.xh = nullptr;
            sa.ghost_n = 0; sa.ghost_m = nullptr;
            double r0 = 0.0, r1 = 0.0, r2 = 0.0;
            const bool iso = a.Q0 == a.Q1 && a.R0 == a.R1;
            bool ok = false;
            // (the quad form here -- every DPP quad of the wave on the same numbers -- measured 55.3 against 50.6 ms: k_solve_m_sequential)
            if (iso) ok = solve_pose_in<false, true>(sa, t, in, xt, false, r0, r1, r2);
            if (!ok) solve_pose_in<false, false>(sa, t, in, xt, false, r0, r1, r2);
            xt[0] = r0; xt[1] = r1; xt[2] = r2;
        }
        if (lane == 0) {
            a.x[3 * (size_t)t] = xt[0];
            a.x[3 * (size_t)t + 1] = xt[1];
            a.x[3 * (size_t)t + 2] = xt[2];
        }
    }
    if (lane == 0) *a.lact = lact;
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
    Items it{a.bx, a.by, a.tx, a.ty, nullptr, 0.0, 0.0, 0.0, a.n, nullptr, nullptr};
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

}  // namespace icm
