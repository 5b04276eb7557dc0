// Host-side pieces of the sweep (see icm_host.hpp).  Compiled with -ffp-contract=off.
#include "icm_host.hpp"

#include <algorithm>
#include <cmath>
#include <limits>

namespace icm {

static inline int cell_of(double v, double g0, double inv, int n) {
    double f = std::floor((v - g0) * inv);
    if (!(f >= 0.0)) f = 0.0;  // also catches NaN
    if (f > (double)(n - 1)) f = (double)(n - 1);
    return (int)f;
}

void build_grid(const double* mx, const double* my, int64_t K, double dist_thr, Grid& g) {
    g.lx.assign((size_t)K, 0.0);
    g.ly.assign((size_t)K, 0.0);
    g.id.assign((size_t)K, 0);
    if (K <= 0) {
        g.gx0 = g.gy0 = 0.0;
        g.inv = 1.0;
        g.nx = g.ny = 1;
        g.cell_start.assign(2, 0);
        return;
    }
    double x0 = mx[0], x1 = mx[0], y0 = my[0], y1 = my[0];
    for (int64_t i = 1; i < K; ++i) {
        x0 = std::min(x0, mx[i]); x1 = std::max(x1, mx[i]);
        y0 = std::min(y0, my[i]); y1 = std::max(y1, my[i]);
    }
    // a hair larger than the gate so rounding in the cell computation can never push a
    // landmark at distance <= dist_thr two cells away
    double cell = dist_thr > 0.0 ? dist_thr * (1.0 + 1e-9) : 1.0;
    const double max_cells = 8.0 * (double)K + 4096.0;
    g.nx = g.ny = 1;
    for (int tries = 0; tries < 128; ++tries) {   // (bounded: a non-finite extent never fits)
        double nxd = std::floor((x1 - x0) / cell) + 1.0, nyd = std::floor((y1 - y0) / cell) + 1.0;
        if (nxd * nyd <= max_cells) {
            g.nx = (int)nxd;
            g.ny = (int)nyd;
            break;
        }
        cell *= 2.0;
    }
    if (!(x1 - x0 < HUGE_VAL) || !(y1 - y0 < HUGE_VAL)) {
        x0 = y0 = 0.0;
        cell = 1.0;
    }
    g.gx0 = x0;
    g.gy0 = y0;
    g.inv = 1.0 / cell;
    const size_t ncell = (size_t)g.nx * (size_t)g.ny;
    g.cell_start.assign(ncell + 1, 0);
    std::vector<int> cid((size_t)K);
    for (int64_t i = 0; i < K; ++i) {
        int cx = cell_of(mx[i], g.gx0, g.inv, g.nx), cy = cell_of(my[i], g.gy0, g.inv, g.ny);
        cid[(size_t)i] = cy * g.nx + cx;
        g.cell_start[(size_t)cid[(size_t)i] + 1]++;
    }
    for (size_t c = 0; c < ncell; ++c) g.cell_start[c + 1] += g.cell_start[c];
    std::vector<int> fill(g.cell_start.begin(), g.cell_start.end() - 1);
    for (int64_t i = 0; i < K; ++i) {  // ascending i => ascending id inside a cell
        int p = fill[(size_t)cid[(size_t)i]]++;
        g.lx[(size_t)p] = mx[i];
        g.ly[(size_t)p] = my[i];
        g.id[(size_t)p] = (int)i;
    }
}

int filtrar_host(const icm_config& cfg, const double* y, const double* counts, int64_t lact,
                 double* y_out, double* counts_out, int64_t* lact_out, std::string& err) {
    const int64_t L = cfg.L;
    if (lact < 0 || lact > L) {
        err = "filtrar: landmarks_actuales outside [0, L]";
        return ICM_ERR_INDEX;
    }
    // prune landmarks seen fewer than `cota` times (scripts/ICM_SLAM_tools.py:229-238).
    // (When nothing is pruned the reference indexes the unsliced (2,L) map with a
    // length-Lact mask and raises IndexError unless Lact == L; here the map is sliced to
    // its Lact live columns in both cases -- a conscious fix, see DESIGN.md.)
    std::vector<double> px, py, pc;
    px.reserve((size_t)lact); py.reserve((size_t)lact); pc.reserve((size_t)lact);
    for (int64_t i = 0; i < lact; ++i)
        if (counts[i] >= cfg.cota) {
            px.push_back(y[i]);
            py.push_back(y[L + i]);
            pc.push_back(counts[i]);
        }
    const int n = (int)px.size();
    if (n == 0) {
        err = "filtrar: no landmark reached `cota` observations (the reference raises here)";
        return ICM_ERR_EMPTY_MAP;
    }
    // nearest other landmark (scripts/ICM_SLAM_tools.py:241-245)
    std::vector<int> nn((size_t)n, 0);
    std::vector<double> nd((size_t)n, 0.0);
    const double thr = cfg.dist_thr;
    bool coincident = false;
    if (n > 1) {
        Grid g;
        build_grid(px.data(), py.data(), n, thr, g);
        for (int i = 0; i < n && !coincident; ++i) {
            double best = std::numeric_limits<double>::infinity();
            int bj = -1;
            int cx = cell_of(px[(size_t)i], g.gx0, g.inv, g.nx), cy = cell_of(py[(size_t)i], g.gy0, g.inv, g.ny);
            for (int ry = std::max(cy - 1, 0); ry <= std::min(cy + 1, g.ny - 1); ++ry) {
                int c0 = std::max(cx - 1, 0), c1 = std::min(cx + 1, g.nx - 1);
                for (int p = g.cell_start[(size_t)ry * g.nx + c0]; p < g.cell_start[(size_t)ry * g.nx + c1 + 1]; ++p) {
                    int j = g.id[(size_t)p];
                    if (j == i) continue;
                    double dx = px[(size_t)i] - g.lx[(size_t)p], dy = py[(size_t)i] - g.ly[(size_t)p];
                    double d = std::sqrt(dx * dx + dy * dy);
                    if (d == 0.0) { coincident = true; break; }
                    if (d < best || (d == best && j < bj)) { best = d; bj = j; }
                }
                if (coincident) break;
            }
            nd[(size_t)i] = best;  // >= thr (or inf) means "no merge"; the exact value is unused then
            nn[(size_t)i] = bj < 0 ? 0 : bj;
        }
    }
    if (n == 1 || coincident) {
        // literal O(n^2) form, including the zeros -> global max replacement (:242)
        double amax = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double dx = px[(size_t)i] - px[(size_t)j], dy = py[(size_t)i] - py[(size_t)j];
                amax = std::max(amax, std::sqrt(dx * dx + dy * dy));
            }
        for (int i = 0; i < n; ++i) {
            double best = std::numeric_limits<double>::infinity();
            int bj = 0;
            for (int j = 0; j < n; ++j) {
                double dx = px[(size_t)j] - px[(size_t)i], dy = py[(size_t)j] - py[(size_t)i];
                double d = std::sqrt(dx * dx + dy * dy);
                if (d == 0.0) d = amax;
                if (d < best) { best = d; bj = j; }
            }
            nd[(size_t)i] = best;
            nn[(size_t)i] = bj;
        }
    }
    // merge by label propagation, in index order (:246-249)
    std::vector<int> c((size_t)n);
    for (int i = 0; i < n; ++i) c[(size_t)i] = i;
    for (int i = 0; i < n; ++i) {
        if (!(nd[(size_t)i] < thr)) continue;
        const int from = c[(size_t)nn[(size_t)i]], to = c[(size_t)i];
        if (from == to) continue;
        for (int k = 0; k < n; ++k)
            if (c[(size_t)k] == from) c[(size_t)k] = to;
    }
    // close the gaps in the label set, keeping order (:251-253)
    std::vector<int> used((size_t)n, 0), rank((size_t)n, 0);
    for (int i = 0; i < n; ++i) used[(size_t)c[(size_t)i]] = 1;
    int m = 0;
    for (int i = 0; i < n; ++i) {
        rank[(size_t)i] = m;
        m += used[(size_t)i];
    }
    // count-weighted means (:255-260)
    std::fill(y_out, y_out + 2 * L, 0.0);
    std::fill(counts_out, counts_out + L, 0.0);
    std::vector<double> sx((size_t)m, 0.0), sy((size_t)m, 0.0), sc((size_t)m, 0.0);
    for (int i = 0; i < n; ++i) {
        const int r = rank[(size_t)c[(size_t)i]];
        sc[(size_t)r] += pc[(size_t)i];
        sx[(size_t)r] += px[(size_t)i] * pc[(size_t)i];
        sy[(size_t)r] += py[(size_t)i] * pc[(size_t)i];
    }
    for (int r = 0; r < m; ++r) {
        counts_out[r] = sc[(size_t)r];
        y_out[r] = sx[(size_t)r] / sc[(size_t)r];
        y_out[L + r] = sy[(size_t)r] / sc[(size_t)r];
    }
    *lact_out = m;
    return ICM_OK;
}

// ---- first-scan clustering ---------------------------------------------------------------
namespace {
struct Link {
    int a, b;
    double h;
};
}  // namespace

int cluster_first_scan_host(const double* pts, int64_t n64, double t, int32_t* labels, std::string& err) {
    if (n64 <= 0 || n64 > 20000) {
        err = "cluster_first_scan: need 1..20000 observations";
        return ICM_ERR_ARG;
    }
    const int n = (int)n64;
    if (n == 1) {
        labels[0] = 0;
        return ICM_OK;
    }
    auto dist = [&](int i, int j) {
        const double dx = pts[2 * i] - pts[2 * j], dy = pts[2 * i + 1] - pts[2 * j + 1];
        return std::sqrt(dx * dx + dy * dy);
    };
    // single linkage as SciPy builds it: Prim's walk from observation 0 ...
    std::vector<Link> z((size_t)n - 1);
    std::vector<char> merged((size_t)n, 0);
    std::vector<double> D((size_t)n, std::numeric_limits<double>::infinity());
    int x = 0;
    for (int k = 0; k < n - 1; ++k) {
        merged[(size_t)x] = 1;
        double cur = std::numeric_limits<double>::infinity();
        int y = -1;
        for (int i = 0; i < n; ++i) {
            if (merged[(size_t)i]) continue;
            const double d = dist(x, i);
            if (D[(size_t)i] > d) D[(size_t)i] = d;
            if (D[(size_t)i] < cur) {
                y = i;
                cur = D[(size_t)i];
            }
        }
        z[(size_t)k] = Link{x, y, cur};
        x = y;
    }
    // ... links ordered by height (stable), then union-find labels, smaller id first
    std::stable_sort(z.begin(), z.end(), [](const Link& p, const Link& q) { return p.h < q.h; });
    std::vector<int> parent((size_t)(2 * n - 1));
    for (int i = 0; i < 2 * n - 1; ++i) parent[(size_t)i] = i;
    auto find = [&](int a) {
        while (parent[(size_t)a] != a) a = parent[(size_t)a];
        return a;
    };
    for (int i = 0; i < n - 1; ++i) {
        int a = find(z[(size_t)i].a), b = find(z[(size_t)i].b);
        if (a > b) std::swap(a, b);
        z[(size_t)i].a = a;
        z[(size_t)i].b = b;
        parent[(size_t)a] = parent[(size_t)b] = n + i;
    }
    // inconsistency coefficient of every link over its depth-2 sub-tree (itself + child links)
    std::vector<double> inc((size_t)n - 1, 0.0), mx((size_t)n - 1, 0.0);
    for (int i = 0; i < n - 1; ++i) {
        // SciPy's walk is post-order: left child link, right child link, the link itself
        double hs[3];
        int cnt = 0;
        if (z[(size_t)i].a >= n) hs[cnt++] = z[(size_t)(z[(size_t)i].a - n)].h;
        if (z[(size_t)i].b >= n) hs[cnt++] = z[(size_t)(z[(size_t)i].b - n)].h;
        hs[cnt++] = z[(size_t)i].h;
        double s = 0.0, ss = 0.0;
        for (int q = 0; q < cnt; ++q) {
            s += hs[q];
            ss += hs[q] * hs[q];
        }
        const double var = cnt >= 2 ? (ss - s * s / cnt) / (cnt - 1) : (ss - s * s / cnt) / cnt;
        const double sd = var > 0.0 ? std::sqrt(var) : 0.0;
        if (sd > 0.0) inc[(size_t)i] = (z[(size_t)i].h - s / cnt) / sd;
        double m = inc[(size_t)i];
        if (z[(size_t)i].a >= n) m = std::max(m, mx[(size_t)(z[(size_t)i].a - n)]);
        if (z[(size_t)i].b >= n) m = std::max(m, mx[(size_t)(z[(size_t)i].b - n)]);
        mx[(size_t)i] = m;
    }
    // flat clusters: left-first depth-first walk; a node whose sub-tree is consistent leads one
    std::vector<int> T((size_t)n, 0);
    int ncl = 0;
    struct Frame {
        int node;
        bool leader;  // this node or an ancestor leads a cluster
        int stage;
    };
    std::vector<Frame> st;
    st.push_back(Frame{n - 2, false, 0});
    while (!st.empty()) {
        Frame& f = st.back();
        const Link& l = z[(size_t)f.node];
        if (f.stage == 0) {
            if (!f.leader && mx[(size_t)f.node] <= t) {
                f.leader = true;
                ++ncl;
            }
            f.stage = 1;
            if (l.a >= n) {
                const bool ld = f.leader;
                st.push_back(Frame{l.a - n, ld, 0});
                continue;
            }
        }
        if (f.stage == 1) {
            f.stage = 2;
            if (l.b >= n) {
                const bool ld = f.leader;
                st.push_back(Frame{l.b - n, ld, 0});
                continue;
            }
        }
        if (l.a < n) {
            if (!f.leader) ++ncl;
            T[(size_t)l.a] = ncl;
        }
        if (l.b < n) {
            if (!f.leader) ++ncl;
            T[(size_t)l.b] = ncl;
        }
        st.pop_back();
    }
    for (int i = 0; i < n; ++i) labels[i] = T[(size_t)i] - 1;
    return ICM_OK;
}

bool staging_layout(int64_t nnz, int64_t nloc, StagingLayout& out) {
    const int64_t nz = nnz > 1 ? nnz : 1;   // (an all-empty sequence still gets one place)
    const int64_t np = nloc > 0 ? nloc : 0;
    out.sparse0 = nz + (int64_t)kStageSlack * np + kStageWave;
    out.entries = out.sparse0 + nz + kStagePad;
    out.prefix_stride = out.entries + kStageWave;
    return nnz >= 0 && nloc >= 0 && out.prefix_stride <= (int64_t)0x7fffffff;
}

}  // namespace icm
