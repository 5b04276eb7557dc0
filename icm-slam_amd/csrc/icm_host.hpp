// Host-side pieces of the sweep: Mapa.filtrar (prune/merge of the running map) and the
// uniform grid over mapa_viejo that the association kernel searches.  Plain C++, no HIP.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/icmslam.h"

namespace icm {

// Uniform grid over the reference map for the gated nearest-landmark search.
// Cell edge >= dist_thr, so every landmark within dist_thr of a point lies in the 3x3
// cells around the point's cell.  Landmarks are stored sorted by cell (counting sort);
// `id` keeps the original column index for the first-index tie-break of np.argmin.
struct Grid {
    double gx0 = 0, gy0 = 0, inv = 1;
    int nx = 1, ny = 1;
    std::vector<int> cell_start;  // nx*ny + 1
    std::vector<double> lx, ly;   // sorted by cell
    std::vector<int> id;
};

void build_grid(const double* map_x, const double* map_y, int64_t K, double dist_thr, Grid& g);

// Mapa.filtrar, reference scripts/ICM_SLAM_tools.py:204-265.  y is (2,L) row-major.
int filtrar_host(const icm_config& cfg, const double* y, const double* counts, int64_t lact,
                 double* y_out, double* counts_out, int64_t* lact_out, std::string& err);

// The ONE place that sizes the staging area of phase A's (pose, landmark) entries and everything indexed like it.
//   [0, sparse0)            packed area: pose t owns [plan[t] + kStageSlack t, plan[t+1] + kStageSlack (t+1)), plan = the
//                           previous sweep's exclusive scan of the entry counts (<= nnz in total)
//   [sparse0, entries)      sparse area: a pose whose entries do not fit its reserved place stages them at
//                           sparse0 + (offset of its first kept beam); at most one entry per kept beam
//   entries                 capacity of st_label / st_k / st_sx / st_sy
//   prefix_stride           capacity of EACH per-entry prefix array (pre_x, pre_y, pre_n): one element per staging
//                           place plus one wave of slack for the branch-free stores of idle lanes
// false: the sequence has too many kept beams for 32-bit entry offsets.
constexpr int kStageSlack = 4;
constexpr int kStagePad = 512;
constexpr int kStageWave = 64;
struct StagingLayout {
    int64_t sparse0 = 0, entries = 0, prefix_stride = 0;
};
bool staging_layout(int64_t nnz, int64_t nloc, StagingLayout& out);

// First-scan clustering of Mapa.actualizar's Lact == 0 branch (reference
// scripts/ICM_SLAM_tools.py:160-165): fcluster(linkage(pdist(pts)), t) - 1, i.e. SciPy's single
// linkage, depth-2 inconsistency coefficients and the 'inconsistent' flat-cluster rule.
// pts is (n,2) row-major; labels_out[n] in 0..ncl-1.
int cluster_first_scan_host(const double* pts, int64_t n, double t, int32_t* labels_out, std::string& err);

}  // namespace icm
