// Sanitizer build of the library's host C++ (icm-slam_amd/csrc/icm_host.cpp: Mapa.filtrar on the host, the search grid,
// the first scan's clustering) -- TEST INFRASTRUCTURE.  The product compiles icm_host.cpp into libicmslam_hip.so with
// hipcc; GPU address sanitizers are not available on this pool, so the host routines are built once more here with
// g++ -fsanitize=address,undefined (make -C icm-slam_amd/csrc asan) behind the two C entry points they have in the
// product (icm_filtrar, icm_cluster_first_scan: include/icmslam.h) plus the grid builder, and tests/test_asan_cpu.py
// drives them in a child process with the sanitizer runtime preloaded.
#include <cstdint>
#include <string>
#include <vector>

#include "../../icm-slam_amd/csrc/icm_host.hpp"

extern "C" {

int asan_filtrar(const icm_config* cfg, const double* y, const double* counts, int64_t lact, double* y_out, double* counts_out,
                 int64_t* lact_out) {
    std::string err;
    return icm::filtrar_host(*cfg, y, counts, lact, y_out, counts_out, lact_out, err);
}

int asan_cluster_first_scan(const double* pts, int64_t n, double t, int32_t* labels_out) {
    std::string err;
    return icm::cluster_first_scan_host(pts, n, t, labels_out, err);
}

// cells of the grid over K landmarks; out3 = [nx, ny, landmarks stored]
int asan_build_grid(const double* mx, const double* my, int64_t K, double dist_thr, int64_t* out3) {
    icm::Grid g;
    icm::build_grid(mx, my, K, dist_thr, g);
    out3[0] = g.nx;
    out3[1] = g.ny;
    out3[2] = (int64_t)g.id.size();
    return (int)g.cell_start.back();
}
}
