// Build-time probe (NOT part of the library): one evaluation of the pose energy in moment form,
// inputs from memory so that nothing folds, the |d| > 0.25 generic-sincos branch compiled out.
// tools/count_eval_flops.py compiles this to gfx950 assembly and counts the FP64 vector
// instructions of k_eval_probe: that is the "flop per energy evaluation" bench.py prices the
// solve kernel with (v_fma_f64 = 2 flops, v_mul/v_add_f64 = 1).
#define ICM_PROBE_FAST_TRIG_ONLY 1
#include "icm_device.hpp"

using namespace icm;

extern "C" __global__ void k_eval_probe(const SolveCtx* __restrict__ c, const PoseMoments* __restrict__ m,
                                        const PoseFold* __restrict__ f, const double* __restrict__ p, double* __restrict__ out) {
    const int i = threadIdx.x;
    out[i] = pose_energy_moments(*c, *m, *f, p[3 * i], p[3 * i + 1], p[3 * i + 2]);
}
