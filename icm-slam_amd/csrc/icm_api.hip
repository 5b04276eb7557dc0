// C-ABI of the MI355X-native ICM sweep (include/icmslam.h, include/icmslam_tuning.h): handle, HBM buffers, sweep
// orchestration.  Device work goes to one HIP stream, plus a side stream for the fused map filter (the k_fl_* chain,
// overlapped with the pose solves) and one for a shard's ghost-pose chain.  A red-black sweep through the default
// pipeline is queued WHOLE: the host waits once, for the side stream's 16-byte counts-and-flags copy and the filter
// result, while the solves are still running; the kernels that would replace state look at the sweep's overflow flags
// themselves.  Only a table overflow (the sweep is then repeated with the host looking in the middle) or a map that
// needs the exact host routine of Mapa.filtrar adds host round trips.  A sharded sweep has ONE collective, issued here
// (icm_sweep_sharded: RCCL resolved with dlopen, or the caller's all-gather) or by the caller between the phase calls.
#include <cstdlib>
#include <cstring>
#include <ctime>

#include <dlfcn.h>

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>   // hipExtLaunchKernelGGL: a launch with its own stop event
#include <rccl/rccl.h>   // types and prototypes only: the library is dlopen()ed (icm_comm_init), never linked
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/icmslam.h"
#include "../../include/icmslam_tuning.h"
#include "icm_host.hpp"
#include "eval_flops.h"
#if __has_include("build_id.h")
#include "build_id.h"   // written by the Makefile: sha256 of the sources
#else
#define ICM_BUILD_ID "unknown"
#endif
#include "icm_kernels.hip"

using namespace icm;

namespace {

std::string g_create_err;

enum KernelId {
    KID_PREFILTER = 0, KID_SCAN, KID_ASSOC_BRUTE, KID_ASSOC_GROUP, KID_COMPACT, KID_SORT, KID_LM_BOUNDS, KID_LM_TOTALS,
    KID_STATS_PREFIX, KID_LM_SCAN, KID_BEAM_TARGETS, KID_POSE_MOMENTS, KID_SOLVE, KID_FILTRAR, KID_NEIGH, KID_CHUNK_L1, KID_CHUNK_L2, KID_LM_L3, KID_REC_PUSH, KID_POSE_ROT, KID_ASSOC_RUNS, KID_RUN_BUILD, KID_COUNT
};
const char* kKernelNames[KID_COUNT] = {"k_prefilter", "k_scan", "k_associate_brute", "k_assoc_group", "k_compact",
                                       "radix_sort_pairs", "k_lm_bounds", "k_lm_scan_totals", "k_stats_prefix",
                                       "k_lm_scan", "k_beam_targets", "k_pose_moments", "k_solve", "k_filtrar", "k_neigh_table",
                                       "k_chunk_l1", "k_chunk_l2", "k_lm_l3", "k_rec_push", "k_pose_rot", "k_assoc_runs", "k_run_build"};

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace

struct icm_handle {
    icm_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;

    // sequence
    int64_t T = 0, B = 0, t_begin = 0, nloc = 0;
    bool uploaded = false, prefiltered = false, have_state = false;
    DevBuf<double> ranges, cosb, sinb, odo, u;
    DevBuf<double> odo_cs;    // (T,2): (cos, sin) of the odometry headings (k_odo_trig, once per sequence)
    DevBuf<double> pose_cs;   // (T,2): (cos, sin)(theta) of every pose as it stands (valid with rot_valid)
    DevBuf<int> nkept, boff, bk;
    DevBuf<double> bd, bx, by, pose_s2;
    DevBuf<unsigned long long> kmask, gh_kmask;   // the pre-filter's decisions: one bit per in-range beam (pass 1 -> pass 2)
    DevBuf<double2> bxy, gh_bxy;   // the kept beams' body points once more, interleaved (the beam-by-beam path: one load per beam)
    std::vector<int> h_boff;
    int64_t nnz = 0;
    // geometric runs of the kept beams (k_run_build, once per sequence): what phase A associates (k_assoc_runs)
    DevBuf<int> nrun, roff, gh_roff;
    DevBuf<double2> r_s, gh_rs;               // sum of the run's body points (the bounding circle's centre is sum / k)
    DevBuf<uint2> r_m, gh_rm;                 // radius (float bits, rounded up) | beams | first beam's offset in the pose << 16
    int64_t nruns = 0;
    int assoc_form = 1;                       // 1 by runs (k_assoc_runs), 0 beam by beam (k_assoc_group): icm_set_assoc_form
    DevBuf<unsigned long long> run_counts;    // [1] runs that went beam by beam, over the handle's life

    // state
    DevBuf<double> x_own, x0;
    double* x = nullptr;  // (T,3): x_own or a bound external buffer
    bool x_external = false;
    std::vector<double> h_map;  // current mapa_viejo (2,K) row-major
    int64_t K = 0, lact = 0;
    std::vector<double> h_counts;  // cant_obs_i after the last filtrar (L)
    bool scan0_empty = false;

    // association grid
    Grid grid;
    DevBuf<int> g_cell, fl_cid, fl_cell_cnt, fl_cell_fill, fl_info, fl_nn, fl_lab, fl_comp, fl_csize, fl_isl, fl_rank, fl_scan_tot;
    DevBuf<FlState> fl_state;
    DevBuf<double> fl_nd;
    DevBuf<LmRec> g_lm;
    int filtrar_path = 0;   // last sweep: 0 GPU without merges, 1 GPU with merges, 2 host routine
    DevBuf<GridParams> gpar;
    DevBuf<NeighRec> g_nb;   // per-cell 3x3 neighbourhood records (k_neigh_table)
    int max_cells = 0;
    DevBuf<double> fl_px, fl_py, fl_pc, counts_new;
    bool h_map_valid = true, gpu_filtrar = true;
    std::vector<LmRec> h_lm;
    DevBuf<double> mapx, mapy;

    // per-sweep
    DevBuf<unsigned short> st_k;    // beams of every staged entry (<= B <= 8192: two bytes)
    DevBuf<int> label, bloc, st_label, nent, isnew, ent_off, new_rank, e_val, e_k, sval, lm_off, flags, scan_tot;
    DevBuf<unsigned> e_key, skey;
    DevBuf<double> st_sx, st_sy, pose_c, pose_m, btx, bty;
    DevBuf<double2> e_b, e_wr, tgt;
    DevBuf<EntW> e_w;
    DevBuf<double> stats_own, off_sx, off_sy, off_n, y_raw, cnt_raw, diag;
    DevBuf<unsigned char> sort_tmp;
    // hierarchical running sums (k_chunk_l1 .. k_rec_push): records = chunks x kT1 slots
    DevBuf<int> rec_label;
    DevBuf<double> rec_s, rec_off, ms;   // [3][nrec], [3][nrec], [3][nsuper][L]
    int nchunks = 0, chunk_poses = 64, chunk_group = 1, nsuper = 0;
    DevBuf<int> st_off;      // where each pose's staged entries start (phase A: packed area or sparse area)
    StagingLayout stl;       // sizes of the staging area (staging_layout, icm_host.hpp: the one place that computes them)
    DevBuf<double> pre_x, pre_y;   // per-entry prefixes of the hierarchical path, one element per staging place (stl.prefix_stride)
    DevBuf<unsigned> pre_n;
    // icm_snapshot_state / icm_restore_state: device copy of the sweep state (poses, map, search structures)
    struct Snapshot {
        DevBuf<double> x, mapx, mapy, counts_new;
        DevBuf<int> g_cell;
        DevBuf<LmRec> g_lm;
        DevBuf<NeighRec> g_nb;
        DevBuf<GridParams> gpar;
        std::vector<double> h_map, h_counts;
        int64_t K = 0, lact = 0;
        bool h_map_valid = true, valid = false, dev_map_current = false;
    } snap;
    DevBuf<int> solve_flags;  // fused red-black solve: [nw] completion flags of the odd waves | [nw] deferral stamps of the even waves | [2] sync words (waves done, waves deferred)
    DevBuf<unsigned long long> solve_counts;   // over the handle's life: [0] even waves that deferred, [1] poses solved once more with the complete energy (an evaluation left the folded form's range)
    int solve_flag_waves = 0;
    int solve_epoch = 0;
    int* fl = nullptr;        // this sweep's block of 16 flag / counter words: the sweeps alternate between the two halves of `flags`,
    int fl_parity = 0;        //   and k_lm_l3 clears the other half for the next sweep (no memset launch at a sweep's head)
    bool fl_next_clean = false;
    int cu_count = 256;       // compute units of the device (icm_create)
    int fold_mode = -1;       // -1 automatic (fold-only main kernel + fix-up when the weights are isotropic), 0 never, 1 always
    // ghost pose of a shard (rank > 0): scan, kept beams, staged entries and moments of pose t_begin - 1
    bool ghost_uploaded = false;
    int ghost_n = 0;
    DevBuf<double> gh_ranges, gh_bd, gh_bx, gh_by, gh_s2, gh_sx, gh_sy, gh_rot, gh_m;
    DevBuf<int> gh_nkept, gh_boff, gh_bk, gh_label, gh_bloc, gh_st_label, gh_misc;   // gh_misc: [0] nent [1] isnew [2] st_off [3..4] plan (zeros) [8..23] flags
    DevBuf<unsigned short> gh_st_k;
    hipEvent_t ev_gh0 = nullptr, ev_gh1 = nullptr;
    bool ghost_pending = false;   // the ghost chain of this sweep is queued on the solve stream (ev_gh1)
    int fused_spin_limit = 1 << 17;   // polls (x ~0.2 us) an even wave waits for its odd neighbours before deferring
    int fuse_colours = 1;    // 1: both colours of an unsharded red-black sweep in one launch (k_solve_m_fused)
    bool ms_clean = false;   // the [superchunk x L] matrix is zero (cleared by the last fused solve launch, launch_fused_solve)
    int entry_path = -1;     // -1 automatic, 0 sort-based pipeline, 1 hierarchical (falls back when a table overflows)
    bool hier_ok = true;     // cleared by an overflow until the next icm_set_state
    int path_used = 0;       // pipeline of the last sweep: 0 sort-based, 1 hierarchical
    double* stats_all = nullptr;
    double* stats_send = nullptr;   // optional (icm_bind_exchange_send)
    int rank = 0, world = 1;
    // collectives issued by the library itself (icm_comm_init): RCCL communicator + the exchange buffers it owns
    ncclComm_t comm = nullptr;
    icm_allgather_fn transport = nullptr;   // icm_comm_init_transport: the caller's all-gather instead of RCCL's
    void* transport_user = nullptr;
    bool comm_ready = false;
    DevBuf<double> own_stats_all, own_stats_send, own_poses;
    int64_t comm_blk = 0;
    int64_t E = 0, n_new_loc = 0, lact_raw = 0;
    int lact0 = 0;
    bool brute = false, debug = false, per_beam = false, assoc_kept = false;
    double thr2 = 0.0;  // largest s with sqrt(s) <= dist_thr
    int hash_slots = 128;  // phase A's per-pose label table; grows to 256 on overflow
    int solve_quad = -1;   // -1 automatic, 0 one lane per pose, 1 one quad per pose
    int form = 0;  // 0 moments (lane per pose), 1 per beam, 2 per entry (wave per pose)
    int *pin_i = nullptr, *pin_i_dev = nullptr;   // pinned host words and their device-side address
    double* pin_d = nullptr;  // pinned staging: raw map download (3L)
    double *pin_map = nullptr, *pin_map_dev = nullptr;   // mapped, pinned [x (L) | y (L) | counters (L)]: the refined map straight from k_fl_finalize (icm_sweep)
    bool host_map_wanted = false;   // this sweep's Mapa.filtrar writes pin_map as well
    DevBuf<double> pack;         // refined map + counters packed for one download
    std::vector<double> h_pack;
    DevBuf<double> x_rows;       // (3,T) staging of the caller's pose layout (transposed to / from (T,3) on the device)
    struct Pinned { char* host; size_t bytes; char* dev; };
    std::vector<Pinned> pinned;  // caller-owned host ranges registered with the runtime (icm_pin_host): kernels read / write them in place
    // The drop-in call on a registered pose array (icm_sweep): the solves write every pose into the caller's array as well
    // (x_mirror, SolveArgs::xh), so there is no download; and when the caller hands back the array the last call filled,
    // there is no upload either -- a side-stream kernel checks, under phase A, that host and device poses still agree
    // (k_x_compare: stale word := x_epoch if not, and then solves and Mapa.filtrar leave everything alone and the call
    // starts over with an upload).
    double* x_mirror = nullptr;      // device-side address of the caller's (3,T) array for THIS call (null: none)
    const double* mirror_host = nullptr;   // the host array whose contents equal the device poses (as far as this library knows)
    DevBuf<int> x_stale;             // [1]: the epoch of the call whose check failed
    int x_epoch = 0;
    // icm_set_phase_timing: events on the handle's stream at the phase boundaries of a (sharded) sweep
    bool phase_timing = false;
    hipEvent_t ev_ph[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    double ph_ms[5] = {0, 0, 0, 0, 0};   // local | exchange (+ waiting for the slowest rank) | targets | solve | host time in finish
    int64_t ph_n = 0;
    int solve_ppw = 0;   // poses per wave of the one-launch solve: 0 automatic, 32 or 64 (ICM_SOLVE_PPW)
    int fault = 0;   // test hook (icm_set_fault): 1 = the next icm_sweep_local reports a HIP error, 2 = the next icm_sweep_targets does (behind the exchange)
    int64_t dropin_counts[3] = {0, 0, 0};   // icm_sweep calls: [0] started without an upload, [1] of those: the check failed (started over), [2] poses mirrored into the caller's array
    double h_x0[3] = {0, 0, 0};      // host copy of x0 as uploaded
    bool x_mirrored = false;         // the sweep's solve launch wrote the poses into x_mirror
    bool x_check = false;            // this sweep runs from the device's poses on the strength of a check still in flight (ev_cmp)
    bool x_check_wait = false;       // ... which the solve launch has not been ordered behind yet
    hipEvent_t ev_cmp = nullptr;
    bool dev_map_current = false;   // the device search structures hold exactly h_map (K, lact)
    std::vector<double> h_yraw, h_cntraw;
    bool raw_on_device = false;   // y_raw / cnt_raw of the last sweep have not been copied to h_yraw / h_cntraw yet

    // timing
    bool timing = false;
    double k_ms[KID_COUNT] = {0};
    int64_t k_n[KID_COUNT] = {0};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // raw-map download overlapped with the solves
    hipStream_t copy_stream = nullptr;
    hipStream_t solve_stream = nullptr;   // side stream of a shard's ghost-pose chain (launch_ghost)
    bool rot_valid = false;          // rot[] holds (cos, sin)(theta - pi/2) of the current poses (written by the solves; k_pose_rot otherwise)
    // Optimistic sweep (icm_sweep_classic): the whole sweep is queued without the host looking at phase A's counts
    // and overflow flags in the middle; the kernels that would replace state (solves, Mapa.filtrar) look at the
    // flags themselves, the host reads them with the filtrar result and repeats the sweep the careful way if set.
    bool optimistic = false;   // this sweep is queued whole (decided at its start, icm_sweep_local)
    DevBuf<unsigned long long> chunk_pub;   // k_chunk_l1 without the scan kernels: new-landmark count of every chunk, epoch-tagged
    unsigned scan_epoch = 0;
    bool scan_wanted = true;   // the next queued-whole sweep runs the scan kernels (new-landmark ranks, a fresh reservation plan)
    bool scan_ran = true;      // ... this sweep did
    bool opt_req = false;      // asked for: by icm_sweep_classic for its first attempt, by icm_set_optimistic for the phase calls
    DevBuf<double> rot;   // (cos, sin)(theta - pi/2) per pose of the shard, refreshed at the start of every sweep (k_pose_rot)
    hipEvent_t ev_map = nullptr, ev_copied = nullptr;
    bool map_ev_in_local = false;   // this sweep's ev_map is the stop event of k_lm_l3 (icm_sweep_local)
    bool defer_filtrar = false;     // set by the library's own sweep drivers: Mapa.filtrar's launches are queued behind the solve launch (icm_sweep_targets)
    bool filtrar_deferred = false;  // ... and this sweep's are still to be queued
    bool map_by_spinner = false;    // ... or there is no event: the side stream polls the word k_lm_l3's last workgroup sets
    bool l3_spin = true;            // (ICM_L3_EVENT=1: the stop event, for A/B runs; cleared for good by a wait that gave up)
    int64_t wait_giveups = 0;       // sweeps whose k_wait_word gave up (icm_get_wait_giveups)
    DevBuf<int> l3_done;            // [0] workgroups of k_lm_l3 through, [1] epoch of the launch that finished last, [2] epoch of a wait that gave up
    int l3_epoch = 0;
    bool map_copy_pending = false;
};

#define HIPCHK(h, call)                                                                             \
    do {                                                                                            \
        hipError_t e__ = (call);                                                                    \
        if (e__ != hipSuccess) {                                                                    \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e__);                          \
            return ICM_ERR_HIP;                                                                     \
        }                                                                                           \
    } while (0)

#define FAIL(h, code, msg)  \
    do {                    \
        (h)->err = (msg);   \
        return (code);      \
    } while (0)

// Launch bracket: with timing on, events around the launch and an accumulate (serialises
// the stream; used only for the per-kernel timing pass).
#define TIMED(h, kid, stmt)                                                    \
    do {                                                                       \
        if ((h)->timing) (void)hipEventRecord((h)->ev0, (h)->stream);          \
        stmt;                                                                  \
        if ((h)->timing) {                                                     \
            (void)hipEventRecord((h)->ev1, (h)->stream);                       \
            (void)hipEventSynchronize((h)->ev1);                               \
            float ms__ = 0.f;                                                  \
            (void)hipEventElapsedTime(&ms__, (h)->ev0, (h)->ev1);              \
            (h)->k_ms[kid] += ms__;                                            \
            (h)->k_n[kid] += 1;                                                \
        }                                                                      \
    } while (0)

constexpr int kWaitPolls = 1 << 14;   // k_wait_word: ~1 us a poll, i.e. ~16 ms -- phase A + levels 1-3 of the longest sequence this build takes are a few ms
constexpr int kStalePoses = 1001;   // internal (icm_sweep_finish -> icm_sweep): the sweep ran from device poses that were not the caller's; nothing was replaced
constexpr int kStaleWord = 20;      // of the host's mapped block: the epoch of the call whose pose check failed
static inline double host_now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}
static inline int nblocks_waves(int64_t nwaves) { return (int)((nwaves + kWavesPerBlock - 1) / kWavesPerBlock); }
static inline int nblocks_threads(int64_t n) { return (int)((n + kBlock - 1) / kBlock); }

extern "C" {

const char* icm_version(void) { return "icmslam-hip 0.5 (gfx950)"; }
const char* icm_build_id(void) { return ICM_BUILD_ID; }

int icm_flop_per_eval(void) { return ICM_FLOP_PER_EVAL; }
int icm_valu_per_eval(void) { return ICM_VALU_PER_EVAL; }

const char* icm_last_error(const icm_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

// ---- RCCL from inside the library ------------------------------------------------------------
// librccl is resolved at run time (the PyTorch-ROCm wheel bundles its own copy; a process must use
// ONE of them, so whatever is already loaded wins).
namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;   // optional
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
} g_rccl;

std::string g_rccl_path;   // icm_comm_set_library: the copy to load when none is loaded yet

bool rccl_load(std::string& err) {
    if (g_rccl.lib) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
    void* lib = nullptr;
    for (const char* n : names)   // already loaded by the host application?
        if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!lib && !g_rccl_path.empty()) lib = dlopen(g_rccl_path.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!lib)
        for (const char* n : names)
            if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) {
        err = std::string("RCCL not found: ") + dlerror();
        return false;
    }
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(dlsym(lib, "ncclCommAbort"));
    g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(lib, "ncclAllGather"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather) {
        err = "RCCL: missing symbols";
        return false;
    }
    g_rccl.lib = lib;
    return true;
}
}  // namespace


// The solve stream gets the highest priority: its few, register-hungry workgroups must win a slot
// whenever one opens beside the phase A/B grids of the main stream.
static hipError_t create_solve_stream(hipStream_t* s) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // (hi = greatest priority, numerically lowest)
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, hi);
}

int icm_create(const icm_config* cfg, int device, icm_handle** out) {
    if (!cfg || !out) {
        g_create_err = "icm_create: null argument";
        return ICM_ERR_ARG;
    }
    *out = nullptr;
    if (cfg->L <= 0 || cfg->L > (1 << 30)) {
        g_create_err = "icm_create: L out of range";
        return ICM_ERR_ARG;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_err = std::string("icm_create: no HIP device (") + hipGetErrorString(e) +
                       "); this library has no CPU fallback";
        return ICM_ERR_HIP;
    }
    if (device < 0 || device >= ndev) {
        g_create_err = "icm_create: device index out of range";
        return ICM_ERR_ARG;
    }
    icm_handle* h = new icm_handle();
    h->cfg = *cfg;
    h->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&h->ev0)) != hipSuccess || (e = hipEventCreate(&h->ev1)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = create_solve_stream(&h->solve_stream)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&h->ev_map, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&h->ev_copied, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&h->ev_cmp, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&h->ev_gh0, hipEventDisableTiming)) != hipSuccess || (e = hipEventCreateWithFlags(&h->ev_gh1, hipEventDisableTiming)) != hipSuccess ||
        (e = hipHostMalloc(reinterpret_cast<void**>(&h->pin_i), 64 * sizeof(int), hipHostMallocMapped)) != hipSuccess ||
        (e = hipHostGetDevicePointer(reinterpret_cast<void**>(&h->pin_i_dev), h->pin_i, 0)) != hipSuccess ||
        (e = hipHostMalloc(reinterpret_cast<void**>(&h->pin_map), (size_t)(3 * cfg->L) * sizeof(double), hipHostMallocMapped)) != hipSuccess ||
        (e = hipHostGetDevicePointer(reinterpret_cast<void**>(&h->pin_map_dev), h->pin_map, 0)) != hipSuccess ||
        (e = hipHostMalloc(reinterpret_cast<void**>(&h->pin_d), (size_t)(3 * cfg->L + 16 + 64) * sizeof(double))) != hipSuccess) {
        g_create_err = std::string("icm_create: ") + hipGetErrorString(e);
        delete h;
        return ICM_ERR_HIP;
    }
    std::memset(h->pin_i, 0, 64 * sizeof(int));
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) h->cu_count = cus;
    }
    {   // exact squared gate: the largest double whose correctly rounded sqrt is <= dist_thr
        const double thr = cfg->dist_thr;
        double s2 = thr * thr;
        if (thr >= 0.0 && std::isfinite(s2)) {
            while (std::sqrt(s2) > thr) s2 = std::nextafter(s2, 0.0);
            while (std::sqrt(std::nextafter(s2, INFINITY)) <= thr) s2 = std::nextafter(s2, INFINITY);
        } else {
            s2 = thr < 0.0 ? -1.0 : s2;
        }
        h->thr2 = s2;
    }
    if (const char* ev = std::getenv("ICM_L3_EVENT")) h->l3_spin = std::atoi(ev) == 0;
    if (const char* ev = std::getenv("ICM_SOLVE_PPW")) h->solve_ppw = std::atoi(ev);
    h->own_stream = true;
    h->h_counts.assign((size_t)cfg->L, 0.0);
    *out = h;
    return ICM_OK;
}

int icm_destroy(icm_handle* h) {
    if (!h) return ICM_OK;
    (void)hipSetDevice(h->device);
    if (h->solve_stream) (void)hipStreamSynchronize(h->solve_stream);
    if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    (void)hipStreamSynchronize(h->stream);
    DevBuf<double>* dd[] = {&h->ranges, &h->cosb, &h->sinb, &h->odo, &h->u, &h->bd, &h->bx, &h->by, &h->x_own, &h->x0,
                            &h->mapx, &h->mapy, &h->st_sx, &h->st_sy, &h->pose_m, &h->pose_c, &h->pose_s2, &h->btx, &h->bty, &h->stats_own, &h->off_sx, &h->off_sy, &h->off_n, &h->y_raw,
                            &h->cnt_raw, &h->diag};
    for (auto* b : dd) b->release();
    DevBuf<int>* di[] = {&h->nkept, &h->boff, &h->bk, &h->g_cell, &h->label, &h->bloc, &h->st_label,
                         &h->nent, &h->isnew, &h->ent_off, &h->new_rank, &h->e_val, &h->e_k, &h->sval, &h->lm_off, &h->flags, &h->scan_tot};
    for (auto* b : di) b->release();
    h->g_lm.release(); h->gpar.release(); h->g_nb.release(); h->st_k.release(); h->bxy.release();
    h->fl_nn.release(); h->fl_lab.release(); h->fl_comp.release(); h->fl_csize.release(); h->fl_isl.release(); h->fl_rank.release();
    h->fl_scan_tot.release(); h->fl_state.release(); h->fl_nd.release();
    h->fl_cid.release(); h->fl_cell_cnt.release(); h->fl_cell_fill.release(); h->fl_info.release();
    h->fl_px.release(); h->fl_py.release(); h->fl_pc.release(); h->counts_new.release();
    h->e_b.release(); h->e_wr.release(); h->tgt.release(); h->e_w.release();
    h->e_key.release();
    h->skey.release();
    h->sort_tmp.release();
    h->rec_label.release(); h->rec_s.release(); h->rec_off.release(); h->ms.release(); h->solve_flags.release(); h->solve_counts.release(); h->odo_cs.release(); h->pose_cs.release();
    h->gh_ranges.release(); h->gh_bd.release(); h->gh_bx.release(); h->gh_by.release(); h->gh_bxy.release(); h->gh_kmask.release(); h->kmask.release(); h->gh_s2.release(); h->gh_sx.release(); h->gh_sy.release();
    h->gh_rot.release(); h->gh_m.release(); h->gh_nkept.release(); h->gh_boff.release(); h->gh_bk.release(); h->gh_label.release(); h->gh_bloc.release();
    h->gh_st_label.release(); h->gh_misc.release(); h->gh_st_k.release();
    for (auto& e : h->ev_ph) if (e) (void)hipEventDestroy(e);
    if (h->ev_cmp) (void)hipEventDestroy(h->ev_cmp);
    h->x_stale.release();
    h->l3_done.release();
    if (h->ev_gh0) (void)hipEventDestroy(h->ev_gh0);
    if (h->ev_gh1) (void)hipEventDestroy(h->ev_gh1);
    h->snap.x.release(); h->snap.mapx.release(); h->snap.mapy.release(); h->snap.counts_new.release();
    h->snap.g_cell.release(); h->snap.g_lm.release(); h->snap.g_nb.release(); h->snap.gpar.release();
    for (auto& pr : h->pinned) (void)hipHostUnregister(pr.host);
    h->pinned.clear();
    if (h->pin_i) (void)hipHostFree(h->pin_i);
    if (h->pin_d) (void)hipHostFree(h->pin_d);
    if (h->pin_map) (void)hipHostFree(h->pin_map);
    h->x_rows.release(); h->pack.release();
    if (h->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(h->comm);
    h->own_stats_all.release(); h->own_stats_send.release(); h->own_poses.release();
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev_map) (void)hipEventDestroy(h->ev_map);
    if (h->ev_copied) (void)hipEventDestroy(h->ev_copied);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->solve_stream) { (void)hipStreamSynchronize(h->solve_stream); (void)hipStreamDestroy(h->solve_stream); }
    h->pre_x.release(); h->pre_y.release(); h->pre_n.release();
    h->rot.release(); h->st_off.release(); h->chunk_pub.release();
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return ICM_OK;
}

int icm_set_stream(icm_handle* h, void* s) {
    if (!h) return ICM_ERR_ARG;
    (void)hipSetDevice(h->device);
    if (h->solve_stream) (void)hipStreamSynchronize(h->solve_stream);
    (void)hipStreamSynchronize(h->stream);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    h->stream = reinterpret_cast<hipStream_t>(s);
    h->own_stream = false;
    return ICM_OK;
}

// The reference keeps poses as a (3,T) array (all x, all y, all theta); the kernels want one
// (x, y, theta) row per pose.  The change of layout runs on the device, so the host side of a
// drop-in call is one copy of the caller's array each way.
__global__ __launch_bounds__(256) void k_x_rows_to_poses(const double* __restrict__ rows, double* __restrict__ x, int T) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    x[3 * (size_t)t] = rows[t];
    x[3 * (size_t)t + 1] = rows[(size_t)T + t];
    x[3 * (size_t)t + 2] = rows[2 * (size_t)T + t];
}
__global__ __launch_bounds__(256) void k_x_poses_to_rows(const double* __restrict__ x, double* __restrict__ rows, int T) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    rows[t] = x[3 * (size_t)t];
    rows[(size_t)T + t] = x[3 * (size_t)t + 1];
    rows[2 * (size_t)T + t] = x[3 * (size_t)t + 2];
}

// Device-side address of a caller's host range that lies inside a range registered by icm_pin_host (null: not registered).
static double* pinned_alias(const icm_handle* h, const void* p, size_t bytes) {
    const char* c = static_cast<const char*>(p);
    for (const auto& pr : h->pinned)
        if (c >= pr.host && c + bytes <= pr.host + pr.bytes) return reinterpret_cast<double*>(pr.dev + (c - pr.host));
    return nullptr;
}

int icm_pin_host(icm_handle* h, void* ptr, size_t bytes) {
    if (!h) return ICM_ERR_ARG;
    if (!ptr || !bytes) FAIL(h, ICM_ERR_ARG, "icm_pin_host: null range");
    HIPCHK(h, hipSetDevice(h->device));
    if (pinned_alias(h, ptr, bytes)) return ICM_OK;
    for (const auto& pr : h->pinned)
        if (static_cast<char*>(ptr) < pr.host + pr.bytes && pr.host < static_cast<char*>(ptr) + bytes)
            FAIL(h, ICM_ERR_ARG, "icm_pin_host: the range overlaps a registered one (icm_unpin_host it first)");
    void* dev = nullptr;
    HIPCHK(h, hipHostRegister(ptr, bytes, hipHostRegisterMapped));
    hipError_t e = hipHostGetDevicePointer(&dev, ptr, 0);
    if (e != hipSuccess) {
        (void)hipHostUnregister(ptr);
        FAIL(h, ICM_ERR_HIP, std::string("icm_pin_host: hipHostGetDevicePointer: ") + hipGetErrorString(e));
    }
    h->pinned.push_back({static_cast<char*>(ptr), bytes, static_cast<char*>(dev)});
    return ICM_OK;
}

int icm_unpin_host(icm_handle* h, void* ptr) {
    if (!h) return ICM_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    for (size_t i = 0; i < h->pinned.size(); ++i)
        if (h->pinned[i].host == static_cast<char*>(ptr)) {
            HIPCHK(h, hipStreamSynchronize(h->stream));   // (a kernel may still be writing into it)
            HIPCHK(h, hipHostUnregister(ptr));
            h->pinned.erase(h->pinned.begin() + (long)i);
            h->mirror_host = nullptr;
            return ICM_OK;
        }
    FAIL(h, ICM_ERR_ARG, "icm_unpin_host: not a registered range");
}

// Do the caller's (3,T) poses (registered host memory, read over PCIe) still equal the device's (T,3) ones, bit for bit?
__global__ __launch_bounds__(256) void k_x_compare(const double* __restrict__ rows, const double* __restrict__ x, int T,
                                                   int* __restrict__ stale_dev, int* __restrict__ stale_host, int epoch) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const unsigned long long* r = reinterpret_cast<const unsigned long long*>(rows);
    const unsigned long long* d = reinterpret_cast<const unsigned long long*>(x);
    const bool differ = r[t] != d[3 * (size_t)t] || r[(size_t)T + t] != d[3 * (size_t)t + 1] || r[2 * (size_t)T + t] != d[3 * (size_t)t + 2];
    if (differ) {
        *stale_dev = epoch;
        *stale_host = epoch;
    }
}

int icm_upload(icm_handle* h, const double* ranges, const double* odo, const double* u, const double* cosb,
               const double* sinb, int64_t T, int64_t B, int64_t t_begin, int64_t t_end) {
    if (!h) return ICM_ERR_ARG;
    if (!ranges || !odo || !u || !cosb || !sinb) FAIL(h, ICM_ERR_ARG, "icm_upload: null pointer");
    if (T < 2 || B < 1 || t_begin < 0 || t_end > T || t_begin >= t_end) FAIL(h, ICM_ERR_ARG, "icm_upload: bad T/B/shard");
    if (T > (1 << 30) || B > 8192 || (t_end - t_begin) * B > (int64_t)2000000000) FAIL(h, ICM_ERR_UNSUPPORTED, "icm_upload: sequence too large for 32-bit beam indices");
    HIPCHK(h, hipSetDevice(h->device));
    h->T = T; h->B = B; h->t_begin = t_begin; h->nloc = t_end - t_begin;
    const size_t nr = (size_t)h->nloc * (size_t)B;
    HIPCHK(h, h->ranges.reserve(nr));
    HIPCHK(h, h->cosb.reserve((size_t)B));
    HIPCHK(h, h->sinb.reserve((size_t)B));
    HIPCHK(h, h->odo.reserve(3 * (size_t)T));
    HIPCHK(h, h->u.reserve(2 * (size_t)T));
    HIPCHK(h, hipMemcpyAsync(h->ranges.p, ranges, nr * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->cosb.p, cosb, (size_t)B * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->sinb.p, sinb, (size_t)B * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->odo.p, odo, 3 * (size_t)T * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->u.p, u, 2 * (size_t)T * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, h->odo_cs.reserve(2 * (size_t)T));
    HIPCHK(h, h->pose_cs.reserve(2 * (size_t)T));
    k_odo_trig<<<nblocks_threads(T), kBlock, 0, h->stream>>>(h->odo.p, (int)T, h->odo_cs.p);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->uploaded = true;
    h->ghost_uploaded = false;
    h->ghost_n = 0;
    h->snap.valid = false;
    h->prefiltered = false;
    h->have_state = false;
    return ICM_OK;
}

// The scan of pose t_begin - 1 (the last pose of the shard below): ranks > 0 of a pose-sharded job solve that pose as
// well (SolveSeg, the ghost pose), so a sharded sweep needs no exchange between its colours.  After icm_upload, before
// icm_prefilter.
int icm_upload_ghost_scan(icm_handle* h, const double* ranges_row) {
    if (!h) return ICM_ERR_ARG;
    if (!h->uploaded || h->prefiltered) FAIL(h, ICM_ERR_ARG, "icm_upload_ghost_scan: between icm_upload and icm_prefilter");
    if (!ranges_row) FAIL(h, ICM_ERR_ARG, "icm_upload_ghost_scan: null pointer");
    if (h->t_begin < 2 || (h->t_begin & 1)) FAIL(h, ICM_ERR_ARG, "icm_upload_ghost_scan: a shard with a ghost pose starts at an even pose >= 2");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, h->gh_ranges.reserve((size_t)h->B));
    HIPCHK(h, hipMemcpyAsync(h->gh_ranges.p, ranges_row, (size_t)h->B * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->ghost_uploaded = true;
    return ICM_OK;
}

// Buffers of the landmark table, its search grid and Mapa.filtrar (sized by L alone).
static int reserve_map_buffers(icm_handle* h) {
    const size_t L = (size_t)h->cfg.L;
    if (8 * (int64_t)L + 4096 > (int64_t)1 << 25) FAIL(h, ICM_ERR_CAPACITY, "landmark capacity L too large for the search grid's 32-bit record offsets (L <= 4 193 792)");
    h->max_cells = (int)(8 * L + 4096);   // bound of both grid builders (build_grid on the host, k_fl_* on the device)
    const size_t nc = (size_t)h->max_cells + 2;
    HIPCHK(h, h->y_raw.reserve(2 * L)); HIPCHK(h, h->cnt_raw.reserve(L));
    HIPCHK(h, h->g_lm.reserve(L)); HIPCHK(h, h->gpar.reserve(1));
    HIPCHK(h, h->fl_nn.reserve(L)); HIPCHK(h, h->fl_lab.reserve(L)); HIPCHK(h, h->fl_comp.reserve(kCompStride * L)); HIPCHK(h, h->fl_csize.reserve(L));
    HIPCHK(h, h->fl_isl.reserve(L + 1)); HIPCHK(h, h->fl_rank.reserve(L + 2)); HIPCHK(h, h->fl_nd.reserve(L)); HIPCHK(h, h->fl_state.reserve(1));
    HIPCHK(h, h->fl_scan_tot.reserve(2 * (nc / kScanTile + 2)));
    HIPCHK(h, h->fl_cid.reserve(L)); HIPCHK(h, h->fl_cell_cnt.reserve(nc)); HIPCHK(h, h->fl_cell_fill.reserve(nc));
    HIPCHK(h, h->fl_info.reserve(8)); HIPCHK(h, h->fl_px.reserve(L)); HIPCHK(h, h->fl_py.reserve(L)); HIPCHK(h, h->fl_pc.reserve(L));
    HIPCHK(h, h->counts_new.reserve(L));
    HIPCHK(h, h->mapx.reserve(L)); HIPCHK(h, h->mapy.reserve(L));
    HIPCHK(h, h->g_cell.reserve(nc));
    HIPCHK(h, h->g_nb.reserve((size_t)h->max_cells));
    return ICM_OK;
}

// Cuts the kept beams of nloc scans into geometric runs (k_run_build: count, exclusive scan, fill).  Thresholds scale
// with the gate: a new run where consecutive body points are more than 0.35 dist_thr apart or a point is more than
// 0.5 dist_thr from the run's first one (a trunk's visible arc is one run; a hedge is cut into pieces small enough
// for the bounding-circle test of k_assoc_runs to settle them).
static int build_runs(icm_handle* h, const int* boff, const double2* bxy, int nloc, DevBuf<int>& nrun, DevBuf<int>& roff,
                      DevBuf<double2>& r_s, DevBuf<uint2>& r_m, int64_t* nruns_out) {
    const double thr = h->cfg.dist_thr > 0.0 ? h->cfg.dist_thr : 1.0;
    const double gap = 0.35 * thr, ext = 0.5 * thr;
    HIPCHK(h, nrun.reserve((size_t)nloc + 1));
    HIPCHK(h, roff.reserve((size_t)nloc + 1));
    TIMED(h, KID_RUN_BUILD, (k_run_build<false><<<nblocks_threads(nloc), kBlock, 0, h->stream>>>(boff, bxy, nloc, gap * gap, ext * ext, nrun.p, nullptr, nullptr, nullptr)));
    k_exscan_i32<<<1, 1024, 0, h->stream>>>(nrun.p, roff.p, nloc);
    int total = 0;
    HIPCHK(h, hipMemcpyAsync(&total, roff.p + nloc, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const size_t nr = (size_t)std::max(total, 1);
    HIPCHK(h, r_s.reserve(nr)); HIPCHK(h, r_m.reserve(nr));
    TIMED(h, KID_RUN_BUILD, (k_run_build<true><<<nblocks_threads(nloc), kBlock, 0, h->stream>>>(boff, bxy, nloc, gap * gap, ext * ext, nullptr, roff.p, r_s.p, r_m.p)));
    HIPCHK(h, hipGetLastError());
    if (nruns_out) *nruns_out = total;
    return ICM_OK;
}

int icm_prefilter(icm_handle* h, int64_t* nnz_out) {
    if (!h) return ICM_ERR_ARG;
    if (!h->uploaded) FAIL(h, ICM_ERR_ARG, "icm_prefilter: call icm_upload first");
    HIPCHK(h, hipSetDevice(h->device));
    const int nloc = (int)h->nloc, B = (int)h->B;
    HIPCHK(h, h->nkept.reserve((size_t)nloc + 1));
    HIPCHK(h, h->boff.reserve((size_t)nloc + 1));
    const size_t lds = (size_t)kWavesPerBlock * (size_t)B * (3 * sizeof(double) + sizeof(int));
    if (lds > 160 * 1024) FAIL(h, ICM_ERR_UNSUPPORTED, "icm_prefilter: too many beams per scan for the LDS staging");
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_prefilter<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_prefilter<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int nb = nblocks_waves(nloc);
    HIPCHK(h, h->kmask.reserve((size_t)nloc * (size_t)((B + kWave - 1) / kWave)));
    TIMED(h, KID_PREFILTER, (k_prefilter<false><<<nb, kBlock, lds, h->stream>>>(h->ranges.p, h->cosb.p, h->sinb.p, nloc, B, h->cfg.rango_laser_max, h->cfg.dist_thr, h->nkept.p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, h->thr2, h->kmask.p)));
    k_exscan_i32<<<1, 1024, 0, h->stream>>>(h->nkept.p, h->boff.p, nloc);
    h->h_boff.assign((size_t)nloc + 1, 0);
    HIPCHK(h, hipMemcpyAsync(h->h_boff.data(), h->boff.p, ((size_t)nloc + 1) * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->nnz = h->h_boff[(size_t)nloc];
    const size_t nz = (size_t)std::max<int64_t>(h->nnz, 1);
    HIPCHK(h, h->bk.reserve(nz));
    HIPCHK(h, h->bd.reserve(nz));
    HIPCHK(h, h->bx.reserve(nz));
    HIPCHK(h, h->by.reserve(nz));
    HIPCHK(h, h->bxy.reserve(nz));
    HIPCHK(h, h->pose_s2.reserve(3 * (size_t)nloc));
    HIPCHK(h, h->pose_c.reserve(3 * (size_t)nloc));
    HIPCHK(h, h->pose_m.reserve(17 * (size_t)nloc));
    HIPCHK(h, h->rot.reserve(2 * (size_t)nloc));
    TIMED(h, KID_PREFILTER, (k_prefilter<true><<<nb, kBlock, lds, h->stream>>>(h->ranges.p, h->cosb.p, h->sinb.p, nloc, B, h->cfg.rango_laser_max, h->cfg.dist_thr, nullptr, h->boff.p, h->bk.p, h->bd.p, h->bx.p, h->by.p, h->pose_s2.p, h->bxy.p, h->thr2, h->kmask.p)));
    // the geometric runs of every scan (k_assoc_runs associates runs, not beams): counted, scanned, filled
    {
        int rr = build_runs(h, h->boff.p, h->bxy.p, nloc, h->nrun, h->roff, h->r_s, h->r_m, &h->nruns);
        if (rr) return rr;
        HIPCHK(h, h->run_counts.reserve(2));
        HIPCHK(h, hipMemsetAsync(h->run_counts.p, 0, 2 * sizeof(unsigned long long), h->stream));
    }
    // per-sweep buffers sized by the kept beams
    // staged entries (packed area, sparse area behind it) and the per-entry prefixes that live at the same places
    if (!staging_layout(h->nnz, nloc, h->stl)) FAIL(h, ICM_ERR_CAPACITY, "too many kept beams for 32-bit entry offsets");
    const size_t nst = (size_t)h->stl.entries, npre = (size_t)h->stl.prefix_stride;
    HIPCHK(h, h->label.reserve(nz)); HIPCHK(h, h->bloc.reserve(nz)); HIPCHK(h, h->st_label.reserve(nst));
    HIPCHK(h, h->st_k.reserve(nst)); HIPCHK(h, h->st_sx.reserve(nst)); HIPCHK(h, h->st_sy.reserve(nst));
    HIPCHK(h, h->st_off.reserve((size_t)nloc + 1));
    HIPCHK(h, h->pre_x.reserve(npre)); HIPCHK(h, h->pre_y.reserve(npre)); HIPCHK(h, h->pre_n.reserve(npre));
    HIPCHK(h, h->btx.reserve(nz)); HIPCHK(h, h->bty.reserve(nz));
    // sort-based pipeline: compact per-entry records (at most one entry per kept beam)
    HIPCHK(h, h->e_key.reserve(nz)); HIPCHK(h, h->skey.reserve(nz)); HIPCHK(h, h->e_val.reserve(nz + kWave));
    HIPCHK(h, h->sval.reserve(nz)); HIPCHK(h, h->e_k.reserve(nz)); HIPCHK(h, h->e_b.reserve(nz));
    HIPCHK(h, h->e_w.reserve(nz + kWave));
    HIPCHK(h, h->e_wr.reserve(nz)); HIPCHK(h, h->tgt.reserve(nz));
    HIPCHK(h, h->scan_tot.reserve(2 * ((size_t)nloc / kScanTile + 2)));
    HIPCHK(h, h->nent.reserve((size_t)nloc + 1)); HIPCHK(h, h->isnew.reserve((size_t)nloc + 1));
    HIPCHK(h, h->ent_off.reserve((size_t)nloc + 1)); HIPCHK(h, h->new_rank.reserve((size_t)nloc + 1));
    HIPCHK(h, hipMemsetAsync(h->ent_off.p, 0, ((size_t)nloc + 1) * sizeof(int), h->stream));   // (no reservation plan yet: phase A)
    const size_t L = (size_t)h->cfg.L;
    HIPCHK(h, h->lm_off.reserve(L + 2)); HIPCHK(h, h->flags.reserve(32)); h->fl = h->flags.p; h->fl_parity = 0; h->fl_next_clean = false;   // [0..7] the sweep's flags and host words, [8..9] totals of a sweep without scan kernels
    HIPCHK(h, h->stats_own.reserve(3 * L + 8)); HIPCHK(h, h->off_sx.reserve(L)); HIPCHK(h, h->off_sy.reserve(L));
    HIPCHK(h, h->off_n.reserve(L));
    {
        int rcm = reserve_map_buffers(h);
        if (rcm) return rcm;
    }
    // poses per chunk: a chunk is one wave's serial work, so short sequences take short chunks
    h->chunk_poses = nloc >= 65536 ? 64 : (nloc >= 16384 ? 32 : 16);
    h->nchunks = (nloc + h->chunk_poses - 1) / h->chunk_poses;
    h->chunk_group = (h->nchunks + kMaxSuper - 1) / kMaxSuper;
    h->nsuper = (h->nchunks + h->chunk_group - 1) / h->chunk_group;
    {
        const size_t nrec = (size_t)h->nchunks * kT1;
        HIPCHK(h, h->rec_label.reserve(nrec)); HIPCHK(h, h->rec_s.reserve(3 * nrec)); HIPCHK(h, h->rec_off.reserve(3 * nrec));
        {
            const size_t had = h->chunk_pub.cap;
            HIPCHK(h, h->chunk_pub.reserve((size_t)h->nchunks + 1));
            if (h->chunk_pub.cap != had) {   // fresh storage: its tags must not match any epoch
                HIPCHK(h, hipMemsetAsync(h->chunk_pub.p, 0, h->chunk_pub.cap * sizeof(unsigned long long), h->stream));
                h->scan_epoch = 0;
            }
        }
        HIPCHK(h, h->ms.reserve(3 * (size_t)h->nsuper * L));
        h->ms_clean = false;
    }
    h->solve_epoch = 0;
    h->solve_flag_waves = 0;   // (flags are cleared with the epoch: launch_fused_solve)
    h->ghost_n = 0;
    if (h->ghost_uploaded) {
        // filtrar_z of the ghost scan (pose t_begin - 1) into the ghost's own arrays, and room for its staged entries:
        // a one-pose phase A launch with an all-zero reservation plan stages them at sparse0 = kWave
        const size_t Bz = (size_t)B, gst = (size_t)kWave + Bz + 512;
        HIPCHK(h, h->gh_nkept.reserve(2)); HIPCHK(h, h->gh_boff.reserve(2)); HIPCHK(h, h->gh_bk.reserve(Bz));
        HIPCHK(h, h->gh_bd.reserve(Bz)); HIPCHK(h, h->gh_bx.reserve(Bz)); HIPCHK(h, h->gh_by.reserve(Bz)); HIPCHK(h, h->gh_bxy.reserve(Bz)); HIPCHK(h, h->gh_kmask.reserve((Bz + kWave - 1) / kWave)); HIPCHK(h, h->gh_s2.reserve(3));
        HIPCHK(h, h->gh_label.reserve(Bz)); HIPCHK(h, h->gh_bloc.reserve(Bz));
        HIPCHK(h, h->gh_st_label.reserve(gst)); HIPCHK(h, h->gh_st_k.reserve(gst)); HIPCHK(h, h->gh_sx.reserve(gst)); HIPCHK(h, h->gh_sy.reserve(gst));
        HIPCHK(h, h->gh_misc.reserve(32)); HIPCHK(h, h->gh_rot.reserve(2)); HIPCHK(h, h->gh_m.reserve(17));
        HIPCHK(h, hipMemsetAsync(h->gh_misc.p, 0, 32 * sizeof(int), h->stream));
        HIPCHK(h, hipMemsetAsync(h->gh_m.p, 0, 17 * sizeof(double), h->stream));
        k_prefilter<false><<<1, kBlock, lds, h->stream>>>(h->gh_ranges.p, h->cosb.p, h->sinb.p, 1, B, h->cfg.rango_laser_max, h->cfg.dist_thr, h->gh_nkept.p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, h->thr2, h->gh_kmask.p);
        k_exscan_i32<<<1, 1024, 0, h->stream>>>(h->gh_nkept.p, h->gh_boff.p, 1);
        k_prefilter<true><<<1, kBlock, lds, h->stream>>>(h->gh_ranges.p, h->cosb.p, h->sinb.p, 1, B, h->cfg.rango_laser_max, h->cfg.dist_thr, nullptr, h->gh_boff.p, h->gh_bk.p, h->gh_bd.p, h->gh_bx.p, h->gh_by.p, h->gh_s2.p, h->gh_bxy.p, h->thr2, h->gh_kmask.p);
        int gb[2] = {0, 0};
        HIPCHK(h, hipMemcpyAsync(gb, h->gh_boff.p, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->ghost_n = gb[1];
        {
            DevBuf<int> gnr;   // (the ghost's run count: scratch)
            int rr = build_runs(h, h->gh_boff.p, h->gh_bxy.p, 1, gnr, h->gh_roff, h->gh_rs, h->gh_rm, nullptr);
            gnr.release();
            if (rr) return rr;
        }
    }
    // the hand-off words of the sweep's side launches (k_wait_word / k_lm_l3, k_x_compare): cleared here, in stream order, in
    // front of the synchronisation below -- never beside a launch that polls them (hipMemset on the null stream is not
    // ordered against the handle's non-blocking streams)
    HIPCHK(h, h->l3_done.reserve(4)); HIPCHK(h, h->x_stale.reserve(2));
    HIPCHK(h, hipMemsetAsync(h->l3_done.p, 0, 4 * sizeof(int), h->stream));
    HIPCHK(h, hipMemsetAsync(h->x_stale.p, 0, 2 * sizeof(int), h->stream));
    h->l3_epoch = h->x_epoch = 0;
    h->pin_i[kStaleWord] = 0;
    size_t tmp_bytes = 0;
    HIPCHK(h, rocprim::radix_sort_pairs(nullptr, tmp_bytes, h->e_key.p, h->skey.p, h->e_val.p, h->sval.p, nz, 0, 32, h->stream));
    HIPCHK(h, h->sort_tmp.reserve(tmp_bytes + 256));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->prefiltered = true;
    if (nnz_out) *nnz_out = h->nnz;
    return ICM_OK;
}

int icm_get_kept(icm_handle* h, int64_t* offsets, int32_t* beam_index, double* d, double* bx, double* by) {
    if (!h) return ICM_ERR_ARG;
    if (!h->prefiltered) FAIL(h, ICM_ERR_ARG, "icm_get_kept: call icm_prefilter first");
    HIPCHK(h, hipSetDevice(h->device));
    if (offsets)
        for (size_t i = 0; i <= (size_t)h->nloc; ++i) offsets[i] = h->h_boff[i];
    const size_t nz = (size_t)h->nnz;
    if (nz) {
        if (beam_index) HIPCHK(h, hipMemcpy(beam_index, h->bk.p, nz * sizeof(int), hipMemcpyDeviceToHost));
        if (d) HIPCHK(h, hipMemcpy(d, h->bd.p, nz * sizeof(double), hipMemcpyDeviceToHost));
        if (bx) HIPCHK(h, hipMemcpy(bx, h->bx.p, nz * sizeof(double), hipMemcpyDeviceToHost));
        if (by) HIPCHK(h, hipMemcpy(by, h->by.p, nz * sizeof(double), hipMemcpyDeviceToHost));
    }
    return ICM_OK;
}

// Upload the current mapa_viejo and its search grid (per sweep; K <= L landmarks).
static int upload_map(icm_handle* h) {
    // columns the reference can match against: mapa_referencia[:, :Lact] (numpy clamps the
    // slice to the K columns that exist), scripts/ICM_SLAM_tools.py:169
    const int64_t km = std::min(h->K, h->lact);
    const double* mx = h->h_map.data();
    const double* my = h->h_map.data() + h->K;
    build_grid(mx, my, km, h->cfg.dist_thr, h->grid);
    const Grid& g = h->grid;
    HIPCHK(h, h->g_cell.reserve(g.cell_start.size()));
    HIPCHK(h, hipMemcpyAsync(h->g_cell.p, g.cell_start.data(), g.cell_start.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    if (km > 0) {
        h->h_lm.resize((size_t)km);
        for (int64_t i = 0; i < km; ++i) h->h_lm[(size_t)i] = LmRec{g.lx[(size_t)i], g.ly[(size_t)i], g.id[(size_t)i], 0, 0, 0};
        HIPCHK(h, hipMemcpyAsync(h->g_lm.p, h->h_lm.data(), (size_t)km * sizeof(LmRec), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->mapx.p, mx, (size_t)km * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->mapy.p, my, (size_t)km * sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    GridParams gp{g.gx0, g.gy0, g.inv, g.nx, g.ny};
    HIPCHK(h, hipMemcpyAsync(h->gpar.p, &gp, sizeof(gp), hipMemcpyHostToDevice, h->stream));
    if ((int64_t)g.nx * g.ny > (int64_t)h->max_cells) FAIL(h, ICM_ERR_CAPACITY, "upload_map: search grid larger than the neighbourhood table");
    TIMED(h, KID_NEIGH, (k_neigh_table<<<nblocks_threads((int64_t)g.nx * g.ny), kBlock, 0, h->stream>>>(GridView{h->gpar.p, h->g_cell.p, h->g_lm.p, nullptr}, h->g_nb.p, h->max_cells)));
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));  // host vectors are reused
    h->h_map_valid = true;
    h->dev_map_current = true;
    return ICM_OK;
}

// wait_host: the caller's arrays may be reused as soon as this returns (the public entry point);
// icm_sweep keeps them for the whole call and lets the copies run behind the queue.
static int set_state_impl(icm_handle* h, const double* x, const double* x0, const double* map_in, int64_t K, int64_t lact_in, bool wait_host) {
    if (!h) return ICM_ERR_ARG;
    if (!h->prefiltered) FAIL(h, ICM_ERR_ARG, "icm_set_state: call icm_upload + icm_prefilter first");
    if (!x || !x0 || (K > 0 && !map_in)) FAIL(h, ICM_ERR_ARG, "icm_set_state: null pointer");
    if (K < 0 || K > h->cfg.L || lact_in < 0) FAIL(h, ICM_ERR_ARG, "icm_set_state: K outside [0, L]");
    if (lact_in == 0) FAIL(h, ICM_ERR_UNSUPPORTED, "icm_set_state: landmarks_actuales == 0 (first-scan clustering branch, scripts/ICM_SLAM_tools.py:160-165) is not part of the sweep");
    if (lact_in < K) FAIL(h, ICM_ERR_UNSUPPORTED, "icm_set_state: landmarks_actuales < columns of mapa_viejo is not supported");
    if (lact_in > h->cfg.L) FAIL(h, ICM_ERR_INDEX, "icm_set_state: landmarks_actuales > L");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t T = (size_t)h->T;
    if (!h->x_external) {  // (re)size with the sequence: a handle may be re-used for a longer one
        HIPCHK(h, h->x_own.reserve(3 * T));
        h->x = h->x_own.p;
    }
    HIPCHK(h, h->x0.reserve(3));
    // (a registered array, icm_pin_host, is copied by DMA without staging: 0.054 against 0.078 ms for S2's 2.4 MB; a layout
    // kernel reading the host array itself over PCIe took 0.09)
    HIPCHK(h, h->x_rows.reserve(3 * T));
    HIPCHK(h, hipMemcpyAsync(h->x_rows.p, x, 3 * T * sizeof(double), hipMemcpyHostToDevice, h->stream));
    k_x_rows_to_poses<<<(int)((T + 255) / 256), 256, 0, h->stream>>>(h->x_rows.p, h->x, (int)T);
    h->mirror_host = nullptr;
    std::memcpy(h->h_x0, x0, sizeof(h->h_x0));
    HIPCHK(h, hipMemcpyAsync(h->x0.p, x0, 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipGetLastError());
    // The driver loop hands back the map the previous sweep returned (mapa_viejo = copy(mapa_refinado),
    // scripts/ICM_ROS.py:311): if these are exactly the values the device structures were built from,
    // the search grid of the last Mapa.filtrar is reused instead of being rebuilt and uploaded.
    const bool same_map = h->dev_map_current && h->h_map_valid && K == h->K && lact_in == h->lact &&
                          h->h_map.size() == 2 * (size_t)K &&
                          (K == 0 || std::memcmp(map_in, h->h_map.data(), 2 * (size_t)K * sizeof(double)) == 0);
    if (!same_map) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->h_map.assign(map_in, map_in + 2 * K);
        h->K = K;
        h->lact = lact_in;
        int rc = upload_map(h);
        if (rc) return rc;
    }
    h->have_state = true;
    h->rot_valid = false;
    h->scan_wanted = true;
    h->hier_ok = true;
    if (wait_host) HIPCHK(h, hipStreamSynchronize(h->stream));
    return ICM_OK;
}

int icm_set_state(icm_handle* h, const double* x, const double* x0, const double* map_in, int64_t K, int64_t lact_in) {
    return set_state_impl(h, x, x0, map_in, K, lact_in, true);
}

// A rank's statistics message: [sum x (L) | sum y (L) | n (L) | header (kStatsHeader)].  Header: [0] landmarks the rank
// created, [1] 0 = fine, 1 = one of its tables overflowed (a sweep queued whole: every rank then leaves its state alone),
// >= 2 = the rank failed in the careful form with error code -([1]) ... see sweep_sharded_once, [2..4] its first pose,
// [5..7] its last pose, [8..10] its last pose but one -- values at the START of the sweep, i.e. the previous sweep's
// results: what the neighbours need as OLD values (the rank above: the two poses around its ghost solve; the rank
// below: the pose after its last odd pose).  One message, one collective per sweep.
constexpr int kStatsHeader = 16;
int64_t icm_stats_stride(const icm_handle* h) { return h ? 3 * h->cfg.L + kStatsHeader : 0; }

int icm_bind_exchange(icm_handle* h, void* stats_all_dev, int rank, int world) {
    if (!h) return ICM_ERR_ARG;
    if (world < 1 || rank < 0 || rank >= world) FAIL(h, ICM_ERR_ARG, "icm_bind_exchange: bad rank/world");
    if (world > 1 && !stats_all_dev) FAIL(h, ICM_ERR_ARG, "icm_bind_exchange: null exchange buffer");
    h->stats_all = reinterpret_cast<double*>(stats_all_dev);
    h->rank = rank;
    h->world = world;
    return ICM_OK;
}

int icm_bind_exchange_send(icm_handle* h, void* stats_send_dev) {
    if (!h) return ICM_ERR_ARG;
    h->stats_send = reinterpret_cast<double*>(stats_send_dev);
    return ICM_OK;
}

// first / last / last-but-one pose of the shard, clamped into the sequence (values nobody reads where they do not exist)
static int edge_first(const icm_handle* h) { return (int)std::min<int64_t>(h->t_begin, std::max<int64_t>(h->T - 1, 0)); }
static int edge_last(const icm_handle* h) {
    return (int)std::min<int64_t>(std::max<int64_t>(h->t_begin + h->nloc - 1, 0), std::max<int64_t>(h->T - 1, 0));
}
static int edge_last2(const icm_handle* h) { return std::max(edge_last(h) - 1, 0); }

// this rank's slot of the landmark statistics: the send buffer if one is bound
static double* stats_slot(icm_handle* h) {
    return h->stats_send ? h->stats_send : h->stats_all + (size_t)h->rank * (size_t)icm_stats_stride(h);
}

int icm_bind_pose_buffer(icm_handle* h, void* x_dev) {
    if (!h) return ICM_ERR_ARG;
    if (!x_dev) FAIL(h, ICM_ERR_ARG, "icm_bind_pose_buffer: null buffer");
    h->x = reinterpret_cast<double*>(x_dev);
    h->mirror_host = nullptr;
    h->x_external = true;
    h->rot_valid = false;
    return ICM_OK;
}

void* icm_pose_buffer(icm_handle* h) {
    if (!h) return nullptr;
    if (!h->x_external) {
        (void)hipSetDevice(h->device);
        if (h->x_own.reserve(3 * (size_t)h->T) != hipSuccess) return nullptr;
        h->x = h->x_own.p;
    }
    return h->x;
}

// Header of a rank's statistics message (layout: icm_stats_stride above).
// n_new_dev / flags_dev (nullable): the values on the device, for a sweep queued without a host look at them
// ([1] = 1 when one of this rank's tables overflowed or its labels exceed L: every rank then leaves state alone).
__global__ void k_set_header(double* stats, int L, double n_new, double flags, const double* __restrict__ x, int first, int last,
                             int last2, const int* __restrict__ n_new_dev = nullptr, const int* __restrict__ flags_dev = nullptr) {
    double* hd = stats + 3 * (size_t)L;
    hd[0] = n_new_dev ? (double)*n_new_dev : n_new;
    hd[1] = flags_dev ? ((flags_dev[0] | flags_dev[1] | flags_dev[2]) ? 1.0 : 0.0) : flags;
    for (int i = 0; i < 3; ++i) {
        hd[2 + i] = x[3 * (size_t)first + i];
        hd[5 + i] = x[3 * (size_t)last + i];
        hd[8 + i] = x[3 * (size_t)last2 + i];
    }
}

// The neighbours' boundary poses out of their headers into this rank's copy of the pose array: below = the pose index
// t_begin - 1 (the ghost pose; its predecessor t_begin - 2 comes with it, and the ghost's rotation pair is formed from
// its owner's value), above = the pose index t_begin + nloc; -1 = none.
__global__ void k_halo_from_headers(double* __restrict__ x, const double* __restrict__ stats_all, int stride, int L, int rank,
                                    int below, int above, double* __restrict__ ghost_rot, double* __restrict__ cs) {
    const int i = threadIdx.x;
    if (i < 3) {
        if (below >= 0) {
            const double* hd = stats_all + (size_t)(rank - 1) * stride + 3 * (size_t)L;
            x[3 * (size_t)below + i] = hd[5 + i];
            if (below >= 1) x[3 * (size_t)(below - 1) + i] = hd[8 + i];
        }
    } else if (i < 6) {
        if (above >= 0) x[3 * (size_t)above + (i - 3)] = stats_all[(size_t)(rank + 1) * stride + 3 * (size_t)L + 2 + (i - 3)];
    } else if (i == 6) {
        if (below >= 0 && ghost_rot) {
            double ct, st;
            pose_rot(stats_all[(size_t)(rank - 1) * stride + 3 * (size_t)L + 7], ct, st);
            ghost_rot[0] = ct;
            ghost_rot[1] = st;
        }
    } else if (i == 7) {
        // the (cos, sin)(theta) pairs of the two poses copied in below the shard: the ghost solve reads them (the pose above
        // is only ever a "next" pose, whose heading enters no cos / sin)
        if (below >= 0 && cs) {
            const double* hd = stats_all + (size_t)(rank - 1) * stride + 3 * (size_t)L;
            cs[2 * (size_t)below] = cos(hd[7]);
            cs[2 * (size_t)below + 1] = sin(hd[7]);
            if (below >= 1) {
                cs[2 * (size_t)(below - 1)] = cos(hd[10]);
                cs[2 * (size_t)(below - 1) + 1] = sin(hd[10]);
            }
        }
    }
}

static FiltrarArgs filtrar_args(icm_handle* h) {
    const int L = (int)h->cfg.L;
    FiltrarArgs fa;
    fa.y_raw = h->y_raw.p; fa.cnt_raw = h->cnt_raw.p; fa.stats_all = h->world > 1 ? h->stats_all : nullptr;
    // landmarks created this sweep (single rank): the total of the new-landmark scan, or, in a sweep without the scan
    // kernels, the count k_chunk_l1 added up
    fa.n_new_dev = h->scan_ran ? h->new_rank.p + h->nloc : h->fl + 9;
    fa.sweep_flags = nullptr;
    fa.L = L; fa.lact0 = h->lact0; fa.world = h->world; fa.stride = (int)icm_stats_stride(h);
    fa.cota = h->cfg.cota; fa.thr = h->cfg.dist_thr; fa.max_cells = h->max_cells;
    fa.st = h->fl_state.p; fa.px = h->fl_px.p; fa.py = h->fl_py.p; fa.pc = h->fl_pc.p; fa.nd = h->fl_nd.p;
    fa.cid = h->fl_cid.p; fa.cell_cnt = h->fl_cell_cnt.p; fa.cell_fill = h->fl_cell_fill.p; fa.nn = h->fl_nn.p;
    fa.lab = h->fl_lab.p; fa.comp = h->fl_comp.p; fa.csize = h->fl_csize.p; fa.isl = h->fl_isl.p; fa.rank = h->fl_rank.p;
    fa.mapx = h->mapx.p; fa.mapy = h->mapy.p; fa.counts_new = h->counts_new.p;
    fa.gpar = h->gpar.p; fa.g_cell = h->g_cell.p; fa.g_lm = h->g_lm.p; fa.info = h->fl_info.p;
    fa.info_host = nullptr;
    if (h->x_check) {
        fa.stale = h->x_stale.p;
        fa.stale_epoch = h->x_epoch;
    }
    if (h->map_by_spinner) {
        fa.gave_up = h->l3_done.p + 2;
        fa.gave_up_epoch = h->l3_epoch;
    }
    return fa;
}

// counting-sort grid over the n points (x, y) into the search structures (g_cell, g_lm); the
// grid parameters are in fl_state.gp and the cell counters are zero (k_fl_scatter / k_fl_setup)
static void launch_grid_chain(icm_handle* h, hipStream_t fs, const FiltrarArgs& fa, const double* x, const double* y, const int* n_dev) {
    const int L = (int)h->cfg.L, nall = h->max_cells + 1, ntiles = (nall + kScanTile - 1) / kScanTile;
    const int* ab = &h->fl_state.p->abort;
    TIMED(h, KID_FILTRAR, (k_fl_cell_count<<<nblocks_threads(L), kBlock, 0, fs>>>(fa, x, y, n_dev)));
    // one scan, two copies: cell starts and the fill cursors
    TIMED(h, KID_FILTRAR, (k_scan_tiles<<<ntiles, kBlock, 0, fs>>>(h->fl_cell_cnt.p, h->fl_cell_cnt.p, h->g_cell.p, h->fl_cell_fill.p, h->fl_scan_tot.p, nall, ab)));
    TIMED(h, KID_FILTRAR, (k_scan_fix<<<ntiles, kBlock, 0, fs>>>(h->g_cell.p, h->fl_cell_fill.p, h->fl_scan_tot.p, nall, ntiles, nullptr, nullptr, ab)));
    TIMED(h, KID_FILTRAR, (k_fl_fill<<<nblocks_threads(L), kBlock, 0, fs>>>(fa, x, y, n_dev)));
}

// Mapa.filtrar + the search grid of the refined map, queued on `fs` (no host involvement).
static int launch_filtrar(icm_handle* h, hipStream_t fs, bool guarded = false, int* info_host = nullptr) {
    const int L = (int)h->cfg.L;
    FiltrarArgs fa = filtrar_args(h);
    if (guarded) fa.sweep_flags = h->fl;   // queued without a host look at the sweep's flags: the kernels look themselves
    fa.info_host = info_host;              // (a sweep: the outcome goes straight into the host's mapped block)
    fa.host_map = (info_host && h->host_map_wanted) ? h->pin_map_dev : nullptr;
    const int nb = std::min(kFlMaxBlocks, (L + kFB - 1) / kFB);
    const int chunk = ((L + nb - 1) / nb + kFB - 1) / kFB * kFB;
    TIMED(h, KID_FILTRAR, (k_fl_count<<<nb, kFB, 0, fs>>>(fa, chunk)));
    TIMED(h, KID_FILTRAR, (k_fl_scatter<<<nb, kFB, 0, fs>>>(fa, chunk)));
    launch_grid_chain(h, fs, fa, h->fl_px.p, h->fl_py.p, &h->fl_state.p->n);
    TIMED(h, KID_FILTRAR, (k_fl_pairs<<<nblocks_threads(L), kBlock, 0, fs>>>(fa)));
    TIMED(h, KID_FILTRAR, (k_fl_finalize<<<nb, kFB, 0, fs>>>(fa)));
    // the grid's size is only known on the device: one thread per cell of the capacity
    TIMED(h, KID_NEIGH, (k_neigh_table<<<nblocks_threads(h->max_cells), kBlock, 0, fs>>>(GridView{h->gpar.p, h->g_cell.p, h->g_lm.p, nullptr}, h->g_nb.p, h->max_cells, &h->fl_state.p->abort)));
    HIPCHK(h, hipGetLastError());
    return ICM_OK;
}

// Survivors closer than dist_thr: label propagation, renumbering and count-weighted means on the
// device, then the grid over the refined map.  n = survivors (host copy of info[0]).
static int launch_filtrar_merge(icm_handle* h, hipStream_t fs, int n) {
    const int L = (int)h->cfg.L;
    const FiltrarArgs fa = filtrar_args(h);
    const int nb = std::min(kFlMaxBlocks, (L + kFB - 1) / kFB);
    HIPCHK(h, hipMemsetAsync(h->fl_csize.p, 0, (size_t)L * sizeof(int), fs));   // (members of an oversize component write nothing)
    TIMED(h, KID_FILTRAR, (k_fl_components<<<nblocks_threads(L), kBlock, 0, fs>>>(fa)));
    TIMED(h, KID_FILTRAR, (k_fl_label_flags<<<nblocks_threads(L), kBlock, 0, fs>>>(fa)));
    TIMED(h, KID_FILTRAR, (k_exscan_i32<<<1, 1024, 0, fs>>>(h->fl_isl.p, h->fl_rank.p, n)));
    TIMED(h, KID_FILTRAR, (k_fl_gather<<<nblocks_threads(L), kBlock, 0, fs>>>(fa)));
    const int* n_ref = &h->fl_state.p->n_ref;
    TIMED(h, KID_FILTRAR, (k_fl_extent<<<nb, kFB, 0, fs>>>(fa, h->mapx.p, h->mapy.p, n_ref)));
    TIMED(h, KID_FILTRAR, (k_fl_setup<<<nb, kFB, 0, fs>>>(fa, n_ref)));
    launch_grid_chain(h, fs, fa, h->mapx.p, h->mapy.p, n_ref);
    TIMED(h, KID_FILTRAR, (k_fl_finalize_merged<<<1, 1, 0, fs>>>(fa)));
    TIMED(h, KID_NEIGH, (k_neigh_table<<<nblocks_threads(h->max_cells), kBlock, 0, fs>>>(GridView{h->gpar.p, h->g_cell.p, h->g_lm.p, nullptr}, h->g_nb.p, h->max_cells)));
    HIPCHK(h, hipGetLastError());
    return ICM_OK;
}

// A sweep can be queued whole, without a host look at phase A's outcome in the middle: a red-black sweep through the
// hierarchical pipeline with the moment-form solves and Mapa.filtrar on the device -- the kernels that would replace
// state check the sweep's flags themselves (sharded: every rank's flags travel in the header of its statistics).  Not
// while per-kernel timing serialises the streams.
static bool optimistic_applies(const icm_handle* h) {
    return h->form == 0 && h->entry_path != 0 && h->hier_ok && !h->debug && !h->per_beam && !h->brute && h->gpu_filtrar && !h->timing;
}

// Phase A + local statistics.
int icm_sweep_local(icm_handle* h) {
    if (!h) return ICM_ERR_ARG;
    if (!h->have_state) FAIL(h, ICM_ERR_ARG, "icm_sweep_local: no state (icm_set_state)");
    HIPCHK(h, hipSetDevice(h->device));
    if (h->fault == 1) {
        h->fault = 0;
        FAIL(h, ICM_ERR_HIP, "icm_sweep_local: injected fault (icm_set_fault)");
    }
    h->optimistic = h->opt_req && optimistic_applies(h);
    h->filtrar_deferred = false;
    const int nloc = (int)h->nloc, L = (int)h->cfg.L;
    if (h->phase_timing) (void)hipEventRecord(h->ev_ph[0], h->stream);
    // pose 0 without kept beams: the reference returns its inputs untouched
    // (scripts/ICM_ROS.py:133-135).  Every rank sees the same scan 0 only if it owns it; the
    // host side checks this before sharding.
    h->scan0_empty = (h->t_begin == 0 && h->h_boff[1] == h->h_boff[0]);
    if (h->scan0_empty) return ICM_OK;
    // a no-beam last pose indexes x[:, T] in the reference (IndexError, scripts/ICM_ROS.py:144)
    if (h->t_begin + h->nloc == h->T && h->h_boff[(size_t)nloc] == h->h_boff[(size_t)nloc - 1])
        FAIL(h, ICM_ERR_INDEX, "sweep: the last pose has no kept beams (the reference raises IndexError at scripts/ICM_ROS.py:144)");
    h->lact0 = (int)h->lact;
    const int km = (int)std::min(h->K, h->lact);
    const int nbw = nblocks_waves(nloc);
    h->fl_parity ^= 1;
    h->fl = h->flags.p + 16 * h->fl_parity;
    if (!h->fl_next_clean) HIPCHK(h, hipMemsetAsync(h->fl, 0, 16 * sizeof(int), h->stream));   // (else the last sweep's k_lm_l3 cleared it)
    h->fl_next_clean = false;
    GridView gv{h->gpar.p, h->g_cell.p, h->g_lm.p, h->g_nb.p};
    const bool dbg = h->debug || h->per_beam;
    h->assoc_kept = dbg;
#define ASSOC_ARGS h->x, h->boff.p, h->bxy.p, h->rot.p, h->gpar.p, h->ent_off.p, nloc, (int)h->t_begin, h->x0.p, gv, h->cfg.dist_thr, h->thr2, h->label.p, \
        h->bloc.p, h->st_label.p, h->st_k.p, h->st_sx.p, h->st_sy.p, h->nent.p, h->isnew.p, h->fl, (int)h->nnz, h->st_off.p, 0, (int)h->stl.sparse0
#define ASSOC_GROUP(PRE, DBG, HS) TIMED(h, KID_ASSOC_GROUP, (k_assoc_group<PRE, DBG, HS><<<nblocks_waves(nloc), kBlock, 0, h->stream>>>(ASSOC_ARGS)))
#define ASSOC_GROUP_HS(PRE, DBG) do { if (h->hash_slots == 128) ASSOC_GROUP(PRE, DBG, 128); else ASSOC_GROUP(PRE, DBG, 256); } while (0)
    const float thr_margin = (float)(1e-4 * h->cfg.dist_thr), thr_m = (float)h->cfg.dist_thr - thr_margin;
#define ASSOC_RUNS(DBG, HS) TIMED(h, KID_ASSOC_RUNS, (k_assoc_runs<DBG, HS><<<nblocks_waves(nloc), kBlock, 0, h->stream>>>( \
        h->x, h->roff.p, h->r_s.p, h->rot.p, h->gpar.p, h->ent_off.p, nloc, (int)h->t_begin, h->x0.p, h->r_m.p, h->boff.p, h->bxy.p, gv, \
        h->cfg.dist_thr, h->thr2, thr_m, thr_margin, h->label.p, h->bloc.p, h->st_label.p, h->st_k.p, h->st_sx.p, h->st_sy.p, h->nent.p, h->isnew.p, h->fl, \
        (int)h->nnz, h->st_off.p, 0, (int)h->stl.sparse0, h->run_counts.p)))
#define ASSOC_RUNS_HS(DBG) do { if (h->hash_slots == 128) ASSOC_RUNS(DBG, 128); else ASSOC_RUNS(DBG, 256); } while (0)
    if (!h->rot_valid) {   // (the poses came from the host, a snapshot or a solve form that does not keep the table)
        TIMED(h, KID_POSE_ROT, (k_pose_rot<<<nblocks_threads(nloc), kBlock, 0, h->stream>>>(h->x, h->x0.p, (int)h->t_begin, nloc, h->rot.p, h->pose_cs.p)));
        h->rot_valid = true;
    }
    if (h->brute)
        TIMED(h, KID_ASSOC_BRUTE, (k_associate_brute<<<nbw, kBlock, 0, h->stream>>>(h->x, h->x0.p, (int)h->t_begin, nloc, h->boff.p, h->bx.p, h->by.p, h->mapx.p, h->mapy.p, km, h->cfg.dist_thr, h->label.p)));
    const int ntiles = (nloc + kScanTile - 1) / kScanTile;
    // Entry pipeline: hierarchical running sums (no sort) for the moment-form solves; the
    // sort-based pipeline when the per-beam / per-entry cross-check forms or the association
    // dump need its per-entry arrays, when asked for, or after a table overflow.
    bool hier = h->entry_path != 0 && h->hier_ok && h->form == 0 && (!dbg || h->entry_path == 1);
    // The scan kernels lay next sweep's reservation plan (and rank the new landmarks for the sort-based pipeline); the
    // hierarchical kernels no longer need them: their per-entry arrays live at the staging places, and the few poses that
    // create a landmark get their ranks inside k_chunk_l1.  A sweep queued whole leaves the scan out while nearly every
    // pose fitted its reserved place last time.
    const bool run_scan = !(h->optimistic && hier) || h->scan_wanted;
    h->scan_ran = run_scan;
    const int nrec = h->nchunks * kT1;
    double* const ms = h->ms.p;
    const size_t msn = (size_t)h->nsuper * (size_t)L;
    for (;;) {
        if (h->brute) {
            if (dbg) ASSOC_GROUP_HS(true, true); else ASSOC_GROUP_HS(true, false);
        } else if (h->assoc_form == 1) {   // by runs: the bounding-circle test, beam by beam where it does not settle
            if (dbg) ASSOC_RUNS_HS(true); else ASSOC_RUNS_HS(false);
        } else {
            if (dbg) ASSOC_GROUP_HS(false, true); else ASSOC_GROUP_HS(false, false);
        }
        h->map_ev_in_local = false;
        if (!run_scan && ++h->scan_epoch == 0u) ++h->scan_epoch;   // (tag 0 = never written)
        if (run_scan) {
            TIMED(h, KID_SCAN, (k_scan_tiles<<<ntiles, kBlock, 0, h->stream>>>(h->nent.p, h->isnew.p, h->ent_off.p, h->new_rank.p, h->scan_tot.p, nloc)));
            TIMED(h, KID_SCAN, (k_scan_fix<<<ntiles, kBlock, 0, h->stream>>>(h->ent_off.p, h->new_rank.p, h->scan_tot.p, nloc, ntiles)));
        }
        if (hier) {  // launched before the host looks at the counts: one synchronisation per sweep
#define CHUNK_L1(CH)                                                                                                   \
    TIMED(h, KID_CHUNK_L1, (k_chunk_l1<CH><<<nblocks_waves(h->nchunks), kBlock, 0, h->stream>>>(                          \
        h->x, h->x0.p, (int)h->t_begin, nloc, h->nchunks, h->st_off.p, h->nent.p, h->ent_off.p, h->new_rank.p, h->lact0, \
        h->st_label.p, h->st_k.p, h->st_sx.p, h->st_sy.p, h->pre_x.p, h->pre_y.p, h->pre_n.p,                                   \
        h->rec_label.p, h->rec_s.p, h->rec_s.p + nrec, h->rec_s.p + 2 * (size_t)nrec, h->fl, 0,                           \
        run_scan ? nullptr : h->fl + 8, h->isnew.p, h->chunk_pub.p, h->scan_epoch, 1 << 16)))
            if (h->chunk_poses == 64) CHUNK_L1(64); else if (h->chunk_poses == 32) CHUNK_L1(32); else CHUNK_L1(16);
#undef CHUNK_L1
            if (!h->ms_clean) HIPCHK(h, hipMemsetAsync(ms, 0, 3 * msn * sizeof(double), h->stream));
            h->ms_clean = false;
            TIMED(h, KID_CHUNK_L2, (k_chunk_l2<<<h->nsuper, kT1, 0, h->stream>>>(
                h->nchunks, h->chunk_group, L, h->rec_label.p, h->rec_s.p, h->rec_s.p + nrec, h->rec_s.p + 2 * (size_t)nrec,
                h->rec_off.p, h->rec_off.p + nrec, h->rec_off.p + 2 * (size_t)nrec, ms, ms + msn, ms + 2 * msn, h->fl)));
            double* stats_mine = h->world > 1 ? stats_slot(h) : nullptr;
            h->map_ev_in_local = false;
            if (h->optimistic && !h->timing) {
                // A sweep queued whole.  The four words the host reads at its one wait of the sweep (and, single rank, the
                // sweep's flags) go straight into its mapped block -- no copy launch on any stream -- and with a single rank the
                // raw map is final with this launch: its STOP EVENT (hipExtLaunchKernel: the dispatch packet's own completion
                // signal, no marker packet on the queue) starts the side stream's Mapa.filtrar, a kernel earlier than the end
                // of k_rec_push.
                const bool map_final = h->world == 1;
                const bool spin = map_final && h->gpu_filtrar && h->l3_spin;
                if (spin) {
                    // ... or no event at all (round 4): a one-wave kernel on the side stream, queued NOW, polls a word that
                    // k_lm_l3's last workgroup sets (raw map and flags out write-through); Mapa.filtrar, queued behind it
                    // in icm_sweep_targets, starts a kernel boundary after the raw map is out, and nothing stands on the
                    // main queue between k_lm_l3 and k_rec_push (the stop event cost 5-6 us there).
                    ++h->l3_epoch;
                    k_wait_word<<<1, kWave, 0, h->copy_stream>>>(h->l3_done.p + 1, h->l3_epoch, kWaitPolls, h->l3_done.p + 2);
                }
                hipExtLaunchKernelGGL(k_lm_l3, dim3((L + kWave - 1) / kWave), dim3(kBlock), 0, h->stream, nullptr, (map_final && !spin) ? h->ev_map : nullptr, 0,
                    h->nsuper, L, h->lact0, (const int*)(run_scan ? h->new_rank.p + nloc : h->fl + 9), ms, ms + msn, ms + 2 * msn, stats_mine, h->y_raw.p, h->cnt_raw.p,
                    (const int*)(run_scan ? h->ent_off.p + nloc : h->fl + 8), h->fl, 0, -1, (const double*)nullptr, (double*)nullptr, 1,
                    h->flags.p + 16 * (h->fl_parity ^ 1), h->pin_i_dev, map_final ? 1 : 0, spin ? h->l3_done.p : (int*)nullptr, h->l3_epoch);
                h->map_ev_in_local = map_final;
                h->map_by_spinner = spin;
            } else
            TIMED(h, KID_LM_L3, (k_lm_l3<<<(L + kWave - 1) / kWave, kBlock, 0, h->stream>>>(h->nsuper, L, h->lact0, run_scan ? h->new_rank.p + nloc : h->fl + 9, ms, ms + msn, ms + 2 * msn, stats_mine, h->y_raw.p, h->cnt_raw.p, run_scan ? h->ent_off.p + nloc : h->fl + 8, h->fl, 0, -1, nullptr, nullptr, 1, h->flags.p + 16 * (h->fl_parity ^ 1))));
            h->fl_next_clean = true;
        }
        if (hier && h->optimistic && !h->timing) {
            // (k_lm_l3 wrote the four words into the host's block itself)
        } else if (hier) {  // k_lm_l3 gathered the four words
            HIPCHK(h, hipMemcpyAsync(h->pin_i, h->fl + 4, 4 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        } else {
            HIPCHK(h, hipMemcpyAsync(h->pin_i, h->ent_off.p + nloc, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipMemcpyAsync(h->pin_i + 1, h->new_rank.p + nloc, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipMemcpyAsync(h->pin_i + 2, h->fl, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        }
        if (h->optimistic && hier) break;   // (no host look here: icm_sweep_finish reads the four words)
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->pin_i[2] && h->hash_slots == 128) {  // a scan with > 96 distinct landmarks: use the larger table from now on
            h->hash_slots = 256;
            HIPCHK(h, hipMemsetAsync(h->fl, 0, 16 * sizeof(int), h->stream));
            continue;
        }
        break;
    }
#undef ASSOC_RUNS_HS
#undef ASSOC_RUNS
#undef ASSOC_GROUP_HS
#undef ASSOC_GROUP
#undef ASSOC_ARGS
    if (h->optimistic && hier) {
        h->path_used = 1;
        if (h->world > 1)
            k_set_header<<<1, 1, 0, h->stream>>>(stats_slot(h), L, 0.0, 0.0, h->x, edge_first(h), edge_last(h), edge_last2(h), run_scan ? h->new_rank.p + nloc : h->fl + 9, h->fl);
        HIPCHK(h, hipGetLastError());
        if (h->phase_timing) (void)hipEventRecord(h->ev_ph[1], h->stream);
        return ICM_OK;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->E = h->pin_i[0];
    h->n_new_loc = h->pin_i[1];
    if (h->pin_i[2]) FAIL(h, ICM_ERR_CAPACITY, "sweep: a scan touched more than 192 distinct landmarks (limit of the per-pose table)");
    if ((int64_t)h->lact0 + h->n_new_loc > L)
        FAIL(h, ICM_ERR_INDEX, "sweep: new landmarks exceed the map capacity L (the reference raises IndexError, scripts/ICM_SLAM_tools.py:191)");
    const int nlab = h->lact0 + (int)h->n_new_loc;
    const int E = (int)h->E;
    if (hier && h->pin_i[3]) {  // a chunk or superchunk table overflowed: this map is too dense for the tables
        hier = false;
        h->hier_ok = false;
    }
    h->path_used = hier ? 1 : 0;
    if (hier) {
        if (h->world > 1) k_set_header<<<1, 1, 0, h->stream>>>(stats_slot(h), L, (double)h->n_new_loc, 0.0, h->x, edge_first(h), edge_last(h), edge_last2(h));
        HIPCHK(h, hipGetLastError());
        return ICM_OK;
    }
    TIMED(h, KID_COMPACT, (k_compact<<<nblocks_threads((int64_t)nloc * 16), kBlock, 0, h->stream>>>(h->x, h->x0.p, (int)h->t_begin, nloc, h->st_off.p, h->ent_off.p, h->new_rank.p, h->lact0, h->st_label.p, h->st_k.p, h->st_sx.p, h->st_sy.p, h->pose_s2.p, h->e_key.p, h->e_val.p, h->e_k.p, h->form == 2 ? h->e_b.p : nullptr, h->e_w.p, h->e_wr.p, h->pose_c.p)));
    int bits = 1;
    while ((1ll << bits) < (int64_t)nlab + 1) ++bits;
    size_t tmp_bytes = h->sort_tmp.cap;
    if (E > 0)
        TIMED(h, KID_SORT, HIPCHK(h, rocprim::radix_sort_pairs(h->sort_tmp.p, tmp_bytes, h->e_key.p, h->skey.p, h->e_val.p, h->sval.p, (size_t)E, 0, bits, h->stream)));
    TIMED(h, KID_LM_BOUNDS, (k_lm_bounds<<<nblocks_threads(nlab + 1), kBlock, 0, h->stream>>>(h->skey.p, E, nlab, h->lm_off.p)));
    if (h->world > 1) {
        double* stats_mine = stats_slot(h);
        TIMED(h, KID_LM_TOTALS, (k_lm_scan<true><<<nblocks_waves(L), kBlock, 0, h->stream>>>(nlab, L, h->lm_off.p, h->sval.p, h->e_w.p, nullptr, nullptr, nullptr, nullptr, stats_mine, nullptr, nullptr)));
        k_set_header<<<1, 1, 0, h->stream>>>(stats_mine, L, (double)h->n_new_loc, 0.0, h->x, edge_first(h), edge_last(h), edge_last2(h));
    }
    HIPCHK(h, hipGetLastError());
    return ICM_OK;
}

// The ghost pose of a shard (the last pose of the rank below, solved here too: SolveSeg): its kept beams associated and
// grouped against mapa_viejo with its owner's previous-sweep value, then its moments against the running means through
// that pose = the totals of all lower ranks (k_stats_prefix has just left them in off_*).  Two one-wave launches on the
// solve stream, beside k_rec_push / k_pose_moments_h; the solve launch waits for them (ev_gh1).
static int launch_ghost(icm_handle* h) {
    if (h->ghost_n <= 0) return ICM_OK;   // (a ghost scan without kept beams: the no-beam branch of the solve needs no moments)
    const int L = (int)h->cfg.L;
    hipStream_t gs = h->timing ? h->stream : h->solve_stream;
    if (!h->timing) {
        HIPCHK(h, hipEventRecord(h->ev_gh0, h->stream));
        HIPCHK(h, hipStreamWaitEvent(gs, h->ev_gh0, 0));
    }
    int* gm = h->gh_misc.p;   // [0] nent [1] isnew [2] st_off [3..4] reservation plan (zeros) [8..23] the ghost launch's flags
    HIPCHK(h, hipMemsetAsync(gm + 8, 0, 16 * sizeof(int), gs));
    if (h->assoc_form == 1) {   // (the form its owner's rank uses: the ghost's entries carry the owner's sums bit for bit)
        const float thr_margin = (float)(1e-4 * h->cfg.dist_thr), thr_m = (float)h->cfg.dist_thr - thr_margin;
        k_assoc_runs<false, 256><<<1, kBlock, 0, gs>>>(h->x, h->gh_roff.p, h->gh_rs.p, h->gh_rot.p, h->gpar.p, gm + 3, 1, (int)h->t_begin - 1, h->x0.p,
            h->gh_rm.p, h->gh_boff.p, h->gh_bxy.p, GridView{h->gpar.p, h->g_cell.p, h->g_lm.p, h->g_nb.p}, h->cfg.dist_thr, h->thr2, thr_m, thr_margin,
            h->gh_label.p, h->gh_bloc.p, h->gh_st_label.p, h->gh_st_k.p, h->gh_sx.p, h->gh_sy.p, gm, gm + 1, gm + 8, h->ghost_n, gm + 2, 0, kWave, nullptr);
    } else
    k_assoc_group<false, false, 256><<<1, kBlock, 0, gs>>>(h->x, h->gh_boff.p, h->gh_bxy.p, h->gh_rot.p, h->gpar.p, gm + 3, 1, (int)h->t_begin - 1, h->x0.p,
        GridView{h->gpar.p, h->g_cell.p, h->g_lm.p, h->g_nb.p}, h->cfg.dist_thr, h->thr2, h->gh_label.p, h->gh_bloc.p, h->gh_st_label.p,
        h->gh_st_k.p, h->gh_sx.p, h->gh_sy.p, gm, gm + 1, gm + 8, h->ghost_n, gm + 2, 0, kWave);
    k_ghost_moments<<<1, kWave, 0, gs>>>(h->x, (int)h->t_begin - 1, gm, gm + 2, h->gh_st_label.p, h->gh_st_k.p, h->gh_sx.p, h->gh_sy.p,
        h->gh_s2.p, h->gh_rot.p, h->off_sx.p, h->off_sy.p, h->off_n.p,
        h->stats_all + (size_t)(h->rank - 1) * (size_t)icm_stats_stride(h), L, h->lact0, h->gh_m.p, gm + 8, h->fl);
    HIPCHK(h, hipGetLastError());
    if (!h->timing) {   // (icm_sweep_solve makes the main stream wait for it in front of the solve launch)
        HIPCHK(h, hipEventRecord(h->ev_gh1, gs));
        h->ghost_pending = true;
    }
    return ICM_OK;
}

// Callers that issue the collective themselves (icm_bind_exchange): a rank whose icm_sweep_local failed in the careful
// form must still send its message -- with the error code in the header ([1] = 1 - code >= 2) -- so that the other ranks
// do not wait for it; after the exchange every rank looks at every header (icm_failed_rank) and fails together.
int icm_mark_failed(icm_handle* h, int code) {
    if (!h) return ICM_ERR_ARG;
    if (code > 0) FAIL(h, ICM_ERR_ARG, "icm_mark_failed: an error code is negative (0: a clean header, the closing exchange)");
    if (!h->have_state || (h->world > 1 && !h->stats_all)) FAIL(h, ICM_ERR_ARG, "icm_mark_failed: no state / no exchange buffer");
    HIPCHK(h, hipSetDevice(h->device));
    k_set_header<<<1, 1, 0, h->stream>>>(stats_slot(h), (int)h->cfg.L, 0.0, code ? (double)(1 - code) : 0.0, h->x, edge_first(h), edge_last(h), edge_last2(h));
    HIPCHK(h, hipGetLastError());
    return ICM_OK;
}

// After the exchange: what the ranks' headers say.  *rank_out / *code_out: the first rank whose header carries an error
// code (-1 / 0: none); *retry_out (nullable): some rank's tables overflowed in a sweep it had queued whole ([1] == 1) --
// that rank, and every rank whose flags its message was folded into, will repeat the sweep, so a rank that looks here
// (careful form) must do the same instead of replacing its state.
int icm_exchange_status(icm_handle* h, int* rank_out, int* code_out, int* retry_out) {
    if (!h || !rank_out || !code_out) return ICM_ERR_ARG;
    *rank_out = -1;
    *code_out = 0;
    if (retry_out) *retry_out = 0;
    if (h->world <= 1 || !h->stats_all) return ICM_OK;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t stride = (size_t)icm_stats_stride(h), L = (size_t)h->cfg.L;
    std::vector<double> hd((size_t)h->world);
    for (int r = 0; r < h->world; ++r)
        HIPCHK(h, hipMemcpyAsync(&hd[(size_t)r], h->stats_all + (size_t)r * stride + 3 * L + 1, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int r = 0; r < h->world; ++r) {
        if (hd[(size_t)r] >= 2.0 && *rank_out < 0) {
            *rank_out = r;
            *code_out = 1 - (int)hd[(size_t)r];
        }
        if (hd[(size_t)r] == 1.0 && retry_out) *retry_out = 1;
    }
    return ICM_OK;
}

int icm_failed_rank(icm_handle* h, int* rank_out, int* code_out) { return icm_exchange_status(h, rank_out, code_out, nullptr); }

static int queue_filtrar(icm_handle* h, bool ev_map_recorded);

// After the (optional) all-gather: offsets, raw map, targets.
int icm_sweep_targets(icm_handle* h) {
    if (!h) return ICM_ERR_ARG;
    if (h->fault == 2) {
        h->fault = 0;
        FAIL(h, ICM_ERR_HIP, "icm_sweep_targets: injected fault (icm_set_fault)");
    }
    if (!h->have_state) FAIL(h, ICM_ERR_ARG, "icm_sweep_targets: no state");
    if (h->scan0_empty) return ICM_OK;
    HIPCHK(h, hipSetDevice(h->device));
    if (h->phase_timing && h->optimistic) (void)hipEventRecord(h->ev_ph[2], h->stream);   // (behind the exchange, which is ordered on this stream)
    const int nloc = (int)h->nloc, L = (int)h->cfg.L;
    const int nlab = h->lact0 + (int)h->n_new_loc;
    const bool ghost = h->world > 1 && h->rank > 0;
    bool ev_map_recorded = false;
    if (h->world > 1) {
        // the neighbours' boundary poses (previous-sweep values) came with their statistics
        const int a = (int)h->t_begin, b = (int)(h->t_begin + h->nloc);
        const int below = ghost ? a - 1 : -1;
        const int above = (b < (int)h->T && h->rank + 1 < h->world) ? b : -1;
        if (below >= 0 || above >= 0)
            k_halo_from_headers<<<1, 8, 0, h->stream>>>(h->x, h->stats_all, (int)icm_stats_stride(h), L, h->rank, below, above, ghost ? h->gh_rot.p : nullptr, h->pose_cs.p);
    }
    if (h->path_used == 1) {
        const int nrec = h->nchunks * kT1;
        const size_t msn = (size_t)h->nsuper * (size_t)L;
        double* ro = h->rec_off.p;
        if (h->world > 1)
            TIMED(h, KID_STATS_PREFIX, (k_stats_prefix<<<nblocks_threads(L), kBlock, 0, h->stream>>>(h->stats_all, (int)icm_stats_stride(h), h->rank, h->world, L, h->lact0, h->off_sx.p, h->off_sy.p, h->off_n.p, h->y_raw.p, h->cnt_raw.p, h->optimistic ? h->fl : nullptr)));
        if (ghost) {
            int rcg = launch_ghost(h);
            if (rcg) return rcg;
        }
        // The side stream's work (Mapa.filtrar) needs the raw map and the flags -- not the moments behind.  Its start signal is
        // the STOP EVENT of a launch (hipExtLaunchKernel: the dispatch packet's own completion signal) rather than an event
        // recorded behind the moments: no marker packet on the main queue between the moments and the solves, and the side
        // stream -- hence the host's one wait of the sweep, hence the next sweep's launches -- starts early: with k_lm_l3
        // where the raw map is final there (single rank, icm_sweep_local), else with this launch.  (A stop event costs the
        // main queue ~5 us behind its launch: one per sweep, not two -- the matrix is cleared by the solve launch's waiting
        // even waves, launch_fused_solve, not by a memset behind an event of k_rec_push.)
        if (h->map_ev_in_local) ev_map_recorded = true;
        if (!h->timing && !h->map_ev_in_local) {
            hipExtLaunchKernelGGL(k_rec_push, dim3(nblocks_threads(nrec)), dim3(kBlock), 0, h->stream, nullptr, h->ev_map, 0,
                nrec, h->chunk_group, L, (const int*)h->rec_label.p, (const double*)h->ms.p, (const double*)(h->ms.p + msn), (const double*)(h->ms.p + 2 * msn),
                (const double*)(h->world > 1 ? h->off_sx.p : nullptr), (const double*)(h->world > 1 ? h->off_sy.p : nullptr), (const double*)(h->world > 1 ? h->off_n.p : nullptr),
                ro, ro + nrec, ro + 2 * (size_t)nrec, 0);
            ev_map_recorded = true;
        } else
        TIMED(h, KID_REC_PUSH, (k_rec_push<<<nblocks_threads(nrec), kBlock, 0, h->stream>>>(
            nrec, h->chunk_group, L, h->rec_label.p, h->ms.p, h->ms.p + msn, h->ms.p + 2 * msn,
            h->world > 1 ? h->off_sx.p : nullptr, h->world > 1 ? h->off_sy.p : nullptr, h->world > 1 ? h->off_n.p : nullptr,
            ro, ro + nrec, ro + 2 * (size_t)nrec)));
        TIMED(h, KID_POSE_MOMENTS, (k_pose_moments_h<<<nblocks_threads((int64_t)nloc * 16), kBlock, 0, h->stream>>>(
            h->x, h->x0.p, (int)h->t_begin, nloc, h->st_off.p, h->nent.p, h->ent_off.p, h->st_k.p, h->st_sx.p, h->st_sy.p, h->pose_s2.p,
            h->pre_x.p, h->pre_y.p, h->pre_n.p, h->chunk_poses,
            ro, ro + nrec, ro + 2 * (size_t)nrec, h->pose_m.p, h->assoc_kept ? h->tgt.p : nullptr, 0, -1, h->rot.p)));
    } else if (h->world > 1) {
        TIMED(h, KID_STATS_PREFIX, (k_stats_prefix<<<nblocks_threads(L), kBlock, 0, h->stream>>>(h->stats_all, (int)icm_stats_stride(h), h->rank, h->world, L, h->lact0, h->off_sx.p, h->off_sy.p, h->off_n.p, h->y_raw.p, h->cnt_raw.p)));
        if (ghost) {
            int rcg = launch_ghost(h);
            if (rcg) return rcg;
        }
        TIMED(h, KID_LM_SCAN, (k_lm_scan<false><<<nblocks_waves(L), kBlock, 0, h->stream>>>(nlab, L, h->lm_off.p, h->sval.p, h->e_w.p, h->off_sx.p, h->off_sy.p, h->off_n.p, h->tgt.p, nullptr, nullptr, nullptr)));
    } else {
        TIMED(h, KID_LM_SCAN, (k_lm_scan<false><<<nblocks_waves(L), kBlock, 0, h->stream>>>(nlab, L, h->lm_off.p, h->sval.p, h->e_w.p, nullptr, nullptr, nullptr, h->tgt.p, nullptr, h->y_raw.p, h->cnt_raw.p)));
    }
    if (h->form == 0 && h->path_used != 1)
        TIMED(h, KID_POSE_MOMENTS, (k_pose_moments<<<nblocks_threads((int64_t)nloc * 16), kBlock, 0, h->stream>>>(h->x, h->x0.p, (int)h->t_begin, nloc, h->ent_off.p, h->e_k.p, h->e_wr.p, h->tgt.p, h->pose_c.p, h->pose_m.p)));
    if (h->assoc_kept)
        TIMED(h, KID_BEAM_TARGETS, (k_beam_targets<<<nblocks_waves(nloc), kBlock, 0, h->stream>>>(nloc, h->boff.p, h->ent_off.p, h->bloc.p, h->tgt.p, h->btx.p, h->bty.p)));
    HIPCHK(h, hipGetLastError());
    // Mapa.filtrar on the side stream.  When that stream waits for the raw map by itself (k_wait_word) and the caller is
    // this library's own sweep driver, its nine launches are queued BEHIND the solve launch (icm_sweep_solve): on short
    // sequences the host is what the main queue waits for, and the solve launch must not stand behind them in the
    // host's launch order.
    if (h->defer_filtrar && h->map_ev_in_local && h->map_by_spinner && !h->timing) {
        h->filtrar_deferred = true;
        return ICM_OK;
    }
    return queue_filtrar(h, ev_map_recorded);
}

// The raw map (and, sharded, the ranks' new-landmark counts) is final: Mapa.filtrar on the side stream, overlapping the
// pose solves; the host's one wait of the sweep is for ev_copied.
static int queue_filtrar(icm_handle* h, bool ev_map_recorded) {
    const int L = (int)h->cfg.L;
    if (!(h->map_ev_in_local && h->map_by_spinner)) {   // (else the side stream already waits for the raw map: k_wait_word)
        if (!ev_map_recorded) HIPCHK(h, hipEventRecord(h->ev_map, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->copy_stream, h->ev_map, 0));
    }
    const size_t Ls = (size_t)L;
    if (h->world > 1)
        for (int r = 0; r < h->world && r < 64; ++r)
            HIPCHK(h, hipMemcpyAsync(h->pin_d + 3 * Ls + 16 + r, h->stats_all + (size_t)r * (size_t)icm_stats_stride(h) + 3 * Ls, sizeof(double), hipMemcpyDeviceToHost, h->copy_stream));
    if (h->gpu_filtrar) {
        hipStream_t fs = h->timing ? h->stream : h->copy_stream;   // (timing: serialised on the main stream so that the events bracket it)
        int rc = launch_filtrar(h, fs, h->optimistic, h->pin_i_dev + 8);
        if (rc) return rc;
        if (h->timing) {
            HIPCHK(h, hipEventRecord(h->ev_map, h->stream));
            HIPCHK(h, hipStreamWaitEvent(h->copy_stream, h->ev_map, 0));
        }
        // (k_fl_finalize wrote its three words into the host's block itself: no copy launch behind the chain)
    }
    if (h->optimistic && !h->map_ev_in_local)   // the sweep's flags as every rank sees them (k_stats_prefix folded the other ranks' in;
        HIPCHK(h, hipMemcpyAsync(h->pin_i + 12, h->fl, 4 * sizeof(int), hipMemcpyDeviceToHost, h->copy_stream));   // single rank: k_lm_l3 wrote them)  ([3]: poses outside their reserved staging place)
    HIPCHK(h, hipEventRecord(h->ev_copied, h->copy_stream));
    h->map_copy_pending = true;
    h->filtrar_deferred = false;
    return ICM_OK;
}

static SolveArgs solve_args(icm_handle* h) {
    SolveArgs a;
    a.x = h->x; a.x0 = h->x0.p; a.odo = h->odo.p; a.u = h->u.p;
    a.T = (int)h->T; a.t_begin = (int)h->t_begin; a.nloc = (int)h->nloc;
    a.boff = h->boff.p; a.bx = h->bx.p; a.by = h->by.p; a.btx = h->btx.p; a.bty = h->bty.p;
    a.per_beam = h->per_beam ? 1 : 0;
    a.ent_off = h->ent_off.p; a.e_k = h->e_k.p; a.e_b = h->e_b.p;
    a.tgt = h->tgt.p; a.pose_c = h->pose_c.p; a.pose_m = h->pose_m.p;
    a.dt = h->cfg.deltat; a.R0 = h->cfg.R[0]; a.R1 = h->cfg.R[1]; a.R2 = h->cfg.R[2];
    a.Q0 = h->cfg.Q[0]; a.Q1 = h->cfg.Q[1]; a.cte = h->cfg.cte_odom;
    a.diag = nullptr;
    a.rot = h->rot.p;
    a.odo_cs = h->odo_cs.p;
    a.cs = h->pose_cs.p;
    a.epoch = 0;
    a.xh = nullptr;   // (launch_fused_solve: the one launch whose even waves mirror the poses into the caller's array)
    a.ghost_n = h->ghost_n;
    a.ghost_m = h->gh_m.p;
    return a;
}

// The segment of the pose sequence this handle solves: the whole sequence, or -- a shard of a multi-rank job --
// its block plus the ghost pose in front (SolveSeg: t0 = t_begin - 2).
static SolveSeg shard_segment(const icm_handle* h, const int* abort) {
    const bool ghost = h->world > 1 && h->rank > 0;
    SolveSeg g{ghost ? (int)h->t_begin - 2 : (int)h->t_begin, (int)(h->t_begin + h->nloc), 0, abort};
    if (h->x_check) {
        g.stale = h->x_stale.p;
        g.stale_epoch = h->x_epoch;
    }
    return g;
}

// Both colours of the poses [g.t0, g.t1) in ONE launch (k_solve_m_fused; lane form), on stream st.  Nothing is queued
// behind it: poses outside the folded form's range and even waves that deferred are dealt with inside the launch.
static int launch_fused_solve(icm_handle* h, SolveArgs a, SolveSeg g, hipStream_t st) {
    const int64_t npc = (g.t1 - g.t0) / 2 + 1;   // poses per colour (upper bound)
    // poses per wave: 64, or 32 while both colours' half-filled waves still find a SIMD each twice over (S1, the shards of
    // an 8-rank job): a wave lasts as long as its slowest lane (ICM_SOLVE_PPW: A/B runs)
    int ppw = (4 * (int64_t)((npc + 31) / 32) <= 4 * (int64_t)h->cu_count) ? 32 : kWave;
    if (h->solve_ppw == 32 || h->solve_ppw == 64) ppw = h->solve_ppw;
    const int nwv = (int)((npc + ppw - 1) / ppw);
    if (h->solve_flag_waves < nwv) {
        HIPCHK(h, hipStreamSynchronize(h->stream));   // (re-allocation: nothing may still be polling the old flags)
        if (h->solve_stream) HIPCHK(h, hipStreamSynchronize(h->solve_stream));
        HIPCHK(h, h->solve_flags.reserve(2 * (size_t)nwv + 2));
        HIPCHK(h, h->solve_counts.reserve(2));
        HIPCHK(h, hipMemsetAsync(h->solve_flags.p, 0, (2 * (size_t)nwv + 2) * sizeof(int), st));
        HIPCHK(h, hipMemsetAsync(h->solve_counts.p, 0, 2 * sizeof(unsigned long long), st));
        h->solve_flag_waves = nwv;
        h->solve_epoch = 0;
    }
    a.epoch = ++h->solve_epoch;   // (flags and stamps hold the epoch of the launch that set them: no reset between launches)
    a.xh = h->x_mirror;
    if (h->x_mirror) h->x_mirrored = true;
    int* const deferred = h->solve_flags.p + h->solve_flag_waves;
    int* const sync = h->solve_flags.p + 2 * (size_t)h->solve_flag_waves;
    // Fold-only Nelder-Mead loop (thirteen coefficients per pose) whenever the folded form can hold at all, i.e. with
    // isotropic weights; the complete energy in the loop otherwise (every pose would be solved twice).
    const bool iso = h->cfg.Q[0] == h->cfg.Q[1] && h->cfg.R[0] == h->cfg.R[1];
    const bool fold = h->fold_mode < 0 ? iso : h->fold_mode == 1;
    const int nb2 = nblocks_waves(2 * nwv);
    // phase B's matrix, cleared for the next sweep by the launch's waiting even waves (when that is at most a few hundred
    // stores per lane; else, and after any other solve launch, the next sweep clears it itself: ms_clean)
    const size_t zn = h->path_used == 1 && !h->ms_clean ? 3 * (size_t)h->nsuper * (size_t)h->cfg.L : 0;
    const bool zero_here = zn > 0 && zn <= (size_t)nwv * kWave * 256 && zn < (1ull << 32);   // (<= 256 stores per lane, one per poll of a ~40 us wait;
                                                                                             //  64 left S1 and the shards of an 8-rank job a 6 us memset per sweep)
    double* const zo = zero_here ? h->ms.p : nullptr;
    if (fold)
        TIMED(h, KID_SOLVE, (k_solve_m_fused<true><<<nb2, kBlock, 0, st>>>(a, g, nwv, h->solve_flags.p, h->fused_spin_limit, deferred, sync, h->solve_counts.p, zo, (unsigned)(zero_here ? zn : 0), ppw)));
    else
        TIMED(h, KID_SOLVE, (k_solve_m_fused<false><<<nb2, kBlock, 0, st>>>(a, g, nwv, h->solve_flags.p, h->fused_spin_limit, deferred, sync, h->solve_counts.p, zo, (unsigned)(zero_here ? zn : 0), ppw)));
    if (zero_here) h->ms_clean = true;
    HIPCHK(h, hipGetLastError());
    return ICM_OK;
}

int icm_sweep_solve(icm_handle* h, int schedule, int colour) {
    if (!h) return ICM_ERR_ARG;
    if (!h->have_state) FAIL(h, ICM_ERR_ARG, "icm_sweep_solve: no state");
    if (h->scan0_empty) return ICM_OK;
    HIPCHK(h, hipSetDevice(h->device));
    if (h->x_check_wait) {    // the check of the caller's poses against the device's (k_x_compare, side stream): done long ago
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_cmp, 0));
        h->x_check_wait = false;
    }
    if (h->ghost_pending) {   // the ghost pose's moments (launch_ghost, solve stream)
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_gh1, 0));
        h->ghost_pending = false;
    }
    h->mirror_host = nullptr;   // (poses about to change on the device; icm_sweep says when the caller's array has them too)
    h->x_mirrored = false;
    const bool ph = h->phase_timing && h->optimistic;
    if (ph) (void)hipEventRecord(h->ev_ph[3], h->stream);
    SolveArgs a = solve_args(h);
    if (h->debug) {
        const bool fresh = h->diag.cap < 3 * (size_t)h->T;
        HIPCHK(h, h->diag.reserve(3 * (size_t)h->T));
        if (fresh) HIPCHK(h, hipMemsetAsync(h->diag.p, 0, 3 * (size_t)h->T * sizeof(double), h->stream));
    }
    if (h->optimistic && schedule != ICM_SCHEDULE_REDBLACK)
        FAIL(h, ICM_ERR_ARG, "icm_sweep_solve: a sweep queued without a host look (icm_set_optimistic) is a red-black sweep");
    a.diag = h->debug ? h->diag.p : nullptr;
    // the moment-form solves write the rotation table entry of every pose they write (store_pose): after both colours
    // the next sweep needs no k_pose_rot launch
    a.rot = h->form == 0 ? h->rot.p : nullptr;
    if (h->form != 0) h->rot_valid = false;   // (the cross-check forms do not keep it)
    const int* abort = h->optimistic ? h->fl : nullptr;
    if (h->world > 1 && h->form != 0)
        FAIL(h, ICM_ERR_UNSUPPORTED, "the per-beam / per-entry cross-check forms of the energy are single-rank only (a shard's ghost pose is solved in moment form)");
    if (h->world > 1 && h->rank > 0 && ((h->t_begin & 1) || h->t_begin < 2 || !h->ghost_uploaded))
        FAIL(h, ICM_ERR_ARG, "icm_sweep_solve: a shard of a multi-rank job starts at an even pose and needs the scan of the pose in front of it (icm_upload_ghost_scan)");
    if (schedule == ICM_SCHEDULE_SEQUENTIAL) {
        if (h->world != 1) FAIL(h, ICM_ERR_UNSUPPORTED, "the sequential (reference-order) schedule is one dependent chain and cannot be sharded");
        if (h->form == 1) TIMED(h, KID_SOLVE, (k_solve_sequential<true><<<1, kWave, 0, h->stream>>>(a)));
        else if (h->form == 2) TIMED(h, KID_SOLVE, (k_solve_sequential<false><<<1, kWave, 0, h->stream>>>(a)));
        else {
            const bool iso = h->cfg.Q[0] == h->cfg.Q[1] && h->cfg.R[0] == h->cfg.R[1];
            const bool fold = h->fold_mode < 0 ? iso : h->fold_mode == 1;
            if (h->solve_quad == 1) {   // (icm_set_solve_lanes(1): the quad form of the chain, a cross-check)
                if (fold) TIMED(h, KID_SOLVE, (k_solve_m_sequential<true, true><<<1, kWave, 0, h->stream>>>(a)));
                else TIMED(h, KID_SOLVE, (k_solve_m_sequential<false, true><<<1, kWave, 0, h->stream>>>(a)));
            } else if (fold) TIMED(h, KID_SOLVE, (k_solve_m_sequential<true><<<1, kWave, 0, h->stream>>>(a)));
            else TIMED(h, KID_SOLVE, (k_solve_m_sequential<false><<<1, kWave, 0, h->stream>>>(a)));
            h->rot_valid = false;   // (the chain does not keep the rotation pairs: k_pose_rot at the head of the next sweep)
        }
    } else if (schedule == ICM_SCHEDULE_REDBLACK && colour < 0 && h->form == 0 && h->fuse_colours && h->solve_quad != 1) {
        // both colours in one launch, even waves chase the odd ones (a shard: its ghost pose is the first odd pose)
        int rc = launch_fused_solve(h, a, shard_segment(h, abort), h->stream);
        if (rc) return rc;
    } else if (schedule == ICM_SCHEDULE_REDBLACK) {
        const SolveSeg g = shard_segment(h, abort);
        const int nw = (g.t1 - g.t0) / 2 + 1;
        for (int col = 1; col >= 0; --col) {
            if (!(colour == col || colour < 0)) continue;
            if (h->form == 1) TIMED(h, KID_SOLVE, (k_solve_colour<true><<<nblocks_waves(nw), kBlock, 0, h->stream>>>(a, col)));
            else if (h->form == 2) TIMED(h, KID_SOLVE, (k_solve_colour<false><<<nblocks_waves(nw), kBlock, 0, h->stream>>>(a, col)));
            else {
                // Throughput form: one lane per pose (64 poses per wave).  Latency form: one DPP quad
                // per pose evaluating the four candidate points of an iteration at once.
                // With the folded energy an evaluation is a sixth of an iteration's instructions, so the quad's
                // one-evaluation iteration no longer pays for its broadcasts: the lane form is faster at every
                // size measured (S1: 0.122 against 0.157 ms, 600 poses: 0.099 against 0.115) and is the
                // automatic choice; the quad form stays selectable (icm_set_solve_lanes) and bit-identical.
                const bool quad = h->solve_quad == 1;
                if (quad) TIMED(h, KID_SOLVE, (k_solve_mq_colour<<<nblocks_threads((int64_t)nw * 4), kBlock, 0, h->stream>>>(a, g, col)));
                else TIMED(h, KID_SOLVE, (k_solve_m_colour<<<nblocks_waves((nw + kWave - 1) / kWave), kBlock, 0, h->stream>>>(a, g, col)));
            }
        }
    } else {
        FAIL(h, ICM_ERR_ARG, "icm_sweep_solve: unknown schedule");
    }
    HIPCHK(h, hipGetLastError());
    if (ph) (void)hipEventRecord(h->ev_ph[4], h->stream);
    if (h->filtrar_deferred) return queue_filtrar(h, true);
    return ICM_OK;
}

// Raw running map and counters of the last sweep (before Mapa.filtrar) -> host, on demand.
static int fetch_raw_map(icm_handle* h) {
    if (!h->raw_on_device) return ICM_OK;
    const size_t L = (size_t)h->cfg.L;
    h->h_yraw.resize(2 * L);
    h->h_cntraw.resize(L);
    HIPCHK(h, hipMemcpy(h->h_yraw.data(), h->y_raw.p, 2 * L * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(h->h_cntraw.data(), h->cnt_raw.p, L * sizeof(double), hipMemcpyDeviceToHost));
    h->raw_on_device = false;
    return ICM_OK;
}

int icm_sweep_finish(icm_handle* h) {
    if (!h) return ICM_ERR_ARG;
    if (!h->have_state) FAIL(h, ICM_ERR_ARG, "icm_sweep_finish: no state");
    if (h->scan0_empty) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return ICM_OK;
    }
    HIPCHK(h, hipSetDevice(h->device));
    const size_t L = (size_t)h->cfg.L;
    if (!h->map_copy_pending) FAIL(h, ICM_ERR_ARG, "icm_sweep_finish: call icm_sweep_targets first");
    const double t_fin0 = h->phase_timing ? host_now_ms() : 0.0;
    HIPCHK(h, hipEventSynchronize(h->ev_copied));  // the solves may still be running
    h->map_copy_pending = false;
    if (h->phase_timing && h->optimistic) {   // (diagnostic sweeps only: this wait for the solves is not part of a normal sweep)
        h->ph_ms[4] += host_now_ms() - t_fin0;
        if (hipEventSynchronize(h->ev_ph[4]) == hipSuccess) {
            for (int i = 0; i < 4; ++i) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, h->ev_ph[i], h->ev_ph[i + 1]) == hipSuccess) h->ph_ms[i] += ms;
            }
            ++h->ph_n;
        }
    }
    if (h->x_check && h->pin_i[kStaleWord] == h->x_epoch) {   // (k_x_compare ran on the side stream in front of everything waited for above)
        h->scan_wanted = true;
        return kStalePoses;
    }
    if (h->optimistic) {   // phase A's counts and flags, read only now (the copy was queued behind k_lm_l3)
        if (h->scan_ran) h->E = h->pin_i[0];   // (a sweep without the scan kernels does not count its entries)
        h->n_new_loc = h->pin_i[1];
        // a table overflowed (here or, sharded, on any rank: the flags travelled with the statistics): poses and map
        // were left alone everywhere
        if (h->pin_i[2] || (h->pin_i[3] & 1) || h->pin_i[12] || h->pin_i[13]) {
            h->scan_wanted = true;
            return ICM_RETRY_CAREFUL;
        }
        if (h->world == 1 && (int64_t)h->lact0 + h->n_new_loc > (int64_t)L)
            FAIL(h, ICM_ERR_INDEX, "sweep: new landmarks exceed the map capacity L (the reference raises IndexError, scripts/ICM_SLAM_tools.py:191)");
        // scan kernels in the next sweep?  (for a fresh reservation plan, while many poses do not fit the old one; the
        // new landmarks' ranks k_chunk_l1 finds itself)
        h->scan_wanted = (int64_t)h->pin_i[15] * 32 > h->nloc;
    } else {
        h->scan_wanted = true;
    }
    // total number of landmarks created this sweep, over all ranks
    int64_t n_new = h->n_new_loc;
    if (h->world > 1) {
        if (h->world > 64) FAIL(h, ICM_ERR_UNSUPPORTED, "more than 64 ranks");
        n_new = 0;
        for (int r = 0; r < h->world; ++r) n_new += (int64_t)h->pin_d[3 * L + 16 + (size_t)r];
    }
    h->lact_raw = h->lact0 + n_new;
    if (h->lact_raw > (int64_t)L) FAIL(h, ICM_ERR_INDEX, "sweep: new landmarks exceed the map capacity L");
    h->raw_on_device = true;   // (raw map and counters stay in y_raw / cnt_raw until somebody asks: fetch_raw_map)
    h->filtrar_path = 2;
    if (h->map_by_spinner && h->gpu_filtrar && h->pin_i[9] == 3) {
        // the side stream's wait for the raw map gave up (k_wait_word): the streams are serialised somewhere.  This sweep's
        // Mapa.filtrar runs on the host below; later sweeps start the side stream by k_lm_l3's stop event, which cannot starve.
        ++h->wait_giveups;
        h->l3_spin = false;
    }
    if (h->gpu_filtrar && h->pin_i[9] == 1) {
        // survivors closer than dist_thr: merged on the device (rare; one extra round trip)
        int rc = launch_filtrar_merge(h, h->copy_stream, h->pin_i[8]);
        if (rc) return rc;
        HIPCHK(h, hipMemcpyAsync(h->pin_i + 8, h->fl_info.p, 4 * sizeof(int), hipMemcpyDeviceToHost, h->copy_stream));
        HIPCHK(h, hipStreamSynchronize(h->copy_stream));
        if (h->pin_i[9] == 0) h->filtrar_path = 1;
    } else if (h->gpu_filtrar && h->pin_i[9] == 0) {
        h->filtrar_path = 0;
    }
    if (h->gpu_filtrar && h->pin_i[9] == 0) {
        // Mapa.filtrar and the search grid of the refined map were produced on the GPU
        // (k_fl_* chain); the refined map becomes the next mapa_viejo (scripts/ICM_ROS.py:311)
        h->K = h->lact = h->pin_i[8];
        h->dev_map_current = true;
        h->h_map_valid = false;   // fetched when asked for (sync_host_map)
        if (h->host_map_wanted && h->filtrar_path == 0) {   // ... or already here: k_fl_finalize wrote it into the host's block
            const size_t K = (size_t)h->K;
            h->h_map.resize(2 * K);
            std::copy(h->pin_map, h->pin_map + K, h->h_map.begin());
            std::copy(h->pin_map + L, h->pin_map + L + K, h->h_map.begin() + K);
            h->h_counts.assign(h->pin_map + 2 * L, h->pin_map + 3 * L);
            h->h_map_valid = true;
        }
        // No wait for the solves here: everything the host needs (raw map, filtrar result) came
        // over the side stream, and whatever is queued next on the main stream is ordered behind
        // them -- the next sweep's phase A starts the moment the last solve ends.  (Readers of x
        // synchronise themselves; a fused solve that gave up is reported at the next
        // synchronisation of the main stream.)
        return ICM_OK;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    {
        int rcf = fetch_raw_map(h);
        if (rcf) return rcf;
    }
    std::vector<double> yo(2 * L), co(L);
    int64_t lact_new = 0;
    int rc = filtrar_host(h->cfg, h->h_yraw.data(), h->h_cntraw.data(), h->lact_raw, yo.data(), co.data(), &lact_new, h->err);
    if (rc) return rc;
    h->K = lact_new;
    h->lact = lact_new;
    h->h_map.resize(2 * (size_t)lact_new);
    for (int64_t i = 0; i < lact_new; ++i) {
        h->h_map[(size_t)i] = yo[(size_t)i];
        h->h_map[(size_t)(lact_new + i)] = yo[L + (size_t)i];
    }
    h->h_counts = co;
    return upload_map(h);
}

// Bring the host copy of the refined map / counters up to date (after GPU-side filtrar).
static int sync_host_map(icm_handle* h) {
    if (h->h_map_valid) return ICM_OK;
    const size_t K = (size_t)h->K, L = (size_t)h->cfg.L;
    h->h_map.assign(2 * K, 0.0);
    h->h_counts.assign(L, 0.0);
    // the refined map's K columns and the counters, packed on the device into ONE download
    HIPCHK(h, h->pack.reserve(2 * L + L));
    if (K) {
        HIPCHK(h, hipMemcpyAsync(h->pack.p, h->mapx.p, K * sizeof(double), hipMemcpyDeviceToDevice, h->copy_stream));
        HIPCHK(h, hipMemcpyAsync(h->pack.p + K, h->mapy.p, K * sizeof(double), hipMemcpyDeviceToDevice, h->copy_stream));
    }
    HIPCHK(h, hipMemcpyAsync(h->pack.p + 2 * K, h->counts_new.p, L * sizeof(double), hipMemcpyDeviceToDevice, h->copy_stream));
    std::vector<double>& st = h->h_pack;
    st.resize(2 * K + L);
    HIPCHK(h, hipMemcpyAsync(st.data(), h->pack.p, (2 * K + L) * sizeof(double), hipMemcpyDeviceToHost, h->copy_stream));
    HIPCHK(h, hipStreamSynchronize(h->copy_stream));
    std::copy(st.begin(), st.begin() + 2 * K, h->h_map.begin());
    std::copy(st.begin() + 2 * K, st.end(), h->h_counts.begin());
    h->h_map_valid = true;
    return ICM_OK;
}

// Device-side copy of everything a sweep reads and replaces: poses, mapa_viejo with its counters
// and search structures.  icm_restore_state puts it back with a few device-to-device copies (no
// host round trip), e.g. to run many sweeps from the same start.
int icm_snapshot_state(icm_handle* h) {
    if (!h) return ICM_ERR_ARG;
    if (!h->have_state) FAIL(h, ICM_ERR_ARG, "icm_snapshot_state: no state (icm_set_state)");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t T = (size_t)h->T, L = (size_t)h->cfg.L, nc = (size_t)h->max_cells;
    icm_handle::Snapshot& sn = h->snap;
    HIPCHK(h, sn.x.reserve(3 * T)); HIPCHK(h, sn.mapx.reserve(L)); HIPCHK(h, sn.mapy.reserve(L)); HIPCHK(h, sn.counts_new.reserve(L));
    HIPCHK(h, sn.g_cell.reserve(nc + 2)); HIPCHK(h, sn.g_lm.reserve(L)); HIPCHK(h, sn.g_nb.reserve(nc)); HIPCHK(h, sn.gpar.reserve(1));
    hipStream_t st = h->stream;
    HIPCHK(h, hipMemcpyAsync(sn.x.p, h->x, 3 * T * sizeof(double), hipMemcpyDeviceToDevice, st));
    HIPCHK(h, hipMemcpyAsync(sn.mapx.p, h->mapx.p, L * sizeof(double), hipMemcpyDeviceToDevice, st));
    HIPCHK(h, hipMemcpyAsync(sn.mapy.p, h->mapy.p, L * sizeof(double), hipMemcpyDeviceToDevice, st));
    HIPCHK(h, hipMemcpyAsync(sn.counts_new.p, h->counts_new.p, L * sizeof(double), hipMemcpyDeviceToDevice, st));
    HIPCHK(h, hipMemcpyAsync(sn.g_cell.p, h->g_cell.p, (nc + 2) * sizeof(int), hipMemcpyDeviceToDevice, st));
    HIPCHK(h, hipMemcpyAsync(sn.g_lm.p, h->g_lm.p, L * sizeof(LmRec), hipMemcpyDeviceToDevice, st));
    HIPCHK(h, hipMemcpyAsync(sn.g_nb.p, h->g_nb.p, nc * sizeof(NeighRec), hipMemcpyDeviceToDevice, st));
    HIPCHK(h, hipMemcpyAsync(sn.gpar.p, h->gpar.p, sizeof(GridParams), hipMemcpyDeviceToDevice, st));
    HIPCHK(h, hipStreamSynchronize(st));
    sn.h_map = h->h_map; sn.h_counts = h->h_counts;
    sn.K = h->K; sn.lact = h->lact; sn.h_map_valid = h->h_map_valid; sn.dev_map_current = h->dev_map_current;
    sn.valid = true;
    return ICM_OK;
}

int icm_restore_state(icm_handle* h) {
    if (!h) return ICM_ERR_ARG;
    icm_handle::Snapshot& sn = h->snap;
    const size_t T = (size_t)h->T, L = (size_t)h->cfg.L, nc = (size_t)h->max_cells;
    if (!sn.valid || sn.x.cap < 3 * T || sn.g_nb.cap < nc) FAIL(h, ICM_ERR_ARG, "icm_restore_state: no snapshot of this sequence (icm_snapshot_state)");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = h->stream;   // stream-ordered behind the last sweep: no synchronisation needed
    HIPCHK(h, hipMemcpyAsync(h->x, sn.x.p, 3 * T * sizeof(double), hipMemcpyDeviceToDevice, st));
    h->mirror_host = nullptr;
    h->rot_valid = false;
    h->scan_wanted = true;
    HIPCHK(h, hipMemcpyAsync(h->mapx.p, sn.mapx.p, L * sizeof(double), hipMemcpyDeviceToDevice, st));
    HIPCHK(h, hipMemcpyAsync(h->mapy.p, sn.mapy.p, L * sizeof(double), hipMemcpyDeviceToDevice, st));
    HIPCHK(h, hipMemcpyAsync(h->counts_new.p, sn.counts_new.p, L * sizeof(double), hipMemcpyDeviceToDevice, st));
    HIPCHK(h, hipMemcpyAsync(h->g_cell.p, sn.g_cell.p, (nc + 2) * sizeof(int), hipMemcpyDeviceToDevice, st));
    HIPCHK(h, hipMemcpyAsync(h->g_lm.p, sn.g_lm.p, L * sizeof(LmRec), hipMemcpyDeviceToDevice, st));
    HIPCHK(h, hipMemcpyAsync(h->g_nb.p, sn.g_nb.p, nc * sizeof(NeighRec), hipMemcpyDeviceToDevice, st));
    HIPCHK(h, hipMemcpyAsync(h->gpar.p, sn.gpar.p, sizeof(GridParams), hipMemcpyDeviceToDevice, st));
    h->h_map = sn.h_map; h->h_counts = sn.h_counts;
    h->K = sn.K; h->lact = sn.lact; h->h_map_valid = sn.h_map_valid; h->dev_map_current = sn.dev_map_current;
    return ICM_OK;
}

static int icm_sweep_classic(icm_handle* h, int schedule) {
    const bool req = h->opt_req;
    int rc = ICM_OK;
    for (int attempt = 0; attempt < 2; ++attempt) {
        h->opt_req = attempt == 0 && schedule == ICM_SCHEDULE_REDBLACK;
        rc = icm_sweep_local(h);
        h->defer_filtrar = true;    // (targets and solve are called back to back here)
        if (!rc) rc = icm_sweep_targets(h);
        if (!rc) rc = icm_sweep_solve(h, schedule, -1);
        h->defer_filtrar = false;
        if (!rc && h->filtrar_deferred) rc = queue_filtrar(h, true);   // (the solve call returned early: scan 0 without beams)
        if (!rc) rc = icm_sweep_finish(h);
        if (rc != ICM_RETRY_CAREFUL) break;
        if (h->x_check && h->pin_i[kStaleWord] == h->x_epoch) { rc = kStalePoses; break; }   // (the flags mean nothing then)
        // a per-pose or per-chunk table overflowed: solves and Mapa.filtrar saw the flags and changed nothing;
        // once more with the host looking in the middle (it sizes the tables / takes the sort-based pipeline)
    }
    h->opt_req = req;
    return rc;
}

int icm_sweep_device(icm_handle* h, int schedule) {
    if (!h) return ICM_ERR_ARG;
    if (!h->have_state) FAIL(h, ICM_ERR_ARG, "icm_sweep_device: no state (icm_set_state)");
    return icm_sweep_classic(h, schedule);
}

int icm_set_optimistic(icm_handle* h, int on) {
    if (!h) return ICM_ERR_ARG;
    h->opt_req = on != 0;
    return ICM_OK;
}

int icm_get_optimistic(const icm_handle* h) { return h ? (h->optimistic ? 1 : 0) : ICM_ERR_ARG; }

// ---- RCCL from inside the library: collectives (the loader is above icm_destroy) ----------------
#define RCCLCHK(h, call)                                                                                     \
    do {                                                                                                     \
        ncclResult_t r__ = (call);                                                                           \
        if (r__ != ncclSuccess) {                                                                            \
            (h)->err = std::string(#call) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r__) : "RCCL error"); \
            return ICM_ERR_HIP;                                                                              \
        }                                                                                                    \
    } while (0)

int icm_comm_set_library(const char* path) {
    g_rccl_path = path ? path : "";
    return ICM_OK;
}

int icm_comm_available(void) {
    std::string err;
    return rccl_load(err) ? 1 : 0;
}

int icm_comm_unique_id(void* id128) {
    if (!id128) return ICM_ERR_ARG;
    if (!rccl_load(g_create_err)) return ICM_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) {
        g_create_err = "ncclGetUniqueId failed";
        return ICM_ERR_HIP;
    }
    std::memcpy(id128, &id, sizeof(id));
    return ICM_OK;
}

// Pose blocks of a sharded job: ceil(T / world) rounded up to an EVEN number of poses, so that every shard starts at an
// even pose (SolveSeg: a shard's ghost pose is odd); rank r owns [r blk, min((r + 1) blk, T)).
int64_t icm_shard_block(int64_t T, int world) {
    if (T <= 0 || world <= 0) return 0;
    const int64_t blk = (T + world - 1) / world;
    return blk + (blk & 1);
}

// exchange buffers owned by the library, bound like a caller's would be
static int comm_setup(icm_handle* h, int rank, int world) {
    HIPCHK(h, hipSetDevice(h->device));
    const int64_t blk = icm_shard_block(h->T, world);
    if (world > 1 && (int64_t)(world - 1) * blk >= h->T)
        FAIL(h, ICM_ERR_ARG, "icm_comm_init: " + std::to_string(h->T) + " poses cannot be split into " + std::to_string(world) + " blocks of " + std::to_string(blk) +
                             " (every shard starts at an even pose and holds at least one pose): use fewer ranks");
    if (h->t_begin != std::min<int64_t>((int64_t)rank * blk, h->T) || h->t_begin + h->nloc != std::min<int64_t>((int64_t)(rank + 1) * blk, h->T))
        FAIL(h, ICM_ERR_ARG, "icm_comm_init: the uploaded shard is not block `rank` of icm_shard_block(T, world)-pose blocks");
    const size_t stride = (size_t)icm_stats_stride(h);
    HIPCHK(h, h->own_stats_all.reserve((size_t)world * stride));
    HIPCHK(h, h->own_stats_send.reserve(stride));
    HIPCHK(h, h->own_poses.reserve((size_t)world * (size_t)blk * 3));
    HIPCHK(h, hipMemsetAsync(h->own_stats_all.p, 0, (size_t)world * stride * sizeof(double), h->stream));
    HIPCHK(h, hipMemsetAsync(h->own_stats_send.p, 0, stride * sizeof(double), h->stream));
    HIPCHK(h, hipMemsetAsync(h->own_poses.p, 0, (size_t)world * (size_t)blk * 3 * sizeof(double), h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->comm_blk = blk;
    int rc;
    if ((rc = icm_bind_exchange(h, h->own_stats_all.p, rank, world))) return rc;
    if ((rc = icm_bind_pose_buffer(h, h->own_poses.p))) return rc;
    if ((rc = icm_bind_exchange_send(h, h->own_stats_send.p))) return rc;
    h->comm_ready = true;
    return ICM_OK;
}

int icm_comm_init(icm_handle* h, const void* id128, int rank, int world) {
    if (!h) return ICM_ERR_ARG;
    if (!id128 || world < 1 || rank < 0 || rank >= world) FAIL(h, ICM_ERR_ARG, "icm_comm_init: bad arguments");
    if (!h->uploaded) FAIL(h, ICM_ERR_ARG, "icm_comm_init: upload this rank's shard first (icm_upload)");
    if (h->comm_ready) FAIL(h, ICM_ERR_ARG, "icm_comm_init: communicator already initialised");
    if (!rccl_load(h->err)) return ICM_ERR_UNSUPPORTED;
    HIPCHK(h, hipSetDevice(h->device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    RCCLCHK(h, g_rccl.CommInitRank(&h->comm, world, id, rank));
    return comm_setup(h, rank, world);
}

// The same sharded driver over a caller-supplied all-gather instead of RCCL (MPI, a test harness carrying the messages
// through host memory, ...): `fn` must gather `count` doubles from every rank's `send` into `recv` (rank-major), ordered
// after the work already queued on `hip_stream` and complete -- or stream-ordered -- when it returns; non-zero = failure.
int icm_comm_init_transport(icm_handle* h, int rank, int world, icm_allgather_fn fn, void* user) {
    if (!h) return ICM_ERR_ARG;
    if (!fn || world < 1 || rank < 0 || rank >= world) FAIL(h, ICM_ERR_ARG, "icm_comm_init_transport: bad arguments");
    if (!h->uploaded) FAIL(h, ICM_ERR_ARG, "icm_comm_init_transport: upload this rank's shard first (icm_upload)");
    if (h->comm_ready) FAIL(h, ICM_ERR_ARG, "icm_comm_init_transport: communicator already initialised");
    h->transport = fn;
    h->transport_user = user;
    return comm_setup(h, rank, world);
}

int icm_comm_destroy(icm_handle* h) {
    if (!h) return ICM_ERR_ARG;
    if (h->comm) {
        (void)hipSetDevice(h->device);
        (void)hipStreamSynchronize(h->stream);
        (void)g_rccl.CommDestroy(h->comm);
        h->comm = nullptr;
    }
    h->transport = nullptr;
    h->transport_user = nullptr;
    h->comm_ready = false;
    return ICM_OK;
}

static int all_gather(icm_handle* h, const double* send, double* recv, size_t count) {
    if (h->transport) {
        if (h->transport(send, recv, count, reinterpret_cast<void*>(h->stream), h->transport_user))
            FAIL(h, ICM_ERR_HIP, "the caller's all-gather (icm_comm_init_transport) failed");
        return ICM_OK;
    }
    RCCLCHK(h, g_rccl.AllGather(send, recv, count, ncclDouble, h->comm, h->stream));
    return ICM_OK;
}

// One red-black sweep of a sharded sequence with its ONE collective issued here, on the handle's stream (SURVEY 8e,
// scripts/ICM_ROS.py:141-158 sharded over poses): local phase A + statistics -> all-gather of the [3L + 16]
// statistics (landmark sums, new-landmark count, flags, the shard's boundary poses of the previous sweep) -> targets
// + the ghost pose's moments -> both colours of the shard in one launch -> Mapa.filtrar (replicated).
static int sweep_sharded_once(icm_handle* h);

static const char* phase_a_error_text(int code) {
    return code == ICM_ERR_INDEX ? "IndexError: labels beyond L or a no-beam last pose"
         : code == ICM_ERR_CAPACITY ? "a scan touched more distinct landmarks than supported"
         : code == ICM_ERR_HIP ? "a HIP runtime error" : code == ICM_ERR_ARG ? "a bad argument / call order" : "error";
}

int icm_sweep_sharded(icm_handle* h) {
    if (!h) return ICM_ERR_ARG;
    if (!h->comm_ready) FAIL(h, ICM_ERR_ARG, "icm_sweep_sharded: no communicator (icm_comm_init)");
    // queued whole first; if ANY rank's tables overflowed every rank sees it (the flags travel with the statistics),
    // nothing was replaced anywhere, and every rank repeats the sweep with its host looking in the middle
    const bool req = h->opt_req;
    int rc = ICM_OK;
    for (int attempt = 0; attempt < 2; ++attempt) {
        h->opt_req = attempt == 0;
        rc = sweep_sharded_once(h);
        if (rc != ICM_RETRY_CAREFUL) break;
        if (attempt == 0) {
            // Before the second collective: a rank that FAILED in the first one has left with its error and will not
            // join a second (its code, >= 2 in header [1], reads as "flags set" to the ranks that queued the sweep whole):
            // every rank looks at the headers still in its exchange buffer and leaves with that error too.
            int fr = -1, code = 0, rc2;
            if ((rc2 = icm_exchange_status(h, &fr, &code, nullptr))) { rc = rc2; break; }
            if (fr >= 0) {
                h->opt_req = req;
                FAIL(h, code, "sharded sweep: rank " + std::to_string(fr) + " failed (" + phase_a_error_text(code) + ")");
            }
        }
    }
    h->opt_req = req;
    return rc;
}

// A rank that cannot take part in the exchange at all (its device or stream is gone): tear the communicator down, so that
// the peers' collective returns with an error instead of waiting for this rank for ever.  (A caller-supplied transport,
// icm_comm_init_transport, is the caller's to tear down: the error return is all this library can do there.)
static void comm_abort(icm_handle* h) {
    if (h->comm && g_rccl.CommAbort) {
        (void)g_rccl.CommAbort(h->comm);
        h->comm = nullptr;
        h->comm_ready = false;
    }
}

static int sweep_sharded_once(icm_handle* h) {
    const size_t stride = (size_t)icm_stats_stride(h);
    // Failing together.  A rank that fails on its own in phase A -- a property of its data (labels beyond L, a no-beam
    // last pose, a table too small at its largest size) or of its device (a HIP error) -- must not leave the others
    // waiting in the collective: it still sends its message, with the error code in the header, and every rank returns
    // that error after the exchange.
    const int rc_local = icm_sweep_local(h);
    int rc;
    if (rc_local) {
        const std::string err_local = h->err;
        rc = icm_mark_failed(h, rc_local);
        if (!rc) rc = all_gather(h, h->own_stats_send.p, h->own_stats_all.p, stride);
        if (rc) comm_abort(h);   // (could not even say so)
        h->err = err_local;
        return rc_local;
    }
    rc = all_gather(h, h->own_stats_send.p, h->own_stats_all.p, stride);
    if (rc) return rc;
    if (!h->optimistic) {
        // careful form: the host looks at every rank's header.  [1] >= 2: that rank failed with code 1 - [1]; [1] == 1: a
        // rank that had queued the sweep whole overflowed -- it and its like will repeat the sweep, and so must this one
        int fr = -1, code = 0, retry = 0;
        if ((rc = icm_exchange_status(h, &fr, &code, &retry))) return rc;
        if (fr >= 0)
            FAIL(h, code, "sharded sweep: rank " + std::to_string(fr) + " failed (" + phase_a_error_text(code) + ")");
        if (retry) return ICM_RETRY_CAREFUL;
    }
    // Behind the exchange a rank can still fail on its own (a device error in the targets, the solve launch or
    // Mapa.filtrar; its peers, whose phases ran, are on their way to the NEXT exchange -- the next sweep's, or the closing
    // one of icm_gather_poses / icm_sharded_end).  It meets them there: one more message with the code in its header and
    // nothing else (farewell), so that every rank returns the error one exchange later instead of waiting for a rank that
    // has left; a rank that cannot even do that aborts the communicator.
    rc = icm_sweep_targets(h);
    if (!rc) rc = icm_sweep_solve(h, ICM_SCHEDULE_REDBLACK, -1);
    if (!rc) rc = icm_sweep_finish(h);
    if (rc < 0) {
        const std::string err_local = h->err;
        int rf = icm_mark_failed(h, rc);
        if (!rf) rf = all_gather(h, h->own_stats_send.p, h->own_stats_all.p, stride);
        if (rf) comm_abort(h);
        h->err = err_local;
    }
    return rc;
}

// The closing exchange of a sharded job: one more all-gather of the statistics message with a clean header, after the
// last sweep -- where a rank that failed BEHIND the last sweep's exchange delivers its error (sweep_sharded_once).
// icm_gather_poses runs it first.
int icm_sharded_end(icm_handle* h) {
    if (!h) return ICM_ERR_ARG;
    if (!h->comm_ready) FAIL(h, ICM_ERR_ARG, "icm_sharded_end: no communicator (icm_comm_init)");
    if (h->world <= 1 || !h->have_state) return ICM_OK;
    int rc = icm_mark_failed(h, 0);
    if (!rc) rc = all_gather(h, h->own_stats_send.p, h->own_stats_all.p, (size_t)icm_stats_stride(h));
    if (rc) return rc;
    int fr = -1, code = 0;
    if ((rc = icm_exchange_status(h, &fr, &code, nullptr))) return rc;
    if (fr >= 0) FAIL(h, code, "sharded job: rank " + std::to_string(fr) + " failed behind the last sweep's exchange (" + phase_a_error_text(code) + ")");
    return ICM_OK;
}

// Every rank's pose block -> every rank (before icm_get_state on a sharded handle).
int icm_gather_poses(icm_handle* h) {
    if (!h) return ICM_ERR_ARG;
    if (!h->comm_ready) FAIL(h, ICM_ERR_ARG, "icm_gather_poses: no communicator (icm_comm_init)");
    HIPCHK(h, hipSetDevice(h->device));
    {
        int rce = icm_sharded_end(h);   // (the closing exchange: a rank that failed behind the last sweep's exchange says so here)
        if (rce) return rce;
    }
    const size_t cnt = (size_t)h->comm_blk * 3;
    return all_gather(h, h->own_poses.p + (size_t)h->rank * cnt, h->own_poses.p, cnt);
}

int icm_get_state(icm_handle* h, double* x, double* map_out, double* counts_out, int64_t* K_out) {
    if (!h) return ICM_ERR_ARG;
    if (!h->have_state) FAIL(h, ICM_ERR_ARG, "icm_get_state: no state");
    HIPCHK(h, hipSetDevice(h->device));
    {
        int rc = sync_host_map(h);
        if (rc) return rc;
    }
    const size_t T = (size_t)h->T, L = (size_t)h->cfg.L;
    if (x) {
        if (h->mirror_host == x && pinned_alias(h, x, 3 * T * sizeof(double))) {
            // the solves of the last sweep wrote every pose into this very array themselves (SolveArgs::xh)
        } else if (double* xa = pinned_alias(h, x, 3 * T * sizeof(double))) {   // a registered array: written in place by the layout kernel
            k_x_poses_to_rows<<<(int)((T + 255) / 256), 256, 0, h->stream>>>(h->x, xa, (int)T);
            HIPCHK(h, hipGetLastError());
        } else {
            HIPCHK(h, h->x_rows.reserve(3 * T));
            k_x_poses_to_rows<<<(int)((T + 255) / 256), 256, 0, h->stream>>>(h->x, h->x_rows.p, (int)T);
            HIPCHK(h, hipGetLastError());
            HIPCHK(h, hipMemcpyAsync(x, h->x_rows.p, 3 * T * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    if (map_out) {
        std::fill(map_out, map_out + 2 * L, 0.0);
        for (int64_t i = 0; i < h->K; ++i) {
            map_out[(size_t)i] = h->h_map[(size_t)i];
            map_out[L + (size_t)i] = h->h_map[(size_t)(h->K + i)];
        }
    }
    if (counts_out) std::copy(h->h_counts.begin(), h->h_counts.end(), counts_out);
    if (K_out) *K_out = h->scan0_empty ? -1 : h->lact;
    return ICM_OK;
}

// One sweep from the device's own poses, which the caller says (and k_x_compare checks, on the side stream under phase
// A) are what its registered array `xa` holds: no upload.  kStalePoses: they were not; nothing was replaced.
static int sweep_without_upload(icm_handle* h, const double* xa, const double* x0, int schedule) {
    const size_t T = (size_t)h->T;
    HIPCHK(h, hipSetDevice(h->device));
    ++h->x_epoch;
    if (std::memcmp(x0, h->h_x0, sizeof(h->h_x0)) != 0) {
        std::memcpy(h->h_x0, x0, sizeof(h->h_x0));
        HIPCHK(h, hipMemcpyAsync(h->x0.p, h->h_x0, sizeof(h->h_x0), hipMemcpyHostToDevice, h->stream));
    }
    k_x_compare<<<(int)((T + 255) / 256), 256, 0, h->copy_stream>>>(xa, h->x, (int)T, h->x_stale.p, h->pin_i_dev + kStaleWord, h->x_epoch);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev_cmp, h->copy_stream));
    h->x_check = h->x_check_wait = true;
    const int rc = icm_sweep_device(h, schedule);
    h->x_check = h->x_check_wait = false;
    return rc;
}

int icm_sweep(icm_handle* h, double* x, const double* x0, const double* map_in, int64_t K, int64_t lact_in,
              int schedule, double* map_out, double* counts_out, int64_t* K_out) {
    if (!h) return ICM_ERR_ARG;
    int rc;
    const size_t T = (size_t)h->T;
    // A registered pose array (icm_pin_host): the solves mirror every pose they write into it (no download), and if it is
    // the array the last call filled and the map is the one that call returned -- the reference's driver loop,
    // scripts/ICM_ROS.py:298-311 -- this call starts from the device's state without an upload.
    double* const xa = (x && h->prefiltered && h->world == 1) ? pinned_alias(h, x, 3 * T * sizeof(double)) : nullptr;
    bool fast = xa && h->have_state && h->mirror_host == x && schedule == ICM_SCHEDULE_REDBLACK && x0 && (K == 0 || map_in) &&
                h->dev_map_current && K == h->K && lact_in == h->lact;
    if (fast) {
        if ((rc = sync_host_map(h))) return rc;   // (already on the host: the last call returned it)
        fast = h->h_map.size() == 2 * (size_t)K && (K == 0 || std::memcmp(map_in, h->h_map.data(), 2 * (size_t)K * sizeof(double)) == 0);
    }
    // (the even poses' lanes write the pairs (t - 1, t): a two-pose sequence has no even pose to solve, so nobody would
    // write pose 1 into the caller's array -- it goes through the ordinary download)
    h->x_mirror = (h->form == 0 && T >= 3) ? xa : nullptr;
    h->x_mirrored = false;
    h->host_map_wanted = true;
    if (fast) ++h->dropin_counts[0];
    rc = fast ? sweep_without_upload(h, xa, x0, schedule) : kStalePoses;
    if (fast && rc == kStalePoses) ++h->dropin_counts[1];
    if (rc == kStalePoses) {   // the usual road: upload, sweep
        rc = set_state_impl(h, x, x0, map_in, K, lact_in, false);
        if (!rc) rc = icm_sweep_device(h, schedule);
    }
    h->x_mirror = nullptr;
    h->host_map_wanted = false;
    if (rc) {
        h->mirror_host = nullptr;
        return rc;
    }
    if (h->scan0_empty) {
        if (K_out) *K_out = -1;
        return ICM_OK;
    }
    if (h->x_mirrored) {   // (the solve launch filled it: icm_get_state below only waits for it)
        h->mirror_host = x;
        ++h->dropin_counts[2];
    }
    rc = icm_get_state(h, x, map_out, counts_out, K_out);
    if (!rc && xa) h->mirror_host = x;   // host array == device poses, however they got there
    return rc;
}

int icm_get_association(icm_handle* h, int32_t* labels, double* target_x, double* target_y) {
    if (!h) return ICM_ERR_ARG;
    if (!h->have_state) FAIL(h, ICM_ERR_ARG, "icm_get_association: no sweep has run");
    if (!h->assoc_kept) FAIL(h, ICM_ERR_ARG, "icm_get_association: enable icm_set_debug before the sweep");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const size_t nz = (size_t)h->nnz;
    if (!nz) return ICM_OK;
    if (labels) {
        HIPCHK(h, hipMemcpy(labels, h->label.p, nz * sizeof(int), hipMemcpyDeviceToHost));
        std::vector<int> nr((size_t)h->nloc + 1);
        HIPCHK(h, hipMemcpy(nr.data(), h->new_rank.p, nr.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (size_t t = 0; t < (size_t)h->nloc; ++t)
            for (int j = h->h_boff[t]; j < h->h_boff[t + 1]; ++j)
                if (labels[j] < 0) labels[j] = h->lact0 + nr[t];
    }
    if (target_x) HIPCHK(h, hipMemcpy(target_x, h->btx.p, nz * sizeof(double), hipMemcpyDeviceToHost));
    if (target_y) HIPCHK(h, hipMemcpy(target_y, h->bty.p, nz * sizeof(double), hipMemcpyDeviceToHost));
    return ICM_OK;
}

int icm_get_raw_map(icm_handle* h, double* y, double* counts, int64_t* lact) {
    if (!h) return ICM_ERR_ARG;
    if (h->h_yraw.empty() && !h->raw_on_device) FAIL(h, ICM_ERR_ARG, "icm_get_raw_map: no sweep has finished");
    {
        HIPCHK(h, hipSetDevice(h->device));
        int rcf = fetch_raw_map(h);
        if (rcf) return rcf;
    }
    if (y) std::copy(h->h_yraw.begin(), h->h_yraw.end(), y);
    if (counts) std::copy(h->h_cntraw.begin(), h->h_cntraw.end(), counts);
    if (lact) *lact = h->lact_raw;
    return ICM_OK;
}

static int run_one(icm_handle* h, int energy_only, int two_sided, const double* x, const double* x_ant,
                   const double* x_pos, const double* u, const double* odo, int odo_cols, const double* bx,
                   const double* by, const double* tx, const double* ty, int64_t n, double* out) {
    if (!h) return ICM_ERR_ARG;
    if (!x_ant || !u || !odo || !out || (n > 0 && (!bx || !by || !tx || !ty))) FAIL(h, ICM_ERR_ARG, "solve_one: null pointer");
    if (two_sided && (!x_pos || odo_cols != 3)) FAIL(h, ICM_ERR_ARG, "solve_one: two-sided needs x_pos and odo (3,3)");
    if (!two_sided && odo_cols != 2 && odo_cols != 3) FAIL(h, ICM_ERR_ARG, "solve_one: odo must be (3,2) or (3,3)");
    if (n < 0 || n > (1 << 24)) FAIL(h, ICM_ERR_ARG, "solve_one: bad n");
    HIPCHK(h, hipSetDevice(h->device));
    double p[22] = {0};
    if (x) std::memcpy(p, x, 3 * sizeof(double));
    std::memcpy(p + 3, x_ant, 3 * sizeof(double));
    if (two_sided) std::memcpy(p + 6, x_pos, 3 * sizeof(double));
    // u is (2,2) row-major = u[:, t-1:t+1] (two-sided) or (2,1) = u[:, t-1]
    const int uc = two_sided ? 2 : 1;
    p[9] = u[0]; p[10] = u[uc];
    if (two_sided) { p[11] = u[1]; p[12] = u[uc + 1]; }
    for (int r = 0; r < 3; ++r) {
        p[13 + r] = odo[r * odo_cols];
        p[16 + r] = odo[r * odo_cols + 1];
        if (two_sided) p[19 + r] = odo[r * odo_cols + 2];
    }
    DevBuf<double> buf;
    const size_t nn = (size_t)n;
    HIPCHK(h, buf.reserve(22 + 4 * nn + 6));
    HIPCHK(h, hipMemcpy(buf.p, p, 22 * sizeof(double), hipMemcpyHostToDevice));
    if (nn) {
        HIPCHK(h, hipMemcpy(buf.p + 22, bx, nn * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(buf.p + 22 + nn, by, nn * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(buf.p + 22 + 2 * nn, tx, nn * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(buf.p + 22 + 3 * nn, ty, nn * sizeof(double), hipMemcpyHostToDevice));
    }
    OneArgs a;
    a.two_sided = two_sided; a.energy_only = energy_only; a.n = (int)n;
    a.p = buf.p; a.bx = buf.p + 22; a.by = buf.p + 22 + nn; a.tx = buf.p + 22 + 2 * nn; a.ty = buf.p + 22 + 3 * nn;
    a.dt = h->cfg.deltat; a.R0 = h->cfg.R[0]; a.R1 = h->cfg.R[1]; a.R2 = h->cfg.R[2];
    a.Q0 = h->cfg.Q[0]; a.Q1 = h->cfg.Q[1]; a.cte = h->cfg.cte_odom;
    a.out = buf.p + 22 + 4 * nn;
    k_solve_one<<<1, kWave, 0, h->stream>>>(a);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double o[6];
    HIPCHK(h, hipMemcpy(o, a.out, 6 * sizeof(double), hipMemcpyDeviceToHost));
    buf.release();
    if (energy_only) out[0] = o[3]; else std::memcpy(out, o, 6 * sizeof(double));
    return ICM_OK;
}

int icm_solve_one(icm_handle* h, int two_sided, const double* x_ant, const double* x_pos, const double* u,
                  const double* odo, int odo_cols, const double* bx, const double* by, const double* tx,
                  const double* ty, int64_t n, double* out) {
    return run_one(h, 0, two_sided, nullptr, x_ant, x_pos, u, odo, odo_cols, bx, by, tx, ty, n, out);
}

int icm_energy_one(icm_handle* h, int two_sided, const double* x, const double* x_ant, const double* x_pos,
                   const double* u, const double* odo, int odo_cols, const double* bx, const double* by,
                   const double* tx, const double* ty, int64_t n, double* out) {
    if (h && !x) FAIL(h, ICM_ERR_ARG, "icm_energy_one: null x");
    if (two_sided == 2)  // observation energy h(x) only
        return run_one(h, 2, 0, x, x_ant, nullptr, u, odo, odo_cols, bx, by, tx, ty, n, out);
    return run_one(h, 1, two_sided, x, x_ant, x_pos, u, odo, odo_cols, bx, by, tx, ty, n, out);
}

int icm_filtrar(const icm_config* cfg, const double* y, const double* counts, int64_t lact, double* y_out,
                double* counts_out, int64_t* lact_out) {
    if (!cfg || !y || !counts || !y_out || !counts_out || !lact_out) {
        g_create_err = "icm_filtrar: null argument";
        return ICM_ERR_ARG;
    }
    return filtrar_host(*cfg, y, counts, lact, y_out, counts_out, lact_out, g_create_err);
}

static int launch_filtrar_merge(icm_handle* h, hipStream_t fs, int n);

int icm_filtrar_device(icm_handle* h, const double* y, const double* counts, int64_t lact, double* y_out,
                       double* counts_out, int64_t* lact_out, int* path_out) {
    if (!h) return ICM_ERR_ARG;
    if (!y || !counts || !y_out || !counts_out || !lact_out) FAIL(h, ICM_ERR_ARG, "icm_filtrar_device: null argument");
    const size_t L = (size_t)h->cfg.L;
    if (lact < 0 || lact > (int64_t)L) FAIL(h, ICM_ERR_ARG, "icm_filtrar_device: landmarks_actuales outside [0, L]");
    if (h->world > 1) FAIL(h, ICM_ERR_UNSUPPORTED, "icm_filtrar_device: not on a handle bound to a multi-rank exchange");
    HIPCHK(h, hipSetDevice(h->device));
    {
        int rc = reserve_map_buffers(h);
        if (rc) return rc;
    }
    HIPCHK(h, h->new_rank.reserve((size_t)h->nloc + 1));
    hipStream_t st = h->stream;
    HIPCHK(h, hipMemcpyAsync(h->y_raw.p, y, 2 * L * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(h, hipMemcpyAsync(h->cnt_raw.p, counts, L * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(h, hipMemsetAsync(h->new_rank.p + h->nloc, 0, sizeof(int), st));
    const int lact0_keep = h->lact0;
    h->lact0 = (int)lact;
    int rc = launch_filtrar(h, st, false);
    int info[4] = {0, 2, 0, 0};
    if (!rc) {
        HIPCHK(h, hipMemcpyAsync(info, h->fl_info.p, sizeof(info), hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipStreamSynchronize(st));
        if (info[1] == 1) {
            rc = launch_filtrar_merge(h, st, info[0]);
            if (!rc) {
                HIPCHK(h, hipMemcpyAsync(info, h->fl_info.p, sizeof(info), hipMemcpyDeviceToHost, st));
                HIPCHK(h, hipStreamSynchronize(st));
                if (info[1] == 0) info[1] = -1;   // merged on the device
            }
        }
    }
    h->lact0 = lact0_keep;
    h->have_state = false;   // the search structures now belong to this map, not to the sweep state
    h->dev_map_current = false;
    if (rc) return rc;
    if (info[1] == 2) {      // coincident landmarks / empty map / a component beyond kCompMax: exact host routine
        if (path_out) *path_out = 2;
        return filtrar_host(h->cfg, y, counts, lact, y_out, counts_out, lact_out, h->err);
    }
    const size_t n = (size_t)info[0];
    std::fill(y_out, y_out + 2 * L, 0.0);
    if (n) {
        HIPCHK(h, hipMemcpy(y_out, h->mapx.p, n * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(y_out + L, h->mapy.p, n * sizeof(double), hipMemcpyDeviceToHost));
    }
    HIPCHK(h, hipMemcpy(counts_out, h->counts_new.p, L * sizeof(double), hipMemcpyDeviceToHost));
    *lact_out = (int64_t)n;
    if (path_out) *path_out = info[1] == -1 ? 1 : 0;
    return ICM_OK;
}

int icm_cluster_first_scan(const double* pts, int64_t n, double t, int32_t* labels_out) {
    if (!pts || !labels_out) {
        g_create_err = "icm_cluster_first_scan: null argument";
        return ICM_ERR_ARG;
    }
    return cluster_first_scan_host(pts, n, t, labels_out, g_create_err);
}

// Mapa.actualizar for ONE scan (reference scripts/ICM_SLAM_tools.py:128-201), both branches.
int icm_associate(icm_handle* h, const double* obs, int64_t n, const double* map_ref, int64_t K_ref, double* map,
                  double* counts, int64_t* lact_inout, int64_t* labels_out) {
    if (!h) return ICM_ERR_ARG;
    if (!map || !counts || !lact_inout || (n > 0 && (!obs || !labels_out))) FAIL(h, ICM_ERR_ARG, "icm_associate: null pointer");
    if (n < 0 || n > (1 << 24) || K_ref < 0) FAIL(h, ICM_ERR_ARG, "icm_associate: bad n / K_ref");
    const int64_t L = h->cfg.L;
    int64_t lact = *lact_inout;
    if (lact < 0) FAIL(h, ICM_ERR_ARG, "icm_associate: landmarks_actuales < 0");
    std::vector<int> lab((size_t)std::max<int64_t>(n, 1), -1);
    if (lact == 0) {
        // first scan ever: single-linkage clusters at dist_thr, their centres and sizes (:160-165)
        if (n == 0) FAIL(h, ICM_ERR_ARG, "icm_associate: no observation to seed the map with (the reference's pdist/linkage raises on an empty scan)");
        int rc = cluster_first_scan_host(obs, n, h->cfg.dist_thr, lab.data(), h->err);
        if (rc) return rc;
        int ncl = 0;
        for (int64_t j = 0; j < n; ++j) ncl = std::max(ncl, lab[(size_t)j] + 1);
        if (ncl > L) FAIL(h, ICM_ERR_INDEX, "icm_associate: more first-scan clusters than the map capacity L");
        for (int i = 0; i < ncl; ++i) {
            double sx = 0.0, sy = 0.0;   // np.mean(obs[c == i, :], axis=0): rows added in order, then / k
            int64_t k = 0;
            for (int64_t j = 0; j < n; ++j)
                if (lab[(size_t)j] == i) {
                    sx += obs[2 * j];
                    sy += obs[2 * j + 1];
                    ++k;
                }
            map[i] = sx / (double)k;
            map[L + i] = sy / (double)k;
            counts[i] = (double)k;
        }
        for (int64_t j = 0; j < n; ++j) labels_out[j] = lab[(size_t)j];
        *lact_inout = ncl;
        return ICM_OK;
    }
    if (n == 0) return ICM_OK;   // nothing observed: labels empty, map and counters untouched
    // columns the scan can be matched against: mapa_referencia[:, :Lact] (numpy clamps the slice)
    const int64_t km = std::min(lact, K_ref);
    if (km <= 0 || !map_ref) FAIL(h, ICM_ERR_ARG, "icm_associate: empty reference map (np.amin of an empty cdist raises ValueError in the reference)");
    HIPCHK(h, hipSetDevice(h->device));
    DevBuf<double> dobs, dmap;
    DevBuf<int> dlab;
    HIPCHK(h, dobs.reserve(2 * (size_t)n));
    HIPCHK(h, dmap.reserve(2 * (size_t)km));
    HIPCHK(h, dlab.reserve((size_t)n));
    HIPCHK(h, hipMemcpyAsync(dobs.p, obs, 2 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(dmap.p, map_ref, (size_t)km * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(dmap.p + km, map_ref + K_ref, (size_t)km * sizeof(double), hipMemcpyHostToDevice, h->stream));
    k_scan_labels<<<nblocks_threads(n), kBlock, 0, h->stream>>>(dobs.p, (int)n, dmap.p, dmap.p + km, (int)km, h->cfg.dist_thr, dlab.p);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(lab.data(), dlab.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    dobs.release(); dmap.release(); dlab.release();
    // every gated-out observation of the scan gets the SAME fresh label Lact: the reference clusters
    // `ztt[:, 2:4]` of an (m, 2) array, i.e. zero-width points -> one cluster (:173-182)
    bool anynew = false;
    for (int64_t j = 0; j < n; ++j) anynew |= lab[(size_t)j] < 0;
    if (anynew) {
        if (lact >= L) FAIL(h, ICM_ERR_INDEX, "icm_associate: a new landmark does not fit in L (the reference raises IndexError, scripts/ICM_SLAM_tools.py:191)");
        for (int64_t j = 0; j < n; ++j)
            if (lab[(size_t)j] < 0) lab[(size_t)j] = (int)lact;
        ++lact;
    }
    // running means, label by label (:184-195): y <- sum(obs_i)/(n_i + k) + y n_i/(n_i + k); n_i += k
    std::vector<char> done((size_t)n, 0);
    for (int64_t j = 0; j < n; ++j) {
        if (done[(size_t)j]) continue;
        const int i = lab[(size_t)j];
        double sx = 0.0, sy = 0.0;
        int64_t k = 0;
        for (int64_t q = j; q < n; ++q)
            if (lab[(size_t)q] == i) {
                sx += obs[2 * q];
                sy += obs[2 * q + 1];
                ++k;
                done[(size_t)q] = 1;
            }
        const double nn = counts[i], tot = nn + (double)k;
        map[i] = sx / tot + map[i] * nn / tot;
        map[L + i] = sy / tot + map[L + i] * nn / tot;
        counts[i] = tot;
    }
    for (int64_t j = 0; j < n; ++j) labels_out[j] = lab[(size_t)j];
    *lact_inout = lact;
    return ICM_OK;
}

int icm_init_pass(icm_handle* h, const double* x0, double* y, double* counts, int64_t* lact, double* x_out) {
    if (!h) return ICM_ERR_ARG;
    if (!h->prefiltered) FAIL(h, ICM_ERR_ARG, "icm_init_pass: call icm_upload + icm_prefilter first");
    if (!x0 || !y || !counts || !lact || !x_out) FAIL(h, ICM_ERR_ARG, "icm_init_pass: null pointer");
    if (h->t_begin != 0 || h->nloc != h->T) FAIL(h, ICM_ERR_UNSUPPORTED, "icm_init_pass: the causal pass is one chain; it needs the whole sequence on one GPU");
    const size_t T = (size_t)h->T, L = (size_t)h->cfg.L;
    if (*lact <= 0 || *lact > (int64_t)L) FAIL(h, ICM_ERR_ARG, "icm_init_pass: seed the map with the first scan's clusters (icm_cluster_first_scan)");
    int maxb = 0;
    for (size_t t = 0; t < T; ++t) maxb = std::max(maxb, h->h_boff[t + 1] - h->h_boff[t]);
    maxb = std::max(maxb, 1);
    const size_t lds = (size_t)maxb * (4 * sizeof(double) + sizeof(int));
    if (lds > 160 * 1024) FAIL(h, ICM_ERR_UNSUPPORTED, "icm_init_pass: too many kept beams per scan for the LDS staging");
    HIPCHK(h, hipSetDevice(h->device));
    DevBuf<double> dx, dy, dc;
    DevBuf<int> di;
    HIPCHK(h, dx.reserve(3 * T)); HIPCHK(h, dy.reserve(2 * L)); HIPCHK(h, dc.reserve(L)); HIPCHK(h, di.reserve(2));
    std::vector<double> xt(3 * T, 0.0);
    xt[0] = x0[0]; xt[1] = x0[1]; xt[2] = x0[2];
    int hi[2] = {(int)*lact, 0};
    HIPCHK(h, hipMemcpy(dx.p, xt.data(), 3 * T * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(dy.p, y, 2 * L * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(dc.p, counts, L * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(di.p, hi, 2 * sizeof(int), hipMemcpyHostToDevice));
    InitArgs a;
    a.x = dx.p; a.odo = h->odo.p; a.u = h->u.p; a.T = (int)T; a.boff = h->boff.p; a.bx = h->bx.p; a.by = h->by.p;
    a.y = dy.p; a.cnt = dc.p; a.lact = di.p; a.L = (int)L; a.maxb = maxb; a.thr = h->cfg.dist_thr;
    a.dt = h->cfg.deltat; a.R0 = h->cfg.R[0]; a.R1 = h->cfg.R[1]; a.R2 = h->cfg.R[2];
    a.Q0 = h->cfg.Q[0]; a.Q1 = h->cfg.Q[1]; a.cte = h->cfg.cte_odom; a.flags = di.p + 1;
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_init_pass), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_init_pass<<<1, kWave, lds, h->stream>>>(a);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(hi, di.p, 2 * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(xt.data(), dx.p, 3 * T * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(y, dy.p, 2 * L * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(counts, dc.p, L * sizeof(double), hipMemcpyDeviceToHost));
    dx.release(); dy.release(); dc.release(); di.release();
    if (hi[1]) FAIL(h, ICM_ERR_INDEX, "init pass: new landmarks exceed the map capacity L (the reference raises IndexError, scripts/ICM_SLAM_tools.py:191)");
    *lact = hi[0];
    for (size_t t = 0; t < T; ++t) {
        x_out[t] = xt[3 * t];
        x_out[T + t] = xt[3 * t + 1];
        x_out[2 * T + t] = xt[3 * t + 2];
    }
    return ICM_OK;
}

int icm_enable_timing(icm_handle* h, int on) {
    if (!h) return ICM_ERR_ARG;
    h->timing = on != 0;
    return ICM_OK;
}
int icm_reset_timing(icm_handle* h) {
    if (!h) return ICM_ERR_ARG;
    for (int i = 0; i < KID_COUNT; ++i) {
        h->k_ms[i] = 0.0;
        h->k_n[i] = 0;
    }
    return ICM_OK;
}
int icm_kernel_count(const icm_handle*) { return KID_COUNT; }
int icm_kernel_time(icm_handle* h, int idx, const char** name, double* ms, int64_t* launches) {
    if (!h || idx < 0 || idx >= KID_COUNT) return ICM_ERR_ARG;
    if (name) *name = kKernelNames[idx];
    if (ms) *ms = h->k_ms[idx];
    if (launches) *launches = h->k_n[idx];
    return ICM_OK;
}
int icm_last_stats(const icm_handle* h, int64_t* out4) {
    if (!h || !out4) return ICM_ERR_ARG;
    out4[0] = h->nnz;
    out4[1] = h->E;
    out4[2] = h->n_new_loc;
    out4[3] = h->lact_raw;
    return ICM_OK;
}

int icm_set_debug(icm_handle* h, int on) {
    if (!h) return ICM_ERR_ARG;
    h->debug = on != 0;
    return ICM_OK;
}

#ifdef ICM_ASSOC_TS
int icm_debug_assoc_ts(icm_handle* h, unsigned long long* out, int n8) {
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpyFromSymbol(out, HIP_SYMBOL(icm::g_assoc_ts), sizeof(unsigned long long) * (size_t)n8));
    return ICM_OK;
}
#endif
#ifdef ICM_WAVE_TS
int icm_debug_wave_ts(icm_handle* h, unsigned long long* out, int n4) {
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpyFromSymbol(out, HIP_SYMBOL(icm::g_wave_ts), sizeof(unsigned long long) * (size_t)n4));
    return ICM_OK;
}
int icm_debug_eval_stats(icm_handle* h, unsigned long long* out4, int reset) {
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpyFromSymbol(out4, HIP_SYMBOL(icm::g_eval_stats), sizeof(unsigned long long) * 4));
    if (reset) {
        unsigned long long z[4] = {0, 0, 0, 0};
        HIPCHK(h, hipMemcpyToSymbol(HIP_SYMBOL(icm::g_eval_stats), z, sizeof(z)));
    }
    return ICM_OK;
}
#endif

int icm_get_solve_diag(icm_handle* h, double* out) {
    if (!h || !out) return ICM_ERR_ARG;
    if (!h->diag.p) FAIL(h, ICM_ERR_ARG, "icm_get_solve_diag: enable icm_set_debug before the sweep");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out, h->diag.p, 3 * (size_t)h->T * sizeof(double), hipMemcpyDeviceToHost));
    return ICM_OK;
}

int icm_set_energy_form(icm_handle* h, int form) {
    if (!h) return ICM_ERR_ARG;
    if (form < 0 || form > 2) FAIL(h, ICM_ERR_ARG, "icm_set_energy_form: form must be 0, 1 or 2");
    h->form = form;
    h->per_beam = form == 1;
    return ICM_OK;
}

int icm_set_solve_lanes(icm_handle* h, int mode) {
    if (!h) return ICM_ERR_ARG;
    if (mode < -1 || mode > 1) FAIL(h, ICM_ERR_ARG, "icm_set_solve_lanes: mode must be -1, 0 or 1");
    h->solve_quad = mode;
    return ICM_OK;
}

int icm_set_colour_fusion(icm_handle* h, int on) {
    if (!h) return ICM_ERR_ARG;
    h->fuse_colours = on != 0;
    return ICM_OK;
}

int icm_set_fused_spin_limit(icm_handle* h, int polls) {
    if (!h) return ICM_ERR_ARG;
    if (polls < 0) FAIL(h, ICM_ERR_ARG, "icm_set_fused_spin_limit: polls must be >= 0");
    h->fused_spin_limit = polls;
    return ICM_OK;
}

int icm_get_fused_deferred(icm_handle* h, int64_t* waves) {
    if (!h || !waves) return ICM_ERR_ARG;
    *waves = 0;
    if (!h->solve_counts.p) return ICM_OK;
    HIPCHK(h, hipSetDevice(h->device));
    unsigned long long v = 0;
    HIPCHK(h, hipMemcpyAsync(&v, h->solve_counts.p, sizeof(v), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *waves = (int64_t)v;
    return ICM_OK;
}

int icm_get_fixup_poses(icm_handle* h, int64_t* poses) {
    if (!h || !poses) return ICM_ERR_ARG;
    *poses = 0;
    if (!h->solve_counts.p) return ICM_OK;
    HIPCHK(h, hipSetDevice(h->device));
    unsigned long long v = 0;
    HIPCHK(h, hipMemcpyAsync(&v, h->solve_counts.p + 1, sizeof(v), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *poses = (int64_t)v;
    return ICM_OK;
}

int icm_get_dropin_counts(const icm_handle* h, int64_t* out3) {
    if (!h || !out3) return ICM_ERR_ARG;
    for (int i = 0; i < 3; ++i) out3[i] = h->dropin_counts[i];
    return ICM_OK;
}

int icm_set_phase_timing(icm_handle* h, int on) {
    if (!h) return ICM_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (on)
        for (auto& e : h->ev_ph)
            if (!e) HIPCHK(h, hipEventCreate(&e));
    h->phase_timing = on != 0;
    for (double& v : h->ph_ms) v = 0.0;
    h->ph_n = 0;
    return ICM_OK;
}

int icm_get_phase_times(const icm_handle* h, double* out5, int64_t* sweeps) {
    if (!h || !out5 || !sweeps) return ICM_ERR_ARG;
    for (int i = 0; i < 5; ++i) out5[i] = h->ph_ms[i];
    *sweeps = h->ph_n;
    return ICM_OK;
}

int icm_set_fault(icm_handle* h, int where) {
    if (!h) return ICM_ERR_ARG;
    if (where < 0 || where > 2) FAIL(h, ICM_ERR_ARG, "icm_set_fault: 0 (none), 1 (the next icm_sweep_local reports a HIP error) or 2 (the next icm_sweep_targets does)");
    h->fault = where;
    return ICM_OK;
}

int icm_set_fold_mode(icm_handle* h, int mode) {
    if (!h) return ICM_ERR_ARG;
    if (mode < -1 || mode > 1) FAIL(h, ICM_ERR_ARG, "icm_set_fold_mode: mode must be -1, 0 or 1");
    h->fold_mode = mode;
    return ICM_OK;
}

int icm_staging_layout(int64_t nnz, int64_t nloc, int64_t* out3) {
    if (!out3) return ICM_ERR_ARG;
    StagingLayout l;
    const bool ok = staging_layout(nnz, nloc, l);
    out3[0] = l.sparse0; out3[1] = l.entries; out3[2] = l.prefix_stride;
    return ok ? ICM_OK : ICM_ERR_CAPACITY;
}

int icm_set_entry_path(icm_handle* h, int mode) {
    if (!h) return ICM_ERR_ARG;
    if (mode < -1 || mode > 1) FAIL(h, ICM_ERR_ARG, "icm_set_entry_path: mode must be -1, 0 or 1");
    h->entry_path = mode;
    h->hier_ok = true;
    return ICM_OK;
}

int icm_get_entry_path(const icm_handle* h) { return h ? h->path_used : ICM_ERR_ARG; }

int icm_last_filtrar_info(const icm_handle* h, int64_t* out3) {
    if (!h || !out3) return ICM_ERR_ARG;
    out3[0] = h->lact;
    out3[1] = h->filtrar_path;
    out3[2] = h->filtrar_path == 2 ? -1 : h->pin_i[10];
    return ICM_OK;
}

int icm_set_gpu_filtrar(icm_handle* h, int on) {
    if (!h) return ICM_ERR_ARG;
    h->gpu_filtrar = on != 0;
    return ICM_OK;
}

// test hook: switch phase A to the brute-force (all landmarks, LDS-tiled) kernel
int icm_get_wait_giveups(const icm_handle* h, int64_t* sweeps) {
    if (!h || !sweeps) return ICM_ERR_ARG;
    *sweeps = h->wait_giveups;
    return ICM_OK;
}

int icm_set_assoc_form(icm_handle* h, int form) {
    if (!h || (form != 0 && form != 1)) return ICM_ERR_ARG;
    h->assoc_form = form;
    return ICM_OK;
}

int icm_get_run_counts(icm_handle* h, int64_t* out2) {
    if (!h || !out2) return ICM_ERR_ARG;
    if (!h->prefiltered) FAIL(h, ICM_ERR_ARG, "icm_get_run_counts: call icm_prefilter first");
    HIPCHK(h, hipSetDevice(h->device));
    unsigned long long c[2] = {0, 0};
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(c, h->run_counts.p, sizeof(c), hipMemcpyDeviceToHost));
    out2[0] = h->nruns;
    out2[1] = (int64_t)c[1];
    return ICM_OK;
}

int icm_get_runs(icm_handle* h, int64_t* offsets, double* centre_xy, double* sum_xy, float* radius, int32_t* count, int32_t* first) {
    if (!h) return ICM_ERR_ARG;
    if (!h->prefiltered) FAIL(h, ICM_ERR_ARG, "icm_get_runs: call icm_prefilter first");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const size_t nloc = (size_t)h->nloc, nr = (size_t)h->nruns;
    if (offsets) {
        std::vector<int> ro(nloc + 1);
        HIPCHK(h, hipMemcpy(ro.data(), h->roff.p, (nloc + 1) * sizeof(int), hipMemcpyDeviceToHost));
        for (size_t i = 0; i <= nloc; ++i) offsets[i] = ro[i];
    }
    if (nr) {
        std::vector<double> sums;
        if (centre_xy && !sum_xy) { sums.resize(2 * nr); sum_xy = sums.data(); }
        if (sum_xy) HIPCHK(h, hipMemcpy(sum_xy, h->r_s.p, nr * sizeof(double2), hipMemcpyDeviceToHost));
        if (radius || count || first || centre_xy) {
            std::vector<uint2> m(nr);
            HIPCHK(h, hipMemcpy(m.data(), h->r_m.p, nr * sizeof(uint2), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < nr; ++i) {
                if (centre_xy) {   // (not stored: the same division the kernels make)
                    const double kd = (double)(m[i].y & 0xffffu);
                    centre_xy[2 * i] = sum_xy[2 * i] / kd;
                    centre_xy[2 * i + 1] = sum_xy[2 * i + 1] / kd;
                }
                if (radius) std::memcpy(&radius[i], &m[i].x, sizeof(float));
                if (count) count[i] = (int32_t)(m[i].y & 0xffffu);
                if (first) first[i] = (int32_t)(m[i].y >> 16);
            }
        }
    }
    return ICM_OK;
}

int icm_set_brute_force(icm_handle* h, int on) {
    if (!h) return ICM_ERR_ARG;
    h->brute = on != 0;
    return ICM_OK;
}

}  // extern "C"
