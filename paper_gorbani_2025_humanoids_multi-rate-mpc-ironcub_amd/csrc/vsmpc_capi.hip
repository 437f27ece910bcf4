// C-ABI of include/vsmpc.h: handle management, host<->device staging, dense-QP debug assembly.
// No compute fallback lives here: every numeric result comes from the HIP kernels.
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "vsmpc_launch.hpp"
#include "../../include/vsmpc_jet.h"

using namespace vsmpc;

struct vsmpc_handle {
    vsmpc_config cfg;
    DevCfg dev;
    int variant;
    int form;        // condensing form of the solve kernel (vsmpc_set_kernel_form)
    KinOpts kin;     // vsmpc_set_kinematics_options
    int device;
    int max_batch;
    int n_var, n_con, n_in, n_p;
    // device staging buffers for the host-pointer entry points
    double* d_in;
    double* d_x;
    double* d_fm;
    int* d_status;
    int* d_iters;
    double* d_lin;  // A | Bj | Bt | c for max_batch instances
    double* d_dbg;  // M | L for one instance
    double* d_kin;  // vsmpc_kinematics_batch: records in, terms out (max_batch instances)
    double* d_kout;
    unsigned long long* d_stamps;  // vsmpc_debug_phase_cycles (max_batch x 16)
    hipEvent_t ev0, ev1;
    // host-pointer entry for larger batches: chunks alternate between two streams so that the upload of one chunk, the
    // solve of the previous one and the download of the one before overlap (full overlap needs pinned caller buffers)
    hipStream_t pipe[4];
    hipEvent_t pipe_done[4];
    hipEvent_t pipe_start;
    // small batches through the host-pointer entry (the reference's own use: one instance per tick): pinned,
    // device-mapped staging that the kernel reads and writes directly, instead of five small copies
    double* h_stage;      // host view:  in[ZC_MAX][n_in] | x[ZC_MAX][n_var] | fm[ZC_MAX][24] | status[ZC_MAX] | iters[ZC_MAX] |
                          //             kin[ZC_MAX][VSMPC_KIN_SIZE] (vsmpc_tick)
    double* d_stage;      // device view of the same allocation (its own base pointer: the two views are unrelated addresses)
};

constexpr int ZC_MAX = 8;  // largest batch served through the mapped staging buffer
#ifndef VS_PIPE_CHUNK
#define VS_PIPE_CHUNK 1024
#endif
#ifndef VS_PIPE_STREAMS
#define VS_PIPE_STREAMS 2
#endif
constexpr int PIPE_CHUNK = VS_PIPE_CHUNK;   // instances per chunk of the pipelined host-pointer entry
constexpr int PIPE_STREAMS = VS_PIPE_STREAMS;
static_assert(PIPE_STREAMS >= 1 && PIPE_STREAMS <= 4, "vsmpc_handle::pipe holds four streams");

// resident closed-loop state of a batch (uses the handle's record / first-move / status buffers as its per-tick scratch)
struct vsmpc_rollout {
    vsmpc_handle* h;
    int batch;
    int substeps;
    RolloutDev rd;
    double* d_state;
    double* d_params;
    int* d_tick;
    double* d_tpos;
    double* d_tvel;
    double* d_talpha;
    double* d_trpy;           // vsmpc_rollout_set_attitude_tracks (or nullptr)
    double* d_trpyd;
    double* d_log;
    int log_ticks;
    double* d_tstate;         // per-instance tick state: reference window FIFO, RPY unwrap (see vsmpc_rollout.hip)
    int valid;                // 0 after a run failed half-way: the device counters are ahead, reset() before the next run
    double* d_rec;            // record of the next tick ([batch][n_in]): written by reset and by every tick's advance
    RolloutCtl* d_ctl;        // per-run control block read by advance_kernel (log destination, tick base)
    int ticks_done;           // ticks since the last reset (the same for every instance)
    hipStream_t own_stream;   // used when the caller passes the null stream (which cannot be captured)
    hipGraphExec_t gexec;     // GRAPH_TICKS ticks (3 launches each) captured once, replayed per chunk
    int graph_state;          // 0 not built yet, 1 ready, -1 capture unavailable (direct launches only)
    int graph_form;           // h->form the graph was captured with (vsmpc_set_kernel_form on the handle rebuilds it)
    // kinematic-tree plant (vsmpc_rollout_set_tree)
    int use_tree;
    vsmpc_tree tree;
    double* d_rs;             // provider states [batch][VSMPC_RS_SIZE]
    double* d_ro;             // Robot-level outputs of the provider [batch][VSMPC_RO_SIZE]
};

namespace {

thread_local char g_hip_msg[256] = "";

int hip_fail(hipError_t e, const char* what) {
    snprintf(g_hip_msg, sizeof(g_hip_msg), "HIP error in %s: %s", what, hipGetErrorString(e));
    return VSMPC_ERR_HIP;
}

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return hip_fail(_e, #expr); \
    } while (0)

#define ON_DEVICE(dev) DeviceScope _scope(dev); HIP_TRY(_scope.err)

// carve-up of the mapped staging buffer (host or device view)
struct Stage {
    double* in; double* x; double* fm; int* st; int* it; double* kin;
};
Stage stage_view(const vsmpc_handle* h, double* base) {
    Stage v;
    v.in = base;
    v.x = v.in + size_t(ZC_MAX) * h->n_in;
    v.fm = v.x + size_t(ZC_MAX) * h->n_var;
    v.st = reinterpret_cast<int*>(v.fm + size_t(ZC_MAX) * VSMPC_FM_SIZE);
    v.it = v.st + ZC_MAX;
    v.kin = v.fm + size_t(ZC_MAX) * (VSMPC_FM_SIZE + 1);
    return v;
}

// Tree plant of a rollout: provider on the body-frame states (joints + this tick's move when `fm` is given), then the
// kinematics terms of those records (I_B for the plant's integration; the Lambda terms are formed again after the advance,
// with the thrusts it measured).  The handle's kinematics buffers are the rollout's scratch, like its solve buffers.
hipError_t enqueue_tree(vsmpc_rollout* r, const double* fm, const int* status, hipStream_t s) {
    vsmpc_handle* h = r->h;
    hipError_t e = launch_tree_state(r->rd, r->batch, r->d_state, fm, status, r->d_rs, s);
    if (e == hipSuccess) e = launch_provider(r->tree, r->d_rs, r->batch, h->d_kin, r->d_ro, nullptr, h->n_in, s);
    if (e == hipSuccess) {
        KinOpts o = h->kin;
        o.constant_lambda = 0;
        e = launch_kinematics(h->d_kin, r->batch, h->d_kout, o, s);
    }
    return e;
}
// Lambda_lin,B | Lambda_ang,B of the tree at the thrusts the advance kernel left in the kinematics records -> r->d_rec
hipError_t enqueue_tree_lambda(vsmpc_rollout* r, hipStream_t s) {
    KinOpts o = r->h->kin;
    o.constant_lambda = 0;
    o.skip_inertia = 1;   // the record's I_G = R I_B R^T is the advance kernel's (the tree is evaluated in the body frame)
    return launch_kinematics_patch(r->h->d_kin, r->batch, r->d_rec, r->h->n_in, o, s);
}

// dt schedule: constraintsVSMPC.cpp:45-51 (beta1, beta2), :78-84, :156-159
void fill_dt(const vsmpc_config& c, double* dt) {
    const double nS = double(c.n_iter_small);
    const double beta2 = (c.period_large - nS * c.period_small) / (nS * (nS - 1.0));
    const double beta1 = c.period_small - beta2;
    auto warp = [&](double t) { return beta1 * t + beta2 * t * t; };
    for (int i = 0; i < c.n_iter; ++i)
        dt[i] = i < c.n_iter_small ? warp(double(i + 1)) - warp(double(i)) : c.period_large;
}

void fill_devcfg(const vsmpc_config& c, DevCfg& d) {
    memset(&d, 0, sizeof(d));
    fill_dt(c, d.dt);
    // diagonal of Q on the weighted rows (costsVSMPC.cpp:78-93): p, h_lin, rpy, h_ang | e_pos, e_rpy
    const double q[NWROWS] = {c.w_com_pos[0], c.w_com_pos[1], c.w_com_pos[2], c.w_lin_mom[0], c.w_lin_mom[1],
                              c.w_lin_mom[2], c.w_rpy[0], c.w_rpy[1], c.w_rpy[2], c.w_ang_mom[0],
                              c.w_ang_mom[1], c.w_ang_mom[2], c.w_com_pos_err[0], c.w_com_pos_err[1],
                              c.w_com_pos_err[2], c.w_rpy_err[0], c.w_rpy_err[1], c.w_rpy_err[2]};
    for (int i = 0; i < NWROWS; ++i) d.sq[i] = std::sqrt(q[i]);
    for (int i = 0; i < NJ; ++i) d.wj[i] = c.w_delta_joint[i] + c.w_reg_joint_pos;
    d.w_reg = c.w_reg_joint_pos;
    d.w_thr = c.w_throttle;
    d.w_init = c.w_initial_throttle;
    d.vmin = Jet::v_of_throttle_div(c.throttle_min);  // constraintsVSMPC.cpp:329-332
    d.vmax = Jet::v_of_throttle_div(c.throttle_max);
    d.use_jet = c.use_jet_dynamic ? 1 : 0;
    d.max_as_iter = 64;
}

bool config_valid(const vsmpc_config& c) {
    if (c.n_iter < 2 || c.n_iter > MAX_STAGES) return false;
    if (c.n_iter_small < 2 || c.n_iter_small > c.control_horizon) return false;
    if (c.control_horizon > c.n_iter) return false;
    if (!(c.period_small > 0.0) || !(c.period_large > 0.0)) return false;
    const double q[] = {c.w_com_pos[0], c.w_com_pos[1], c.w_com_pos[2], c.w_lin_mom[0], c.w_lin_mom[1],
                        c.w_lin_mom[2], c.w_rpy[0], c.w_rpy[1], c.w_rpy[2], c.w_ang_mom[0], c.w_ang_mom[1],
                        c.w_ang_mom[2], c.w_com_pos_err[0], c.w_com_pos_err[1], c.w_com_pos_err[2],
                        c.w_rpy_err[0], c.w_rpy_err[1], c.w_rpy_err[2], c.w_throttle, c.w_initial_throttle,
                        c.w_reg_joint_pos};
    for (double v : q)
        if (!(v >= 0.0)) return false;
    for (int i = 0; i < NJ; ++i)
        if (!(c.w_delta_joint[i] + c.w_reg_joint_pos > 0.0)) return false;  // condensed Hessian must stay PD
    if (!(c.w_initial_throttle > 0.0)) return false;
    if (!(c.throttle_max > c.throttle_min)) return false;
    return true;
}

}  // namespace

extern "C" {

int vsmpc_create(const vsmpc_config* cfg, int device, int max_batch, vsmpc_handle** out) {
    if (cfg == nullptr || out == nullptr || max_batch <= 0) return VSMPC_ERR_INVALID_ARG;
    *out = nullptr;
    if (!config_valid(*cfg)) return VSMPC_ERR_INVALID_ARG;
    const int variant = select_variant(cfg->n_iter, cfg->n_iter_small, cfg->control_horizon);
    if (variant == VARIANT_NONE) return VSMPC_ERR_UNSUPPORTED_CONFIG;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return VSMPC_ERR_INVALID_ARG;
    ON_DEVICE(device);

    vsmpc_handle* h = new (std::nothrow) vsmpc_handle();
    if (h == nullptr) return VSMPC_ERR_ALLOC;
    memset(h, 0, sizeof(*h));
    h->cfg = *cfg;
    fill_devcfg(*cfg, h->dev);
    h->variant = variant;
    h->form = initial_kernel_form();
    for (int i = 0; i < VSMPC_N_JOINTS; ++i) h->kin.sel[i] = 3 + i;   // the shipped robot: joints 3..10
    h->kin.constant_lambda = 0;
    if (h->form == 1 && !variant_has_structured(variant)) h->form = 0;
    h->device = device;
    h->max_batch = max_batch;
    const int N = cfg->n_iter, nS = cfg->n_iter_small, H = cfg->control_horizon;
    h->n_var = NX * (N + 1) + NJ * H + NTH * (H - nS + 1);
    h->n_con = NX * (N + 1) + NTH * (N - nS + 1);
    h->n_in = VSMPC_IN_XREF + 12 * (N - nS + 1);
    h->n_p = variant_condensed_dim(variant);

    const size_t B = size_t(max_batch);
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipMalloc(&h->d_in, B * h->n_in * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&h->d_x, B * h->n_var * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&h->d_fm, B * VSMPC_FM_SIZE * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&h->d_status, B * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&h->d_iters, B * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&h->d_lin, B * (NX * NX + NX * NJ + NX * NTH + NX) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&h->d_dbg, size_t(2) * h->n_p * h->n_p * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&h->d_kin, B * VSMPC_KIN_SIZE * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&h->d_kout, B * VSMPC_KIN_OUT * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&h->d_stamps, B * 16 * sizeof(unsigned long long));
    for (int i = 0; i < PIPE_STREAMS && e == hipSuccess; ++i) {
        e = hipStreamCreateWithFlags(&h->pipe[i], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&h->pipe_done[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->pipe_start, hipEventDisableTiming);
    if (e == hipSuccess) {
        const size_t zc = size_t(ZC_MAX) * (h->n_in + h->n_var + VSMPC_FM_SIZE + 1 + VSMPC_KIN_SIZE) * sizeof(double);  // ints share one double
        e = hipHostMalloc(reinterpret_cast<void**>(&h->h_stage), zc, hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_stage), h->h_stage, 0);
    }
    if (e == hipSuccess) e = hipEventCreate(&h->ev0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e != hipSuccess) {
        vsmpc_destroy(h);
        return e == hipErrorOutOfMemory ? VSMPC_ERR_ALLOC : hip_fail(e, "vsmpc_create");
    }
    *out = h;
    return VSMPC_OK;
}

void vsmpc_destroy(vsmpc_handle* h) {
    if (h == nullptr) return;
    DeviceScope scope(h->device);
    if (h->d_in) (void)hipFree(h->d_in);
    if (h->d_x) (void)hipFree(h->d_x);
    if (h->d_fm) (void)hipFree(h->d_fm);
    if (h->d_status) (void)hipFree(h->d_status);
    if (h->d_iters) (void)hipFree(h->d_iters);
    if (h->d_lin) (void)hipFree(h->d_lin);
    if (h->d_dbg) (void)hipFree(h->d_dbg);
    if (h->d_kin) (void)hipFree(h->d_kin);
    if (h->d_kout) (void)hipFree(h->d_kout);
    if (h->d_stamps) (void)hipFree(h->d_stamps);
    for (int i = 0; i < PIPE_STREAMS; ++i) {
        if (h->pipe[i]) (void)hipStreamDestroy(h->pipe[i]);
        if (h->pipe_done[i]) (void)hipEventDestroy(h->pipe_done[i]);
    }
    if (h->pipe_start) (void)hipEventDestroy(h->pipe_start);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    delete h;
}

int vsmpc_num_variables(const vsmpc_handle* h) { return h ? h->n_var : VSMPC_ERR_INVALID_ARG; }
int vsmpc_num_constraints(const vsmpc_handle* h) { return h ? h->n_con : VSMPC_ERR_INVALID_ARG; }
int vsmpc_input_doubles(const vsmpc_handle* h) { return h ? h->n_in : VSMPC_ERR_INVALID_ARG; }
int vsmpc_max_batch(const vsmpc_handle* h) { return h ? h->max_batch : VSMPC_ERR_INVALID_ARG; }
int vsmpc_condensed_dim(const vsmpc_handle* h) { return h ? h->n_p : VSMPC_ERR_INVALID_ARG; }
const char* vsmpc_kernel_name(const vsmpc_handle* h) { return h ? variant_kernel_name(h->variant) : "none"; }

int vsmpc_solve_batch_device(vsmpc_handle* h, const double* d_in, int batch, double* d_x, double* d_first_move,
                             int* d_status, int* d_iters, void* stream) {
    if (h == nullptr || d_in == nullptr || d_status == nullptr || batch < 0) return VSMPC_ERR_INVALID_ARG;
    if (batch > h->max_batch) return VSMPC_ERR_BATCH_TOO_LARGE;
    if (batch == 0) return VSMPC_OK;
    ON_DEVICE(h->device);   // an enqueue-only entry must not change the caller's current device either
    HIP_TRY(launch_solve(h->variant, h->form, h->dev, d_in, batch, d_x, d_first_move, d_status, d_iters, nullptr, nullptr, nullptr,
                         static_cast<hipStream_t>(stream)));
    return VSMPC_OK;
}

int vsmpc_solve_batch(vsmpc_handle* h, const double* in, int batch, double* x, double* first_move, int* status,
                      int* iters, void* stream) {
    if (h == nullptr || in == nullptr || status == nullptr || batch < 0) return VSMPC_ERR_INVALID_ARG;
    if (batch > h->max_batch) return VSMPC_ERR_BATCH_TOO_LARGE;
    if (batch == 0) return VSMPC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    ON_DEVICE(h->device);
    const size_t B = size_t(batch);
    if (batch <= ZC_MAX) {
        // zero-copy path: the kernel reads the records from and writes the results to pinned host memory
        const Stage hv = stage_view(h, h->h_stage), dv = stage_view(h, h->d_stage);   // the same carve-up on both views
        memcpy(hv.in, in, B * h->n_in * sizeof(double));
        HIP_TRY(launch_solve(h->variant, h->form, h->dev, dv.in, batch, dv.x, dv.fm, dv.st, dv.it, nullptr, nullptr, nullptr, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (x) memcpy(x, hv.x, B * h->n_var * sizeof(double));
        if (first_move) memcpy(first_move, hv.fm, B * VSMPC_FM_SIZE * sizeof(double));
        memcpy(status, hv.st, B * sizeof(int));
        if (iters) memcpy(iters, hv.it, B * sizeof(int));
        return VSMPC_OK;
    }
    // Pinned output buffers (hipHostMalloc / vsmpc_alloc_host) are written by the kernel itself over PCIe (16 B per lane
    // posted writes): no device-to-host copies, no copy launches (batch 4096, all outputs: 1.01 ms against 1.11 ms with
    // one copy of the trajectories behind the last chunk; profiles/r02_v11_hostpath.json).
    auto device_view = [](const void* host) -> void* {
        if (host == nullptr) return nullptr;
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, host) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return at.type == hipMemoryTypeHost ? at.devicePointer : nullptr;
    };
    double* zx = static_cast<double*>(device_view(x));
    double* zfm = static_cast<double*>(device_view(first_move));
    int* zst = static_cast<int*>(device_view(status));
    int* zit = static_cast<int*>(device_view(iters));
    const bool direct = (x == nullptr || zx) && (first_move == nullptr || zfm) && zst && (iters == nullptr || zit);
    // chunks of PIPE_CHUNK instances rotate over the handle's streams: upload(k+1) | solve(k) | download(k-1)
    // overlap when the caller's buffers are pinned (hipHostMalloc / vsmpc_alloc_host); with pageable buffers the
    // runtime stages the copies itself and the chunks still overlap with the kernels
    HIP_TRY(hipEventRecord(h->pipe_start, s));                 // work queued on the caller's stream comes first
    const int nstreams = std::min(PIPE_STREAMS, (batch + PIPE_CHUNK - 1) / PIPE_CHUNK);
    for (int i = 0; i < nstreams; ++i) HIP_TRY(hipStreamWaitEvent(h->pipe[i], h->pipe_start, 0));
    // A failure in the middle leaves earlier chunks queued: kernels that write straight into the caller's pinned buffers,
    // copies into h->d_in.  Nothing returns before every pipe stream that was used has drained.
    hipError_t err = hipSuccess;
    auto ok = [&](hipError_t e) { if (e != hipSuccess && err == hipSuccess) err = e; return err == hipSuccess; };
    int k = 0;
    for (int first = 0; first < batch && err == hipSuccess; first += PIPE_CHUNK, ++k) {
        const int n = std::min(PIPE_CHUNK, batch - first);
        const size_t o = size_t(first), N = size_t(n);
        hipStream_t ps = h->pipe[k % nstreams];
        if (!ok(hipMemcpyAsync(h->d_in + o * h->n_in, in + o * h->n_in, N * h->n_in * sizeof(double), hipMemcpyHostToDevice, ps))) break;
        if (direct) {
            ok(launch_solve(h->variant, h->form, h->dev, h->d_in + o * h->n_in, n, x ? zx + o * h->n_var : nullptr,
                            first_move ? zfm + o * VSMPC_FM_SIZE : nullptr, zst + o, iters ? zit + o : nullptr,
                            nullptr, nullptr, nullptr, ps));
            continue;
        }
        if (!ok(launch_solve(h->variant, h->form, h->dev, h->d_in + o * h->n_in, n, h->d_x + o * h->n_var,
                             h->d_fm + o * VSMPC_FM_SIZE, h->d_status + o, h->d_iters + o, nullptr, nullptr, nullptr, ps))) break;
        if (x) ok(hipMemcpyAsync(x + o * h->n_var, h->d_x + o * h->n_var, N * h->n_var * sizeof(double), hipMemcpyDeviceToHost, ps));
        if (first_move)
            ok(hipMemcpyAsync(first_move + o * VSMPC_FM_SIZE, h->d_fm + o * VSMPC_FM_SIZE,
                              N * VSMPC_FM_SIZE * sizeof(double), hipMemcpyDeviceToHost, ps));
        ok(hipMemcpyAsync(status + o, h->d_status + o, N * sizeof(int), hipMemcpyDeviceToHost, ps));
        if (iters) ok(hipMemcpyAsync(iters + o, h->d_iters + o, N * sizeof(int), hipMemcpyDeviceToHost, ps));
    }
    for (int i = 0; i < nstreams; ++i) {
        if (err == hipSuccess && ok(hipEventRecord(h->pipe_done[i], h->pipe[i])))
            ok(hipStreamWaitEvent(s, h->pipe_done[i], 0));   // the caller's stream continues after all of them
    }
    for (int i = 0; i < nstreams; ++i) {
        const hipError_t e = hipStreamSynchronize(h->pipe[i]);
        if (err == hipSuccess) err = e;
    }
    HIP_TRY(err);
    return VSMPC_OK;
}

int vsmpc_linearize_batch(vsmpc_handle* h, const double* in, int batch, double* A, double* Bj, double* Bt,
                          double* c, double* dt) {
    if (h == nullptr || in == nullptr || batch < 0) return VSMPC_ERR_INVALID_ARG;
    if (batch > h->max_batch) return VSMPC_ERR_BATCH_TOO_LARGE;
    if (dt) fill_dt(h->cfg, dt);
    if (batch == 0) return VSMPC_OK;
    ON_DEVICE(h->device);
    const size_t B = size_t(batch);
    double* dA = h->d_lin;
    double* dBj = dA + size_t(h->max_batch) * NX * NX;
    double* dBt = dBj + size_t(h->max_batch) * NX * NJ;
    double* dC = dBt + size_t(h->max_batch) * NX * NTH;
    HIP_TRY(hipMemcpy(h->d_in, in, B * h->n_in * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(launch_linearize(h->variant, h->dev, h->d_in, batch, dA, dBj, dBt, dC, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    if (A) HIP_TRY(hipMemcpy(A, dA, B * NX * NX * sizeof(double), hipMemcpyDeviceToHost));
    if (Bj) HIP_TRY(hipMemcpy(Bj, dBj, B * NX * NJ * sizeof(double), hipMemcpyDeviceToHost));
    if (Bt) HIP_TRY(hipMemcpy(Bt, dBt, B * NX * NTH * sizeof(double), hipMemcpyDeviceToHost));
    if (c) HIP_TRY(hipMemcpy(c, dC, B * NX * sizeof(double), hipMemcpyDeviceToHost));
    return VSMPC_OK;
}

// Stacks the reference-ordered dense QP from the DEVICE linearisation; layout bookkeeping only
// (IMPCProblem.cpp:150-194; cost order variableSamplingMPC.cpp:70-76, row order :77-84).
int vsmpc_assemble_dense(vsmpc_handle* h, const double* in_one, double* H, double* g, double* Ac, double* lo,
                         double* hi) {
    if (h == nullptr || in_one == nullptr || H == nullptr || g == nullptr || Ac == nullptr || lo == nullptr ||
        hi == nullptr)
        return VSMPC_ERR_INVALID_ARG;
    std::vector<double> A(NX * NX), Bj(NX * NJ), Bt(NX * NTH), c(NX), dt(MAX_STAGES);
    int rc = vsmpc_linearize_batch(h, in_one, 1, A.data(), Bj.data(), Bt.data(), c.data(), dt.data());
    if (rc != VSMPC_OK) return rc;
    const vsmpc_config& cf = h->cfg;
    const int N = cf.n_iter, nS = cf.n_iter_small, Hc = cf.control_horizon;
    const int nvar = h->n_var, ncon = h->n_con;
    const int offJ = NX * (N + 1), offV = offJ + NJ * Hc, nvb = Hc - nS + 1;
    memset(H, 0, sizeof(double) * size_t(nvar) * nvar);
    memset(g, 0, sizeof(double) * nvar);
    memset(Ac, 0, sizeof(double) * size_t(ncon) * nvar);
    memset(lo, 0, sizeof(double) * ncon);
    memset(hi, 0, sizeof(double) * ncon);
    auto Hat = [&](int r, int cc) -> double& { return H[size_t(r) * nvar + cc]; };
    auto Aat = [&](int r, int cc) -> double& { return Ac[size_t(r) * nvar + cc]; };
    double q[NX] = {0};
    for (int i = 0; i < 3; ++i) {
        q[i] = cf.w_com_pos[i]; q[3 + i] = cf.w_lin_mom[i]; q[6 + i] = cf.w_rpy[i];
        q[9 + i] = cf.w_ang_mom[i]; q[20 + i] = cf.w_com_pos_err[i]; q[23 + i] = cf.w_rpy_err[i];
    }
    // ReferenceTrackingCost (costsVSMPC.cpp:166-178)
    for (int i = 1; i <= N; ++i) {
        const int col = (i - 1) < nS ? 0 : (i - 1) - nS;
        for (int r = 0; r < NX; ++r) {
            Hat(i * NX + r, i * NX + r) += q[r];
            if (r < 12) g[i * NX + r] += -q[r] * in_one[VSMPC_IN_XREF + col * 12 + r];
        }
    }
    // RegualarizationCost (costsVSMPC.cpp:375-409)
    for (int i = 0; i < Hc; ++i)
        for (int r = 0; r < NJ; ++r) Hat(offJ + i * NJ + r, offJ + i * NJ + r) += cf.w_delta_joint[r];
    for (int i = 0; i < Hc - nS; ++i)
        for (int r = 0; r < NTH; ++r) {
            const int a = offV + i * NTH + r, b = offV + (i + 1) * NTH + r;
            Hat(a, a) += cf.w_throttle; Hat(b, a) -= cf.w_throttle;
            Hat(a, b) -= cf.w_throttle; Hat(b, b) += cf.w_throttle;
        }
    // ThrottleInitialValueCost (costsVSMPC.cpp:468-487)
    double vprev[NTH];
    for (int r = 0; r < NTH; ++r) {
        vprev[r] = Jet::v_of_throttle_div(in_one[VSMPC_IN_UPREV + r]);
        Hat(offV + r, offV + r) += cf.w_initial_throttle;
        g[offV + r] += -cf.w_initial_throttle * vprev[r];
    }
    // JointPositionRegularizationCost (costsVSMPC.cpp:558-592)
    for (int i = 0; i < Hc; ++i)
        for (int r = 0; r < NJ; ++r) {
            Hat(offJ + i * NJ + r, offJ + i * NJ + r) += cf.w_reg_joint_pos;
            g[offJ + i * NJ + r] += cf.w_reg_joint_pos * in_one[VSMPC_IN_QERR + r];
        }
    // ConstraintSystemDynamicVS (constraintsVSMPC.cpp:76-131)
    for (int i = 0; i < N; ++i) {
        const double d = dt[i];
        const int jb = i < Hc ? i : Hc - 1;
        const int tb = i < nS ? 0 : (i < Hc ? i - (nS - 1) : Hc - nS);
        for (int r = 0; r < NX; ++r) {
            for (int cc = 0; cc < NX; ++cc) Aat(i * NX + r, i * NX + cc) = (r == cc ? 1.0 : 0.0) + d * A[r * NX + cc];
            Aat(i * NX + r, (i + 1) * NX + r) = -1.0;
            for (int cc = 0; cc < NJ; ++cc) Aat(i * NX + r, offJ + jb * NJ + cc) = d * Bj[r * NJ + cc];
            for (int cc = 0; cc < NTH; ++cc) Aat(i * NX + r, offV + tb * NTH + cc) = d * Bt[r * NTH + cc];
            lo[i * NX + r] = -d * c[r];
            hi[i * NX + r] = -d * c[r];
        }
    }
    // ConstraintInitialState (IQPUtilsMPC.cpp:71-92)
    const int r0 = N * NX;
    for (int r = 0; r < NX; ++r) {
        Aat(r0 + r, r) = 1.0;
        lo[r0 + r] = hi[r0 + r] = in_one[VSMPC_IN_X0 + r];
    }
    // ThrottleConstraint (constraintsVSMPC.cpp:338-365); trailing rows stay 0 in [0,0]
    const int r1 = r0 + NX;
    const bool hold = in_one[VSMPC_IN_HOLD] != 0.0;
    for (int i = 0; i < nvb; ++i)
        for (int r = 0; r < NTH; ++r) {
            Aat(r1 + i * NTH + r, offV + i * NTH + r) = 1.0;
            if (hold && i == 0) {
                lo[r1 + r] = hi[r1 + r] = vprev[r];
            } else {
                lo[r1 + i * NTH + r] = h->dev.vmin;
                hi[r1 + i * NTH + r] = h->dev.vmax;
            }
        }
    return VSMPC_OK;
}

int vsmpc_debug_condensed(vsmpc_handle* h, const double* in_one, double* M, double* Lfac) {
    if (h == nullptr || in_one == nullptr) return VSMPC_ERR_INVALID_ARG;
    ON_DEVICE(h->device);
    const size_t np2 = size_t(h->n_p) * h->n_p;
    HIP_TRY(hipMemcpy(h->d_in, in_one, h->n_in * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(h->d_dbg, 0, 2 * np2 * sizeof(double)));   // the kernel writes the lower triangles only
    HIP_TRY(launch_solve(h->variant, h->form, h->dev, h->d_in, 1, h->d_x, h->d_fm, h->d_status, h->d_iters, h->d_dbg,
                         h->d_dbg + np2, nullptr, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    if (M) HIP_TRY(hipMemcpy(M, h->d_dbg, np2 * sizeof(double), hipMemcpyDeviceToHost));
    if (Lfac) HIP_TRY(hipMemcpy(Lfac, h->d_dbg + np2, np2 * sizeof(double), hipMemcpyDeviceToHost));
    return VSMPC_OK;
}

int vsmpc_kinematics_batch(vsmpc_handle* h, const double* kin, int batch, double* out, double* records) {
    if (h == nullptr || kin == nullptr || out == nullptr || batch < 0) return VSMPC_ERR_INVALID_ARG;
    if (batch > h->max_batch) return VSMPC_ERR_BATCH_TOO_LARGE;
    if (batch == 0) return VSMPC_OK;
    ON_DEVICE(h->device);
    HIP_TRY(hipMemcpy(h->d_kin, kin, size_t(batch) * VSMPC_KIN_SIZE * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(launch_kinematics(h->d_kin, batch, h->d_kout, h->kin, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, h->d_kout, size_t(batch) * VSMPC_KIN_OUT * sizeof(double), hipMemcpyDeviceToHost));
    if (records != nullptr) {  // patch the three fields of the input records (host side, layout bookkeeping only)
        for (int b = 0; b < batch; ++b) {
            double* rec = records + size_t(b) * h->n_in;
            const double* o = out + size_t(b) * VSMPC_KIN_OUT;
            memcpy(rec + VSMPC_IN_LLIN, o, 24 * sizeof(double));
            memcpy(rec + VSMPC_IN_LANG, o + 24, 24 * sizeof(double));
            memcpy(rec + VSMPC_IN_INERTIA, o + 48, 9 * sizeof(double));
        }
    }
    return VSMPC_OK;
}

// One tick of the reference's drop-in surface in ONE submission: kinematics terms -> record -> solve, one synchronisation.
int vsmpc_tick(vsmpc_handle* h, const double* kin, double* in, int batch, double* x, double* first_move, int* status,
               int* iters, void* stream) {
    if (h == nullptr || kin == nullptr || in == nullptr || status == nullptr || batch < 0) return VSMPC_ERR_INVALID_ARG;
    if (batch > h->max_batch) return VSMPC_ERR_BATCH_TOO_LARGE;
    if (batch == 0) return VSMPC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    ON_DEVICE(h->device);
    const size_t B = size_t(batch);
    if (batch <= ZC_MAX) {
        // the kinematics kernel reads the raw Robot quantities from, and writes LLIN | LANG | INERTIA into, the mapped
        // staging buffer; the solve kernel (next in stream order) reads the completed record from there
        const Stage hv = stage_view(h, h->h_stage), dv = stage_view(h, h->d_stage);
        memcpy(hv.kin, kin, B * VSMPC_KIN_SIZE * sizeof(double));
        memcpy(hv.in, in, B * h->n_in * sizeof(double));
        HIP_TRY(launch_kinematics_patch(dv.kin, batch, dv.in, h->n_in, h->kin, s));
        HIP_TRY(launch_solve(h->variant, h->form, h->dev, dv.in, batch, dv.x, dv.fm, dv.st, dv.it, nullptr, nullptr, nullptr, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (size_t b = 0; b < B; ++b)   // hand the completed fields back (the caller's record is the record of the tick)
            memcpy(in + b * h->n_in + VSMPC_IN_LLIN, hv.in + b * h->n_in + VSMPC_IN_LLIN, (24 + 24 + 9) * sizeof(double));
        if (x) memcpy(x, hv.x, B * h->n_var * sizeof(double));
        if (first_move) memcpy(first_move, hv.fm, B * VSMPC_FM_SIZE * sizeof(double));
        memcpy(status, hv.st, B * sizeof(int));
        if (iters) memcpy(iters, hv.it, B * sizeof(int));
        return VSMPC_OK;
    }
    HIP_TRY(hipMemcpyAsync(h->d_kin, kin, B * VSMPC_KIN_SIZE * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->d_in, in, B * h->n_in * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(launch_kinematics_patch(h->d_kin, batch, h->d_in, h->n_in, h->kin, s));
    HIP_TRY(launch_solve(h->variant, h->form, h->dev, h->d_in, batch, h->d_x, h->d_fm, h->d_status, h->d_iters, nullptr, nullptr,
                         nullptr, s));
    HIP_TRY(hipMemcpyAsync(in, h->d_in, B * h->n_in * sizeof(double), hipMemcpyDeviceToHost, s));
    if (x) HIP_TRY(hipMemcpyAsync(x, h->d_x, B * h->n_var * sizeof(double), hipMemcpyDeviceToHost, s));
    if (first_move) HIP_TRY(hipMemcpyAsync(first_move, h->d_fm, B * VSMPC_FM_SIZE * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(status, h->d_status, B * sizeof(int), hipMemcpyDeviceToHost, s));
    if (iters) HIP_TRY(hipMemcpyAsync(iters, h->d_iters, B * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return VSMPC_OK;
}

int vsmpc_provider_batch(vsmpc_handle* h, const vsmpc_tree* tree, const double* state, int batch, double* kin,
                         double* robot, double* records) {
    if (h == nullptr || tree == nullptr || state == nullptr || batch < 0) return VSMPC_ERR_INVALID_ARG;
    if (batch > h->max_batch) return VSMPC_ERR_BATCH_TOO_LARGE;
    if (batch == 0) return VSMPC_OK;
    // the provider delivers the CURRENT frame Jacobians; jointsLambdaOption "constant" re-reads those slots as the
    // configure-time relative Jacobians and thrusts (vsmpc_set_kinematics_options): the combination has no meaning
    if (records != nullptr && h->kin.constant_lambda) return VSMPC_ERR_UNSUPPORTED_CONFIG;
    if (tree->parent[0] != -1) return VSMPC_ERR_INVALID_ARG;
    for (int b = 1; b < VSMPC_TREE_NB; ++b)
        if (tree->parent[b] < 0 || tree->parent[b] >= b) return VSMPC_ERR_INVALID_ARG;      // parents precede children
    for (int j = 0; j < VSMPC_TREE_NJ; ++j)
        if (tree->robot_joint[j] < 0 || tree->robot_joint[j] >= VSMPC_KIN_NJ) return VSMPC_ERR_INVALID_ARG;
    for (int i = 0; i < VSMPC_N_THRUSTS; ++i)
        if (tree->jet_body[i] < 0 || tree->jet_body[i] >= VSMPC_TREE_NB) return VSMPC_ERR_INVALID_ARG;
    ON_DEVICE(h->device);
    // scratch: the state records go through d_lin (1014 doubles per instance), the Robot-level outputs through d_x
    // (n_var >= 67 doubles per instance), the kinematics record through d_kin
    double* d_state = h->d_lin;
    double* d_robot = h->d_x;   // n_var >= 67 doubles per instance
    HIP_TRY(hipMemcpy(d_state, state, size_t(batch) * VSMPC_RS_SIZE * sizeof(double), hipMemcpyHostToDevice));
    if (records != nullptr)
        HIP_TRY(hipMemcpy(h->d_in, records, size_t(batch) * h->n_in * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(launch_provider(*tree, d_state, batch, h->d_kin, d_robot, records != nullptr ? h->d_in : nullptr, h->n_in, nullptr));
    if (records != nullptr) HIP_TRY(launch_kinematics_patch(h->d_kin, batch, h->d_in, h->n_in, h->kin, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    if (kin != nullptr)
        HIP_TRY(hipMemcpy(kin, h->d_kin, size_t(batch) * VSMPC_KIN_SIZE * sizeof(double), hipMemcpyDeviceToHost));
    if (robot != nullptr)
        HIP_TRY(hipMemcpy(robot, d_robot, size_t(batch) * VSMPC_RO_SIZE * sizeof(double), hipMemcpyDeviceToHost));
    if (records != nullptr)
        HIP_TRY(hipMemcpy(records, h->d_in, size_t(batch) * h->n_in * sizeof(double), hipMemcpyDeviceToHost));
    return VSMPC_OK;
}

int vsmpc_set_kinematics_options(vsmpc_handle* h, const int* joint_selector, int constant_lambda) {
    if (h == nullptr) return VSMPC_ERR_INVALID_ARG;
    if (joint_selector != nullptr) {
        for (int i = 0; i < VSMPC_N_JOINTS; ++i)
            if (joint_selector[i] < 0 || joint_selector[i] >= VSMPC_KIN_NJ) return VSMPC_ERR_INVALID_ARG;
        for (int i = 0; i < VSMPC_N_JOINTS; ++i) h->kin.sel[i] = joint_selector[i];
    }
    h->kin.constant_lambda = constant_lambda ? 1 : 0;
    return VSMPC_OK;
}

int vsmpc_debug_phase_cycles(vsmpc_handle* h, const double* in, int batch, unsigned long long* stamps16) {
    if (h == nullptr || in == nullptr || stamps16 == nullptr || batch <= 0) return VSMPC_ERR_INVALID_ARG;
    if (batch > h->max_batch) return VSMPC_ERR_BATCH_TOO_LARGE;
    ON_DEVICE(h->device);
    unsigned long long* d_st = h->d_stamps;
    HIP_TRY(hipMemset(d_st, 0, size_t(batch) * 16 * sizeof(unsigned long long)));
    HIP_TRY(hipMemcpy(h->d_in, in, size_t(batch) * h->n_in * sizeof(double), hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; ++rep)  // warm instruction caches, keep the last run
        HIP_TRY(launch_solve(h->variant, h->form, h->dev, h->d_in, batch, h->d_x, h->d_fm, h->d_status, h->d_iters, nullptr,
                             nullptr, d_st, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(stamps16, d_st, size_t(batch) * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return VSMPC_OK;
}

int vsmpc_timing_begin(vsmpc_handle* h, void* stream) {
    if (h == nullptr) return VSMPC_ERR_INVALID_ARG;
    HIP_TRY(hipEventRecord(h->ev0, static_cast<hipStream_t>(stream)));
    return VSMPC_OK;
}

int vsmpc_timing_end(vsmpc_handle* h, void* stream, int launches, float* ms_per_launch) {
    if (h == nullptr || ms_per_launch == nullptr || launches <= 0) return VSMPC_ERR_INVALID_ARG;
    HIP_TRY(hipEventRecord(h->ev1, static_cast<hipStream_t>(stream)));
    HIP_TRY(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *ms_per_launch = ms / float(launches);
    return VSMPC_OK;
}

// ---- closed-loop rollout -------------------------------------------------------------------------------------

int vsmpc_rollout_create(vsmpc_handle* h, int batch, const double* traj_pos, const double* traj_vel, int n_traj,
                         const double* traj_alpha, int n_alpha, double alpha_dt, vsmpc_rollout** out) {
    if (h == nullptr || out == nullptr || traj_pos == nullptr || traj_vel == nullptr || traj_alpha == nullptr ||
        batch <= 0 || n_traj <= 0 || n_alpha <= 0 || !(alpha_dt > 0.0))
        return VSMPC_ERR_INVALID_ARG;
    *out = nullptr;
    if (batch > h->max_batch) return VSMPC_ERR_BATCH_TOO_LARGE;
    ON_DEVICE(h->device);
    vsmpc_rollout* r = new (std::nothrow) vsmpc_rollout();
    if (r == nullptr) return VSMPC_ERR_ALLOC;
    memset(r, 0, sizeof(*r));
    r->h = h;
    r->batch = batch;
    r->rd.n_in = h->n_in;
    r->rd.n_ref = h->cfg.n_iter - h->cfg.n_iter_small + 1;
    r->rd.ratio = int(std::lround(h->cfg.period_large / h->cfg.period_small));   // constraintsVSMPC.cpp:322
    r->rd.n_traj = n_traj;
    r->rd.n_alpha = n_alpha;
    r->rd.period_mpc = h->cfg.period_mpc;
    r->rd.alpha_dt = alpha_dt;
    r->rd.n_ts = 12 * r->rd.n_ref + 8;
    {   // TrajectoryManager::configure(.., 1 / periodMPC): des_fps truncated to int (systemDynamicsVSMPC.cpp:272), integer
        // up-sampling factor against the track's own rate
        const int des_fps = int(1.0 / h->cfg.period_mpc + 1e-9), fps = int(std::lround(1.0 / alpha_dt));
        if (fps <= 0 || des_fps < fps || des_fps % fps != 0) { delete r; return VSMPC_ERR_INVALID_ARG; }
        r->rd.alpha_up = des_fps / fps;
    }
    r->substeps = std::min(16, std::max(1, int(std::lround(h->cfg.period_mpc / 1e-3))));  // 1 kHz plant, as the MuJoCo harness
    const size_t B = size_t(batch);
    hipError_t e = hipMalloc(&r->d_state, B * VSMPC_PLANT_STATE * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&r->d_params, B * VSMPC_PLANT_PARAMS * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&r->d_tick, B * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&r->d_tpos, size_t(n_traj) * 3 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&r->d_tvel, size_t(n_traj) * 3 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&r->d_talpha, size_t(n_alpha) * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(r->d_tpos, traj_pos, size_t(n_traj) * 3 * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(r->d_tvel, traj_vel, size_t(n_traj) * 3 * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(r->d_talpha, traj_alpha, size_t(n_alpha) * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(r->d_tick, 0, B * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&r->d_ctl, sizeof(RolloutCtl));
    if (e == hipSuccess) e = hipMalloc(&r->d_tstate, B * r->rd.n_ts * sizeof(double));
    if (e == hipSuccess) e = hipMemset(r->d_tstate, 0, B * r->rd.n_ts * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&r->d_rec, B * h->n_in * sizeof(double));
    if (e == hipSuccess) e = hipMemset(r->d_rec, 0, B * h->n_in * sizeof(double));
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&r->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        vsmpc_rollout_destroy(r);
        return e == hipErrorOutOfMemory ? VSMPC_ERR_ALLOC : hip_fail(e, "vsmpc_rollout_create");
    }
    *out = r;
    return VSMPC_OK;
}

void vsmpc_rollout_destroy(vsmpc_rollout* r) {
    if (r == nullptr) return;
    DeviceScope scope(r->h->device);
    if (r->d_state) (void)hipFree(r->d_state);
    if (r->d_params) (void)hipFree(r->d_params);
    if (r->d_tick) (void)hipFree(r->d_tick);
    if (r->d_tpos) (void)hipFree(r->d_tpos);
    if (r->d_tvel) (void)hipFree(r->d_tvel);
    if (r->d_talpha) (void)hipFree(r->d_talpha);
    if (r->d_trpy) (void)hipFree(r->d_trpy);
    if (r->d_trpyd) (void)hipFree(r->d_trpyd);
    if (r->d_log) (void)hipFree(r->d_log);
    if (r->d_ctl) (void)hipFree(r->d_ctl);
    if (r->d_rec) (void)hipFree(r->d_rec);
    if (r->d_tstate) (void)hipFree(r->d_tstate);
    if (r->d_rs) (void)hipFree(r->d_rs);
    if (r->d_ro) (void)hipFree(r->d_ro);
    if (r->gexec) (void)hipGraphExecDestroy(r->gexec);
    if (r->own_stream) (void)hipStreamDestroy(r->own_stream);
    delete r;
}

int vsmpc_rollout_reset(vsmpc_rollout* r, const double* state, const double* params) {
    if (r == nullptr || state == nullptr || params == nullptr) return VSMPC_ERR_INVALID_ARG;
    ON_DEVICE(r->h->device);
    const size_t B = size_t(r->batch);
    HIP_TRY(hipMemcpy(r->d_state, state, B * VSMPC_PLANT_STATE * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(r->d_params, params, B * VSMPC_PLANT_PARAMS * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(r->d_tick, 0, B * sizeof(int)));
    r->ticks_done = 0;
    // record of tick 0; from here on every tick's advance kernel leaves the record of the following tick
    r->valid = 0;
    if (r->use_tree) HIP_TRY(enqueue_tree(r, nullptr, nullptr, nullptr));
    HIP_TRY(launch_record(r->rd, r->batch, r->d_state, r->d_params, r->d_tick, r->d_tpos, r->d_tvel, r->d_talpha,
                          r->d_tstate, r->d_rec, nullptr));
    if (r->use_tree) HIP_TRY(enqueue_tree_lambda(r, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    r->valid = 1;
    return VSMPC_OK;
}

int vsmpc_rollout_set_attitude_tracks(vsmpc_rollout* r, const double* traj_rpy, const double* traj_rpy_dot) {
    if (r == nullptr) return VSMPC_ERR_INVALID_ARG;
    ON_DEVICE(r->h->device);
    const size_t bytes = size_t(r->rd.n_traj) * 3 * sizeof(double);
    // whatever happens below, the captured ticks and the record of the next tick refer to the old tracks: drop them first
    if (r->gexec) { (void)hipGraphExecDestroy(r->gexec); r->gexec = nullptr; }
    r->graph_state = 0;
    r->valid = 0;                                                                // vsmpc_rollout_reset before the next run
    auto set = [&](double*& dst, const double* src) -> hipError_t {
        if (src == nullptr) {
            if (dst) (void)hipFree(dst);
            dst = nullptr;
            return hipSuccess;
        }
        if (dst == nullptr) {
            hipError_t e = hipMalloc(&dst, bytes);
            if (e != hipSuccess) { dst = nullptr; return e; }
        }
        return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
    };
    const hipError_t e0 = set(r->d_trpy, traj_rpy);
    r->rd.traj_rpy = r->d_trpy;              // (a failed set leaves either the old, still valid buffer or nullptr)
    HIP_TRY(e0);
    const hipError_t e1 = set(r->d_trpyd, traj_rpy_dot);
    r->rd.traj_rpyd = r->d_trpyd;
    HIP_TRY(e1);
    return VSMPC_OK;
}

int vsmpc_rollout_set_tree(vsmpc_rollout* r, const vsmpc_tree* tree) {
    if (r == nullptr) return VSMPC_ERR_INVALID_ARG;
    ON_DEVICE(r->h->device);
    if (r->gexec) { (void)hipGraphExecDestroy(r->gexec); r->gexec = nullptr; }   // the captured ticks have other launches
    r->graph_state = 0;
    r->valid = 0;                                                                // vsmpc_rollout_reset before the next run
    r->use_tree = 0;
    r->rd.tree = 0;
    if (tree == nullptr) return VSMPC_OK;
    if (tree->parent[0] != -1) return VSMPC_ERR_INVALID_ARG;
    for (int b = 1; b < VSMPC_TREE_NB; ++b)
        if (tree->parent[b] < 0 || tree->parent[b] >= b) return VSMPC_ERR_INVALID_ARG;
    for (int j = 0; j < VSMPC_TREE_NJ; ++j)
        if (tree->robot_joint[j] < 0 || tree->robot_joint[j] >= VSMPC_KIN_NJ) return VSMPC_ERR_INVALID_ARG;
    for (int i = 0; i < VSMPC_N_THRUSTS; ++i)
        if (tree->jet_body[i] < 0 || tree->jet_body[i] >= VSMPC_TREE_NB) return VSMPC_ERR_INVALID_ARG;
    const size_t B = size_t(r->batch);
    if (r->d_rs == nullptr) HIP_TRY(hipMalloc(&r->d_rs, B * VSMPC_RS_SIZE * sizeof(double)));
    if (r->d_ro == nullptr) HIP_TRY(hipMalloc(&r->d_ro, B * VSMPC_RO_SIZE * sizeof(double)));
    r->tree = *tree;
    r->use_tree = 1;
    r->rd.tree = 1;
    r->rd.tree_ro = r->d_ro;
    r->rd.tree_kout = r->h->d_kout;
    r->rd.tree_kin = r->h->d_kin;
    return VSMPC_OK;
}

// include/vsmpc_jet.h
int vsmpc_rollout_set_jet_plant(vsmpc_rollout* r, vsmpc_jet* j, const double* Q, const double* R) {
    if (r == nullptr || (j != nullptr && (Q == nullptr || R == nullptr))) return VSMPC_ERR_INVALID_ARG;
    RolloutDev rd = r->rd;
    rd.jet_nn = 0;
    rd.jet_w = nullptr;
    if (j != nullptr) {
        int dev = -1;
        jet_plant_view(j, &rd.jet_w, &rd.jet_hidden, rd.jet_norm, &dev);
        if (dev != r->h->device) return VSMPC_ERR_INVALID_ARG;
        rd.jet_nn = 1;
        for (int k = 0; k < 4; ++k) { rd.ekf_q[k] = Q[k]; rd.ekf_r[k] = R[k]; }
    }
    r->rd = rd;
    // the captured tick graph holds the old launch arguments, and the record of the next tick was built from the other
    // set of measurements: rebuild both
    if (r->gexec) { (void)hipGraphExecDestroy(r->gexec); r->gexec = nullptr; }
    r->graph_state = 0;
    r->valid = 0;                         // vsmpc_rollout_reset before the next run
    return VSMPC_OK;
}

}  // extern "C"

namespace {

constexpr int GRAPH_TICKS = 25;  // ticks per captured graph (50 kernel nodes)

// one closed-loop tick: two launches on `s` (solve, advance + next record), the stream order is the only
// synchronisation the loop needs
hipError_t enqueue_tick(vsmpc_rollout* r, hipStream_t s) {
    vsmpc_handle* h = r->h;
    hipError_t e = launch_solve(h->variant, h->form, h->dev, r->d_rec, r->batch, h->d_x, h->d_fm, h->d_status, h->d_iters, nullptr,
                                nullptr, nullptr, s);
    if (e == hipSuccess && r->use_tree) e = enqueue_tree(r, h->d_fm, h->d_status, s);   // A_mom, I_B of the joints after the move
    if (e == hipSuccess)
        e = launch_advance(r->rd, r->batch, r->d_state, r->d_params, r->d_tick, h->d_fm, h->d_status, h->d_iters,
                           r->d_talpha, r->d_ctl, r->substeps, r->d_tpos, r->d_tvel, r->d_tstate, r->d_rec, s);
    if (e == hipSuccess && r->use_tree) e = enqueue_tree_lambda(r, s);
    return e;
}

// Captures GRAPH_TICKS ticks into a graph (every launch argument is tick-invariant: tick counters, log destination and
// tick base live in device memory).  Launch-bound loop -> one graph launch per chunk instead of 50 kernel launches.
void build_tick_graph(vsmpc_rollout* r, hipStream_t s) {
    r->graph_state = -1;
    r->graph_form = r->h->form;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return; }
    hipError_t e = hipSuccess;
    for (int t = 0; t < GRAPH_TICKS && e == hipSuccess; ++t) e = enqueue_tick(r, s);
    hipGraph_t graph = nullptr;
    const hipError_t e2 = hipStreamEndCapture(s, &graph);
    if (e == hipSuccess && e2 == hipSuccess && graph != nullptr &&
        hipGraphInstantiate(&r->gexec, graph, nullptr, nullptr, 0) == hipSuccess)
        r->graph_state = 1;
    else
        (void)hipGetLastError();
    if (graph) (void)hipGraphDestroy(graph);
}

}  // namespace

extern "C" {

int vsmpc_rollout_run(vsmpc_rollout* r, int ticks, double* log, void* stream) {
    if (r == nullptr || ticks < 0) return VSMPC_ERR_INVALID_ARG;
    if (!r->valid) return VSMPC_ERR_INVALID_ARG;   // never reset, or a previous run failed half-way: reset() first
    if (ticks == 0) return VSMPC_OK;
    vsmpc_handle* h = r->h;
    ON_DEVICE(h->device);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : r->own_stream;
    const size_t row = size_t(r->batch) * VSMPC_ROLLOUT_LOG;
    if (log != nullptr && r->log_ticks < ticks) {
        if (r->d_log) (void)hipFree(r->d_log);
        r->d_log = nullptr;
        r->log_ticks = 0;
        hipError_t e = hipMalloc(&r->d_log, size_t(ticks) * row * sizeof(double));
        if (e != hipSuccess) return e == hipErrorOutOfMemory ? VSMPC_ERR_ALLOC : hip_fail(e, "vsmpc_rollout_run");
        r->log_ticks = ticks;
    }
    const RolloutCtl ctl = {log ? r->d_log : nullptr, r->ticks_done, log ? ticks : 0};
    HIP_TRY(hipMemcpyAsync(r->d_ctl, &ctl, sizeof(ctl), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));  // `ctl` lives on this stack frame
    r->valid = 0;                       // until the whole run has completed: a failure below leaves the counters ahead
    int t = 0;
    if (r->graph_state != 0 && r->graph_form != h->form) {   // the captured launches are of the other condensing form
        if (r->gexec) { (void)hipGraphExecDestroy(r->gexec); r->gexec = nullptr; }
        r->graph_state = 0;
    }
    if (ticks >= GRAPH_TICKS && r->graph_state == 0) build_tick_graph(r, s);
    if (r->graph_state == 1)
        for (; ticks - t >= GRAPH_TICKS; t += GRAPH_TICKS) HIP_TRY(hipGraphLaunch(r->gexec, s));
    for (; t < ticks; ++t) HIP_TRY(enqueue_tick(r, s));
    if (log) HIP_TRY(hipMemcpyAsync(log, r->d_log, size_t(ticks) * row * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    r->ticks_done += ticks;
    r->valid = 1;
    return VSMPC_OK;
}

int vsmpc_rollout_get_state(vsmpc_rollout* r, double* state) {
    if (r == nullptr || state == nullptr) return VSMPC_ERR_INVALID_ARG;
    ON_DEVICE(r->h->device);
    HIP_TRY(hipMemcpy(state, r->d_state, size_t(r->batch) * VSMPC_PLANT_STATE * sizeof(double), hipMemcpyDeviceToHost));
    return VSMPC_OK;
}

int vsmpc_rollout_get_records(vsmpc_rollout* r, double* records) {
    if (r == nullptr || records == nullptr) return VSMPC_ERR_INVALID_ARG;
    ON_DEVICE(r->h->device);
    HIP_TRY(hipMemcpy(records, r->d_rec, size_t(r->batch) * r->h->n_in * sizeof(double), hipMemcpyDeviceToHost));
    return VSMPC_OK;
}

int vsmpc_set_kernel_form(vsmpc_handle* h, int form) {
    if (h == nullptr || form < 0 || form > 2) return VSMPC_ERR_INVALID_ARG;
    if (form == 1 && !variant_has_structured(h->variant)) return VSMPC_ERR_UNSUPPORTED_CONFIG;
    const int prev = h->form;
    h->form = form;
    return prev;
}

void* vsmpc_alloc_host(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

void vsmpc_free_host(void* p) {
    if (p != nullptr) (void)hipHostFree(p);
}

const char* vsmpc_strerror(int code) {
    switch (code) {
        case VSMPC_OK: return "ok";
        case VSMPC_ERR_INVALID_ARG: return "invalid argument";
        case VSMPC_ERR_UNSUPPORTED_CONFIG: return "unsupported MPC configuration (no kernel instantiation)";
        case VSMPC_ERR_BATCH_TOO_LARGE: return "batch exceeds max_batch of the handle";
        case VSMPC_ERR_HIP: return g_hip_msg[0] ? g_hip_msg : "HIP runtime error";
        case VSMPC_ERR_ALLOC: return "allocation failed";
        default: return "unknown error";
    }
}

}  // extern "C"
