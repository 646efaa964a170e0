// qg_capi.hip -- host side of the C ABI declared in include/quadgym.h.
//
// Owns the per-env device state (struct-of-arrays in HBM), converts the double-precision
// model/task descriptions into the kernarg-sized single-precision tables the kernels read,
// and launches the kernels of qg_kernels.hip.  There is no CPU code path: every compute
// entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/quadgym.h"
#include "../../include/qg_model_data.h"
// the kernels are compiled in the same translation unit (one code object, no -fgpu-rdc)
#include "qg_kernels.hip"
#include "qg_kernel_link.hip"
#include "qg_kernel_resident.hip"
#include "qg_walk.hip"
#include "qg_po.hip"
#include "qg_tables.h"

struct qg_sim {
    int32_t n;
    int32_t device;
    int32_t obs_dim;
    qg_model model;
    qg_task task;
    KModel *d_model;
    KTask *d_task;
    KState st;
    // staging for the host-pointer entry points
    float *d_actions, *d_obs, *d_reward, *d_comps, *d_stage;
    int32_t caller_inflight;  // a device-pointer step has been enqueued on a caller's stream since the last device-wide wait
    int32_t captured_once;    // a device-pointer step of this handle has been CAPTURED into a hipGraph: replays enqueue steps the library
                              // never sees, so from then on every host-pointer call takes the device-wide wait (sticky)
    uint8_t *h_pin;           // page-locked staging of the host-pointer entry points (see pin_reserve)
    size_t h_pin_cap;
    uint8_t *d_done, *d_mask;
    hipStream_t stream;       // the library's own stream (host-pointer calls, timing)
    hipEvent_t ev0, ev1;
    uint64_t seed;
    uint64_t env_index_base;
    int32_t track_ctrl;
    int32_t link_helpers;     // walking forms of the one-link-per-lane kernel run with helper waves (QG_LINK_HELPERS at qg_create; default 1)
    int32_t baked;            // 1: the model equals the compiled-in default, the literal-constant kernel variant runs
    int32_t mapping;          // QG_MAP_AUTO / QG_MAP_LANE / QG_MAP_QUAD (request)
    int32_t creating;
    int32_t walk_bound;       // qg_walk layers bound to this handle (qg_set_task refuses while > 0)
    int32_t po_unfused;       // env QG_PO_UNFUSED=1: keep the observation pack of qg_po_step a launch of its own (A/B, parity test)
    int32_t simds;            // SIMDs of the handle's GPU (hipDeviceProp: compute units x 4; 1024 on an MI355X): AUTO's thresholds are
                              // "one wave per SIMD" sizes
    // resident form of the one-link-per-lane step (qg_resident_*, qg_kernel_resident.hip)
    struct {
        int32_t active;       // qg_resident_start has set the mailbox up (the mode is on until qg_resident_stop)
        int32_t launched;     // a resident launch has been enqueued and has not been waited for since
        KResident k;          // mailbox pointers, slots, time-outs
        void *d_mail;         // door, arrival shards, completed counter (one allocation)
        volatile unsigned long long *hstat;   // page-locked host words the kernels report into
        hipStream_t ctl_stream;
        hipStream_t last_stream;              // where the latest ring went (waited for before the kernel is retired)
        int32_t own_buffers;                  // the action / output slots are the library's (else the caller's, qg_resident_start)
        int64_t rung;         // env-steps rung through the API since qg_resident_start
        uint64_t lost_seen, gaveup_seen;
    } res;
};
static int resident_retire(qg_sim *s);
static void resident_free(qg_sim *s);

static thread_local char g_err[512] = "";

int qg_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define fail qg_fail

#define HIP_TRY(expr, code)                                                                         \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return fail(code, "%s: %s", #expr, hipGetErrorString(e_));            \
    } while (0)

extern "C" const char *qg_version(void) { return "quadgym 0.1.0 (gfx950)"; }
#ifndef QG_SOURCE_HASH
#define QG_SOURCE_HASH "unknown"
#endif
extern "C" const char *qg_build_id(void) { return QG_SOURCE_HASH; }
extern "C" const char *qg_last_error(void) { return g_err; }

// The step time is a staircase in the batch size (profiles/r03/map_sweep.txt, microseconds per env-step on an MI355X): flat at 11.8 up to
// 4096 envs (one wave of the one-link-per-lane kernel per SIMD), 18.3-19.0 for 4097 .. 16 384 (one wave of the one-leg-per-lane kernel
// per SIMD), 24.5-25.3 for 16 385 .. 32 768 (one wave of the two-legs-per-lane kernel per SIMD), then ~23 us per further 32 768 envs.
// The top of a stair costs no more per step than its foot: this returns the top of the stair `n_envs` stands on.
extern "C" int32_t qg_recommended_batch(int32_t n_envs, int32_t device_id) {
    int simds = 1024;
    hipDeviceProp_t prop;
    if (device_id >= 0 && hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) simds = 4 * prop.multiProcessorCount;
    else (void)hipGetLastError();
    if (n_envs < 1) n_envs = 1;
    const int64_t link = (int64_t)simds * QGK_LINK_ENVS, quad = (int64_t)simds * QGK_QUAD_ENVS, pair = (int64_t)simds * QGK_PAIR_ENVS;
    int64_t r = n_envs <= link ? link : (n_envs <= quad ? quad : ((n_envs + pair - 1) / pair) * pair);
    return (int32_t)(r > INT32_MAX ? INT32_MAX : r);
}

extern "C" int qg_device_pci_bus_id(int32_t device_id, char *out, int32_t len) {
    if (!out || len < 16) return qg_fail(QG_ERR_ARG, "qg_device_pci_bus_id: need a buffer of at least 16 bytes");
    hipError_t e = hipDeviceGetPCIBusId(out, len, device_id);
    if (e != hipSuccess) return qg_fail(QG_ERR_DEVICE, "hipDeviceGetPCIBusId(%d): %s", device_id, hipGetErrorString(e));
    return QG_OK;
}

extern "C" int qg_default_model(qg_model *out) {
    if (!out) return fail(QG_ERR_ARG, "qg_default_model: null output");
    static const qg_model def = QG_MODEL_DEFAULT_INIT;
    *out = def;
    return QG_OK;
}

extern "C" int qg_default_task(qg_task *out) {
    if (!out) return fail(QG_ERR_ARG, "qg_default_task: null output");
    memset(out, 0, sizeof *out);
    out->frame_skip = 4;          // quadruped.py:44
    out->max_time = 10.0;         // quadruped.py:43
    out->use_time_limit = 1;      // quadruped.py:52
    out->use_fall = 0;
    out->fall_height = 0.2;       // README.md:87
    out->w_forward = 1.0;         // README.md:65-72
    out->w_ctrl = -0.1;
    out->alive_bonus = 1.0;
    out->obs_mode = QG_OBS_FULL;
    out->sensor_lag = 1;
    out->auto_reset = 0;
    out->reset_flags = 0;
    for (int i = 0; i < QG_NU; i++) out->default_ctrl[i] = (i % 3 == 2) ? -0.5 : 0.0;   // quadruped.py:124
    out->reset_joint_jitter = 0.1;
    return QG_OK;
}

extern "C" int64_t qg_time_limit_substeps(double timestep, double max_time) { return qg_time_limit_substeps_impl(timestep, max_time); }

extern "C" int qg_destroy(qg_sim *s) {
    if (!s) return QG_OK;
    (void)hipSetDevice(s->device);
    (void)resident_retire(s);
    (void)hipDeviceSynchronize();                  // steps may still be in flight on a caller's stream (the header's ordering contract)
    resident_free(s);
    void *ptrs[] = {s->d_model, s->d_task, s->st.qpos, s->st.qvel, s->st.act, s->st.ctrl, s->st.nstep, s->st.episode, s->d_actions,
                    s->d_obs,   s->d_reward, s->d_comps, s->d_stage, s->d_done, s->d_mask};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (s->h_pin) (void)hipHostFree(s->h_pin);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
    return QG_OK;
}

extern "C" int qg_create(int32_t n_envs, int32_t device_id, const qg_model *model, const qg_task *task, uint64_t env_index_base,
                         qg_sim **out) {
    if (!out) return fail(QG_ERR_ARG, "qg_create: null output");
    *out = nullptr;
    if (n_envs < 1) return fail(QG_ERR_ARG, "qg_create: n_envs must be >= 1");
    qg_model dm;
    qg_task dt;
    if (!model) { qg_default_model(&dm); model = &dm; }
    if (!task) { qg_default_task(&dt); task = &dt; }
    KModel km;
    KTask kt;
    int rc = build_tables(model, task, &km, &kt);
    if (rc != QG_OK) return rc;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(QG_ERR_DEVICE, "no HIP device is available; quadgym has no CPU backend");
    if (device_id < 0 || device_id >= ndev) return fail(QG_ERR_DEVICE, "device_id %d out of range (0..%d)", device_id, ndev - 1);
    HIP_TRY(hipSetDevice(device_id), QG_ERR_DEVICE);

    qg_sim *s = new (std::nothrow) qg_sim();
    if (!s) return fail(QG_ERR_ALLOC, "out of host memory");
    memset(s, 0, sizeof *s);
    s->n = n_envs;
    s->device = device_id;
    s->obs_dim = task->obs_mode == QG_OBS_IMU ? 21 : QG_NSENSOR;
    s->model = *model;
    s->task = *task;
    s->env_index_base = env_index_base;
    s->track_ctrl = 1;
    { const char *e = getenv("QG_LINK_HELPERS"); s->link_helpers = e ? (atoi(e) != 0) : 1; }
    s->mapping = QG_MAP_AUTO;
    {
        hipDeviceProp_t prop;
        s->simds = (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) ? 4 * prop.multiProcessorCount : 1024;
    }
    if (const char *e = getenv("QG_PO_UNFUSED")) s->po_unfused = atoi(e) != 0;
    {
        static const KModel baked = {QG_BAKED_FLOATS};
        s->baked = QG_BAKED_LEGS_IDENTICAL && memcmp(&km, &baked, sizeof km) == 0;
    }
    size_t n = (size_t)n_envs;
#define ALLOC(ptr, bytes)                                                                   \
    do {                                                                                    \
        hipError_t e_ = hipMalloc((void **)&(ptr), (bytes));                                \
        if (e_ != hipSuccess) {                                                             \
            qg_destroy(s);                                                                  \
            return fail(QG_ERR_ALLOC, "hipMalloc(%zu): %s", (size_t)(bytes), hipGetErrorString(e_)); \
        }                                                                                   \
    } while (0)
    ALLOC(s->d_model, sizeof(KModel));
    ALLOC(s->d_task, sizeof(KTask));
    ALLOC(s->st.qpos, n * QG_NQ * sizeof(float));
    ALLOC(s->st.qvel, n * QG_NV * sizeof(float));
    ALLOC(s->st.act, n * QG_NU * sizeof(float));
    ALLOC(s->st.ctrl, n * QG_NU * sizeof(float));
    ALLOC(s->st.nstep, n * sizeof(int32_t));
    ALLOC(s->st.episode, n * sizeof(int32_t));
    ALLOC(s->d_actions, n * QG_NU * sizeof(float));
    ALLOC(s->d_obs, n * (QG_NSENSOR + 2) * sizeof(float));
    ALLOC(s->d_reward, n * sizeof(float));
    ALLOC(s->d_comps, n * QG_NREWARD * sizeof(float));
    ALLOC(s->d_stage, n * (QG_NQ + QG_NV + 2 * QG_NU) * sizeof(float));     // all four state fields side by side (qg_get_state)
    ALLOC(s->d_done, n);
    ALLOC(s->d_mask, n);
#undef ALLOC
    hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&s->ev0);
    if (e == hipSuccess) e = hipEventCreate(&s->ev1);
    if (e == hipSuccess) e = hipMemcpy(s->d_model, &km, sizeof km, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(s->d_task, &kt, sizeof kt, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        qg_destroy(s);
        return fail(QG_ERR_DEVICE, "device setup: %s", hipGetErrorString(e));
    }
    if (hipMemset(s->st.episode, 0, n * sizeof(int32_t)) != hipSuccess) {
        qg_destroy(s);
        return fail(QG_ERR_DEVICE, "device setup: memset");
    }
    *out = s;
    s->creating = 1;                 // the constructor's own reset does not count as an episode
    rc = qg_reset(s, nullptr, 0, 0);
    s->creating = 0;
    if (rc != QG_OK) {
        qg_destroy(s);
        *out = nullptr;
    }
    return rc;
}

extern "C" int qg_num_envs(const qg_sim *s) { return s ? s->n : fail(QG_ERR_ARG, "null handle"); }
extern "C" int qg_obs_dim(const qg_sim *s) { return s ? s->obs_dim : fail(QG_ERR_ARG, "null handle"); }

extern "C" int qg_reset(qg_sim *s, const uint8_t *mask, uint64_t seed, uint32_t flags) {
    if (!s) return fail(QG_ERR_ARG, "null handle");
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    // steps may be in flight on a caller's stream (qg_step_device*): the reset runs on the library's own non-blocking stream
    // and must not overlap them (a resident step kernel first stores the state it holds in registers and leaves)
    { int rr = resident_retire(s); if (rr != QG_OK) return rr; }
    HIP_TRY(hipDeviceSynchronize(), QG_ERR_LAUNCH);
    // the seed keys the reset streams of EVERY env (auto-resets included): only a whole-batch reset may change it, a masked
    // reset draws from the streams already in force
    if (!mask) s->seed = seed;
    else seed = s->seed;
    const uint8_t *dmask = nullptr;
    if (mask) {
        HIP_TRY(hipMemcpyAsync(s->d_mask, mask, (size_t)s->n, hipMemcpyHostToDevice, s->stream), QG_ERR_DEVICE);
        dmask = s->d_mask;
    }
    int threads = 256, blocks = (s->n + threads - 1) / threads;
    hipLaunchKernelGGL(qg_reset_kernel, dim3(blocks), dim3(threads), 0, s->stream, s->d_model, s->d_task, s->st, s->n, dmask, seed,
                       s->env_index_base, flags, s->creating ? 0 : 1);
    HIP_TRY(hipGetLastError(), QG_ERR_LAUNCH);
    HIP_TRY(hipStreamSynchronize(s->stream), QG_ERR_LAUNCH);
    return QG_OK;
}

// AUTO = the measured optimum per batch size (profiles/r03/map_sweep.txt: every mapping around the boundaries on one box, HIP events,
// microseconds per launch at frame_skip 4):
//   envs        link     quad     pair
//   4 096       12.0     18.1              one link per lane: 1024 waves, one per SIMD
//   5 120       19.2     18.3              a second link wave per SIMD costs more than the quad kernel's idle SIMDs
//   16 384               19.0     24.2     one quad wave per SIMD
//   20 000               28.2     24.5     the quad grid needs a second wave on some SIMDs, the pair grid (32 envs per wave) does not
//   32 768               29.0     25.3
//   40 000               40.4     46.7     second round of pair waves (one wave per SIMD by construction) against two resident quad waves
//   57 344               51.9     47.5     from 1.75 rounds of pair waves on, pair is ahead again
//   262 144             187.7    187.6
// The one-env-per-lane kernel (66 us at 4096 envs) only runs on request.  The compiled-in robot runs the variants with literal
// constants; any other numbers run the variants that stage the model tables in LDS (link up to 4096 envs, quad above; no pair form).
static int effective_mapping(const qg_sim *s) {
    if (s->mapping == QG_MAP_LANE || s->mapping == QG_MAP_QUAD) return s->mapping;
    if (s->mapping == QG_MAP_PAIR) return s->baked ? QG_MAP_PAIR : QG_MAP_QUAD;
    // (the one-link-per-lane kernel addresses the state with 32-bit byte offsets from scalar bases: 19 n floats must stay below 4 GiB)
    if (s->mapping == QG_MAP_LINK) return (s->task.sensor_lag && s->n <= (1 << 24)) ? QG_MAP_LINK : QG_MAP_QUAD;
    // up to one wave of the one-link-per-lane kernel per SIMD (4096 envs on the 1024 SIMDs of an MI355X); the other boundaries are
    // the same measurement in units of "waves per SIMD" (pair: > 1 quad wave per SIMD up to 1 pair wave per SIMD, and from 1.75 on)
    const int simds = s->simds;
    if (s->task.sensor_lag && s->n <= simds * QGK_LINK_ENVS) return QG_MAP_LINK;
    if (s->baked && s->n > simds * QGK_QUAD_ENVS && (s->n <= simds * QGK_PAIR_ENVS || s->n >= (simds + 3 * (simds / 4)) * QGK_PAIR_ENVS)) return QG_MAP_PAIR;
    return QG_MAP_QUAD;
}

// Which step kernels carry the fused observation pack (KPoLaunch): the one-link-per-lane kernel, and -- round 3 -- the four-wave-workgroup
// forms of the two-legs-per-lane and one-leg-per-lane kernels that AUTO runs above 4096 envs (explicit mapping requests on small
// grids, which launch the one-wave-workgroup forms, keep the observation pack a launch of its own).
static bool po_fusable(const qg_sim *s) {
    const int emap = effective_mapping(s);
    if (emap == QG_MAP_LINK) return true;
    if (emap == QG_MAP_PAIR) return (s->n + QGK_PAIR_ENVS - 1) / QGK_PAIR_ENVS > s->simds / 4;
    if (emap == QG_MAP_QUAD) {
        const int qblocks = (s->n + QGK_QUAD_ENVS - 1) / QGK_QUAD_ENVS;
        return qblocks > s->simds / 4;
    }
    return false;
}

// `walk` != NULL: the fused walking launch (one-leg-per-lane kernel with the task layer folded in); walk_comps / walk_sample go
// with it
// `po` != NULL (with `walk`, one-link-per-lane mapping only): the partially observable observation pack fused in as well
static int launch_step(qg_sim *s, const float *d_actions, float *d_obs, float *d_reward, uint8_t *d_done, float *d_comps,
                       float *d_packed, hipStream_t stream, const KWalkLaunch *walk = nullptr, const KPoLaunch *po = nullptr) {
    KStepArgs P;
    P.st = s->st;
    P.n = s->n;
    P.track_ctrl = s->track_ctrl;
    P.actions = d_actions;
    P.obs = d_obs;
    P.reward = d_reward;
    P.done = d_done;
    P.comps = d_comps;
    P.packed = d_packed;
    P.seed = s->seed;
    P.env_index_base = s->env_index_base;
    int blocks = (s->n + QGK_WAVE - 1) / QGK_WAVE;
    const int emap = effective_mapping(s);
    if (s->res.launched) {            // a per-launch step while the resident kernel holds the state in registers: it has to hand it back first
        int rr = resident_retire(s);
        if (rr != QG_OK) return rr;
    }
    if (stream != s->stream) {
        s->caller_inflight = 1;
        if (!s->captured_once) {
            // (not asked of the legacy NULL stream: while ANOTHER stream is in global-mode capture that query itself is a
            // capture-implicit error and invalidates the capture in progress; a failed query counts as "captured")
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (stream) {
                const hipError_t qe = hipStreamIsCapturing(stream, &cs);
                if (qe != hipSuccess) { (void)hipGetLastError(); s->captured_once = 1; }
                else if (cs == hipStreamCaptureStatusActive) s->captured_once = 1;
            }
        }
    }
    if (po && !(walk && po_fusable(s))) return fail(QG_ERR_ARG, "launch_step: no step kernel with the fused observation pack for this handle");
    if (walk && emap == QG_MAP_LINK) {
        const int per_block = QGK_LINK_ENVS * QGK_LINK_WAVES;
        int lblocks = (s->n + per_block - 1) / per_block;
        dim3 lg(lblocks), lb(QGK_WAVE * QGK_LINK_WAVES);
        // helper waves (qg_step_kernel_link<.., HELP>): the compiled-in robot's walking forms; QG_LINK_HELPERS=0 at qg_create keeps the one-role kernels
        const int helpers = s->link_helpers;
        const dim3 lb2(2 * QGK_WAVE * QGK_LINK_WAVES);
        if (helpers && po && s->baked) hipLaunchKernelGGL((qg_step_kernel_link<true, true, true, true>), lg, lb2, 0, stream, s->d_model, s->d_task, P, *walk, *po);
        else if (helpers && s->baked) hipLaunchKernelGGL((qg_step_kernel_link<true, false, true, true>), lg, lb2, 0, stream, s->d_model, s->d_task, P, *walk, KPoNone{});
        else if (po && s->baked) hipLaunchKernelGGL((qg_step_kernel_link<true, true, true>), lg, lb, 0, stream, s->d_model, s->d_task, P, *walk, *po);
        else if (po) hipLaunchKernelGGL((qg_step_kernel_link<true, true, false>), lg, lb, 0, stream, s->d_model, s->d_task, P, *walk, *po);
        else if (s->baked) hipLaunchKernelGGL((qg_step_kernel_link<true, false, true>), lg, lb, 0, stream, s->d_model, s->d_task, P, *walk, KPoNone{});
        else hipLaunchKernelGGL((qg_step_kernel_link<true, false, false>), lg, lb, 0, stream, s->d_model, s->d_task, P, *walk, KPoNone{});
    } else if (walk && emap == QG_MAP_PAIR) {
        int pblocks = (s->n + QGK_PAIR_ENVS - 1) / QGK_PAIR_ENVS;
        if (po)
            hipLaunchKernelGGL((qg_step_kernel_pair<4, true, true>), dim3((pblocks + 3) / 4), dim3(QGK_WAVE * 4), 0, stream, s->d_task, P, *walk, *po);
        else if (pblocks > s->simds / 4)
            hipLaunchKernelGGL((qg_step_kernel_pair<4, true>), dim3((pblocks + 3) / 4), dim3(QGK_WAVE * 4), 0, stream, s->d_task, P, *walk, KPoNone{});
        else
            hipLaunchKernelGGL((qg_step_kernel_pair<1, true>), dim3(pblocks), dim3(QGK_WAVE), 0, stream, s->d_task, P, *walk, KPoNone{});
    } else if (walk) {
        int qblocks = (s->n + QGK_QUAD_ENVS - 1) / QGK_QUAD_ENVS;
        const int wpe = qblocks <= s->simds ? 1 : 2;
        const bool wg4 = qblocks > s->simds / 4;    // four-wave workgroups for grids of more than one wave per compute unit
        dim3 g1(qblocks), b1(QGK_WAVE), g4((qblocks + 3) / 4), b4(QGK_WAVE * 4);
        if (po) {                                   // po_fusable(): four-wave workgroups, register cap for one or two waves per SIMD
            if (!s->baked) hipLaunchKernelGGL((qg_step_kernel_quad<1, false, true, 4, true>), g4, b4, 0, stream, s->d_model, s->d_task, P, *walk, *po);
            else if (wpe == 1 && s->link_helpers)
                hipLaunchKernelGGL((qg_step_kernel_quad<2, true, true, 4, true, true>), g4, dim3(QGK_WAVE * 8), 0, stream, s->d_model, s->d_task, P, *walk, *po);
            else if (wpe == 1) hipLaunchKernelGGL((qg_step_kernel_quad<1, true, true, 4, true>), g4, b4, 0, stream, s->d_model, s->d_task, P, *walk, *po);
            else hipLaunchKernelGGL((qg_step_kernel_quad<2, true, true, 4, true>), g4, b4, 0, stream, s->d_model, s->d_task, P, *walk, *po);
        } else if (!s->baked) {
            if (wg4) hipLaunchKernelGGL((qg_step_kernel_quad<1, false, true, 4>), g4, b4, 0, stream, s->d_model, s->d_task, P, *walk, KPoNone{});
            else hipLaunchKernelGGL((qg_step_kernel_quad<1, false, true, 1>), g1, b1, 0, stream, s->d_model, s->d_task, P, *walk, KPoNone{});
        } else if (wpe == 1) {
            // at most one physics wave per SIMD: helper waves beside them (QG_LINK_HELPERS, as for the one-link-per-lane kernel)
            if (s->link_helpers)
                hipLaunchKernelGGL((qg_step_kernel_quad<2, true, true, 4, false, true>), g4, dim3(QGK_WAVE * 8), 0, stream, s->d_model, s->d_task, P, *walk, KPoNone{});
            else if (wg4) hipLaunchKernelGGL((qg_step_kernel_quad<1, true, true, 4>), g4, b4, 0, stream, s->d_model, s->d_task, P, *walk, KPoNone{});
            else hipLaunchKernelGGL((qg_step_kernel_quad<1, true, true, 1>), g1, b1, 0, stream, s->d_model, s->d_task, P, *walk, KPoNone{});
        } else {
            hipLaunchKernelGGL((qg_step_kernel_quad<2, true, true, 4>), g4, b4, 0, stream, s->d_model, s->d_task, P, *walk, KPoNone{});
        }
    } else if (emap == QG_MAP_LINK) {
        const int per_block = QGK_LINK_ENVS * QGK_LINK_WAVES;
        int lblocks = (s->n + per_block - 1) / per_block;
        dim3 lg(lblocks), lb(QGK_WAVE * QGK_LINK_WAVES);
        if (s->baked) hipLaunchKernelGGL((qg_step_kernel_link<false, false, true>), lg, lb, 0, stream, s->d_model, s->d_task, P, KWalkNone{}, KPoNone{});
        else hipLaunchKernelGGL((qg_step_kernel_link<false, false, false>), lg, lb, 0, stream, s->d_model, s->d_task, P, KWalkNone{}, KPoNone{});
    } else if (emap == QG_MAP_PAIR) {
        int pblocks = (s->n + QGK_PAIR_ENVS - 1) / QGK_PAIR_ENVS;
        if (pblocks > s->simds / 4)
            hipLaunchKernelGGL((qg_step_kernel_pair<4, false>), dim3((pblocks + 3) / 4), dim3(QGK_WAVE * 4), 0, stream, s->d_task, P, KWalkNone{}, KPoNone{});
        else
            hipLaunchKernelGGL((qg_step_kernel_pair<1, false>), dim3(pblocks), dim3(QGK_WAVE), 0, stream, s->d_task, P, KWalkNone{}, KPoNone{});
    } else if (emap == QG_MAP_QUAD) {
        int qblocks = (s->n + QGK_QUAD_ENVS - 1) / QGK_QUAD_ENVS;
        const bool one_wave = qblocks <= s->simds;  // at most one wave per SIMD (256 CUs x 4 on an MI355X): give each wave the whole register file
        const bool wg4 = qblocks > s->simds / 4;    // four-wave workgroups for grids of more than one wave per compute unit
        dim3 g1(qblocks), b1(QGK_WAVE), g4((qblocks + 3) / 4), b4(QGK_WAVE * 4);
        if (s->baked) {
            const int wpe = one_wave ? 1 : 2;
            if (wpe == 1 && !wg4) hipLaunchKernelGGL((qg_step_kernel_quad<1, true, false, 1>), g1, b1, 0, stream, s->d_model, s->d_task, P, KWalkNone{}, KPoNone{});
            else if (wpe == 1) hipLaunchKernelGGL((qg_step_kernel_quad<1, true, false, 4>), g4, b4, 0, stream, s->d_model, s->d_task, P, KWalkNone{}, KPoNone{});
            else if (!wg4) hipLaunchKernelGGL((qg_step_kernel_quad<2, true, false, 1>), g1, b1, 0, stream, s->d_model, s->d_task, P, KWalkNone{}, KPoNone{});
            else hipLaunchKernelGGL((qg_step_kernel_quad<2, true, false, 4>), g4, b4, 0, stream, s->d_model, s->d_task, P, KWalkNone{}, KPoNone{});
        } else {
            // tables in LDS: the 256-register cap spills 888 B per lane and measured 2x slower at every grid size (363 vs 741 us
            // at 262 144 envs), so any other robot runs the one-wave-per-SIMD form throughout
            if (wg4) hipLaunchKernelGGL((qg_step_kernel_quad<1, false, false, 4>), g4, b4, 0, stream, s->d_model, s->d_task, P, KWalkNone{}, KPoNone{});
            else hipLaunchKernelGGL((qg_step_kernel_quad<1, false, false, 1>), g1, b1, 0, stream, s->d_model, s->d_task, P, KWalkNone{}, KPoNone{});
        }
    } else if (s->baked)
        hipLaunchKernelGGL(qg_step_kernel<true>, dim3(blocks), dim3(QGK_WAVE), 0, stream, s->d_model, s->d_task, P);
    else
        hipLaunchKernelGGL(qg_step_kernel<false>, dim3(blocks), dim3(QGK_WAVE), 0, stream, s->d_model, s->d_task, P);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(QG_ERR_LAUNCH, "qg_step_kernel launch: %s", hipGetErrorString(e));
    if (s->task.auto_reset && (s->task.reset_flags & QG_RESET_JOINT_JITTER)) {     // start-pose randomisation of the envs just auto-reset
        const int total = 12 * s->n, threads = 256;
        hipLaunchKernelGGL(qg_jitter_kernel, dim3((total + threads - 1) / threads), dim3(threads), 0, stream, s->d_model, s->d_task, s->st, s->n,
                           (const uint8_t *)d_done, (const float *)d_packed, s->obs_dim + 2, s->seed, s->env_index_base);
        e = hipGetLastError();
        if (e != hipSuccess) return fail(QG_ERR_LAUNCH, "qg_jitter_kernel launch: %s", hipGetErrorString(e));
    }
    return QG_OK;
}

extern "C" int qg_step_device(qg_sim *s, const float *actions, float *obs, float *reward, uint8_t *done, float *comps, void *stream) {
    if (!s || !actions || !obs || !reward || !done) return fail(QG_ERR_ARG, "qg_step_device: null argument");
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    return launch_step(s, actions, obs, reward, done, comps, nullptr, (hipStream_t)stream);
}

extern "C" int qg_step_device_packed(qg_sim *s, const float *actions, float *packed, void *stream) {
    if (!s || !actions || !packed) return fail(QG_ERR_ARG, "qg_step_device_packed: null argument");
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    return launch_step(s, actions, nullptr, nullptr, nullptr, nullptr, packed, (hipStream_t)stream);
}

// ---- page-locked staging of the host-pointer entry points --------------------------------------------------------------------------
// The caller's buffers are ordinary (pageable) memory: a hipMemcpyAsync from / to them is a synchronous, staged copy with ~10-15 us of
// fixed cost each -- five of them made a ONE-env qg_step 70 us for a 10 us kernel.  The entry points therefore copy through one
// page-locked arena per handle: host memcpy in, truly asynchronous transfers enqueued around the launch, one stream synchronisation,
// host memcpy out (tools/host_step_rate.py).
struct PinOut { void *user; size_t off, bytes; };
static int pin_reserve(qg_sim *s, size_t bytes) {
    if (bytes <= s->h_pin_cap) return QG_OK;
    if (s->h_pin) { (void)hipHostFree(s->h_pin); s->h_pin = nullptr; s->h_pin_cap = 0; }
    size_t cap = bytes + bytes / 4 + 4096;
    HIP_TRY(hipHostMalloc((void **)&s->h_pin, cap, hipHostMallocDefault), QG_ERR_ALLOC);
    s->h_pin_cap = cap;
    return QG_OK;
}
static size_t pin_align(size_t x) { return (x + 255) & ~(size_t)255; }
// actions (host) -> device through the arena's first bytes, asynchronously on the library's stream
static int pin_actions_in(qg_sim *s, const float *actions, float *d_actions) {
    const size_t bytes = (size_t)s->n * QG_NU * sizeof(float);
    memcpy(s->h_pin, actions, bytes);
    HIP_TRY(hipMemcpyAsync(d_actions, s->h_pin, bytes, hipMemcpyHostToDevice, s->stream), QG_ERR_DEVICE);
    return QG_OK;
}
// Above ~1 MB the detour loses: the extra host copy into the caller's (often freshly allocated, not yet touched) array costs more than
// the staged transfer's fixed overhead -- the 4.3 MB observation stack of 4096 partially observable envs went from 213 to 369 us per step
// through the arena -- so large outputs go straight to the caller's memory.
#define QG_PIN_MAX_BYTES ((size_t)1 << 20)
static int pin_out_enqueue(qg_sim *s, const PinOut &o, const void *d_src) {
    if (!o.user) return QG_OK;
    void *dst = o.bytes > QG_PIN_MAX_BYTES ? o.user : (void *)(s->h_pin + o.off);
    HIP_TRY(hipMemcpyAsync(dst, d_src, o.bytes, hipMemcpyDeviceToHost, s->stream), QG_ERR_DEVICE);
    return QG_OK;
}
static void pin_out_finish(qg_sim *s, const PinOut &o) {
    if (o.user && o.bytes <= QG_PIN_MAX_BYTES) memcpy(o.user, s->h_pin + o.off, o.bytes);
}

// The host-pointer steps must not overtake device-pointer steps still in flight on a caller's stream; a device-wide wait is only
// needed if one has been enqueued since the last one (the library's own stream is synchronised at the end of every host-pointer call).
static int wait_for_caller_streams(qg_sim *s) {
    { int rr = resident_retire(s); if (rr != QG_OK) return rr; }
    if (!s->caller_inflight && !s->captured_once) return QG_OK;
    HIP_TRY(hipDeviceSynchronize(), QG_ERR_LAUNCH);
    s->caller_inflight = 0;
    return QG_OK;
}

extern "C" int qg_step(qg_sim *s, const float *actions, float *obs, float *reward, uint8_t *done, float *comps) {
    if (!s || !actions || !obs || !reward || !done) return fail(QG_ERR_ARG, "qg_step: null argument");
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    { int rc0 = wait_for_caller_streams(s); if (rc0 != QG_OK) return rc0; }
    size_t n = (size_t)s->n;
    size_t off = pin_align(n * QG_NU * sizeof(float));
    PinOut o_obs = {obs, off, n * s->obs_dim * sizeof(float)};       off += pin_align(o_obs.bytes);
    PinOut o_rew = {reward, off, n * sizeof(float)};                   off += pin_align(o_rew.bytes);
    PinOut o_done = {done, off, n};                                    off += pin_align(o_done.bytes);
    PinOut o_comp = {comps, off, n * QG_NREWARD * sizeof(float)};      off += pin_align(o_comp.bytes);
    int rc = pin_reserve(s, off);
    if (rc == QG_OK) rc = pin_actions_in(s, actions, s->d_actions);
    if (rc == QG_OK) rc = launch_step(s, s->d_actions, s->d_obs, s->d_reward, s->d_done, comps ? s->d_comps : nullptr, nullptr, s->stream);
    if (rc == QG_OK) rc = pin_out_enqueue(s, o_obs, s->d_obs);
    if (rc == QG_OK) rc = pin_out_enqueue(s, o_rew, s->d_reward);
    if (rc == QG_OK) rc = pin_out_enqueue(s, o_done, s->d_done);
    if (rc == QG_OK) rc = pin_out_enqueue(s, o_comp, s->d_comps);
    if (rc != QG_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream), QG_ERR_LAUNCH);
    pin_out_finish(s, o_obs); pin_out_finish(s, o_rew); pin_out_finish(s, o_done); pin_out_finish(s, o_comp);
    return QG_OK;
}

static int copy_in(qg_sim *s, const float *host, float *field_major, int w) {
    if (!host) return QG_OK;
    int total = s->n * w, threads = 256;
    HIP_TRY(hipMemcpyAsync(s->d_stage, host, (size_t)total * sizeof(float), hipMemcpyHostToDevice, s->stream), QG_ERR_DEVICE);
    hipLaunchKernelGGL(qg_transpose_in, dim3((total + threads - 1) / threads), dim3(threads), 0, s->stream, s->d_stage, field_major, s->n, w);
    HIP_TRY(hipGetLastError(), QG_ERR_LAUNCH);
    HIP_TRY(hipStreamSynchronize(s->stream), QG_ERR_LAUNCH);
    return QG_OK;
}

// State snapshot, part 1: where the five outputs land in the page-locked arena (from `off` on); part 2: the four transposes into their
// own regions of the staging buffer and every transfer, enqueued on the library's stream (no synchronisation in here).
struct StateOut { PinOut o[5]; };
static size_t state_out_layout(qg_sim *s, float *qpos, float *qvel, float *act, float *ctrl, int32_t *nstep, size_t off, StateOut &so) {
    const size_t n = (size_t)s->n;
    float *dst[4] = {qpos, qvel, act, ctrl};
    const int w[4] = {QG_NQ, QG_NV, QG_NU, QG_NU};
    for (int f = 0; f < 4; f++) { so.o[f] = {dst[f], off, n * w[f] * sizeof(float)}; off += pin_align(so.o[f].bytes); }
    so.o[4] = {nstep, off, n * sizeof(int32_t)};
    return off + pin_align(so.o[4].bytes);
}
static int state_out_enqueue(qg_sim *s, const StateOut &so) {
    const size_t n = (size_t)s->n;
    const float *src[4] = {s->st.qpos, s->st.qvel, s->st.act, s->st.ctrl};
    const int w[4] = {QG_NQ, QG_NV, QG_NU, QG_NU};
    size_t soff = 0;
    int rc;
    for (int f = 0; f < 4; f++) {
        float *stage = s->d_stage + soff;
        soff += n * w[f];
        if (!so.o[f].user) continue;
        const int total = s->n * w[f], threads = 256;
        hipLaunchKernelGGL(qg_transpose_out, dim3((total + threads - 1) / threads), dim3(threads), 0, s->stream, src[f], stage, s->n, w[f]);
        HIP_TRY(hipGetLastError(), QG_ERR_LAUNCH);
        if ((rc = pin_out_enqueue(s, so.o[f], stage)) != QG_OK) return rc;
    }
    return pin_out_enqueue(s, so.o[4], s->st.nstep);
}

extern "C" int qg_get_state(qg_sim *s, float *qpos, float *qvel, float *act, float *ctrl, int32_t *nstep) {
    if (!s) return fail(QG_ERR_ARG, "null handle");
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    { int rr = resident_retire(s); if (rr != QG_OK) return rr; }
    HIP_TRY(hipDeviceSynchronize(), QG_ERR_LAUNCH);   // steps may be in flight on a caller's stream
    // every transfer enqueued, ONE synchronisation (five synchronised round trips made the single-env facade's mirror of the state
    // 120 us of a 160 us step)
    StateOut so;
    int rc = pin_reserve(s, state_out_layout(s, qpos, qvel, act, ctrl, nstep, 0, so));
    if (rc == QG_OK) rc = state_out_enqueue(s, so);
    if (rc != QG_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream), QG_ERR_LAUNCH);
    for (int f = 0; f < 5; f++) pin_out_finish(s, so.o[f]);
    return QG_OK;
}

// qg_step and qg_get_state in one call and one synchronisation: what an env that mirrors the state on the host after every step
// (the reference's `env.data`, read by user reward / termination callables) needs.
extern "C" int qg_step_mirror(qg_sim *s, const float *actions, float *obs, float *reward, uint8_t *done, float *comps, float *qpos, float *qvel,
                              float *act, float *ctrl, int32_t *nstep) {
    if (!s || !actions || !obs || !reward || !done) return fail(QG_ERR_ARG, "qg_step_mirror: null argument");
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    { int rc0 = wait_for_caller_streams(s); if (rc0 != QG_OK) return rc0; }
    size_t n = (size_t)s->n;
    size_t off = pin_align(n * QG_NU * sizeof(float));
    PinOut o_obs = {obs, off, n * s->obs_dim * sizeof(float)};       off += pin_align(o_obs.bytes);
    PinOut o_rew = {reward, off, n * sizeof(float)};                   off += pin_align(o_rew.bytes);
    PinOut o_done = {done, off, n};                                    off += pin_align(o_done.bytes);
    PinOut o_comp = {comps, off, n * QG_NREWARD * sizeof(float)};      off += pin_align(o_comp.bytes);
    StateOut so;
    off = state_out_layout(s, qpos, qvel, act, ctrl, nstep, off, so);
    int rc = pin_reserve(s, off);
    if (rc == QG_OK) rc = pin_actions_in(s, actions, s->d_actions);
    if (rc == QG_OK) rc = launch_step(s, s->d_actions, s->d_obs, s->d_reward, s->d_done, comps ? s->d_comps : nullptr, nullptr, s->stream);
    if (rc == QG_OK) rc = pin_out_enqueue(s, o_obs, s->d_obs);
    if (rc == QG_OK) rc = pin_out_enqueue(s, o_rew, s->d_reward);
    if (rc == QG_OK) rc = pin_out_enqueue(s, o_done, s->d_done);
    if (rc == QG_OK) rc = pin_out_enqueue(s, o_comp, s->d_comps);
    if (rc == QG_OK) rc = state_out_enqueue(s, so);
    if (rc != QG_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream), QG_ERR_LAUNCH);
    pin_out_finish(s, o_obs); pin_out_finish(s, o_rew); pin_out_finish(s, o_done); pin_out_finish(s, o_comp);
    for (int f = 0; f < 5; f++) pin_out_finish(s, so.o[f]);
    return QG_OK;
}

extern "C" int qg_set_state(qg_sim *s, const float *qpos, const float *qvel, const float *act, const float *ctrl, const int32_t *nstep) {
    if (!s) return fail(QG_ERR_ARG, "null handle");
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    { int rr = resident_retire(s); if (rr != QG_OK) return rr; }
    HIP_TRY(hipDeviceSynchronize(), QG_ERR_LAUNCH);
    int rc;
    if ((rc = copy_in(s, qpos, s->st.qpos, QG_NQ)) != QG_OK) return rc;
    if ((rc = copy_in(s, qvel, s->st.qvel, QG_NV)) != QG_OK) return rc;
    if ((rc = copy_in(s, act, s->st.act, QG_NU)) != QG_OK) return rc;
    if ((rc = copy_in(s, ctrl, s->st.ctrl, QG_NU)) != QG_OK) return rc;
    if (nstep) HIP_TRY(hipMemcpy(s->st.nstep, nstep, (size_t)s->n * sizeof(int32_t), hipMemcpyHostToDevice), QG_ERR_DEVICE);
    return QG_OK;
}

extern "C" int qg_time_step_kernel(qg_sim *s, const float *d_actions, float *d_packed, int32_t iters, float *ms_per_launch) {
    if (!s || !d_actions || !d_packed || iters < 1 || !ms_per_launch) return fail(QG_ERR_ARG, "qg_time_step_kernel: bad argument");
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    { int rr = resident_retire(s); if (rr != QG_OK) return rr; }
    HIP_TRY(hipDeviceSynchronize(), QG_ERR_LAUNCH);
    // the launches are exactly what qg_step_device_packed enqueues (data.ctrl write-back as the handle has it set)
    HIP_TRY(hipEventRecord(s->ev0, s->stream), QG_ERR_DEVICE);
    for (int i = 0; i < iters; i++) {
        int rc = launch_step(s, d_actions, nullptr, nullptr, nullptr, nullptr, d_packed, s->stream);
        if (rc != QG_OK) return rc;
    }
    HIP_TRY(hipEventRecord(s->ev1, s->stream), QG_ERR_DEVICE);
    HIP_TRY(hipEventSynchronize(s->ev1), QG_ERR_LAUNCH);
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1), QG_ERR_DEVICE);
    *ms_per_launch = ms / (float)iters;
    return QG_OK;
}

extern "C" int qg_set_mapping(qg_sim *s, int32_t mapping) {
    if (!s) return fail(QG_ERR_ARG, "null handle");
    if (mapping == QG_MAP_PAIR && !s->baked)
        return fail(QG_ERR_ARG, "qg_set_mapping: the two-legs-per-lane kernel serves the compiled-in robot only");
    if (mapping != QG_MAP_AUTO && mapping != QG_MAP_LANE && mapping != QG_MAP_QUAD && mapping != QG_MAP_PAIR && mapping != QG_MAP_LINK)
        return fail(QG_ERR_ARG, "qg_set_mapping: unknown mapping %d", mapping);
    if (s->res.active) return fail(QG_ERR_ARG, "qg_set_mapping: the resident step mode is on (qg_resident_stop first)");
    s->mapping = mapping;
    return QG_OK;
}
extern "C" int qg_get_mapping(const qg_sim *s) { return s ? effective_mapping(s) : fail(QG_ERR_ARG, "null handle"); }

/* 1 if the handle runs the kernel variant with the default robot's constants baked in as literals */
extern "C" int qg_uses_baked_model(const qg_sim *s) { return s ? s->baked : fail(QG_ERR_ARG, "null handle"); }

extern "C" int qg_set_task(qg_sim *s, const qg_task *task) {
    if (!s || !task) return fail(QG_ERR_ARG, "qg_set_task: null argument");
    if (s->walk_bound) return fail(QG_ERR_ARG, "qg_set_task: a walking task layer is bound to this handle (it holds a copy of the task)");
    if (task->obs_mode != s->task.obs_mode) return fail(QG_ERR_ARG, "qg_set_task: obs_mode is fixed at qg_create (it sizes the output rows)");
    KModel km;
    KTask kt;
    int rc = build_tables(&s->model, task, &km, &kt);
    if (rc != QG_OK) return rc;
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    { int rr = resident_retire(s); if (rr != QG_OK) return rr; }
    HIP_TRY(hipDeviceSynchronize(), QG_ERR_LAUNCH);   // steps reading the old task may be in flight on a caller's stream
    HIP_TRY(hipMemcpy(s->d_task, &kt, sizeof kt, hipMemcpyHostToDevice), QG_ERR_DEVICE);
    s->task = *task;
    return QG_OK;
}

extern "C" int qg_get_task(const qg_sim *s, qg_task *out) {
    if (!s || !out) return fail(QG_ERR_ARG, "qg_get_task: null argument");
    *out = s->task;
    return QG_OK;
}

/* development builds (-DQG_PHASE_TIMES): the s_memrealtime stamps (10 ns units) of the last launch's first wave; QG_ERR_ARG in production builds */
extern "C" int qg_debug_phase_times(uint64_t out[16]) {
#ifdef QG_PHASE_TIMES
    if (!out) return fail(QG_ERR_ARG, "qg_debug_phase_times: null output");
    HIP_TRY(hipDeviceSynchronize(), QG_ERR_LAUNCH);
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(qg_phase_times), 16 * sizeof(uint64_t)), QG_ERR_DEVICE);
    return QG_OK;
#else
    (void)out;
    return fail(QG_ERR_ARG, "qg_debug_phase_times: the library was built without -DQG_PHASE_TIMES");
#endif
}

extern "C" int qg_set_track_ctrl(qg_sim *s, int32_t on) {
    if (!s) return fail(QG_ERR_ARG, "null handle");
    if ((on ? 1 : 0) != s->track_ctrl) { int rr = resident_retire(s); if (rr != QG_OK) return rr; }      // (a resident launch holds the old setting)
    s->track_ctrl = on ? 1 : 0;
    return QG_OK;
}

// ------------------------------------------------------------------------------------------------------
// many env-steps per launch: the sequence form and the resident form of the one-link-per-lane kernel (qg_kernel_resident.hip)
// ------------------------------------------------------------------------------------------------------
static int multi_step_usable(const qg_sim *s, const char *who) {
    if (effective_mapping(s) != QG_MAP_LINK)
        return fail(QG_ERR_ARG, "%s: needs the one-link-per-lane mapping (AUTO up to 4096 envs, lagged sensors)", who);
    if (s->n > s->simds * QGK_LINK_ENVS) return fail(QG_ERR_ARG, "%s: at most %d envs (one wave per SIMD)", who, s->simds * QGK_LINK_ENVS);
    if (s->walk_bound) return fail(QG_ERR_ARG, "%s: a walking task layer is bound to this handle", who);
    if (s->task.auto_reset && (s->task.reset_flags & QG_RESET_JOINT_JITTER))
        return fail(QG_ERR_ARG, "%s: hinge jitter at auto-reset is a launch of its own behind every step; not available in this form", who);
    return QG_OK;
}
static KStepArgs multi_step_args(const qg_sim *s) {
    KStepArgs P = {};
    P.st = s->st;
    P.n = s->n;
    P.track_ctrl = s->track_ctrl;
    P.seed = s->seed;
    P.env_index_base = s->env_index_base;
    return P;
}
static dim3 multi_step_grid(const qg_sim *s) {
    const int per_block = QGK_LINK_ENVS * QGK_LINK_WAVES;
    return dim3((s->n + per_block - 1) / per_block);
}

extern "C" int qg_step_device_seq(qg_sim *s, const float *actions, float *packed, int32_t count, void *stream) {
    if (!s || !actions || !packed || count < 1) return fail(QG_ERR_ARG, "qg_step_device_seq: bad argument");
    if (s->walk_bound) return fail(QG_ERR_ARG, "qg_step_device_seq: a walking task layer is bound to this handle");
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    int rc;
    if (s->res.launched && (rc = resident_retire(s)) != QG_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bool seq_pair = effective_mapping(s) == QG_MAP_PAIR && s->baked && s->task.sensor_lag &&
                          !(s->task.auto_reset && (s->task.reset_flags & QG_RESET_JOINT_JITTER));
    if (seq_pair) {                   // the two-legs-per-lane mapping (16 385 .. 32 768 envs, >= 57 344): its own one-launch form
        if (st != s->stream) {
            s->caller_inflight = 1;
            if (!s->captured_once) {
                hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                if (st && hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive) s->captured_once = 1;
            }
        }
        KResident R = {};
        R.actions = actions;
        R.packed = packed;
        R.count = count;
        R.slots = 1;
        const KStepArgs P = multi_step_args(s);
        const int pblocks = (s->n + QGK_PAIR_ENVS - 1) / QGK_PAIR_ENVS;
        if (pblocks > s->simds / 4) hipLaunchKernelGGL((qg_step_kernel_pair_multi<4>), dim3((pblocks + 3) / 4), dim3(QGK_WAVE * 4), 0, st, s->d_task, P, R);
        else hipLaunchKernelGGL((qg_step_kernel_pair_multi<1>), dim3(pblocks), dim3(QGK_WAVE), 0, st, s->d_task, P, R);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(QG_ERR_LAUNCH, "qg_step_kernel_pair_multi launch: %s", hipGetErrorString(e));
        return QG_OK;
    }
    const int qblocks_seq = (s->n + QGK_QUAD_ENVS - 1) / QGK_QUAD_ENVS;
    const bool seq_quad = effective_mapping(s) == QG_MAP_QUAD && s->task.sensor_lag && qblocks_seq > s->simds / 4 &&
                          !(s->task.auto_reset && (s->task.reset_flags & QG_RESET_JOINT_JITTER));
    if (seq_quad) {                   // the one-leg-per-lane mapping on the grids AUTO gives it (four-wave workgroups): its own one-launch form
        if (st != s->stream) {
            s->caller_inflight = 1;
            if (!s->captured_once) {
                hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                if (st && hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive) s->captured_once = 1;
            }
        }
        KResident R = {};
        R.actions = actions;
        R.packed = packed;
        R.count = count;
        R.slots = 1;
        const KStepArgs P = multi_step_args(s);
        const dim3 g4((qblocks_seq + 3) / 4), b4(QGK_WAVE * 4);
        const bool one_wave = qblocks_seq <= s->simds;          // as launch_step: the whole register file while the grid is one wave per SIMD
        if (!s->baked) hipLaunchKernelGGL((qg_step_kernel_quad_multi<1, false>), g4, b4, 0, st, s->d_model, s->d_task, P, R);   // (tables in LDS: always the one-wave form)
        else if (one_wave) hipLaunchKernelGGL((qg_step_kernel_quad_multi<1, true>), g4, b4, 0, st, s->d_model, s->d_task, P, R);
        else hipLaunchKernelGGL((qg_step_kernel_quad_multi<2, true>), g4, b4, 0, st, s->d_model, s->d_task, P, R);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(QG_ERR_LAUNCH, "qg_step_kernel_quad_multi launch: %s", hipGetErrorString(e));
        return QG_OK;
    }
    if (multi_step_usable(s, "qg_step_device_seq") != QG_OK) {
        // another mapping (more than one wave per SIMD of the one-link-per-lane kernel), or hinge jitter behind every step: the same
        // rows from `count` per-step launches -- the call means the same thing for every handle, the one-launch form is the fast path
        const size_t arow = (size_t)s->n * QG_NU, prow = (size_t)s->n * (s->obs_dim + 2);
        for (int32_t k = 0; k < count; k++)
            if ((rc = launch_step(s, actions + k * arow, nullptr, nullptr, nullptr, nullptr, packed + k * prow, st)) != QG_OK) return rc;
        return QG_OK;
    }
    if (st != s->stream) {
        s->caller_inflight = 1;
        if (!s->captured_once) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (st && hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive) s->captured_once = 1;
        }
    }
    KResident R = {};
    R.actions = actions;
    R.packed = packed;
    R.count = count;
    R.slots = 1;
    const KStepArgs P = multi_step_args(s);
    const dim3 g = multi_step_grid(s), b(QGK_WAVE * QGK_LINK_WAVES);
    if (s->baked) hipLaunchKernelGGL((qg_step_kernel_link_multi<true, false>), g, b, 0, st, s->d_model, s->d_task, P, R);
    else hipLaunchKernelGGL((qg_step_kernel_link_multi<false, false>), g, b, 0, st, s->d_model, s->d_task, P, R);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(QG_ERR_LAUNCH, "qg_step_kernel_link_multi launch: %s", hipGetErrorString(e));
    return QG_OK;
}

static void resident_free(qg_sim *s) {
    if (s->res.d_mail) (void)hipFree(s->res.d_mail);
    if (s->res.own_buffers && s->res.k.actions) (void)hipFree((void *)s->res.k.actions);
    if (s->res.own_buffers && s->res.k.packed) (void)hipFree(s->res.k.packed);
    if (s->res.hstat) (void)hipHostFree((void *)s->res.hstat);
    if (s->res.ctl_stream) (void)hipStreamDestroy(s->res.ctl_stream);
    memset(&s->res, 0, sizeof s->res);
}

// The resident kernel stores the state it holds in registers and leaves: STOP into the door (the waves first finish what has been
// rung), then the library's stream -- where the kernel runs -- is waited for.  Rings still queued on the caller's stream are waited
// for first, so that "every step enqueued before this call" has run, as the ordering contract of quadgym.h says.
static int resident_retire(qg_sim *s) {
    if (!s->res.active || !s->res.launched) return QG_OK;
    if (s->res.last_stream) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s->res.last_stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive)
            return fail(QG_ERR_ARG, "the resident step kernel cannot be retired while its rings are being captured");
        (void)hipStreamSynchronize(s->res.last_stream);      // (a stream the caller has destroyed since is no reason to fail)
        (void)hipGetLastError();
    }
    hipLaunchKernelGGL(qg_resident_ctl_kernel, dim3(1), dim3(64), 0, s->res.ctl_stream, s->res.k.door, 0);
    HIP_TRY(hipGetLastError(), QG_ERR_LAUNCH);
    HIP_TRY(hipStreamSynchronize(s->res.ctl_stream), QG_ERR_LAUNCH);
    HIP_TRY(hipStreamSynchronize(s->stream), QG_ERR_LAUNCH);
    s->res.launched = 0;
    return QG_OK;
}

// (re)launch on the library's stream: clear STOP, then the kernel; it starts at the env-step the previous launch left off at
static int resident_launch(qg_sim *s) {
    s->res.hstat[0] = QG_RES_RUNNING;
    hipLaunchKernelGGL(qg_resident_ctl_kernel, dim3(1), dim3(64), 0, s->stream, s->res.k.door, 1);
    HIP_TRY(hipGetLastError(), QG_ERR_LAUNCH);
    KStepArgs P = multi_step_args(s);
    const dim3 g = multi_step_grid(s), b(QGK_WAVE * QGK_LINK_WAVES);
    if (s->baked) hipLaunchKernelGGL((qg_step_kernel_link_multi<true, true>), g, b, 0, s->stream, s->d_model, s->d_task, P, s->res.k);
    else hipLaunchKernelGGL((qg_step_kernel_link_multi<false, true>), g, b, 0, s->stream, s->d_model, s->d_task, P, s->res.k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(QG_ERR_LAUNCH, "resident kernel launch: %s", hipGetErrorString(e));
    s->res.launched = 1;
    // (a ring on another stream that gets to the door before the two launches above sees STOP with `RUNNING` in hstat and waits for
    // the door to open -- qg_resident_ring_kernel -- so nothing has to be waited for here)
    return QG_OK;
}

extern "C" int qg_resident_start(qg_sim *s, int32_t slots, int32_t idle_timeout_us, float *actions, float *packed) {
    if (!s) return fail(QG_ERR_ARG, "null handle");
    if ((actions == nullptr) != (packed == nullptr)) return fail(QG_ERR_ARG, "qg_resident_start: pass both slot buffers or neither");
    if (s->res.active) return fail(QG_ERR_ARG, "qg_resident_start: already on");
    if (slots < 1 || slots > 4096) return fail(QG_ERR_ARG, "qg_resident_start: slots must be 1..4096");
    if (idle_timeout_us == 0) idle_timeout_us = 2000;
    if (idle_timeout_us < 50 || idle_timeout_us > 100000) return fail(QG_ERR_ARG, "qg_resident_start: idle_timeout_us must be 50..100000 (0 = 2000)");
    int rc = multi_step_usable(s, "qg_resident_start");
    if (rc != QG_OK) return rc;
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    HIP_TRY(hipDeviceSynchronize(), QG_ERR_LAUNCH);
    const size_t n = (size_t)s->n, row = (size_t)s->obs_dim + 2;
    const size_t mail_bytes = 256 + QG_RES_SHARDS * 128 + 256;
    hipError_t e = hipMalloc(&s->res.d_mail, mail_bytes);
    if (e == hipSuccess) e = hipMemset(s->res.d_mail, 0, mail_bytes);
    s->res.own_buffers = actions == nullptr;
    if (s->res.own_buffers) {
        float *acts = nullptr;
        if (e == hipSuccess) e = hipMalloc((void **)&acts, (size_t)slots * n * QG_NU * sizeof(float));
        if (e == hipSuccess) e = hipMemset(acts, 0, (size_t)slots * n * QG_NU * sizeof(float));
        s->res.k.actions = acts;
        if (e == hipSuccess) e = hipMalloc((void **)&s->res.k.packed, (size_t)slots * n * row * sizeof(float));
        if (e == hipSuccess) e = hipMemset(s->res.k.packed, 0, (size_t)slots * n * row * sizeof(float));
    } else {
        s->res.k.actions = actions;
        s->res.k.packed = packed;
    }
    if (e == hipSuccess) e = hipHostMalloc((void **)&s->res.hstat, 64, hipHostMallocCoherent | hipHostMallocMapped);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->res.ctl_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        resident_free(s);
        return fail(QG_ERR_ALLOC, "qg_resident_start: %s", hipGetErrorString(e));
    }
    for (int i = 0; i < 8; i++) s->res.hstat[i] = 0;
    uint8_t *m = (uint8_t *)s->res.d_mail;
    s->res.k.door = (unsigned long long *)m;
    s->res.k.done = (unsigned long long *)(m + 256);
    s->res.k.completed = (unsigned long long *)(m + 256 + QG_RES_SHARDS * 128);
    s->res.k.hstat = (unsigned long long *)s->res.hstat;
    s->res.k.slots = slots;
    s->res.k.count = 0;
    s->res.k.idle_ticks = (uint32_t)idle_timeout_us * 100u;
    s->res.k.ring_ticks = 20000000u;          // 200 ms without a single arrival: the ring gives up and says so
    s->res.active = 1;
    s->res.rung = 0;
    return resident_launch(s);
}

extern "C" int qg_resident_stop(qg_sim *s) {
    if (!s) return fail(QG_ERR_ARG, "null handle");
    if (!s->res.active) return QG_OK;
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    int rc = resident_retire(s);
    if (rc != QG_OK) return rc;
    HIP_TRY(hipDeviceSynchronize(), QG_ERR_LAUNCH);
    resident_free(s);
    return QG_OK;
}

extern "C" int qg_resident_buffers(qg_sim *s, float **actions, float **packed, int32_t *slots) {
    if (!s) return fail(QG_ERR_ARG, "null handle");
    if (!s->res.active) return fail(QG_ERR_ARG, "qg_resident_buffers: the resident step mode is off");
    if (actions) *actions = (float *)s->res.k.actions;
    if (packed) *packed = s->res.k.packed;
    if (slots) *slots = s->res.k.slots;
    return QG_OK;
}

// rings that found the kernel retired, or gave up waiting, since the last look: an error the caller must see once
static int resident_check_reports(qg_sim *s, const char *who) {
    const uint64_t lost = s->res.hstat[2], gave = s->res.hstat[3];
    if (lost != s->res.lost_seen) {
        const uint64_t d = lost - s->res.lost_seen;
        s->res.lost_seen = lost;
        s->res.rung -= (int64_t)d;        // those rings did not advance the door: the count (and the slot of the next env-step) follows the device
        return fail(QG_ERR_LAUNCH, "%s: %llu env-step(s) were rung after the resident kernel had retired and were NOT executed "
                    "(rings must follow one another within the idle time-out, or call qg_resident_ensure before a burst)", who, (unsigned long long)d);
    }
    if (gave != s->res.gaveup_seen) {
        s->res.gaveup_seen = gave;
        return fail(QG_ERR_LAUNCH, "%s: a ring gave up waiting for the resident kernel (no arrival for 200 ms)", who);
    }
    return QG_OK;
}

extern "C" int qg_resident_ensure(qg_sim *s) {
    if (!s) return fail(QG_ERR_ARG, "null handle");
    if (!s->res.active) return fail(QG_ERR_ARG, "qg_resident_ensure: the resident step mode is off");
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    if (s->res.launched && s->res.hstat[0] == QG_RES_RUNNING) return QG_OK;
    if (s->res.launched && s->res.hstat[0] == QG_RES_RETIRING) {
        // no ring for half the time-out: the kernel may shut the door before a ring enqueued now reaches it.  Retire it here (it
        // stores the state and leaves within microseconds) and launch it again: a fresh idle clock for what follows.
        int rc = resident_retire(s);
        if (rc != QG_OK) return rc;
    }
    return resident_launch(s);        // stream-ordered behind the launch that has left (or is leaving)
}

extern "C" int qg_resident_step_device(qg_sim *s, int32_t count, void *stream) {
    if (!s || count < 1) return fail(QG_ERR_ARG, "qg_resident_step_device: bad argument");
    if (!s->res.active) return fail(QG_ERR_ARG, "qg_resident_step_device: the resident step mode is off (qg_resident_start)");
    if (count > s->res.k.slots) return fail(QG_ERR_ARG, "qg_resident_step_device: count %d exceeds the %d slots of the mailbox", count, s->res.k.slots);
    if ((hipStream_t)stream == s->stream) return fail(QG_ERR_ARG, "qg_resident_step_device: that is the stream the resident kernel occupies");
    int rc = resident_check_reports(s, "qg_resident_step_device");
    if (rc != QG_OK) return rc;
    if ((rc = qg_resident_ensure(s)) != QG_OK) return rc;
    const unsigned nwaves = multi_step_grid(s).x * QGK_LINK_WAVES;
    hipLaunchKernelGGL(qg_resident_ring_kernel, dim3(1), dim3(QGK_WAVE), 0, (hipStream_t)stream, s->res.k, (unsigned)count, nwaves,
                       (unsigned long long)s->res.rung);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(QG_ERR_LAUNCH, "ring kernel launch: %s", hipGetErrorString(e));
    s->res.last_stream = (hipStream_t)stream;
    s->res.rung += count;
    return QG_OK;
}

extern "C" int qg_resident_status(qg_sim *s, int64_t *rung, int32_t *running, int64_t *completed_at_exit, int64_t *not_executed) {
    if (!s) return fail(QG_ERR_ARG, "null handle");
    if (!s->res.active) return fail(QG_ERR_ARG, "qg_resident_status: the resident step mode is off");
    if (rung) *rung = s->res.rung;
    if (running) *running = s->res.launched && s->res.hstat[0] == QG_RES_RUNNING;
    if (completed_at_exit) *completed_at_exit = (int64_t)s->res.hstat[1];
    if (not_executed) *not_executed = (int64_t)s->res.hstat[2];
    return QG_OK;
}

// ------------------------------------------------------------------------------------------------------
// native per-step exchange over RCCL (qg_comm.h)
// ------------------------------------------------------------------------------------------------------
#include "qg_comm.h"

extern "C" int qg_comm_unique_id(uint8_t id[QG_COMM_ID_BYTES]) {
    if (!id) return fail(QG_ERR_ARG, "qg_comm_unique_id: null output");
    int rc = qg_rccl_load();
    if (rc != QG_OK) return rc;
    qg_nccl_unique_id u;
    RCCL_TRY(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, QG_COMM_ID_BYTES);
    return QG_OK;
}

extern "C" int qg_comm_destroy(qg_comm *c) {
    if (!c) return QG_OK;
    (void)hipSetDevice(c->sim->device);
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    for (int i = 0; i < 2; i++) {
        if (c->produced[i]) (void)hipEventDestroy(c->produced[i]);
        if (c->consumed[i]) (void)hipEventDestroy(c->consumed[i]);
    }
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    delete c;
    return QG_OK;
}

extern "C" int qg_comm_create(qg_sim *s, int32_t rank, int32_t world, const uint8_t id[QG_COMM_ID_BYTES], qg_comm **out) {
    if (!s || !id || !out || world < 1 || rank < 0 || rank >= world) return fail(QG_ERR_ARG, "qg_comm_create: bad argument");
    *out = nullptr;
    int rc = qg_rccl_load();
    if (rc != QG_OK) return rc;
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    qg_comm *c = new (std::nothrow) qg_comm();
    if (!c) return fail(QG_ERR_ALLOC, "out of host memory");
    memset(c, 0, sizeof *c);
    c->sim = s;
    c->rank = rank;
    c->world = world;
    hipError_t e = hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        e = hipEventCreateWithFlags(&c->produced[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->consumed[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        qg_comm_destroy(c);
        return fail(QG_ERR_DEVICE, "qg_comm_create: %s", hipGetErrorString(e));
    }
    qg_nccl_unique_id u;
    memcpy(u.internal, id, QG_COMM_ID_BYTES);
    int r = g_rccl.CommInitRank(&c->comm, world, u, rank);
    if (r != 0) {
        const char *msg = g_rccl.GetErrorString(r);
        qg_comm_destroy(c);
        return fail(QG_ERR_DEVICE, "ncclCommInitRank: %s", msg);
    }
    *out = c;
    return QG_OK;
}

extern "C" int qg_comm_rollout(qg_comm *c, const float *const *actions, int32_t n_actions, float *const packed[2], float *const gathered[2],
                               int32_t steps, int32_t root) {
    if (!c || !actions || n_actions < 1 || !packed || steps < 0 || root < 0 || root >= c->world) return fail(QG_ERR_ARG, "qg_comm_rollout: bad argument");
    if (c->rank == root && !gathered) return fail(QG_ERR_ARG, "qg_comm_rollout: the root needs the gathered buffers");
    qg_sim *s = c->sim;
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    const size_t count = (size_t)s->n * (size_t)(s->obs_dim + 2);
    for (int k = 0; k < steps; k++) {
        const int b = k & 1;
        // the step may overwrite packed[b] only after the gather that read it (two steps ago) has finished
        if (c->consumed_valid[b]) HIP_TRY(hipStreamWaitEvent(s->stream, c->consumed[b], 0), QG_ERR_DEVICE);
        int rc = launch_step(s, actions[k % n_actions], nullptr, nullptr, nullptr, nullptr, packed[b], s->stream);
        if (rc != QG_OK) return rc;
        HIP_TRY(hipEventRecord(c->produced[b], s->stream), QG_ERR_DEVICE);
        HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->produced[b], 0), QG_ERR_DEVICE);
        RCCL_TRY(g_rccl.GroupStart());
        if (c->rank == root)
            for (int r = 0; r < c->world; r++)
                RCCL_TRY(g_rccl.Recv(gathered[b] + (size_t)r * count, count, QG_NCCL_FLOAT32, r, c->comm, c->comm_stream));
        RCCL_TRY(g_rccl.Send(packed[b], count, QG_NCCL_FLOAT32, root, c->comm, c->comm_stream));
        RCCL_TRY(g_rccl.GroupEnd());
        HIP_TRY(hipEventRecord(c->consumed[b], c->comm_stream), QG_ERR_DEVICE);
        c->consumed_valid[b] = 1;
    }
    return QG_OK;
}

extern "C" int qg_comm_synchronize(qg_comm *c) {
    if (!c) return fail(QG_ERR_ARG, "null handle");
    HIP_TRY(hipSetDevice(c->sim->device), QG_ERR_DEVICE);
    HIP_TRY(hipStreamSynchronize(c->sim->stream), QG_ERR_LAUNCH);
    HIP_TRY(hipStreamSynchronize(c->comm_stream), QG_ERR_LAUNCH);
    return QG_OK;
}

// ------------------------------------------------------------------------------------------------------
// walking task layer (qg_walk.hip)
// ------------------------------------------------------------------------------------------------------
struct qg_walk {
    qg_sim *sim;
    int32_t saved_use_flip, saved_track_ctrl, bound;     // what qg_walk_create changed on the sim; restored by qg_walk_destroy
    qg_walk_params params;
    KWalkParams kp;
    KWalkState st;
    float *d_obs, *d_reward, *d_comps, *d_actions, *d_tmp;
    uint8_t *d_done;
    size_t ring_slots, summary_blocks;      // allocated extent of the estimator's ring (whole blocks) and of its block summaries
};

// The walking env-step is ONE launch with every mapping AUTO can pick -- the task layer is fused into the one-link-per-lane, the
// one-leg-per-lane and the two-legs-per-lane kernels (16.9 us at 4096 envs, 33.9 us at 32 768; estimator -> physics -> reward as
// three launches measured 33.0 and 46.7 us).  Only an explicit LANE request keeps the three launches.
static bool walk_fused(const qg_sim *s) { const int m = effective_mapping(s); return m == QG_MAP_QUAD || m == QG_MAP_LINK || m == QG_MAP_PAIR; }

extern "C" int qg_walk_default_params(qg_walk_params *p) {
    if (!p) return fail(QG_ERR_ARG, "qg_walk_default_params: null output");
    memset(p, 0, sizeof *p);
    p->settling_time = 0.0;
    for (int i = 0; i < QG_NU; i++) {
        p->joint_centers[i] = (i % 3 == 2) ? -0.5 : 0.0;
        p->amp_target[i] = (i % 3 == 0) ? 1.5 : ((i % 3 == 1) ? 0.5 : 0.0);
        p->freq_target[i] = (i % 3 == 2) ? 0.0 : 1.0;
    }
    p->ema_alpha = 0.8;
    p->min_freq = 1.0;
    p->control_cost_alpha = 0.8;
    const double w[10] = {10.0, -2.0, 10.0, -50.0, 10.0, 10.0, -50.0, -1.0, -2.5, -8.0};
    for (int i = 0; i < 10; i++) p->w[i] = w[i];
    p->w_diff_ideal = -20.0;
    p->body_height = 0.13;
    return QG_OK;
}

extern "C" int qg_walk_destroy(qg_walk *w) {
    if (!w) return QG_OK;
    (void)hipSetDevice(w->sim->device);
    (void)hipDeviceSynchronize();                  // steps that read the task state may still be in flight on a caller's stream
    if (w->bound) {                                // give the sim back as qg_walk_create found it
        qg_sim *s = w->sim;
        s->task.use_flip = w->saved_use_flip;
        s->track_ctrl = w->saved_track_ctrl;
        s->walk_bound -= 1;
        KModel km;
        KTask kt;
        if (build_tables(&s->model, &s->task, &km, &kt) == QG_OK) (void)hipMemcpy(s->d_task, &kt, sizeof kt, hipMemcpyHostToDevice);
    }
    void *ptrs[] = {w->st.vel, w->st.head, w->st.gvel, w->st.ideal, w->st.prev_ctrl, w->st.prev_ctrl_cost, w->st.has_ctrl_cost,
                    w->st.prev_derive, w->st.has_derive, w->st.calls, w->st.sig, w->st.bmax, w->st.bmin, w->st.smax, w->st.smin, w->st.cross, w->st.count,
                    w->st.f_est, w->st.a_est, w->st.eff_actions, w->d_obs, w->d_reward, w->d_comps, w->d_actions, w->d_tmp, w->d_done};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete w;
    return QG_OK;
}

extern "C" int qg_walk_create(qg_sim *s, const qg_walk_params *params, qg_walk **out) {
    if (!s || !out) return fail(QG_ERR_ARG, "qg_walk_create: null argument");
    *out = nullptr;
    if (s->obs_dim != QG_NSENSOR) return fail(QG_ERR_ARG, "qg_walk_create: the walking rewards read the 33-value sensordata (obs_mode QG_OBS_FULL)");
    // one task layer per simulator: a second one would save the flags the first has already switched (flip termination, data.ctrl
    // tracking) as "what the sim had", and whichever is destroyed first would switch them off under the other
    if (s->walk_bound) return fail(QG_ERR_ARG, "qg_walk_create: a walking task layer is already bound to this simulator (destroy it first)");
    if (s->res.active) return fail(QG_ERR_ARG, "qg_walk_create: the resident step mode is on (qg_resident_stop first)");
    qg_walk_params dp;
    if (!params) { qg_walk_default_params(&dp); params = &dp; }
    if (!(params->min_freq > 0) || !(params->ema_alpha >= 0 && params->ema_alpha <= 1)) return fail(QG_ERR_ARG, "qg_walk_create: bad estimator parameters");
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    qg_walk *w = new (std::nothrow) qg_walk();
    if (!w) return fail(QG_ERR_ALLOC, "out of host memory");
    memset(w, 0, sizeof *w);
    w->sim = s;
    w->params = *params;
    const double dt = s->model.timestep * s->task.frame_skip;        // walking_quad.py:56,93
    KWalkParams &k = w->kp;
    k.dt = (float)dt;
    k.inv_dt = (float)(1.0 / dt);
    int64_t settle = params->settling_time > 0 ? qg_time_limit_substeps_impl(s->model.timestep, params->settling_time) : 0;
    k.settle_substeps = (int32_t)(settle > INT32_MAX ? INT32_MAX : settle);
    {   // the reference puts no upper bound on the window (frame_skip 1 / 2 / 3 at the shipped timestep: 1000 / 500 / 334 samples);
        // only memory does: the ring holds window x 12 x n_envs samples.  (Checked as a double BEFORE the conversion: a tiny min_freq
        // would make the cast itself undefined.)
        const double w_exact = std::ceil(2.0 / (params->min_freq * dt));   // math_utils.py:26-28
        if (!(w_exact >= 1) || w_exact > 1e6) {
            delete w;
            return fail(QG_ERR_ARG, "qg_walk_create: estimator window %g outside 1..1000000 samples (min_freq * timestep * frame_skip)", w_exact);
        }
        k.window = (int32_t)w_exact;
    }
    k.ema_alpha = (float)params->ema_alpha;
    k.control_cost_alpha = (float)params->control_cost_alpha;
    for (int i = 0; i < 10; i++) k.w[i] = (float)params->w[i];
    k.w_diff_ideal = (float)params->w_diff_ideal;
    k.body_height = (float)params->body_height;
    for (int i = 0; i < QG_NU; i++) {
        k.joint_centers[i] = (float)params->joint_centers[i];
        k.amp_target[i] = (float)params->amp_target[i];
        k.freq_target[i] = (float)params->freq_target[i];
    }
    k.auto_reset = s->task.auto_reset;
    k.unit_zero = params->unit_zero ? 1 : 0;
    const size_t n = (size_t)s->n, W = (size_t)k.window;
#define WALLOC(ptr, bytes)                                                                   \
    do {                                                                                    \
        hipError_t e_ = hipMalloc((void **)&(ptr), (bytes));                                \
        if (e_ == hipSuccess) e_ = hipMemset((ptr), 0, (bytes));                            \
        if (e_ != hipSuccess) {                                                             \
            qg_walk_destroy(w);                                                             \
            return fail(QG_ERR_ALLOC, "hipMalloc(%zu): %s", (size_t)(bytes), hipGetErrorString(e_)); \
        }                                                                                   \
    } while (0)
    WALLOC(w->st.vel, 2 * n * 4); WALLOC(w->st.head, 2 * n * 4); WALLOC(w->st.gvel, 2 * n * 4); WALLOC(w->st.ideal, 2 * n * 4);
    WALLOC(w->st.prev_ctrl, 12 * n * 4); WALLOC(w->st.prev_ctrl_cost, n * 4); WALLOC(w->st.has_ctrl_cost, n);
    WALLOC(w->st.prev_derive, n * 4); WALLOC(w->st.has_derive, n); WALLOC(w->st.calls, n * 4);
    {   // the ring in whole blocks and all 16 summary slots, whatever the window: the estimator's loads are unconditional
        const size_t nb = (W + QG_WALK_BLOCK - 1) / QG_WALK_BLOCK, Wp = nb * QG_WALK_BLOCK;
        WALLOC(w->st.sig, Wp * 12 * n * 4); WALLOC(w->st.cross, Wp * 12 * n);
        const size_t nbs = nb > QG_WALK_MAXBLOCKS ? nb : QG_WALK_MAXBLOCKS;     // at least the 16 slots the unrolled rebuild reads
        w->ring_slots = Wp; w->summary_blocks = nbs;
        WALLOC(w->st.bmax, nbs * 12 * n * 4); WALLOC(w->st.bmin, nbs * 12 * n * 4);
        WALLOC(w->st.smax, (size_t)(QG_WALK_BLOCK + 1) * 12 * n * 4); WALLOC(w->st.smin, (size_t)(QG_WALK_BLOCK + 1) * 12 * n * 4);
    }
    WALLOC(w->st.count, 12 * n * 4);
    WALLOC(w->st.f_est, 12 * n * 4); WALLOC(w->st.a_est, 12 * n * 4);
    WALLOC(w->st.eff_actions, 12 * n * 4);
    WALLOC(w->d_obs, n * QG_NSENSOR * 4); WALLOC(w->d_reward, n * 4); WALLOC(w->d_comps, n * QG_NWALKREWARD * 4);
    WALLOC(w->d_actions, n * 12 * 4); WALLOC(w->d_tmp, n * 12 * 4); WALLOC(w->d_done, n);
#undef WALLOC
    // the reference's termination set for this env: flip or time limit (walking_quad.py:162-166); data.ctrl feeds the estimator
    w->saved_use_flip = s->task.use_flip;
    w->saved_track_ctrl = s->track_ctrl;
    w->bound = 1;
    s->walk_bound += 1;
    s->task.use_flip = 1;
    {
        KModel km;
        KTask kt;
        int rc = build_tables(&s->model, &s->task, &km, &kt);
        if (rc != QG_OK) { qg_walk_destroy(w); return rc; }
        hipError_t e = hipMemcpy(s->d_task, &kt, sizeof kt, hipMemcpyHostToDevice);
        if (e != hipSuccess) { qg_walk_destroy(w); return fail(QG_ERR_DEVICE, "task update: %s", hipGetErrorString(e)); }
    }
    s->track_ctrl = 1;
    *out = w;
    s->creating = 1;                     // the constructor's own reset does not count as an episode
    int rc = qg_walk_reset(w, nullptr, s->seed, 0);
    s->creating = 0;
    if (rc != QG_OK) { qg_walk_destroy(w); *out = nullptr; }
    return rc;
}

extern "C" int qg_walk_set_commands(qg_walk *w, const float *velocity_xy, const float *heading_xy) {
    if (!w || !velocity_xy || !heading_xy) return fail(QG_ERR_ARG, "qg_walk_set_commands: null argument");
    qg_sim *s = w->sim;
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    HIP_TRY(hipDeviceSynchronize(), QG_ERR_LAUNCH);
    size_t n = (size_t)s->n;
    float *host = new (std::nothrow) float[6 * n];
    if (!host) return fail(QG_ERR_ALLOC, "out of host memory");
    float *vel = host, *head = host + 2 * n, *gv = host + 4 * n;
    for (size_t i = 0; i < n; i++) {
        float v0 = velocity_xy[2 * i], v1 = velocity_xy[2 * i + 1], h0 = heading_xy[2 * i], h1 = heading_xy[2 * i + 1];
        vel[i] = v0; vel[n + i] = v1; head[i] = h0; head[n + i] = h1;
        gv[i] = h0 * v0 - h1 * v1;                    // control_inputs.py:14-27
        gv[n + i] = h1 * v0 + h0 * v1;
    }
    hipError_t e = hipMemcpy(w->st.vel, vel, 2 * n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(w->st.head, head, 2 * n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(w->st.gvel, gv, 2 * n * 4, hipMemcpyHostToDevice);
    delete[] host;
    if (e != hipSuccess) return fail(QG_ERR_DEVICE, "qg_walk_set_commands: %s", hipGetErrorString(e));
    return QG_OK;
}

// new commands for the envs `select` marks (device pointer, NULL = all); no-op without a sampler
static int walk_sample_commands(qg_walk *w, const uint8_t *select, hipStream_t st) {
    if (!w->kp.cmd_sample) return QG_OK;
    qg_sim *s = w->sim;
    int threads = 256, blocks = (s->n + threads - 1) / threads;
    hipLaunchKernelGGL(qg_walk_command_kernel, dim3(blocks), dim3(threads), 0, st, w->kp, w->st, s->n, select, s->seed, s->env_index_base,
                       (const int32_t *)s->st.episode);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(QG_ERR_LAUNCH, "qg_walk_command_kernel launch: %s", hipGetErrorString(e));
    return QG_OK;
}

extern "C" int qg_walk_set_command_sampler(qg_walk *w, const qg_command_sampler *c) {
    if (!w) return fail(QG_ERR_ARG, "null handle");
    KWalkParams &k = w->kp;
    HIP_TRY(hipSetDevice(w->sim->device), QG_ERR_DEVICE);
    HIP_TRY(hipDeviceSynchronize(), QG_ERR_LAUNCH);   // steps reading the old parameters may be in flight
    if (!c) { k.cmd_sample = 0; return QG_OK; }
    if (c->fixed & ~7u) return fail(QG_ERR_ARG, "qg_walk_set_command_sampler: unknown bits in `fixed`");
    if (!(c->fixed & QG_CMD_FIXED_SPEED) && !(std::fabs(c->min_speed) < 1e30 && std::fabs(c->max_speed) < 1e30))
        return fail(QG_ERR_ARG, "qg_walk_set_command_sampler: min_speed / max_speed must be finite");
    k.cmd_fixed = c->fixed;
    k.cmd_min_speed = (float)c->min_speed;
    k.cmd_max_speed = (float)c->max_speed;
    k.cmd_theta = (float)c->fixed_heading_angle;
    k.cmd_alpha = (float)c->fixed_velocity_angle;
    k.cmd_speed = (float)c->fixed_speed;
    k.cmd_sample = 1;
    return QG_OK;
}

extern "C" int qg_walk_get_commands(qg_walk *w, float *velocity_xy, float *heading_xy) {
    if (!w) return fail(QG_ERR_ARG, "null handle");
    qg_sim *s = w->sim;
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    HIP_TRY(hipDeviceSynchronize(), QG_ERR_LAUNCH);
    size_t n = (size_t)s->n;
    float *host = new (std::nothrow) float[2 * n];
    if (!host) return fail(QG_ERR_ALLOC, "out of host memory");
    float *dsts[2] = {velocity_xy, heading_xy};
    const float *srcs[2] = {w->st.vel, w->st.head};
    for (int a = 0; a < 2; a++) {
        if (!dsts[a]) continue;
        hipError_t e = hipMemcpy(host, srcs[a], 2 * n * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { delete[] host; return fail(QG_ERR_DEVICE, "qg_walk_get_commands: %s", hipGetErrorString(e)); }
        for (size_t i = 0; i < n; i++) { dsts[a][2 * i] = host[i]; dsts[a][2 * i + 1] = host[n + i]; }
    }
    delete[] host;
    return QG_OK;
}

extern "C" int qg_walk_reset(qg_walk *w, const uint8_t *mask, uint64_t seed, uint32_t flags) {
    if (!w) return fail(QG_ERR_ARG, "null handle");
    qg_sim *s = w->sim;
    int rc = qg_reset(s, mask, seed, flags);           // uploads the mask into s->d_mask
    if (rc != QG_OK) return rc;
    int threads = 256, blocks = (s->n + threads - 1) / threads;
    hipLaunchKernelGGL(qg_walk_reset_kernel, dim3(blocks), dim3(threads), 0, s->stream, w->kp, w->st, s->n, mask ? s->d_mask : nullptr);
    HIP_TRY(hipGetLastError(), QG_ERR_LAUNCH);
    if (!s->creating) {                                // walking_quad.py:121-122 (not for the constructor's own reset)
        rc = walk_sample_commands(w, mask ? s->d_mask : nullptr, s->stream);
        if (rc != QG_OK) return rc;
    }
    HIP_TRY(hipStreamSynchronize(s->stream), QG_ERR_LAUNCH);
    return QG_OK;
}

// pre + physics + post.  The commands of auto-reset envs are redrawn by the caller AFTER everything that still reads the old
// ones (the partially observable pack) has been launched.
static int walk_step_core(qg_walk *w, const float *actions, float *obs, float *reward, uint8_t *done, float *components, void *stream,
                          bool po_follows, const KPoLaunch *po_fused = nullptr) {
    qg_sim *s = w->sim;
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    hipStream_t st = (hipStream_t)stream;
    if (walk_fused(s)) {
        KWalkLaunch wl;
        wl.P = w->kp;
        wl.S = w->st;
        wl.comps = components;
        wl.sample = (w->kp.cmd_sample && !po_follows) ? 1 : 0;
        return launch_step(s, actions, obs, reward, done, nullptr, nullptr, st, &wl, po_fused);
    }
    if (po_fused) return fail(QG_ERR_ARG, "walk_step_core: no fused walking launch for this handle");
    int threads = 256;
    int total = 12 * s->n;
    hipLaunchKernelGGL(qg_walk_pre_kernel, dim3((total + threads - 1) / threads), dim3(threads), 0, st, w->kp, w->st, s->n, actions,
                       (const float *)s->st.ctrl, (const int32_t *)s->st.nstep);
    HIP_TRY(hipGetLastError(), QG_ERR_LAUNCH);
    int rc = launch_step(s, w->st.eff_actions, obs, reward, done, nullptr, nullptr, st);
    if (rc != QG_OK) return rc;
    hipLaunchKernelGGL(qg_walk_post_kernel, dim3((s->n + threads - 1) / threads), dim3(threads), 0, st, w->kp, w->st, s->n, (const float *)obs,
                       (const uint8_t *)done, reward, components, (w->kp.cmd_sample && !po_follows) ? 1 : 0, s->seed, s->env_index_base,
                       (const int32_t *)s->st.episode);
    HIP_TRY(hipGetLastError(), QG_ERR_LAUNCH);
    return QG_OK;
}

extern "C" int qg_walk_step_device(qg_walk *w, const float *actions, float *obs, float *reward, uint8_t *done, float *components, void *stream) {
    if (!w || !actions || !obs || !reward || !done) return fail(QG_ERR_ARG, "qg_walk_step_device: null argument");
    return walk_step_core(w, actions, obs, reward, done, components, stream, false);
}

extern "C" int qg_walk_step(qg_walk *w, const float *actions, float *obs, float *reward, uint8_t *done, float *components) {
    if (!w || !actions || !obs || !reward || !done) return fail(QG_ERR_ARG, "qg_walk_step: null argument");
    qg_sim *s = w->sim;
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    { int rc0 = wait_for_caller_streams(s); if (rc0 != QG_OK) return rc0; }
    size_t n = (size_t)s->n;
    size_t off = pin_align(n * QG_NU * sizeof(float));
    PinOut o_obs = {obs, off, n * QG_NSENSOR * 4};                     off += pin_align(o_obs.bytes);
    PinOut o_rew = {reward, off, n * 4};                               off += pin_align(o_rew.bytes);
    PinOut o_done = {done, off, n};                                    off += pin_align(o_done.bytes);
    PinOut o_comp = {components, off, n * QG_NWALKREWARD * 4};         off += pin_align(o_comp.bytes);
    int rc = pin_reserve(s, off);
    if (rc == QG_OK) rc = pin_actions_in(s, actions, w->d_actions);
    if (rc == QG_OK) rc = qg_walk_step_device(w, w->d_actions, w->d_obs, w->d_reward, w->d_done, components ? w->d_comps : nullptr, s->stream);
    if (rc == QG_OK) rc = pin_out_enqueue(s, o_obs, w->d_obs);
    if (rc == QG_OK) rc = pin_out_enqueue(s, o_rew, w->d_reward);
    if (rc == QG_OK) rc = pin_out_enqueue(s, o_done, w->d_done);
    if (rc == QG_OK) rc = pin_out_enqueue(s, o_comp, w->d_comps);
    if (rc != QG_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream), QG_ERR_LAUNCH);
    pin_out_finish(s, o_obs); pin_out_finish(s, o_rew); pin_out_finish(s, o_done); pin_out_finish(s, o_comp);
    return QG_OK;
}

extern "C" int qg_walk_get_estimates(qg_walk *w, float *f_est, float *a_est, float *ideal_xy) {
    if (!w) return fail(QG_ERR_ARG, "null handle");
    qg_sim *s = w->sim;
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    HIP_TRY(hipDeviceSynchronize(), QG_ERR_LAUNCH);
    int threads = 256;
    // the estimates live env-major ([n][12]) on the device, as the caller wants them; the ideal position is [2][n]
    if (f_est) HIP_TRY(hipMemcpy(f_est, w->st.f_est, (size_t)s->n * 12 * 4, hipMemcpyDeviceToHost), QG_ERR_DEVICE);
    if (a_est) HIP_TRY(hipMemcpy(a_est, w->st.a_est, (size_t)s->n * 12 * 4, hipMemcpyDeviceToHost), QG_ERR_DEVICE);
    if (ideal_xy) {
        int total = s->n * 2;
        hipLaunchKernelGGL(qg_transpose_out, dim3((total + threads - 1) / threads), dim3(threads), 0, s->stream, (const float *)w->st.ideal, w->d_tmp, s->n, 2);
        HIP_TRY(hipGetLastError(), QG_ERR_LAUNCH);
        HIP_TRY(hipMemcpyAsync(ideal_xy, w->d_tmp, (size_t)total * 4, hipMemcpyDeviceToHost, s->stream), QG_ERR_DEVICE);
        HIP_TRY(hipStreamSynchronize(s->stream), QG_ERR_LAUNCH);
    }
    return QG_OK;
}

// ---- task-layer snapshot / restore (checkpoint) -------------------------------------------------------------
// One opaque blob per layer: a header that pins what the bytes mean (layer, library layout version, n_envs, window) followed by the
// layer's device arrays in declaration order, byte for byte.  Restoring a blob into a layer of the same shape reproduces every later
// step bit for bit (tests/test_walking_gpu.py::test_task_state_snapshot_restores_bit_identical_rollouts).
struct QgBlobHeader { uint32_t magic, version; int32_t n, window; int64_t bytes; };
#define QG_BLOB_WALK 0x4b4c5751u   /* "QWLK" */
#define QG_BLOB_PO 0x4f505751u     /* "QWPO" */
#define QG_BLOB_VERSION 5u
struct QgField { void *ptr; size_t bytes; };

static int walk_fields(const qg_walk *w, QgField *f) {
    const size_t n = (size_t)w->sim->n, R = w->ring_slots, NB = w->summary_blocks;
    const KWalkState &S = w->st;
    const QgField all[] = {
        {S.vel, 2 * n * 4}, {S.head, 2 * n * 4}, {S.gvel, 2 * n * 4}, {S.ideal, 2 * n * 4}, {S.prev_ctrl, 12 * n * 4}, {S.prev_ctrl_cost, n * 4},
        {S.has_ctrl_cost, n}, {S.prev_derive, n * 4}, {S.has_derive, n}, {S.calls, n * 4}, {S.sig, R * 12 * n * 4}, {S.cross, R * 12 * n},
        {S.bmax, NB * 12 * n * 4}, {S.bmin, NB * 12 * n * 4}, {S.smax, (size_t)(QG_WALK_BLOCK + 1) * 12 * n * 4}, {S.smin, (size_t)(QG_WALK_BLOCK + 1) * 12 * n * 4},
        {S.count, 12 * n * 4}, {S.f_est, 12 * n * 4},
        {S.a_est, 12 * n * 4}, {S.eff_actions, 12 * n * 4}};
    const int k = (int)(sizeof all / sizeof all[0]);
    if (f) memcpy(f, all, sizeof all);
    return k;
}
#define QG_MAX_FIELDS 32
static int64_t blob_bytes(const QgField *f, int k) {
    size_t b = sizeof(QgBlobHeader);
    for (int i = 0; i < k; i++) b += f[i].bytes;
    return (int64_t)b;
}
static int blob_out(qg_sim *s, uint32_t magic, int32_t window, const QgField *f, int k, void *blob) {
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    HIP_TRY(hipDeviceSynchronize(), QG_ERR_LAUNCH);   // steps may be in flight on a caller's stream
    QgBlobHeader h = {magic, QG_BLOB_VERSION, s->n, window, blob_bytes(f, k)};
    uint8_t *p = (uint8_t *)blob;
    memcpy(p, &h, sizeof h);
    p += sizeof h;
    for (int i = 0; i < k; i++) {
        HIP_TRY(hipMemcpy(p, f[i].ptr, f[i].bytes, hipMemcpyDeviceToHost), QG_ERR_DEVICE);
        p += f[i].bytes;
    }
    return QG_OK;
}
static int blob_in(qg_sim *s, uint32_t magic, int32_t window, const QgField *f, int k, const void *blob, const char *who) {
    QgBlobHeader h;
    memcpy(&h, blob, sizeof h);
    if (h.magic != magic || h.version != QG_BLOB_VERSION) return fail(QG_ERR_ARG, "%s: not a snapshot of this layer / library version", who);
    if (h.n != s->n || h.window != window || h.bytes != blob_bytes(f, k))
        return fail(QG_ERR_ARG, "%s: the snapshot was taken from %d envs with window %d (%lld bytes); this layer has %d envs, window %d (%lld bytes)", who,
                    h.n, h.window, (long long)h.bytes, s->n, window, (long long)blob_bytes(f, k));
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    HIP_TRY(hipDeviceSynchronize(), QG_ERR_LAUNCH);
    const uint8_t *p = (const uint8_t *)blob + sizeof h;
    for (int i = 0; i < k; i++) {
        HIP_TRY(hipMemcpy(f[i].ptr, p, f[i].bytes, hipMemcpyHostToDevice), QG_ERR_DEVICE);
        p += f[i].bytes;
    }
    return QG_OK;
}

extern "C" int64_t qg_walk_state_bytes(const qg_walk *w) {
    if (!w) return fail(QG_ERR_ARG, "null handle");
    QgField f[QG_MAX_FIELDS];
    return blob_bytes(f, walk_fields(w, f));
}
extern "C" int qg_walk_get_state(qg_walk *w, void *blob) {
    if (!w || !blob) return fail(QG_ERR_ARG, "qg_walk_get_state: null argument");
    QgField f[QG_MAX_FIELDS];
    return blob_out(w->sim, QG_BLOB_WALK, w->kp.window, f, walk_fields(w, f), blob);
}
extern "C" int qg_walk_set_state(qg_walk *w, const void *blob) {
    if (!w || !blob) return fail(QG_ERR_ARG, "qg_walk_set_state: null argument");
    QgField f[QG_MAX_FIELDS];
    return blob_in(w->sim, QG_BLOB_WALK, w->kp.window, f, walk_fields(w, f), blob, "qg_walk_set_state");
}

// the reset streams of the simulator itself: the per-env episode counters and the batch seed that key every random draw of a
// (re)set -- what qg_get_state does not cover and a bit-exact resume under auto-reset needs
extern "C" int qg_get_reset_streams(qg_sim *s, int32_t *episode, uint64_t *seed) {
    if (!s) return fail(QG_ERR_ARG, "null handle");
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    { int rr = resident_retire(s); if (rr != QG_OK) return rr; }
    HIP_TRY(hipDeviceSynchronize(), QG_ERR_LAUNCH);
    if (episode) HIP_TRY(hipMemcpy(episode, s->st.episode, (size_t)s->n * sizeof(int32_t), hipMemcpyDeviceToHost), QG_ERR_DEVICE);
    if (seed) *seed = s->seed;
    return QG_OK;
}
extern "C" int qg_set_reset_streams(qg_sim *s, const int32_t *episode, uint64_t seed) {
    if (!s) return fail(QG_ERR_ARG, "null handle");
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    { int rr = resident_retire(s); if (rr != QG_OK) return rr; }
    HIP_TRY(hipDeviceSynchronize(), QG_ERR_LAUNCH);
    if (episode) HIP_TRY(hipMemcpy(s->st.episode, episode, (size_t)s->n * sizeof(int32_t), hipMemcpyHostToDevice), QG_ERR_DEVICE);
    s->seed = seed;
    return QG_OK;
}

// ------------------------------------------------------------------------------------------------------
// partially observable observation pack (qg_po.hip)
// ------------------------------------------------------------------------------------------------------
struct qg_po {
    qg_walk *walk;
    KPoParams kp;
    KPoState st;
    float *d_obs33, *d_out, *d_term;
};

extern "C" int qg_po_destroy(qg_po *p) {
    if (!p) return QG_OK;
    (void)hipSetDevice(p->walk->sim->device);
    (void)hipDeviceSynchronize();                  // steps that read or write the frame ring may still be in flight on a caller's stream
    void *ptrs[] = {p->st.orient, p->st.alias, p->st.nstep, p->st.stack, p->st.head, p->d_obs33, p->d_out, p->d_term};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    delete p;
    return QG_OK;
}

extern "C" int qg_po_obs_dim(const qg_po *p) { return p ? p->kp.window * QG_PO_FRAME : fail(QG_ERR_ARG, "null handle"); }

static int po_reset_kernel(qg_po *p, const uint8_t *dmask, float *d_out) {
    qg_sim *s = p->walk->sim;
    int threads = 256, blocks = (s->n + threads - 1) / threads;
    hipLaunchKernelGGL(qg_po_reset_kernel, dim3(blocks), dim3(threads), 0, s->stream, p->kp, p->st, s->n, dmask, (const float *)p->walk->st.vel,
                       (const float *)p->walk->st.head, d_out);
    HIP_TRY(hipGetLastError(), QG_ERR_LAUNCH);
    HIP_TRY(hipStreamSynchronize(s->stream), QG_ERR_LAUNCH);
    return QG_OK;
}

extern "C" int qg_po_create(qg_walk *w, int32_t obs_window, qg_po **out) {
    if (!w || !out) return fail(QG_ERR_ARG, "qg_po_create: null argument");
    *out = nullptr;
    if (obs_window < 1 || obs_window > 64) return fail(QG_ERR_ARG, "qg_po_create: obs_window must be in 1..64");
    qg_sim *s = w->sim;
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    qg_po *p = new (std::nothrow) qg_po();
    if (!p) return fail(QG_ERR_ALLOC, "out of host memory");
    memset(p, 0, sizeof *p);
    p->walk = w;
    KPoParams &k = p->kp;
    k.dt = (float)(s->model.timestep * s->task.frame_skip);          // po_walking_quad.py:18
    k.gain = 0.033f;                                                  // the library's default IMU gain
    // data.time > settling_time / 2 (:37): first substep count whose f64-accumulated clock exceeds it
    {
        double t = 0, half = w->params.settling_time / 2;
        int64_t c = 0;
        while (!(t > half) && c < INT32_MAX) { t += s->model.timestep; c++; }
        k.half_settle_substeps = (int32_t)c;
    }
    k.window = obs_window;
    k.frame_skip = s->task.frame_skip;
    k.auto_reset = s->task.auto_reset;
    for (int i = 0; i < QG_NU; i++) k.default_ctrl[i] = (float)s->task.default_ctrl[i];
    size_t n = (size_t)s->n, width = (size_t)obs_window * QG_PO_FRAME;
    hipError_t e = hipMalloc((void **)&p->st.orient, 4 * n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&p->st.alias, n);
    if (e == hipSuccess) e = hipMalloc((void **)&p->st.nstep, n * 4);
    // the ring keeps every frame twice (KPoState.stack); QG_PO_RING_SLACK bytes behind it: the fused forms' unpredicated 16-byte loads may
    // read that far past the last env's row (sized and asserted against the copy's batch shape next to QG_PO_COPY_K)
    if (e == hipSuccess) e = hipMalloc((void **)&p->st.stack, 2 * n * width * 4 + QG_PO_RING_SLACK);
    if (e == hipSuccess) e = hipMalloc((void **)&p->st.head, n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_obs33, n * QG_NSENSOR * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_out, n * width * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_term, n * width * 4);
    if (e == hipSuccess) e = hipMemset(p->st.alias, 0, n);
    if (e == hipSuccess) e = hipMemset(p->st.nstep, 0, n * 4);
    if (e == hipSuccess) e = hipMemset(p->st.stack, 0, 2 * n * width * 4 + QG_PO_RING_SLACK);
    if (e == hipSuccess) e = hipMemset(p->st.head, 0, n * 4);
    if (e == hipSuccess) {                                           // computed_orientation = [1, 0, 0, 0] (:19)
        float *h = new float[4 * n];
        for (size_t i = 0; i < n; i++) { h[i] = 1.f; h[n + i] = h[2 * n + i] = h[3 * n + i] = 0.f; }
        e = hipMemcpy(p->st.orient, h, 4 * n * 4, hipMemcpyHostToDevice);
        delete[] h;
    }
    if (e != hipSuccess) {
        qg_po_destroy(p);
        return fail(QG_ERR_ALLOC, "qg_po_create: %s", hipGetErrorString(e));
    }
    *out = p;
    return QG_OK;
}

extern "C" int qg_po_reset(qg_po *p, const uint8_t *mask, uint64_t seed, uint32_t flags, float *obs) {
    if (!p) return fail(QG_ERR_ARG, "null handle");
    qg_sim *s = p->walk->sim;
    // the reset frame shows the estimate and the command as they stand BEFORE the robots / commands are reset (:59-69)
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    HIP_TRY(hipDeviceSynchronize(), QG_ERR_LAUNCH);   // device-pointer steps may be in flight on a caller's stream
    if (mask) HIP_TRY(hipMemcpy(s->d_mask, mask, (size_t)s->n, hipMemcpyHostToDevice), QG_ERR_DEVICE);
    int rc = po_reset_kernel(p, mask ? s->d_mask : nullptr, p->d_out);
    if (rc != QG_OK) return rc;
    if (obs) HIP_TRY(hipMemcpy(obs, p->d_out, (size_t)s->n * p->kp.window * QG_PO_FRAME * 4, hipMemcpyDeviceToHost), QG_ERR_DEVICE);
    return qg_walk_reset(p->walk, mask, seed, flags);
}

extern "C" int qg_po_step_device(qg_po *p, const float *actions, float *obs, float *reward, uint8_t *done, float *components,
                                 float *terminal_obs, void *stream) {
    if (!p || !actions || !obs || !reward || !done) return fail(QG_ERR_ARG, "qg_po_step_device: null argument");
    qg_walk *w = p->walk;
    qg_sim *s = w->sim;
    // up to 4096 envs the whole step -- physics, walking task layer, observation pack -- is ONE launch
    // (QG_PO_UNFUSED=1 at qg_create keeps the separate observation-pack launch: the A/B and the parity test of the two forms)
    if (walk_fused(s) && po_fusable(s) && !s->po_unfused) {
        KPoLaunch pl;
        pl.P = p->kp;
        pl.S = p->st;
        pl.out = obs;
        pl.term_out = terminal_obs;
        pl.sample = w->kp.cmd_sample ? 1 : 0;
        return walk_step_core(w, actions, nullptr, reward, done, components, stream, true, &pl);
    }
    int rc = walk_step_core(w, actions, p->d_obs33, reward, done, components, stream, true);
    if (rc != QG_OK) return rc;
    int blocks = (s->n + QG_PO_ENVS - 1) / QG_PO_ENVS;
    hipLaunchKernelGGL(qg_po_frame_kernel, dim3(blocks), dim3(QG_PO_THREADS), 0, (hipStream_t)stream, p->kp, p->st, s->n, (const float *)p->d_obs33,
                       (const float *)w->st.eff_actions, (const float *)s->st.qpos, w->kp, w->st, (const uint8_t *)done, obs, terminal_obs,
                       w->kp.cmd_sample ? 1 : 0, s->seed, s->env_index_base, (const int32_t *)s->st.episode);
    HIP_TRY(hipGetLastError(), QG_ERR_LAUNCH);
    return QG_OK;
}

extern "C" int qg_po_step(qg_po *p, const float *actions, float *obs, float *reward, uint8_t *done, float *components, float *terminal_obs) {
    if (!p || !actions || !obs || !reward || !done) return fail(QG_ERR_ARG, "qg_po_step: null argument");
    qg_walk *w = p->walk;
    qg_sim *s = w->sim;
    HIP_TRY(hipSetDevice(s->device), QG_ERR_DEVICE);
    { int rc0 = wait_for_caller_streams(s); if (rc0 != QG_OK) return rc0; }
    size_t n = (size_t)s->n, width = (size_t)p->kp.window * QG_PO_FRAME;
    size_t off = pin_align(n * QG_NU * sizeof(float));
    PinOut o_obs = {obs, off, n * width * 4};                          off += pin_align(o_obs.bytes);
    PinOut o_rew = {reward, off, n * 4};                               off += pin_align(o_rew.bytes);
    PinOut o_done = {done, off, n};                                    off += pin_align(o_done.bytes);
    PinOut o_comp = {components, off, n * QG_NWALKREWARD * 4};         off += pin_align(o_comp.bytes);
    int rc = pin_reserve(s, off);
    if (rc == QG_OK) rc = pin_actions_in(s, actions, w->d_actions);
    if (rc == QG_OK) rc = qg_po_step_device(p, w->d_actions, p->d_out, w->d_reward, w->d_done, components ? w->d_comps : nullptr,
                                            terminal_obs ? p->d_term : nullptr, s->stream);
    if (rc == QG_OK) rc = pin_out_enqueue(s, o_obs, p->d_out);
    if (rc == QG_OK) rc = pin_out_enqueue(s, o_rew, w->d_reward);
    if (rc == QG_OK) rc = pin_out_enqueue(s, o_done, w->d_done);
    if (rc == QG_OK) rc = pin_out_enqueue(s, o_comp, w->d_comps);
    if (rc != QG_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream), QG_ERR_LAUNCH);
    pin_out_finish(s, o_obs); pin_out_finish(s, o_rew); pin_out_finish(s, o_done); pin_out_finish(s, o_comp);
    if (terminal_obs) {
        // the terminal stacks only exist for envs that finished: the [n][obs_dim] transfer (4.3 MB at 4096 envs and window 10 -- as much
        // as the observation itself) is skipped on the steps where none did
        bool any = false;
        for (size_t i = 0; i < n && !any; i++) any = done[i] != 0;
        if (any) HIP_TRY(hipMemcpy(terminal_obs, p->d_term, n * width * 4, hipMemcpyDeviceToHost), QG_ERR_DEVICE);
    }
    return QG_OK;
}

static int po_fields(const qg_po *p, QgField *f) {
    const size_t n = (size_t)p->walk->sim->n, width = (size_t)p->kp.window * QG_PO_FRAME;
    const QgField all[] = {{p->st.orient, 4 * n * 4}, {p->st.alias, n}, {p->st.nstep, n * 4}, {p->st.stack, 2 * n * width * 4}, {p->st.head, n * 4}};
    const int k = (int)(sizeof all / sizeof all[0]);
    if (f) memcpy(f, all, sizeof all);
    return k;
}
extern "C" int64_t qg_po_state_bytes(const qg_po *p) {
    if (!p) return fail(QG_ERR_ARG, "null handle");
    QgField f[QG_MAX_FIELDS];
    return blob_bytes(f, po_fields(p, f));
}
extern "C" int qg_po_get_state(qg_po *p, void *blob) {
    if (!p || !blob) return fail(QG_ERR_ARG, "qg_po_get_state: null argument");
    QgField f[QG_MAX_FIELDS];
    return blob_out(p->walk->sim, QG_BLOB_PO, p->kp.window, f, po_fields(p, f), blob);
}
extern "C" int qg_po_set_state(qg_po *p, const void *blob) {
    if (!p || !blob) return fail(QG_ERR_ARG, "qg_po_set_state: null argument");
    QgField f[QG_MAX_FIELDS];
    return blob_in(p->walk->sim, QG_BLOB_PO, p->kp.window, f, po_fields(p, f), blob, "qg_po_set_state");
}
