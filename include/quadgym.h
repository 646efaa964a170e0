/* quadgym.h -- C ABI of the MI355X-native batched quadruped simulator.
 *
 * This is the drop-in boundary for the one hot path of antopio26/quadruped-gym:
 * QuadrupedEnv.step() / reset() (src/envs/quadruped.py:115-182) and the four
 * call sites where that file hands the arithmetic to the third-party `mujoco`
 * package (src/envs/quadruped.py:59,60,120,165).  The reference has no FFI of
 * its own (it is plain Python over the mujoco bindings); each entry point below
 * names the reference call it replaces.  Bindings: ctypes stub in INTEGRATION.md.
 *
 * Conventions: extern "C", opaque handle, int status (0 = ok, <0 = error, text
 * from qg_last_error()), no exceptions cross the boundary, caller owns every
 * buffer it passes, the library owns the per-env device state.  One handle is
 * bound to one GPU; calls on a handle are not re-entrant.  There is NO CPU
 * backend: every compute entry point fails with QG_ERR_DEVICE when no HIP device
 * is usable.
 *
 * Ordering contract.  The *_device entry points (qg_step_device, qg_step_device_packed,
 * qg_walk_step_device, qg_po_step_device) enqueue on the caller's stream and return at once;
 * consecutive calls on ONE stream are ordered by that stream, calls on different streams are the
 * caller's to order.  Every other entry point that touches the per-env state (qg_reset,
 * qg_walk_reset, qg_po_reset, the host-pointer qg_step / qg_walk_step / qg_po_step,
 * qg_get_state / qg_set_state, the task-layer snapshots, the command and estimate accessors, the destroy calls) first waits
 * for ALL work on the handle's device (hipDeviceSynchronize), runs on the library's own stream and
 * returns when it has completed -- it can therefore follow device-pointer steps on any stream
 * without further synchronisation, and must not be called while a stream is being captured.
 * (The host-pointer steps skip that device-wide wait when no device-pointer call of this handle
 * has gone to a caller's stream since the last one: their own stream is synchronised at the end
 * of every call, so there is nothing to wait for.  Once a device-pointer step of the handle has
 * been captured into a hipGraph the wait is taken on every host-pointer call: replays enqueue
 * steps the library does not see.  Device-pointer steps on the legacy NULL stream must not overlap
 * another stream's capture: the library cannot ask the NULL stream whether it is being captured.)
 *
 * Layouts at the boundary (row-major, env-major -- what NumPy / torch hand over):
 *   actions  [n_envs][12] f32      obs   [n_envs][obs_dim] f32
 *   reward   [n_envs] f32          done  [n_envs] u8
 *   reward_components [n_envs][3] f32  (forward, control_cost, alive)
 *   packed   [n_envs][obs_dim + 2] f32 (obs, reward, done as 0/1) -- one buffer
 *            for the per-step RCCL gather.
 * State arrays (qg_get_state / qg_set_state): qpos [n][19], qvel [n][18],
 * act [n][12], ctrl [n][12] f32 and nstep [n] i32 (physics substeps since
 * reset; data.time = nstep * timestep).
 */
#ifndef QUADGYM_H
#define QUADGYM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QG_NBODY 13      /* FRAME + 4 x (fema, shin, foot)      quadruped.xml:62-142 */
#define QG_NLEG 4
#define QG_NJNT 12       /* hinge joints = actuators            quadruped.xml:156-172 */
#define QG_NQ 19         /* 3 pos + 4 quat + 12 hinge */
#define QG_NV 18
#define QG_NU 12
#define QG_NSENSOR 33    /* model.nsensordata                   quadruped.xml:174-217 */
#define QG_MAXCP 12      /* contact sample points per body (table width) */
#define QG_NREWARD 3

/* status codes */
#define QG_OK 0
#define QG_ERR_ARG (-1)
#define QG_ERR_DEVICE (-2)
#define QG_ERR_ALLOC (-3)
#define QG_ERR_LAUNCH (-4)

/* observation packs (qg_task.obs_mode) */
#define QG_OBS_FULL 0    /* the 33-value sensordata of quadruped.py:141-143 */
#define QG_OBS_IMU 1     /* jointpos 12 + accel 3 + gyro 3 + velocimeter 3 = 21 (BASELINE config 5) */

/* work mappings of the step kernel (qg_set_mapping) */
#define QG_MAP_AUTO 0    /* the measured optimum per size: LINK up to 4096 envs; above, for the compiled-in robot QUAD up to 16384 and
                            for 32769..57343, PAIR for 16385..32768 and >= 57344; QUAD for any other model numbers */
#define QG_MAP_LANE 1    /* one environment per wavefront lane (64 envs per wave), any model numbers */
#define QG_MAP_QUAD 2    /* one leg per lane, four lanes per environment (16 envs per wave), any model numbers */
#define QG_MAP_PAIR 3    /* two legs per lane as packed FP32 pairs, two lanes per environment (32 envs per wave);
                            compiled-in robot only: qg_set_mapping refuses it for other model numbers */
#define QG_MAP_LINK 4    /* one link per lane, sixteen lanes per environment (4 envs per wave), any model numbers; the reference's
                            lagged sensors only (otherwise the request falls back to QUAD) */

/* qg_reset flags */
#define QG_RESET_RANDOM_YAW 1u   /* walking_quad.py:68-75: qpos[3:7] = [cos a/2, 0, 0, sin a/2], a ~ U(0, 2 pi) */
#define QG_RESET_JOINT_JITTER 2u /* hinge j starts at qpos0 + reset_joint_jitter * U(-1, 1), clamped to its range (the reference's open
                                    "RANDOMIZE ENVIRONMENT - Starting pose, joints" item, TODO.md:8; SURVEY config 3 option) */

/* Robot constants: what mujoco.MjModel.from_xml_path (quadruped.py:59) compiles
 * out of scene.xml.  Topology is fixed (body 0 = FRAME with the free joint, body
 * 1+3k+{0,1,2} = fema/shin/foot of leg k, joint j drives body j+1); numbers are
 * data.  Units SI, angles in radians, quaternions (w,x,y,z). */
typedef struct qg_model {
    double timestep;                       /* 0.002 (engine default, quadruped.xml:4 sets none) */
    double gravity[3];
    int32_t body_parent[QG_NBODY];         /* -1 for FRAME */
    double body_pos[QG_NBODY][3];          /* body frame in the parent body frame */
    double body_quat[QG_NBODY][4];
    double body_mass[QG_NBODY];
    double body_ipos[QG_NBODY][3];         /* centre of mass in the body frame */
    double body_inertia[QG_NBODY][6];      /* xx yy zz xy xz yz about the COM, body axes */
    double jnt_axis[QG_NJNT][3];           /* in the child body frame; anchor = body origin */
    double jnt_ref[QG_NJNT];               /* qpos0 of the hinge; rotation applied = qpos - ref */
    double jnt_range[QG_NJNT][2];
    double jnt_damping[QG_NJNT];
    double jnt_armature[QG_NJNT];
    double free_damping;                   /* applies to all 6 base DoFs (childclass, quadruped.xml:9,62-63) */
    double free_armature;
    double act_kp[QG_NU], act_kv[QG_NU], act_gear[QG_NU], act_timeconst[QG_NU];
    double act_ctrlrange[QG_NU][2], act_forcerange[QG_NU][2];
    double limit_stiffness, limit_damping; /* soft joint limits (N m/rad, N m s/rad) */
    double limit_ramp;                     /* limit damping ramps in linearly over this penetration (rad) */
    int32_t ncp[QG_NBODY];                 /* must be 12 for FRAME, 8 for every link */
    double cp[QG_NBODY][QG_MAXCP][3];      /* contact sample points, body frame */
    double contact_stiffness;              /* N/m per sample point */
    double contact_damping;                /* N s/m per body in contact (implicit) */
    double contact_margin;                 /* contact starts at this height */
    double contact_friction;               /* Coulomb mu */
    double contact_ramp;                   /* contact damping ramps in linearly over this summed penetration (m) */
    double qpos0[QG_NQ];                   /* what mj_resetData restores (quadruped.py:120) */
} qg_model;

/* Task constants: QuadrupedEnv's constructor arguments and the README reward /
 * termination set (quadruped.py:40-52,97-100,149-151; README.md:64-90). */
typedef struct qg_task {
    int32_t frame_skip;        /* quadruped.py:44, default 4 */
    double max_time;           /* quadruped.py:43, default 10.0; reported as `terminated` */
    int32_t use_time_limit;    /* use_default_termination, quadruped.py:52,99-100 */
    int32_t use_fall;          /* README.md:86-89 */
    double fall_height;        /* README literal 0.2; the base starts at 0.13 (quadruped.xml:62) */
    int32_t use_flip;          /* walking_quad.py:156-166: body z axis (sensordata[29]) < 0 terminates */
    double w_forward;          /* reward = w_forward*qvel[0] + w_ctrl*sum(ctrl^2) + alive; README.md:65-72 */
    double w_ctrl;             /* -0.1 */
    double alive_bonus;        /* 1.0 */
    int32_t obs_mode;          /* QG_OBS_FULL | QG_OBS_IMU */
    int32_t sensor_lag;        /* 1 = sensors describe the start of the last substep (mj_step order) */
    int32_t auto_reset;        /* 1 = envs that finish are reset inside the step (VecEnv semantics) */
    uint32_t reset_flags;      /* QG_RESET_* applied by auto-reset */
    double default_ctrl[QG_NU];/* quadruped.py:124: [0, 0, -0.5] * 4 */
    double reset_joint_jitter; /* [rad] half-width of the QG_RESET_JOINT_JITTER draw, default 0.1 */
} qg_task;

typedef struct qg_sim qg_sim;  /* opaque */

const char *qg_version(void);
/* 16 hex digits: hash of the sources this library was built from (kernels, C ABI, model tables).  Measurement tooling stamps
 * committed profiles with it; bench.py flags a roofline entry that was measured on other sources (roofline.profile_stale). */
const char *qg_build_id(void);
const char *qg_last_error(void);
/* The batch size at the top of the stair `n_envs` stands on: the step time is a staircase in the batch size (4096 / 16 384 / every
 * further 32 768 envs on the 1024 SIMDs of an MI355X -- one wave per SIMD of the kernel AUTO picks), so this many envs cost no more
 * per step than `n_envs` do.  4097 envs cost 55 % more per step than 4096.  device_id < 0: an MI355X. */
int32_t qg_recommended_batch(int32_t n_envs, int32_t device_id);
/* PCI bus id ("0000:05:00.0") of HIP device `device_id` into out[len >= 16] -- what bench.py's N > 1 line lists per rank, so that
 * "did RCCL see N ranks on N distinct GPUs" can be read off the line. */
int qg_device_pci_bus_id(int32_t device_id, char *out, int32_t len);

/* Fill with the constants compiled from the reference model (include/qg_model_data.h). */
int qg_default_model(qg_model *out);
int qg_default_task(qg_task *out);

/* Substep count at which `data.time >= max_time` first holds when time is
 * accumulated as the reference's engine does it (f64, time += timestep per
 * substep); quadruped.py:149-151. */
int64_t qg_time_limit_substeps(double timestep, double max_time);

/* Replaces MjModel.from_xml_path + MjData (quadruped.py:59-60).  `env_index_base`
 * is the global index of this handle's env 0 (shards of one batch get disjoint
 * ranges so per-env random streams do not depend on the sharding).  Random draws at reset come from a counter-based
 * stream keyed by (seed, global env index, number of resets that env has gone through). */
int qg_create(int32_t n_envs, int32_t device_id, const qg_model *model, const qg_task *task,
              uint64_t env_index_base, qg_sim **out);
int qg_destroy(qg_sim *sim);
int qg_num_envs(const qg_sim *sim);
int qg_obs_dim(const qg_sim *sim);

/* Replaces QuadrupedEnv.reset (quadruped.py:115-139): mj_resetData, time = 0,
 * ctrl = default.  mask == NULL resets every env, else only mask[i] != 0
 * (host pointer, n_envs bytes).  The first observation is all zeros, as in the
 * reference (no mj_forward after mj_resetData).  `seed` keys the reset random
 * streams (yaw, hinge jitter, commands) of every env of the batch, auto-resets
 * included: it is adopted by a whole-batch reset only (mask == NULL); a masked
 * reset ignores it and draws from the streams already in force. */
int qg_reset(qg_sim *sim, const uint8_t *mask, uint64_t seed, uint32_t flags);

/* Replaces QuadrupedEnv.step (quadruped.py:153-182) for the whole batch: clip to
 * [-1, 1], frame_skip x {ctrl = action; mj_step}, sensor pack, rewards,
 * terminations.  Host-pointer form (copies in and out, synchronous). */
int qg_step(qg_sim *sim, const float *actions, float *obs, float *reward, uint8_t *done,
            float *reward_components /* nullable */);

/* Device-pointer forms: every pointer is device memory on the handle's GPU,
 * `stream` is a hipStream_t (NULL = default stream); asynchronous. */
int qg_step_device(qg_sim *sim, const float *actions, float *obs, float *reward, uint8_t *done,
                   float *reward_components /* nullable */, void *stream);
int qg_step_device_packed(qg_sim *sim, const float *actions, float *packed, void *stream);

/* Snapshot / restore of data.qpos, qvel, act, ctrl and the substep counter
 * (host pointers; any may be NULL).  Used by the parity tests and checkpoints. */
int qg_get_state(qg_sim *sim, float *qpos, float *qvel, float *act, float *ctrl, int32_t *nstep);
/* qg_step followed by qg_get_state in ONE call and one synchronisation (any of the state pointers may be NULL): what an env that
 * mirrors the state on the host after every step needs -- the reference's `env.data`, which user reward / termination callables
 * read (quadruped.py:170-178). */
int qg_step_mirror(qg_sim *sim, const float *actions, float *obs, float *reward, uint8_t *done, float *reward_components, float *qpos,
                   float *qvel, float *act, float *ctrl, int32_t *nstep);
int qg_set_state(qg_sim *sim, const float *qpos, const float *qvel, const float *act, const float *ctrl,
                 const int32_t *nstep);

/* Time the step kernel alone: `iters` launches back to back on the handle's own
 * stream bracketed by HIP events on that stream; returns the mean milliseconds
 * per launch in *ms_per_launch.  State advances `iters` env-steps. */
int qg_time_step_kernel(qg_sim *sim, const float *d_actions, float *d_packed, int32_t iters, float *ms_per_launch);

/* Replace the task constants of a live handle -- what assigning env.reward_fns / env.termination_fns / env.max_time after
 * construction does in the reference (README.md:74-89): reward weights, fall / flip / time-limit terminations, auto-reset
 * and its flags, frame_skip.  obs_mode is fixed at qg_create.  Refused while a walking task layer is bound.  Takes effect
 * from the next step; waits for steps in flight. */
int qg_set_task(qg_sim *sim, const qg_task *task);
int qg_get_task(const qg_sim *sim, qg_task *out);

/* data.ctrl (the last env-clipped action, quadruped.py:164) is written back each step only
 * while this is on (default on; bulk-throughput callers switch it off). */
int qg_set_track_ctrl(qg_sim *sim, int32_t on);
/* Development builds only (make CXXFLAGS+=-DQG_PHASE_TIMES; tools/phase_times.py): the 100 MHz clock stamps the first wave of the last
 * one-link-per-lane launch took at its phase marks.  Production builds carry no such code and return QG_ERR_ARG. */
int qg_debug_phase_times(uint64_t out[16]);

/* 1 when the handle's model equals the compiled-in default (include/qg_model_data.h) and the
 * kernel variant with those constants baked into the instruction stream runs; 0 for any other
 * numbers (generic variant, tables read from device memory). */
int qg_uses_baked_model(const qg_sim *sim);

/* Choose how environments map onto wavefront lanes (QG_MAP_*); results agree to rounding.
 * qg_get_mapping returns the mapping the next step will actually use (LANE, QUAD or PAIR). */
int qg_set_mapping(qg_sim *sim, int32_t mapping);
int qg_get_mapping(const qg_sim *sim);

/* ---- many env-steps per launch (round 4; packed rows, no task layer) -----------------------------------------------------------
 * The reference keeps an env's state in MjData across steps (quadruped.py:163-165: the hot loop re-reads nothing); the per-launch
 * step kernel re-loads and stores it around every env-step.  Two forms keep it in registers instead:
 *
 * qg_step_device_seq: ONE launch runs `count` env-steps on actions[count][n_envs][12] and writes packed[count][n_envs][obs_dim + 2]
 * (device pointers, `stream` as for qg_step_device) -- open-loop sequences (action repeat, a planned sequence, K-step graphs).
 * Results are bit-identical to `count` calls of qg_step_device_packed.  Every mapping AUTO picks has its one-launch form (one link
 * per lane up to 4096 envs, one leg per lane, two legs per lane); where none applies (the one-env-per-lane mapping, un-lagged
 * sensors, hinge jitter at auto-reset, explicit mapping requests on small grids) the call IS those `count` launches, so it means
 * the same for every handle.  The resident form below exists for the one-link-per-lane mapping (<= 4096 envs) only.
 *
 * The RESIDENT form (opt-in): qg_resident_start launches the step kernel once on the library's own stream; it stays on the GPU and
 * is handed each env-step through a mailbox in device memory -- `slots` action buffers [n_envs][12] and `slots` output buffers
 * [n_envs][obs_dim + 2] (qg_resident_buffers), env-step i of the resident sequence (counted from qg_resident_start) uses slot
 * i % slots.  qg_resident_step_device(count, stream) enqueues a RING on the caller's stream: it makes the next `count` env-steps
 * runnable and holds the stream until their rows are in memory -- a policy on the same stream stays in the loop (read the rows,
 * write the next slot's actions, ring); `count` > 1 lets the kernel run ahead through slots the caller filled beforehand.
 * Nothing in it waits without a deadline: a kernel that is not rung for `idle_timeout_us` (0 = 2000; 50 .. 100000) stores the
 * state and leaves; the next qg_resident_step_device (or qg_resident_ensure, for graph replays) launches it again.  After HALF the
 * time-out without a ring the kernel tells the host that it is about to leave (it still takes rings); the host then never rings it
 * but retires it and launches again (tens of microseconds, synchronous) -- so a ring enqueued by qg_resident_step_device has half
 * the time-out to reach the GPU before the door can shut on it.  A ring that does meet a retired kernel (a stream backlog longer
 * than that, a replayed graph without qg_resident_ensure) does NOT run its steps; the next resident call returns QG_ERR_LAUNCH and
 * says how many (the state is that of the last executed step).  Every entry point that needs the state in memory (reset, get / set_state, set_task, the per-launch
 * steps, destroy) first retires the kernel; qg_resident_stop retires it and frees the mailbox.
 * Measured: closed-loop rings cost MORE per env-step than a kernel launch; the form pays for run-ahead only. */
int qg_step_device_seq(qg_sim *sim, const float *actions, float *packed, int32_t count, void *stream);
/* actions / packed: the mailbox's slot buffers, [slots][n_envs][12] and [slots][n_envs][obs_dim + 2] f32 in device memory that the
 * caller owns and keeps alive until qg_resident_stop -- or both NULL: the library allocates them (qg_resident_buffers). */
int qg_resident_start(qg_sim *sim, int32_t slots, int32_t idle_timeout_us, float *actions, float *packed);
int qg_resident_stop(qg_sim *sim);
int qg_resident_buffers(qg_sim *sim, float **actions, float **packed, int32_t *slots);
int qg_resident_step_device(qg_sim *sim, int32_t count, void *stream);
/* Launch the resident kernel again if it has retired (idle): call before replaying a hipGraph that holds captured rings. */
int qg_resident_ensure(qg_sim *sim);
/* No synchronisation: env-steps rung through the API, whether the kernel is on the GPU, the env-steps completed when it last left,
 * env-steps of rings that met a retired kernel (cumulative).  Any output may be NULL. */
int qg_resident_status(qg_sim *sim, int64_t *rung, int32_t *running, int64_t *completed_at_exit, int64_t *not_executed);

/* ---- native per-step exchange over RCCL (opt-in) --------------------------------------------------------------
 * Env batches shard across the GPUs of a node, one process per GPU, no exchange inside the physics; once per env-step
 * the packed [n_envs][obs_dim + 2] rows are gathered to `root` over xGMI.  These entry points issue that gather from C
 * (RCCL loaded with dlopen, no link-time dependency) so that the host cost per step is a few microseconds instead of
 * torch.distributed's ~30; the unique id is created on one rank and handed to the others by whatever channel the
 * caller has (bench.py broadcasts it through torch.distributed).  Validated with a 1-rank communicator only so far. */
#define QG_COMM_ID_BYTES 128
typedef struct qg_comm qg_comm;
int qg_comm_unique_id(uint8_t id[QG_COMM_ID_BYTES]);
int qg_comm_create(qg_sim *sim, int32_t rank, int32_t world, const uint8_t id[QG_COMM_ID_BYTES], qg_comm **out);
int qg_comm_destroy(qg_comm *comm);
/* `steps` env-steps with one gather each, double-buffered and overlapped: step k reads actions[k % n_actions] (device
 * pointers, host array of n_actions pointers), writes packed[k & 1] and gathers it into gathered[k & 1]
 * ([world][n_envs][obs_dim + 2] on the root; ignored elsewhere).  Returns when everything is enqueued. */
int qg_comm_rollout(qg_comm *comm, const float *const *actions, int32_t n_actions, float *const packed[2],
                    float *const gathered[2], int32_t steps, int32_t root);
/* Block the host until the communicator's streams are idle. */
int qg_comm_synchronize(qg_comm *comm);

/* ---- walking task layer ----------------------------------------------
 * What WalkingQuadrupedEnv adds around QuadrupedEnv.step() (src/envs/walking_quad.py): the velocity /
 * heading command (src/envs/control_inputs.py), the settling-time action mask (:142-143), the online
 * frequency / amplitude estimator of the control signal (src/envs/math_utils.py:11-158), the 11-term
 * reward of input_control_reward (:352-428) and the flip termination (:156-166).  A qg_walk is bound to
 * one qg_sim (which must use the 33-value observation) and keeps the per-env task state on the device. */
#define QG_NWALKREWARD 11   /* walking_quad.py:332-351 reward_keys */

typedef struct qg_walk_params {
    double settling_time;            /* walking_quad.py:11,142-143 */
    double joint_centers[QG_NU];     /* :36-39  [0, 0, -0.5] * 4 */
    double ema_alpha, min_freq;      /* :54-59  0.8, 1 Hz (window = ceil(2 / (min_freq * dt))) */
    double control_cost_alpha;       /* :254    0.8 */
    double w[10];                    /* :362-373 weights of the ten value terms, in reward_keys order */
    double w_diff_ideal;             /* :383    -20 */
    double body_height;              /* :369    0.13 */
    double amp_target[QG_NU];        /* :279-285 [1.5, 0.5, 0] * 4 */
    double freq_target[QG_NU];       /* :272-277 [1, 1, 0] * 4 */
    int32_t unit_zero;               /* 0 (default, the reference): unit() of an exactly zero vector is NaN (math_utils.py:7-8) and that
                                        NaN reaches progress_direction_reward_local and the total (walking_quad.py:197-205,422);
                                        1: the direction term is 0 when the local xy velocity or the commanded velocity is exactly
                                        zero.  Why an option: the engine's f64 state practically never holds an exact zero there, this
                                        f32 pipeline with its exact four-fold symmetry does on the first step of EVERY episode, and a
                                        NaN reward poisons a PPO update (train_quadruped.py:132-134) -- INTEGRATION.md section 4 */
} qg_walk_params;

typedef struct qg_walk qg_walk;

int qg_walk_default_params(qg_walk_params *out);
/* Binds the task layer to `sim` (switches its flip termination and data.ctrl tracking on; qg_walk_destroy switches both back
 * to what they were).  One layer per simulator: a second qg_walk_create on the same sim is refused.  The estimator window
 * ceil(2 / (min_freq * timestep * frame_skip)) has no upper bound other than memory (math_utils.py:26-28).  Lifetime: a qg_walk borrows its qg_sim and a qg_po borrows its qg_walk -- destroy them in the order
 * po, walk, sim. */
int qg_walk_create(qg_sim *sim, const qg_walk_params *params, qg_walk **out);
int qg_walk_destroy(qg_walk *walk);
/* control_inputs.py: per-env local velocity (vx, vy) and heading unit vector (cos, sin); host pointers [n][2]. */
int qg_walk_set_commands(qg_walk *walk, const float *velocity_xy, const float *heading_xy);
/* WalkingQuadrupedEnv.reset (walking_quad.py:96-126): resets the robots (as qg_reset) and the per-episode task
 * state; the estimator is NOT reset, as in the reference (:115). */
int qg_walk_reset(qg_walk *walk, const uint8_t *mask, uint64_t seed, uint32_t flags);
/* WalkingQuadrupedEnv.step (walking_quad.py:128-148).  components: [n][11] in reward_keys order, nullable. */
int qg_walk_step(qg_walk *walk, const float *actions, float *obs, float *reward, uint8_t *done, float *components);
int qg_walk_step_device(qg_walk *walk, const float *actions, float *obs, float *reward, uint8_t *done, float *components,
                        void *stream);
/* VelocityHeadingControls.sample(options) (control_inputs.py:74-115) on the device: with a sampler installed every
 * qg_walk_reset and every in-step auto-reset draws a new command for the envs it resets (random_controls,
 * walking_quad.py:121-122), from streams 13..15 of the env's (seed, global env index, episode) key -- the reference
 * draws from the global NumPy RNG, which a batch cannot reproduce.  theta, alpha ~ U(-pi, pi), speed ~ U(min, max)
 * unless the corresponding QG_CMD_FIXED_* bit selects the fixed value.  The new command takes effect after the
 * step's reward and (PO) observation have been produced, as in the reference.  NULL removes the sampler. */
#define QG_CMD_FIXED_HEADING 1u          /* options['fixed_heading_angle'] */
#define QG_CMD_FIXED_VELOCITY_ANGLE 2u   /* options['fixed_velocity_angle'] */
#define QG_CMD_FIXED_SPEED 4u            /* options['fixed_speed'] */
typedef struct qg_command_sampler {
    uint32_t fixed;                      /* QG_CMD_FIXED_* */
    double min_speed, max_speed;         /* defaults 0, 1 */
    double fixed_heading_angle, fixed_velocity_angle, fixed_speed;
} qg_command_sampler;
int qg_walk_set_command_sampler(qg_walk *walk, const qg_command_sampler *sampler);
/* Current commands, host pointers [n][2] each (either may be NULL). */
int qg_walk_get_commands(qg_walk *walk, float *velocity_xy, float *heading_xy);
/* Snapshot of the estimator outputs (f_est, a_est: [n][12], host pointers) and the ideal position ([n][2]). */
int qg_walk_get_estimates(qg_walk *walk, float *f_est, float *a_est, float *ideal_xy);

/* Task-layer snapshot / restore (checkpoint and resume; the reference only resumes the policy,
 * src/train_quadruped.py:114-141, its env state is rebuilt by reset()).  qg_get_state / qg_set_state cover the physics; these cover what
 * the walking layer keeps per env beyond it -- commands, ideal position, previous control and its first cost, the derived-term memory,
 * the whole estimator (ring, block summaries, counters, estimates) -- as ONE opaque blob of qg_walk_state_bytes() bytes (host pointer).
 * The layout is private to the library version and to (n_envs, window); set_state refuses a blob whose header does not match. */
int64_t qg_walk_state_bytes(const qg_walk *walk);
int qg_walk_get_state(qg_walk *walk, void *blob);
int qg_walk_set_state(qg_walk *walk, const void *blob);
/* The simulator's reset streams: per-env episode counters [n] and the batch seed, which key every random draw of a (re)set
 * (random yaw, hinge jitter, commands).  Together with qg_get_state and the blobs a restored run continues bit for bit,
 * auto-resets included.  Either output may be NULL; a NULL `episode` in the setter leaves the counters as they are. */
int qg_get_reset_streams(qg_sim *sim, int32_t *episode, uint64_t *seed);
int qg_set_reset_streams(qg_sim *sim, const int32_t *episode, uint64_t seed);

/* ---- partially observable observation pack ---------------------------------
 * POWalkingQuadrupedEnv (src/envs/po_walking_quad.py:10-90): per step one 26-value frame [gyro 3, accel 3,
 * Madgwick-IMU Euler angles 3, body_vel xy 2, data.ctrl 12, command vx vy theta 3], stacked over obs_window
 * steps (FIFO).  Sits on top of a qg_walk. */
#define QG_PO_FRAME_DIM 26

typedef struct qg_po qg_po;

int qg_po_create(qg_walk *walk, int32_t obs_window, qg_po **out);          /* po_walking_quad.py:10-27 */
int qg_po_destroy(qg_po *po);
int qg_po_obs_dim(const qg_po *po);                                         /* 26 * obs_window */
/* POWalkingQuadrupedEnv.reset (:59-69); obs (nullable, host pointer [n][obs_dim]) receives the stacked reset frames */
int qg_po_reset(qg_po *po, const uint8_t *mask, uint64_t seed, uint32_t flags, float *obs);
/* POWalkingQuadrupedEnv.step (:72-90).  obs: [n][obs_dim]; terminal_obs: nullable, receives the last stacked
 * observation of envs that finished (rows of other envs are left untouched); components: [n][11], nullable.
 * Up to 4096 envs of the built-in robot the whole step is ONE kernel launch (physics, walking task layer,
 * observation pack); `obs` must not alias `actions`: the rows are written from the first instructions of the launch on. */
int qg_po_step(qg_po *po, const float *actions, float *obs, float *reward, uint8_t *done, float *components, float *terminal_obs);
int qg_po_step_device(qg_po *po, const float *actions, float *obs, float *reward, uint8_t *done, float *components,
                      float *terminal_obs, void *stream);
/* Snapshot / restore of the observation pack's own state (orientation estimate, its aliasing flag, the frame ring): as qg_walk_get_state. */
int64_t qg_po_state_bytes(const qg_po *po);
int qg_po_get_state(qg_po *po, void *blob);
int qg_po_set_state(qg_po *po, const void *blob);

#ifdef __cplusplus
}
#endif
#endif /* QUADGYM_H */
