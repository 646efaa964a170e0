// qg_kernel_resident.hip -- the one-link-per-lane env-step kernel in a form that runs MANY env-steps per launch with the state in
// registers between them (round 4; the per-launch form is qg_step_kernel_link in qg_kernel_link.hip, whose substep this file reuses).
//
// The reference keeps an env's state in MjData across steps (src/envs/quadruped.py:163-165); the per-launch kernel re-loads 49 state
// floats per env at the head of every env-step and stores them at its tail, between two dependent dispatches.  Two ways out of that:
//   DOOR = false  qg_step_device_seq: ONE launch runs `count` env-steps on actions[count][n][12] and writes packed[count][n][D + 2]
//                 -- open-loop sequences (action repeat, evaluation of a planned sequence, K-step graphs);
//   DOOR = true   the RESIDENT form: the launch stays on the GPU and is handed each env-step through a mailbox -- a 64-bit door word
//                 that a tiny ring kernel on the caller's stream advances, action / output slots in device memory, arrival counters
//                 the ring kernel waits on -- so that a policy on the caller's stream stays in the loop (qg_resident_*).
// Both run the SAME arithmetic per env-step as the per-launch kernel, in the same order (the quaternion is re-normalised and the
// hinges' sines / cosines are re-evaluated at the head of every env-step, exactly what a fresh launch does with the state it loads),
// so their results are bit-identical to it (tests/test_resident_gpu.py).
//
// Safety of the resident form -- no unbounded wait anywhere:
//   * a wave waits for the door with s_sleep polling against the 100 MHz clock (s_memrealtime); when `idle_ticks` pass without a
//     ring it RETIRES the kernel: compare-and-swap of the door from exactly "k steps rung" to "k | STOP".  A ring that got in first
//     wins (the swap fails, the step runs); a retired door refuses rings (the ring kernel's own swap fails and it reports the steps
//     as not executed).  Every wave therefore runs exactly the steps 0 .. seq - 1 of the final door value, stores its state and
//     exits; the host sees `state = exited` in page-locked memory and launches again before it rings next;
//   * waves never wait for each other: no workgroup barrier in the loop (arrival is per wave), so no exit order can deadlock;
//   * the ring kernel's wait for the arrivals gives up after a period without progress and reports it.
// Out of scope of these forms (the launcher refuses, or falls back to per-step launches): the walking / observation-pack layers,
// un-lagged sensors, hinge jitter at auto-reset (a launch of its own behind every per-launch step), separate obs / reward / done
// outputs (packed rows only).  The sequence form also exists for the two-legs-per-lane and the one-leg-per-lane kernels (end of file).

#define QG_DOOR_STOP (1ull << 63)
#define QG_RES_SHARDS 32            // arrival counters, one 128-byte line each; wave w of the grid arrives at shard w % 32
#define QG_RES_RUNNING 1ull         // hstat[0]
#define QG_RES_EXIT_STOP 2ull       // retired on request (qg_resident_stop, or any entry point that needs the state in memory)
#define QG_RES_EXIT_IDLE 3ull       // retired itself: no ring within idle_ticks
#define QG_RES_RETIRING 4ull        // no ring for idle_ticks / 2: the kernel still takes rings, and leaves at idle_ticks if none comes.  The
                                    // host does not ring a kernel in this state (it retires it and launches again): a ring it enqueues
                                    // after seeing RUNNING therefore has idle_ticks / 2 to reach the GPU before the door can shut

struct KResident {
    unsigned long long *door;       // device: env-steps rung so far | QG_DOOR_STOP
    unsigned long long *done;       // device: [QG_RES_SHARDS] arrival counters (index 16 s), cumulative env-steps x waves
    unsigned long long *completed;  // device: env-steps the previous launches have completed (where this launch starts)
    unsigned long long *hstat;      // page-locked host memory: [0] QG_RES_*, [1] env-steps completed at exit, [2] env-steps of refused
                                    // rings, [3] rings that gave up waiting
    const float *actions;           // [slots][n][12]
    float *packed;                  // [slots][n][D + 2]
    int32_t slots;
    int32_t count;                  // DOOR = false: env-steps of this launch
    uint32_t idle_ticks;            // DOOR: give up waiting for a ring after this many ticks of the 100 MHz clock
    uint32_t ring_ticks;            // ring kernel: give up after this long without an arrival
};

DEV unsigned long long res_load_u64(const unsigned long long *p) {
    const unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // global_load_dwordx2 sc1
    return ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)v);
}

// an action of the mailbox: agent-scope load (global_load_dword sc1) -- the slot was written by another kernel while this one runs
DEV float res_load_action(const float *base, size_t byte_off) {
    return __hip_atomic_load((const __attribute__((address_space(1))) float *)((lk_gcbytes)base + byte_off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool BAKED, bool DOOR>
__global__ __launch_bounds__(QGK_WAVE * QGK_LINK_WAVES, 1) void qg_step_kernel_link_multi(const KModel *__restrict__ Mp, const KTask *__restrict__ T, KStepArgs P, KResident R) {
    __shared__ __attribute__((aligned(16))) float tile_all[QGK_LINK_WAVES][QGK_LINK_ENVS * 35];
    __shared__ KModel smodel;
    if constexpr (!BAKED) {
        const float *src = reinterpret_cast<const float *>(Mp);
        float *dst = reinterpret_cast<float *>(&smodel);
        for (int i = threadIdx.x; i < (int)(sizeof(KModel) / sizeof(float)); i += QGK_WAVE * QGK_LINK_WAVES) dst[i] = src[i];
        __syncthreads();                // before the loop: every wave reaches it exactly once
    }
    const KModel &C = BAKED ? QG_BAKED_MODEL : smodel;
    struct { int32_t frame_skip, limit_substeps, use_fall, use_flip, obs_mode, auto_reset; uint32_t reset_flags; float fall_height, w_forward, w_ctrl, alive_bonus;
             const float *default_ctrl; } Tk = {T->frame_skip, T->limit_substeps, T->use_fall, T->use_flip, T->obs_mode, T->auto_reset, T->reset_flags,
                                               T->fall_height, T->w_forward, T->w_ctrl, T->alive_bonus, T->default_ctrl};
    const int lane = threadIdx.x & (QGK_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    float *tile = tile_all[wave];
    const int r = lane & 3, k = (lane >> 2) & 3, el = lane >> 4;
    const int env0 = (blockIdx.x * QGK_LINK_WAVES + wave) * QGK_LINK_ENVS;
    const int n = P.n;
    const bool live = env0 + el < n;
    const int env = live ? env0 + el : n - 1;
    const bool lead_env = (lane & 15) == 0;
    const float cm = (k == 0) ? 1.f : (k == 2) ? -1.f : 0.f;
    const float sm = (k == 1) ? 1.f : (k == 3) ? -1.f : 0.f;
#include "qg_link_regs.inc"

    // ---- the state: loaded ONCE per launch ------------------------------------------------------------------------------------------
    BaseState B;
    const unsigned n4 = 4u * (unsigned)n, e4 = 4u * (unsigned)env;
    B.pw = v3(lk_ld(P.st.qpos, e4), lk_ld(P.st.qpos, n4 + e4), lk_ld(P.st.qpos, 2 * n4 + e4));
    B.qw = lk_ld(P.st.qpos, 3 * n4 + e4); B.qx = lk_ld(P.st.qpos, 4 * n4 + e4); B.qy = lk_ld(P.st.qpos, 5 * n4 + e4); B.qz = lk_ld(P.st.qpos, 6 * n4 + e4);
    B.vw = v3(lk_ld(P.st.qvel, e4), lk_ld(P.st.qvel, n4 + e4), lk_ld(P.st.qvel, 2 * n4 + e4));
    B.wb = v3(lk_ld(P.st.qvel, 3 * n4 + e4), lk_ld(P.st.qvel, 4 * n4 + e4), lk_ld(P.st.qvel, 5 * n4 + e4));
    int nstep = lk_ld(P.st.nstep, e4);
    int episode = lk_ld(P.st.episode, e4);
    const int rk = r < 3 ? r : 2;
    const int jch = 3 * k + rk;
    const unsigned j4 = (unsigned)jch * n4 + e4;
    HingeLane J;
    J.q = lk_ld(P.st.qpos, 7 * n4 + j4);
    J.qd = lk_ld(P.st.qvel, 6 * n4 + j4);
    J.act = lk_ld(P.st.act, j4);
    J.u = 0.f; J.sn = 0.f; J.cs = 1.f;
    const KLink &Lj = link_of<BAKED>(C, k, rk);
    const float clo = BAKED ? sel3(rk, C.link[0].ctrl_lo, C.link[1].ctrl_lo, C.link[2].ctrl_lo) : Lj.ctrl_lo;
    const float chi = BAKED ? sel3(rk, C.link[0].ctrl_hi, C.link[1].ctrl_hi, C.link[2].ctrl_hi) : Lj.ctrl_hi;
    const float ref = BAKED ? sel3(rk, C.link[0].ref, C.link[1].ref, C.link[2].ref) : Lj.ref;
    const float q0 = BAKED ? sel3(rk, C.qpos0[7], C.qpos0[8], C.qpos0[9]) : C.qpos0[7 + jch];
    const int od = Tk.obs_mode == 1 ? 21 : 33;
    const int row = od + 2;
    const int fs = Tk.frame_skip;
    const bool wch = live && r < 3;
    const bool lead = live && lead_env;
    float *srow = tile + el * 35;
    const int live_envs = max(0, min(QGK_LINK_ENVS, n - env0));
    const int total = live_envs * row;
    const unsigned slot_act = 12u * n4;                           // bytes of one action slot
    const size_t slot_out = (size_t)n * row;                      // floats of one output slot
    const unsigned a_off = 12u * e4 + 4u * (unsigned)jch;         // this lane's action within a slot

    unsigned long long kdone = DOOR ? *R.completed : 0ull;       // env-steps completed (the index of the next one)
    int slot = DOOR ? (int)(kdone % (unsigned long long)R.slots) : 0;
    float ctrl_reg = 0.f;
    bool stepped = false;
    unsigned long long seen = kdone;                              // DOOR: env-steps known to be rung (wave-uniform)
    unsigned unreported = 0;                                      // DOOR: env-steps this wave has finished since its last arrival
    unsigned long long exit_code = QG_RES_EXIT_STOP;
    float a_next = 0.f;
    bool have_next = false;                                       // wave-uniform
    const int wgrid = blockIdx.x * QGK_LINK_WAVES + wave;
    unsigned long long *my_done = DOOR ? R.done + 16 * (wgrid & (QG_RES_SHARDS - 1)) : nullptr;

    for (;;) {
        if constexpr (DOOR) {
            // everything rung so far is done: report it (the caller's ring waits for that) -- and at least every 16th env-step of a long
            // run-ahead, so that the ring sees the waves alive
            if (unreported && (seen <= kdone || unreported >= 16u)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's rows have left (write-through stores)
                if (lane == 0) (void)__hip_atomic_fetch_add(my_done, (unsigned long long)unreported, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unreported = 0;
                QG_MARK(5);                                                   // rows drained, arrival issued
            }
            if (seen <= kdone) {                                              // ... then wait for the next ring
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                bool go = false, announced = false;
                for (int polls = 0;; ++polls) {
                    const unsigned long long v = res_load_u64(R.door);
                    if ((v & ~QG_DOOR_STOP) > kdone) {
                        seen = v & ~QG_DOOR_STOP; go = true; exit_code = QG_RES_EXIT_STOP;
                        if (announced) __hip_atomic_store(R.hstat + 0, QG_RES_RUNNING, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                    if (v & QG_DOOR_STOP) break;
                    if ((polls & 3) != 3) {                                   // (the clock is a scalar memory read: every fourth poll)
                        if (polls < 256) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(32);
                        continue;
                    }
                    const unsigned long long idle = __builtin_amdgcn_s_memrealtime() - t0;
                    if (!announced && idle > (unsigned long long)(R.idle_ticks >> 1) && blockIdx.x == 0 && threadIdx.x == 0) {
                        // half the time-out has passed: tell the host (it will not ring a kernel in this state), keep taking rings
                        __hip_atomic_store(R.hstat + 0, QG_RES_RETIRING, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        announced = true;
                    }
                    if (idle > (unsigned long long)R.idle_ticks) {
                        // nobody rang: retire the kernel -- unless a ring gets in first (then the swap fails and the next poll sees it)
                        if (lane == 0) {
                            unsigned long long expect = kdone;
                            (void)__hip_atomic_compare_exchange_strong(R.door, &expect, kdone | QG_DOOR_STOP, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        exit_code = QG_RES_EXIT_IDLE;
                        continue;
                    }
                    if (polls < 256) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(32);
                }
                if (!go) break;
                QG_MARK(1);                                                   // the ring seen
            }
        } else {
            if (kdone >= (unsigned long long)R.count) break;
        }
        // ---- one env-step, as the per-launch kernel runs it on the state it has just loaded ------------------------------------------
        // the action: already on its way when the previous env-step knew this one to be rung (run-ahead, and every step of a sequence)
        float a_in;
        if (have_next) a_in = a_next;
        else if constexpr (DOOR) a_in = res_load_action(R.actions, (size_t)slot * slot_act + a_off);
        else a_in = lk_ld(R.actions + (size_t)kdone * (slot_act / 4u), a_off);
        have_next = DOOR ? seen > kdone + 1 : kdone + 1 < (unsigned long long)R.count;
        if (have_next) {
            if constexpr (DOOR) a_next = res_load_action(R.actions, (size_t)(slot + 1 >= R.slots ? 0 : slot + 1) * slot_act + a_off);
            else a_next = lk_ld(R.actions + (size_t)(kdone + 1) * (slot_act / 4u), a_off);
        }
#ifdef QG_PHASE_TIMES
        asm volatile("s_waitcnt vmcnt(0)" :: "v"(a_in) : "memory");
#endif
        QG_MARK(2);                                                 // the action in a register
        const float aclip = fminf(fmaxf(a_in, -1.f), 1.f);         // quadruped.py:160
        J.u = fminf(fmaxf(aclip, clo), chi);
        quat_unit(B);
        sincos_f(J.q - ref, J.sn, J.cs);
        float zaxis_z = 1.f;
        asm volatile(".p2align 6");
#pragma unroll 1
        for (int s = 0; s < fs; ++s) substep_link<BAKED>(C, cm, sm, r, lead_env, B, J, K, s == fs - 1, srow, k, zaxis_z);
        nstep += fs;
        QG_MARK(3);                                                 // physics done

        const float ssq = env_sum(r < 3 ? aclip * aclip : 0.f);
        const float c_fwd = Tk.w_forward * B.vw.x;
        const float c_ctl = Tk.w_ctrl * ssq;
        const float c_alive = Tk.alive_bonus;
        const float reward = reward_total(c_fwd, c_ctl, c_alive);
        bool done = nstep >= Tk.limit_substeps;
        if (Tk.use_fall) done = done || (B.pw.z < Tk.fall_height);
        {
            float probe = J.q + J.qd;
            probe = env_sum(r < 3 ? probe : 0.f) + B.pw.x + B.pw.y + B.pw.z + B.qw + B.vw.x + B.vw.y + B.vw.z + B.wb.x + B.wb.y + B.wb.z;
            done = done || state_is_bad(probe);
        }
        if (Tk.use_flip) done = done || (zaxis_z < 0.f);
        const bool rst = done && Tk.auto_reset;
        ctrl_reg = aclip;
        if (rst) {                                                  // the auto-reset of the per-launch kernel, on the registers
            B.pw = v3(C.qpos0[0], C.qpos0[1], C.qpos0[2]);
            B.qw = C.qpos0[3]; B.qx = C.qpos0[4]; B.qy = C.qpos0[5]; B.qz = C.qpos0[6];
            if (Tk.reset_flags & 1u) {
                float a = 6.283185307179586f * uniform24(P.seed, P.env_index_base + (uint64_t)env, (uint64_t)episode);
                float sn, cs;
                sincos_f(0.5f * a, sn, cs);
                B.qw = cs; B.qx = 0.f; B.qy = 0.f; B.qz = sn;
            }
            B.vw = v3(0.f, 0.f, 0.f);
            B.wb = v3(0.f, 0.f, 0.f);
            nstep = 0;
            episode += 1;
            J.q = q0; J.qd = 0.f; J.act = 0.f;
            ctrl_reg = Tk.default_ctrl[jch];
        }
        if (lead_env) {
            if (od == 21) { srow[18] = srow[30]; srow[19] = srow[31]; srow[20] = srow[32]; }
            srow[od] = reward; srow[od + 1] = done ? 1.f : 0.f;
        }
        wave_sync();
        {
            float *dst = R.packed + (DOOR ? (size_t)slot : (size_t)kdone) * slot_out + (size_t)env0 * row;
            float v[3];
            if (row == 35) {
#pragma unroll
                for (int u = 0; u < 3; ++u) v[u] = tile[min(lane + u * QGK_WAVE, QGK_LINK_ENVS * 35 - 1)];
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (lane + u * QGK_WAVE < total) {
                        if constexpr (DOOR) __hip_atomic_store(dst + lane + u * QGK_WAVE, v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // write-through (sc1)
                        else dst[lane + u * QGK_WAVE] = v[u];
                    }
            } else {
                float w[QGK_LINK_ENVS];
#pragma unroll
                for (int er = 0; er < QGK_LINK_ENVS; ++er) w[er] = tile[er * 35 + min(lane, 34)];
#pragma unroll
                for (int er = 0; er < QGK_LINK_ENVS; ++er)
                    if (er < live_envs && lane < row) {
                        if constexpr (DOOR) __hip_atomic_store(dst + er * row + lane, w[er], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else dst[er * row + lane] = w[er];
                    }
            }
        }
        wave_sync();                    // the tile's reads are done before the next env-step's last substep writes it again
        QG_MARK(4);                                                 // terminations, reward, tile, row stores issued
        stepped = true;
        ++kdone;
        if constexpr (DOOR) {
            ++unreported;
            if (++slot >= R.slots) slot = 0;
        }
    }

    // ---- leaving: the state goes back to memory, exactly as the per-launch kernel stores it -------------------------------------------
    if constexpr (DOOR) {
        if (unreported) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) (void)__hip_atomic_fetch_add(my_done, (unsigned long long)unreported, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (lead) {
        lk_st(P.st.qpos, e4, B.pw.x); lk_st(P.st.qpos, n4 + e4, B.pw.y); lk_st(P.st.qpos, 2 * n4 + e4, B.pw.z);
        lk_st(P.st.qpos, 3 * n4 + e4, B.qw); lk_st(P.st.qpos, 4 * n4 + e4, B.qx); lk_st(P.st.qpos, 5 * n4 + e4, B.qy); lk_st(P.st.qpos, 6 * n4 + e4, B.qz);
        lk_st(P.st.qvel, e4, B.vw.x); lk_st(P.st.qvel, n4 + e4, B.vw.y); lk_st(P.st.qvel, 2 * n4 + e4, B.vw.z);
        lk_st(P.st.qvel, 3 * n4 + e4, B.wb.x); lk_st(P.st.qvel, 4 * n4 + e4, B.wb.y); lk_st(P.st.qvel, 5 * n4 + e4, B.wb.z);
        lk_st(P.st.nstep, e4, nstep);
        lk_st(P.st.episode, e4, episode);
    }
    if (wch && stepped) {
        lk_st(P.st.qpos, 7 * n4 + j4, J.q);
        lk_st(P.st.qvel, 6 * n4 + j4, J.qd);
        lk_st(P.st.act, j4, J.act);
        if (P.track_ctrl) lk_st(P.st.ctrl, j4, ctrl_reg);
    }
    if constexpr (DOOR) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            *R.completed = kdone;
            __hip_atomic_store(R.hstat + 1, kdone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(R.hstat + 0, exit_code, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// The ring: `count` more env-steps are rung (door += count unless the kernel has retired), then the caller's stream waits here until
// every wave of the resident grid has reported them -- what follows on the stream (a policy) reads the step's rows.  One wave; lane
// s < QG_RES_SHARDS watches arrival shard s.  nwaves = waves of the resident grid.
// `hint` = the door value the host expects (env-steps rung through the API so far): right in an eager loop, where it saves the ring a
// load round trip in front of its compare-and-swap (0.4 us of the 0.8 the door took, phase clock); stale in a replayed graph, where
// the first swap fails and returns the value to go on with.
__global__ __launch_bounds__(QGK_WAVE) void qg_resident_ring_kernel(KResident R, unsigned count, unsigned nwaves, unsigned long long hint) {
    const int lane = threadIdx.x;
    unsigned long long target = 0;
    int ok = 0;
    QG_MARK(8);                                                     // ring kernel entered
    if (lane == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long v = hint;
        for (;;) {
            if (!(v & QG_DOOR_STOP)) {      // the swap can only lose against a stale hint or a retiring wave (v gains STOP)
                if (__hip_atomic_compare_exchange_strong(R.door, &v, v + count, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { target = v + count; ok = 1; break; }
                continue;
            }
            // The door is shut.  If the host has queued the next launch (it sets RUNNING before it does) the door opens shortly: wait
            // for it, with the ring's own deadline; if the kernel has left and nothing is queued these env-steps are NOT executed.
            if (__hip_atomic_load(R.hstat + 0, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != QG_RES_RUNNING) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)R.ring_ticks) break;
            __builtin_amdgcn_s_sleep(8);
            v = __hip_atomic_load(R.door, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!ok) {                          // say so: the next resident call of the host returns an error
            const unsigned long long lost = __hip_atomic_load(R.hstat + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(R.hstat + 2, lost + count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    ok = __shfl(ok, 0);
    if (!ok) return;
    QG_MARK(9);                                                     // door advanced
    target = ((unsigned long long)__shfl((unsigned)(target >> 32), 0) << 32) | (unsigned long long)(unsigned)__shfl((unsigned)target, 0);
    const unsigned long long members = lane < QG_RES_SHARDS && (unsigned)lane < nwaves ? (nwaves - lane + QG_RES_SHARDS - 1) / QG_RES_SHARDS : 0ull;
    const unsigned long long want = target * members;
    unsigned long long last = 0, t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned polls = 1;; ++polls) {
        const unsigned long long have = members ? __hip_atomic_load(R.done + 16 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        if (__all(have >= want)) break;
        if (polls & 31u) continue;                                              // (the clock is a scalar memory read: not in every round)
        if (__any(have != last)) t0 = __builtin_amdgcn_s_memrealtime();        // an arrival: the waves are alive
        last = have;
        if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)R.ring_ticks) {
            if (lane == 0) {
                const unsigned long long g = __hip_atomic_load(R.hstat + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(R.hstat + 3, g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            break;
        }
    }
    QG_MARK(10);                                                    // every shard has reported
}

// door control: op 0 sets STOP (the resident waves finish what is rung, store the state and exit), op 1 clears it (before the next launch)
__global__ void qg_resident_ctl_kernel(unsigned long long *door, int op) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (op == 0) (void)__hip_atomic_fetch_or(door, QG_DOOR_STOP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else (void)__hip_atomic_fetch_and(door, ~QG_DOOR_STOP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- the sequence form for the two-legs-per-lane kernel (16 385 .. 32 768 envs and >= 57 344: BASELINE configs 3 and 4's sizes) ----------
// qg_step_kernel_pair with an outer loop over env-steps, exactly as qg_step_kernel_link_multi<.., DOOR = false> is to qg_step_kernel_link:
// state loaded once and stored once, per env-step what a fresh launch does with the state it loads, bit-identical results.  Compiled-in
// robot, lagged sensors, packed rows (the launcher falls back to per-step launches otherwise).
template <int WAVES>
__global__ __launch_bounds__(QGK_WAVE * WAVES, 1) void qg_step_kernel_pair_multi(const KTask *__restrict__ T, KStepArgs P, KResident R) {
    __shared__ float tile_all[WAVES][QGK_PAIR_ENVS * 35];
    const KModel &C = QG_BAKED_MODEL;
    const int lane = threadIdx.x & (QGK_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    float *tile = tile_all[wave];
    const int half = lane & 1;
    const int el = lane >> 1;
    const int env0 = (blockIdx.x * WAVES + wave) * QGK_PAIR_ENVS;
    const int n = P.n;
    const bool live = env0 + el < n;
    int env = live ? env0 + el : n - 1;
    f2 cm, sm;
    cm.x = half ? -1.f : 1.f; cm.y = 0.f;
    sm.x = 0.f; sm.y = half ? -1.f : 1.f;
    struct { int32_t frame_skip, limit_substeps, use_fall, use_flip, obs_mode, auto_reset; uint32_t reset_flags; float fall_height, w_forward, w_ctrl, alive_bonus;
             const float *default_ctrl; } Tk = {T->frame_skip, T->limit_substeps, T->use_fall, T->use_flip, T->obs_mode, T->auto_reset, T->reset_flags,
                                               T->fall_height, T->w_forward, T->w_ctrl, T->alive_bonus, T->default_ctrl};

    BaseState B;
    B.pw = v3<float>(P.st.qpos[0 * n + env], P.st.qpos[1 * n + env], P.st.qpos[2 * n + env]);
    B.qw = P.st.qpos[3 * n + env]; B.qx = P.st.qpos[4 * n + env]; B.qy = P.st.qpos[5 * n + env]; B.qz = P.st.qpos[6 * n + env];
    B.vw = v3<float>(P.st.qvel[0 * n + env], P.st.qvel[1 * n + env], P.st.qvel[2 * n + env]);
    B.wb = v3<float>(P.st.qvel[3 * n + env], P.st.qvel[4 * n + env], P.st.qvel[5 * n + env]);
    int nstep = P.st.nstep[env];
    int episode = P.st.episode[env];
    LegPair L;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int j = 3 * (2 * half + c) + i;
            const float qq = P.st.qpos[(7 + j) * n + env], qv = P.st.qvel[(6 + j) * n + env], aa = P.st.act[j * n + env];
            if (c == 0) { L.q[i].x = qq; L.qd[i].x = qv; L.act[i].x = aa; L.u[i].x = 0.f; }
            else { L.q[i].y = qq; L.qd[i].y = qv; L.act[i].y = aa; L.u[i].y = 0.f; }
        }
    }
    const int od = Tk.obs_mode == 1 ? 21 : 33;
    const int row = od + 2;
    const int fs = Tk.frame_skip;
    const int live_envs = max(0, min(QGK_PAIR_ENVS, n - env0));
    const int total = live_envs * row;
    const bool lead = live && half == 0;
    float *srow = tile + el * 35;
    const size_t slot_act = (size_t)n * 12, slot_out = (size_t)n * row;
    float ctrl_reg[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float a_next[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool have_next = false;

    for (int kstep = 0; kstep < R.count; ++kstep) {
        float a_in[6];
        const float *ap = R.actions + (size_t)kstep * slot_act + (size_t)env * 12 + 6 * half;
#pragma unroll
        for (int c6 = 0; c6 < 6; ++c6) a_in[c6] = have_next ? a_next[c6] : ap[c6];
        have_next = kstep + 1 < R.count;
        if (have_next) {
#pragma unroll
            for (int c6 = 0; c6 < 6; ++c6) a_next[c6] = ap[slot_act + c6];
        }
        float aclip[6];
        float ssq = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float a = fminf(fmaxf(a_in[3 * c + i], -1.f), 1.f);    // quadruped.py:160
                aclip[3 * c + i] = a;
                ssq = fmaf(a, a, ssq);
                const float uu = fminf(fmaxf(a, C.link[i].ctrl_lo), C.link[i].ctrl_hi);
                if (c == 0) L.u[i].x = uu; else L.u[i].y = uu;
            }
        }
        ssq = pair_sum(ssq);
        quat_unit(B);
#pragma unroll
        for (int i = 0; i < 3; ++i) sincos_f(L.q[i] - f2(C.link[i].ref), L.sc[2 * i], L.sc[2 * i + 1]);
        float zaxis_z = 1.f;
#pragma unroll 1
        for (int s = 0; s < fs; ++s) substep_pair(C, cm, sm, B, L, s == fs - 1, srow, half, zaxis_z);
        nstep += fs;
        asm volatile("" : "+v"(env));

        const float c_fwd = Tk.w_forward * B.vw.x;
        const float c_ctl = Tk.w_ctrl * ssq;
        const float c_alive = Tk.alive_bonus;
        const float reward = reward_total(c_fwd, c_ctl, c_alive);
        bool done = nstep >= Tk.limit_substeps;
        if (Tk.use_fall) done = done || (B.pw.z < Tk.fall_height);
        {
            float probe = hsum(L.q[0]) + hsum(L.q[1]) + hsum(L.q[2]) + hsum(L.qd[0]) + hsum(L.qd[1]) + hsum(L.qd[2]);
            probe = pair_sum(probe) + B.pw.x + B.pw.y + B.pw.z + B.qw + B.vw.x + B.vw.y + B.vw.z + B.wb.x + B.wb.y + B.wb.z;
            done = done || state_is_bad(probe);
        }
        if (Tk.use_flip) done = done || (zaxis_z < 0.f);
        if (half == 0) {
            if (od == 21) { srow[18] = srow[30]; srow[19] = srow[31]; srow[20] = srow[32]; }
            srow[od] = reward; srow[od + 1] = done ? 1.f : 0.f;
        }
        wave_sync();
        {
            float *dst = R.packed + (size_t)kstep * slot_out + (size_t)env0 * row;
            if (row == 35) {
                for (int e = lane; e < total; e += QGK_WAVE) dst[e] = tile[e];
            } else {
                const unsigned magic = row == 23 ? 2850u : (65536u + row - 1) / row;
                for (int e = lane; e < total; e += QGK_WAVE) {
                    const int er = (int)(((unsigned)e * magic) >> 16), ec = e - er * row;
                    dst[e] = tile[er * 35 + ec];
                }
            }
        }
        wave_sync();
        const bool rst = done && Tk.auto_reset;
#pragma unroll
        for (int c6 = 0; c6 < 6; ++c6) ctrl_reg[c6] = aclip[c6];
        if (rst) {
            B.pw = v3<float>(C.qpos0[0], C.qpos0[1], C.qpos0[2]);
            B.qw = C.qpos0[3]; B.qx = C.qpos0[4]; B.qy = C.qpos0[5]; B.qz = C.qpos0[6];
            if (Tk.reset_flags & 1u) {
                float a = 6.283185307179586f * uniform24(P.seed, P.env_index_base + (uint64_t)env, (uint64_t)episode);
                float sn, cs;
                sincos_f(0.5f * a, sn, cs);
                B.qw = cs; B.qx = 0.f; B.qy = 0.f; B.qz = sn;
            }
            B.vw = v3<float>(0.f, 0.f, 0.f);
            B.wb = v3<float>(0.f, 0.f, 0.f);
            nstep = 0;
            episode += 1;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                L.q[i] = f2(C.qpos0[7 + i]); L.qd[i] = f2(0.f); L.act[i] = f2(0.f);
            }
#pragma unroll
            for (int c6 = 0; c6 < 6; ++c6) ctrl_reg[c6] = Tk.default_ctrl[6 * half + c6];
        }
    }

    if (lead) {
        P.st.qpos[0 * n + env] = B.pw.x; P.st.qpos[1 * n + env] = B.pw.y; P.st.qpos[2 * n + env] = B.pw.z;
        P.st.qpos[3 * n + env] = B.qw; P.st.qpos[4 * n + env] = B.qx; P.st.qpos[5 * n + env] = B.qy; P.st.qpos[6 * n + env] = B.qz;
        P.st.qvel[0 * n + env] = B.vw.x; P.st.qvel[1 * n + env] = B.vw.y; P.st.qvel[2 * n + env] = B.vw.z;
        P.st.qvel[3 * n + env] = B.wb.x; P.st.qvel[4 * n + env] = B.wb.y; P.st.qvel[5 * n + env] = B.wb.z;
        P.st.nstep[env] = nstep;
        P.st.episode[env] = episode;
    }
    if (live && R.count > 0) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int j = 3 * (2 * half + c) + i;
                P.st.qpos[(7 + j) * n + env] = c == 0 ? L.q[i].x : L.q[i].y;
                P.st.qvel[(6 + j) * n + env] = c == 0 ? L.qd[i].x : L.qd[i].y;
                P.st.act[j * n + env] = c == 0 ? L.act[i].x : L.act[i].y;
                if (P.track_ctrl) P.st.ctrl[j * n + env] = ctrl_reg[3 * c + i];
            }
        }
    }
}

// ---- ... and for the one-leg-per-lane kernel (4 097 .. 16 384 envs at one wave per SIMD, 32 769 .. 57 343 at two; any model numbers) ------
// qg_step_kernel_quad's plain path with an outer loop over env-steps; WPE / BAKED choose the same substep instantiation the per-launch
// launcher picks for the grid, so the bits are the per-launch kernel's.  Four-wave workgroups (every grid AUTO gives this mapping).
template <int WPE, bool BAKED>
__global__ __launch_bounds__(QGK_WAVE * 4, WPE) void qg_step_kernel_quad_multi(const KModel *__restrict__ Mp, const KTask *__restrict__ T, KStepArgs P, KResident R) {
    constexpr int WAVES = 4;
    __shared__ float tile_all[WAVES][QGK_QUAD_ENVS * 35];
    __shared__ KModel smodel;
    const int lane = threadIdx.x & (QGK_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    float *tile = tile_all[wave];
    if constexpr (!BAKED) {
        const float *src = reinterpret_cast<const float *>(Mp);
        float *dst = reinterpret_cast<float *>(&smodel);
        for (int i = threadIdx.x; i < (int)(sizeof(KModel) / sizeof(float)); i += QGK_WAVE * WAVES) dst[i] = src[i];
        __syncthreads();
    }
    const KModel &C = BAKED ? QG_BAKED_MODEL : smodel;
    const int k = lane & 3;
    const int el = lane >> 2;
    const int env0 = (blockIdx.x * WAVES + wave) * QGK_QUAD_ENVS;
    const int n = P.n;
    const bool live = env0 + el < n;
    const int env = live ? env0 + el : n - 1;
    const float cm = (k == 0) ? 1.f : (k == 2) ? -1.f : 0.f;
    const float sm = (k == 1) ? 1.f : (k == 3) ? -1.f : 0.f;
    struct { int32_t frame_skip, limit_substeps, use_fall, use_flip, obs_mode, auto_reset; uint32_t reset_flags; float fall_height, w_forward, w_ctrl, alive_bonus;
             const float *default_ctrl; } Tk = {T->frame_skip, T->limit_substeps, T->use_fall, T->use_flip, T->obs_mode, T->auto_reset, T->reset_flags,
                                               T->fall_height, T->w_forward, T->w_ctrl, T->alive_bonus, T->default_ctrl};

    BaseState B;
    B.pw = v3(P.st.qpos[0 * n + env], P.st.qpos[1 * n + env], P.st.qpos[2 * n + env]);
    B.qw = P.st.qpos[3 * n + env]; B.qx = P.st.qpos[4 * n + env]; B.qy = P.st.qpos[5 * n + env]; B.qz = P.st.qpos[6 * n + env];
    B.vw = v3(P.st.qvel[0 * n + env], P.st.qvel[1 * n + env], P.st.qvel[2 * n + env]);
    B.wb = v3(P.st.qvel[3 * n + env], P.st.qvel[4 * n + env], P.st.qvel[5 * n + env]);
    int nstep = P.st.nstep[env];
    int episode = P.st.episode[env];
    LegState L;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = 3 * k + i;
        L.q[i] = P.st.qpos[(7 + j) * n + env];
        L.qd[i] = P.st.qvel[(6 + j) * n + env];
        L.act[i] = P.st.act[j * n + env];
        L.u[i] = 0.f; L.sc[2 * i] = 0.f; L.sc[2 * i + 1] = 1.f;
    }
    const int od = Tk.obs_mode == 1 ? 21 : 33;
    const int row = od + 2;
    const int fs = Tk.frame_skip;
    const int live_envs = max(0, min(QGK_QUAD_ENVS, n - env0));
    const int total = live_envs * row;
    float *srow = tile + el * 35;
    const size_t slot_act = (size_t)n * 12, slot_out = (size_t)n * row;
    float ctrl_reg[3] = {0.f, 0.f, 0.f}, a_next[3] = {0.f, 0.f, 0.f};
    bool have_next = false;

    for (int kstep = 0; kstep < R.count; ++kstep) {
        const float *ap = R.actions + (size_t)kstep * slot_act + (size_t)env * 12 + 3 * k;
        float aclip[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float a_in = have_next ? a_next[i] : ap[i];
            const float a = fminf(fmaxf(a_in, -1.f), 1.f);    // quadruped.py:160
            aclip[i] = a;
            L.u[i] = fminf(fmaxf(a, link_of<BAKED>(C, k, i).ctrl_lo), link_of<BAKED>(C, k, i).ctrl_hi);
        }
        have_next = WPE == 1 && kstep + 1 < R.count;     // (two waves per SIMD: no registers to park the next action in; the partner covers the load)
        if (have_next) {
#pragma unroll
            for (int i = 0; i < 3; ++i) a_next[i] = ap[slot_act + i];
        }
        quat_unit(B);
#pragma unroll
        for (int i = 0; i < 3; ++i) sincos_f(L.q[i] - link_of<BAKED>(C, k, i).ref, L.sc[2 * i], L.sc[2 * i + 1]);
        float zaxis_z = 1.f;
        asm volatile(".p2align 6");
#pragma unroll 1
        for (int s = 0; s < fs; ++s) substep_quad<BAKED, (WPE > 1)>(C, cm, sm, B, L, s == fs - 1, srow, k, zaxis_z);
        nstep += fs;

        float ssq = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) ssq = fmaf(aclip[i], aclip[i], ssq);
        ssq = quad_sum(ssq);
        const float c_fwd = Tk.w_forward * B.vw.x;
        const float c_ctl = Tk.w_ctrl * ssq;
        const float c_alive = Tk.alive_bonus;
        const float reward = reward_total(c_fwd, c_ctl, c_alive);
        bool done = nstep >= Tk.limit_substeps;
        if (Tk.use_fall) done = done || (B.pw.z < Tk.fall_height);
        {
            float probe = L.q[0] + L.q[1] + L.q[2] + L.qd[0] + L.qd[1] + L.qd[2];
            probe = quad_sum(probe) + B.pw.x + B.pw.y + B.pw.z + B.qw + B.vw.x + B.vw.y + B.vw.z + B.wb.x + B.wb.y + B.wb.z;
            done = done || state_is_bad(probe);
        }
        if (Tk.use_flip) done = done || (zaxis_z < 0.f);
        if (k == 0) {
            if (od == 21) { srow[18] = srow[30]; srow[19] = srow[31]; srow[20] = srow[32]; }
            srow[od] = reward; srow[od + 1] = done ? 1.f : 0.f;
        }
        wave_sync();
        {
            float *dst = R.packed + (size_t)kstep * slot_out + (size_t)env0 * row;
            if (row == 35) {
                for (int e = lane; e < total; e += QGK_WAVE) dst[e] = tile[e];
            } else {
                const unsigned magic = row == 23 ? 2850u : (65536u + row - 1) / row;
                for (int e = lane; e < total; e += QGK_WAVE) {
                    const int er = (int)(((unsigned)e * magic) >> 16), ec = e - er * row;
                    dst[e] = tile[er * 35 + ec];
                }
            }
        }
        wave_sync();
        const bool rst = done && Tk.auto_reset;
        if constexpr (WPE == 1) {
#pragma unroll
            for (int i = 0; i < 3; ++i) ctrl_reg[i] = aclip[i];
        } else if (live && P.track_ctrl) {      // two waves per SIMD: data.ctrl goes out every env-step (as the per-launch kernel writes it) rather
#pragma unroll                                  // than riding through the substep loop in three more registers
            for (int i = 0; i < 3; ++i) P.st.ctrl[(3 * k + i) * n + env] = rst ? Tk.default_ctrl[3 * k + i] : aclip[i];
        }
        if (rst) {
            B.pw = v3(C.qpos0[0], C.qpos0[1], C.qpos0[2]);
            B.qw = C.qpos0[3]; B.qx = C.qpos0[4]; B.qy = C.qpos0[5]; B.qz = C.qpos0[6];
            if (Tk.reset_flags & 1u) {
                float a = 6.283185307179586f * uniform24(P.seed, P.env_index_base + (uint64_t)env, (uint64_t)episode);
                float sn, cs;
                sincos_f(0.5f * a, sn, cs);
                B.qw = cs; B.qx = 0.f; B.qy = 0.f; B.qz = sn;
            }
            B.vw = v3(0.f, 0.f, 0.f);
            B.wb = v3(0.f, 0.f, 0.f);
            nstep = 0;
            episode += 1;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                L.q[i] = C.qpos0[7 + (BAKED ? i : 3 * k + i)]; L.qd[i] = 0.f; L.act[i] = 0.f;
                if constexpr (WPE == 1) ctrl_reg[i] = Tk.default_ctrl[3 * k + i];
            }
        }
    }

    // the addresses of the final stores are derived from (env, leg) AFTER the loops: visible to the optimiser they are computed ahead of
    // them and carried through the substep loop (qg_step_kernel_quad does the same for its epilogue)
    int env_e = env, k_e = k;
    asm volatile("" : "+v"(env_e), "+v"(k_e));
    if (live && k_e == 0) {
        P.st.qpos[0 * n + env_e] = B.pw.x; P.st.qpos[1 * n + env_e] = B.pw.y; P.st.qpos[2 * n + env_e] = B.pw.z;
        P.st.qpos[3 * n + env_e] = B.qw; P.st.qpos[4 * n + env_e] = B.qx; P.st.qpos[5 * n + env_e] = B.qy; P.st.qpos[6 * n + env_e] = B.qz;
        P.st.qvel[0 * n + env_e] = B.vw.x; P.st.qvel[1 * n + env_e] = B.vw.y; P.st.qvel[2 * n + env_e] = B.vw.z;
        P.st.qvel[3 * n + env_e] = B.wb.x; P.st.qvel[4 * n + env_e] = B.wb.y; P.st.qvel[5 * n + env_e] = B.wb.z;
        P.st.nstep[env_e] = nstep;
        P.st.episode[env_e] = episode;
    }
    if (live && R.count > 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int j = 3 * k_e + i;
            P.st.qpos[(7 + j) * n + env_e] = L.q[i];
            P.st.qvel[(6 + j) * n + env_e] = L.qd[i];
            P.st.act[j * n + env_e] = L.act[i];
            if (WPE == 1 && P.track_ctrl) P.st.ctrl[j * n + env_e] = ctrl_reg[i];
        }
    }
}
