#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched quadruped simulator (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: QuadrupedEnv.step() semantics for every env
of the shard (clip, frame_skip x physics substep, sensor pack, rewards, terminations, auto-reset)
in one kernel launch, inputs (actions) and state already resident in HBM.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, env batches sharded (4096 per GPU, weak scaling), no exchange inside the
physics; per env-step ONE RCCL gather of the packed [envs, obs+2] f32 buffer (obs, reward, done) to
rank 0 over xGMI, issued on a communication stream and overlapped with the next env-step.  The K timed
steps are measured with the host issuing kernel + collective per step (eager); then, unless --exchange eager,
the same K steps are measured again as replays of a hipGraph of up to 16 env-steps under a watchdog, and the faster
of the two is reported (`config.exchange` says which; a failed or stalled attempt leaves the eager result).

Rank 0 prints ONE JSON line.  `roofline` is for the step kernel against HBM (algorithmic bytes per
env-step x envs / average kernel duration measured with HIP events on the kernel's stream);
`cpu_baseline` is the CPU oracle (a scalar C port; the reference's engine, mujoco, is not installed)
timed on this host.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# multi-process GPU work on this stack needs dmabuf IPC (the host driver has no legacy IPC): RCCL / tensor sharing across ranks fail
# with hipIpcGetMemHandle "invalid argument" otherwise.  Already exported on the boxes; set here too so a bare launcher inherits it.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
STALL_EXIT = 3                 # exit status when the hipGraph attempt of --exchange auto hangs the GPU (the eager line is still printed)
FP32_PEAK_TFLOPS = 157.3       # vector FP32 spec peak


def algorithmic_bytes_per_env_step(obs_dim: int) -> int:
    """f32 struct-of-arrays, all substeps fused, state read once / written once.
    read qpos 19 + qvel 18 + act 12 (196 B) + action 48 B + time 4 B; write state 196 B + time 4 B
    + obs 4*obs_dim + reward 4 + done 4  ->  588 B with the 33-value obs, 540 B with the 21-value pack."""
    return 196 + 48 + 4 + 196 + 4 + 4 * obs_dim + 4 + 4


def cpu_baseline(n_envs: int, frame_skip: int, seconds: float):
    """CPU oracle (oracle/qg_oracle.c, scalar C, f64) on the same workload, bounded sample."""
    from oracle import oracle as O
    model, task = O.default_model(), O.default_task()
    task.frame_skip = frame_skip
    task.use_fall = 1
    task.fall_height = 0.05
    n = min(n_envs, 256)
    rng = np.random.default_rng(0)
    batch = O.Batch(model, task, n)
    batch.reset()
    acts = rng.uniform(-1, 1, (8, n, 12))
    batch.step(acts[0])                        # warm
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < seconds:
        _, _, done, _ = batch.step(acts[steps % 8])
        if done.any():
            batch.reset(mask=done)
        steps += 1
    dt = time.perf_counter() - t0
    out = {"value": n * steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": f"{n} envs x {steps} env-steps (frame_skip {frame_skip}) of the same workload, single thread, {dt:.1f} s",
           "host_cpus": os.cpu_count(), "cpu_model": _cpu_model()}
    # all host cores: the same workload at the bench's own batch size, env range partitioned over host threads (OpenMP inside the
    # oracle's C batch loop; bit-identical to the single-thread run).  A GPU box may show 256 logical CPUs while the job's CPU
    # share is smaller (cgroup quota); more runnable threads than that share only thrash, so a few thread counts are tried --
    # the cgroup quota if there is one, 16, 64 and every CPU in the affinity mask -- and the best is reported with all of them listed
    try:
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        quota = None
        try:
            with open("/sys/fs/cgroup/cpu.max") as fh:
                q, per = fh.read().split()
                if q != "max":
                    quota = max(1, int(round(int(q) / int(per))))
        except (OSError, ValueError):
            pass
        cands = sorted({c for c in (quota, 16, 64, avail) if c and c <= avail})
        nb = max(n_envs, 4096)
        acts_b = rng.uniform(-1, 1, (4, nb, 12))
        tried = []
        for nthr in cands:
            big = O.Batch(model, task, nb)
            big.reset()
            for w in range(3):
                big.step(acts_b[w], threads=nthr)        # warm (thread pool start-up, first touch)
            t1 = time.perf_counter()
            k = 0
            budget = max(1.5, seconds / (2 * len(cands)))
            while time.perf_counter() - t1 < budget:
                _, _, d, _ = big.step(acts_b[k % 4], threads=nthr)
                if d.any():
                    big.reset(mask=d)
                k += 1
            dt2 = time.perf_counter() - t1
            tried.append({"threads": nthr, "value": nb * k / dt2, "sample": f"{nb} envs x {k} env-steps, {dt2:.1f} s"})
        best = max(tried, key=lambda e: e["value"])
        out["all_cores"] = {"value": best["value"], "cores": best["threads"], "logical_cpus": avail, "cgroup_cpu_quota": quota,
                            "sample": best["sample"] + ", OpenMP static partition", "tried": tried}
    except Exception as exc:       # the single-thread number stands on its own
        out["all_cores"] = {"error": repr(exc)}
    try:
        import mujoco  # noqa: F401
        out["mujoco"] = "importable on this host but not benchmarked: the reference env also needs gymnasium and cv2"
    except Exception:
        out["mujoco"] = "unavailable on this host (reference engine not installed; B0 row of BASELINE.md not measurable)"
    return out


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--frame-skip", type=int, default=4)
    ap.add_argument("--obs-mode", type=int, default=0, help="0: 33 sensors, 1: 21-value IMU+joint pack")
    ap.add_argument("--random-yaw", action="store_true", help="BASELINE config 3: random heading at every (re)set")
    ap.add_argument("--joint-jitter", type=float, default=0.0, metavar="RAD",
                    help="BASELINE config 3 option: hinges restart at qpos0 + RAD * U(-1,1) at every (re)set")
    ap.add_argument("--sync-gather", action="store_true", help="do not overlap the RCCL gather with the next step")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--walking", action="store_true", help="time the walking task layer too (SURVEY 8 f1): pre + physics + post kernels per env-step")
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="development only: every rank uses GPU 0 and the gather goes through gloo on CPU copies, to exercise the "
                         "multi-rank control flow on a one-GPU box (numbers are meaningless)")
    ap.add_argument("--generic-model", action="store_true", help="perturb one mass by 1e-3 so the table-driven (generic) kernel variant runs")
    ap.add_argument("--force-gather", action="store_true", help="run the per-step gather even with one rank (measures its host-side cost)")
    ap.add_argument("--gather-op", choices=["gather", "all_gather"], default="gather", help="collective used for the per-step exchange")
    ap.add_argument("--native-rccl", action="store_true",
                    help="issue the per-step gather from the library's C loop over RCCL (qg_comm_*) instead of torch.distributed; "
                         "validated with a 1-rank communicator only: opt-in")
    ap.add_argument("--graph", type=int, default=0, metavar="G",
                    help="capture G env-steps (kernel + per-step collective, double-buffered) into one hipGraph and replay it; "
                         "steps and warmup are rounded up to multiples of G.  Validated with a 1-rank RCCL group only: opt-in")
    ap.add_argument("--mapping", choices=["auto", "lane", "quad", "pair", "link"], default="auto", help="work mapping of the step kernel")
    ap.add_argument("--exchange", choices=["auto", "eager"], default="auto",
                    help="multi-GPU only.  eager: kernel launch + asynchronous RCCL gather issued per step from the host (~36 us of host "
                         "work per step).  auto (default): time the eager loop first, then ALSO try the same K steps as replays of a "
                         "hipGraph of 8 env-steps (kernel + collective, no per-step host work) under a watchdog, and report the faster "
                         "of the two; if capture or replay fails or stalls, the eager result is what is printed")
    ap.add_argument("--graph-timeout", type=float, default=90.0, help="seconds the hipGraph attempt of --exchange auto may take")
    ap.add_argument("--wakeup-ms", type=float, default=300.0,
                    help="milliseconds of unrelated GPU load (a scratch tensor rewritten in a loop) before the warm-up steps, to bring "
                         "the device out of its idle clocks; 0 disables it.  No env-step runs in it")
    ap.add_argument("--no-track-ctrl", action="store_true",
                    help="development only: skip the data.ctrl write-back (48 B/env) the reference's step maintains; such a line is "
                         "marked `ctrl_tracking: false` and is not the reported configuration")
    ap.add_argument("--config4-steps", type=int, default=-1, metavar="K4",
                    help="N > 1 only: after the headline, BASELINE config 4's per-GPU shard (32 768 envs, random yaw, one gather of "
                         "[32768, obs+2] f32 per env-step) is timed in the same process for K4 steps and attached as `config4`; "
                         "-1 (default) = max(--steps, 200), 0 = skip")
    ap.add_argument("--seq", type=int, default=0, metavar="S",
                    help="opt-in: ONE launch per S env-steps (qg_step_device_seq: the state stays in registers between the env-steps of "
                         "a launch; open-loop action sequences).  One GPU, plain step, one-link-per-lane mapping")
    ap.add_argument("--resident", choices=["closed", "ahead"], default=None,
                    help="opt-in: the RESIDENT step kernel (qg_resident_*): launched once, handed every env-step through a mailbox of 16 "
                         "action / output slots.  closed: one ring per env-step on the timed stream (a policy in the loop would sit "
                         "between two rings); ahead: one ring per 16 env-steps (the kernel runs ahead through the slots).  One GPU")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from quadruped_gym_amd import _abi
    from quadruped_gym_amd.sim import BatchedSim

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"WORLD_SIZE {world} != --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.rehearse_shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or (args.force_gather and not args.native_rccl)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.rehearse_shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # RCCL on ROCm

    group_info = None
    if use_dist:
        # who is in the communicator, asked of the communicator itself (one all-gather over it): config.rccl of the printed line
        import ctypes as C
        from quadruped_gym_amd.dist import describe_group
        pci = C.create_string_buffer(64)
        _abi.load_library().qg_device_pci_bus_id(local_rank, pci, 64)        # (an error leaves the buffer empty: the member shows no bus id)
        group_info = describe_group(local_rank, pci.value.decode(), "cpu" if args.rehearse_shared_gpu else dev)

    n = args.envs_per_gpu
    task = _abi.default_task()
    task.frame_skip = args.frame_skip
    task.obs_mode = args.obs_mode
    task.use_fall = 1
    task.fall_height = 0.05
    task.auto_reset = 1
    task.reset_flags = (_abi.RESET_RANDOM_YAW if args.random_yaw else 0) | (_abi.RESET_JOINT_JITTER if args.joint_jitter > 0 else 0)
    if args.joint_jitter > 0:
        task.reset_joint_jitter = args.joint_jitter
    model = None
    if args.generic_model:
        model = _abi.default_model()
        model.body_mass[3] *= 1.001
    sim = BatchedSim(n, device=local_rank, model=model, task=task, env_index_base=rank * n)
    # data.ctrl is written back every step, as the reference's step maintains it (quadruped.py:164): the timed region skips nothing
    sim.set_track_ctrl(not args.no_track_ctrl)
    sim.set_mapping({"auto": _abi.MAP_AUTO, "lane": _abi.MAP_LANE, "quad": _abi.MAP_QUAD, "pair": _abi.MAP_PAIR, "link": _abi.MAP_LINK}[args.mapping])
    mapping_name = {_abi.MAP_LANE: "one env per lane", _abi.MAP_QUAD: "one leg per lane (4 lanes per env)",
                    _abi.MAP_PAIR: "two legs per lane, packed f32 (2 lanes per env)",
                    _abi.MAP_LINK: "one link per lane (16 lanes per env)"}[sim.mapping]
    sim.reset(seed=0, flags=task.reset_flags)
    od = sim.obs_dim
    row = od + 2

    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)                         # Philox, U(-1, 1), a pool cycled through the run
    pool = [torch.rand((n, 12), generator=gen, device=dev) * 2 - 1 for _ in range(16)]
    packed = [torch.empty((n, row), device=dev) for _ in range(2)]
    compute = torch.cuda.current_stream(dev)
    gatherer = None
    native = None
    if args.native_rccl:
        import ctypes as C
        from quadruped_gym_amd._abi import check
        lib = _abi.load_library()
        uid = (C.c_uint8 * 128)()
        if rank == 0:
            check(lib.qg_comm_unique_id(uid), "qg_comm_unique_id")
        if world > 1:                                   # hand the id to the other ranks through torch.distributed
            t = torch.tensor(list(uid), dtype=torch.uint8, device=dev)
            dist.broadcast(t, src=0)
            for i, v in enumerate(t.cpu().tolist()):
                uid[i] = v
        native = C.c_void_p()
        check(lib.qg_comm_create(sim._h, rank, world, uid, C.byref(native)), "qg_comm_create")
        n_gath = [torch.empty((world, n, row), device=dev) for _ in range(2)] if rank == 0 else None
        a_arr = (C.c_void_p * len(pool))(*[p.data_ptr() for p in pool])
        p_arr = (C.c_void_p * 2)(packed[0].data_ptr(), packed[1].data_ptr())
        g_arr = (C.c_void_p * 2)(n_gath[0].data_ptr(), n_gath[1].data_ptr()) if rank == 0 else None
    elif use_dist:
        from quadruped_gym_amd.dist import PackedGatherer
        gatherer = PackedGatherer(n, row, "cpu" if args.rehearse_shared_gpu else dev, dst=0, op=args.gather_op)

    walk = None
    if args.walking:
        import ctypes as C
        from quadruped_gym_amd._abi import check
        lib = _abi.load_library()
        walk = C.c_void_p()
        check(lib.qg_walk_create(sim._h, None, C.byref(walk)), "qg_walk_create")
        cmd_v = np.tile(np.array([[0.3, 0.0]], np.float32), (n, 1))
        cmd_h = np.tile(np.array([[1.0, 0.0]], np.float32), (n, 1))
        check(lib.qg_walk_set_commands(walk, cmd_v.ctypes.data, cmd_h.ctypes.data), "qg_walk_set_commands")
        w_obs = torch.empty((n, 33), device=dev); w_rew = torch.empty(n, device=dev)
        w_done = torch.empty(n, device=dev, dtype=torch.uint8); w_comp = torch.empty((n, 11), device=dev)

    step_fn = sim.bind_step_packed(pool, packed, stream=compute)

    # ---- opt-in forms that keep the state in registers across env-steps (include/quadgym.h: qg_step_device_seq, qg_resident_*) ----
    multi = None
    if args.seq or args.resident:
        if world > 1 or use_dist or walk is not None or args.graph > 0 or (args.seq and args.resident):
            raise SystemExit("--seq / --resident: one GPU, plain step, no --graph, one of the two")
        chunk = args.seq if args.seq else (16 if args.resident == "ahead" else 1)
        slots = max(16, chunk)
        # the action pool IS the mailbox: [slots, n, 12], env-step i reads slot i % slots -- a fresh action buffer every step
        pool_t = torch.stack([pool[i % 16] for i in range(slots)]).contiguous()
        rows_t = torch.empty((slots, n, row), device=dev)
        args.steps = -(-args.steps // slots) * slots
        args.warmup = -(-max(args.warmup, slots) // slots) * slots
        if args.resident:
            sim.resident_start(pool_t, rows_t)
            ring = sim.bind_resident_step(chunk, stream=compute)
        multi = {"chunk": chunk, "slots": slots}

    def run_multi(k0, count):
        if args.resident:
            for _ in range(count // multi["chunk"]):
                ring()
            return
        S = multi["chunk"]
        for k in range(k0, k0 + count, S):
            o = k % multi["slots"]
            sim.step_device_seq(pool_t[o:o + S], rows_t[o:o + S], stream=compute)

    def run_eager(k0, count):
        for k in range(k0, k0 + count):
            b = k & 1
            if gatherer is not None:
                gatherer.wait_buffer_free(compute)       # the gather that read packed[b] two steps ago
            if walk is not None:
                check(lib.qg_walk_step_device(walk, pool[k & 15].data_ptr(), w_obs.data_ptr(), w_rew.data_ptr(), w_done.data_ptr(),
                                              None if os.environ.get("QG_BENCH_NO_COMPS") else w_comp.data_ptr(),
                                              C.c_void_p(compute.cuda_stream)), "qg_walk_step_device")
                continue
            step_fn(k & 15, b)
            if gatherer is not None:
                gatherer.submit(packed[b].cpu() if args.rehearse_shared_gpu else packed[b])   # RCCL gather on the communication stream (ordering by events, no host sync)
                if args.sync_gather:
                    gatherer.collect()                   # host waits for every gather: the un-overlapped reference point
                elif len(gatherer.pending) > 64:
                    del gatherer.pending[:-2]

    state = {"graph": None, "side": None, "G": 0}

    def build_graph(G):
        """G env-steps (kernel + asynchronous collective, double-buffered, joined at the end) captured into one hipGraph."""
        side = torch.cuda.Stream(dev)
        side.wait_stream(compute)
        with torch.cuda.stream(side):                    # warm-up on the capture stream (lazy RCCL initialisation happens here)
            warm = sim.bind_step_packed(pool, packed, stream=side)
            for k in range(G):
                if gatherer is not None:
                    gatherer.wait_buffer_free(side)
                warm(k & 15, k & 1)
                if gatherer is not None:
                    gatherer.submit(packed[k & 1])
            if gatherer is not None:
                gatherer.drain()
        compute.wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            cap = sim.bind_step_packed(pool, packed, stream=torch.cuda.current_stream(dev))
            for k in range(G):
                if gatherer is not None:
                    gatherer.wait_buffer_free()          # becomes a dependency edge of the graph
                cap(k & 15, k & 1)
                if gatherer is not None:
                    gatherer.submit(packed[k & 1])
            if gatherer is not None:
                for bb in (0, 1):                        # join the outstanding collectives before the graph ends
                    if gatherer.work[bb] is not None:
                        gatherer.work[bb].wait()
                        gatherer.work[bb] = None
                gatherer.pending.clear()
        state["graph"], state["side"], state["G"] = graph, side, G

    if args.graph > 0:
        if walk is not None or args.rehearse_shared_gpu:
            raise SystemExit("--graph serves the plain step (with or without the RCCL exchange)")
        G = args.graph + (args.graph & 1)                # whole double-buffer cycles
        args.steps = -(-args.steps // G) * G
        args.warmup = -(-max(args.warmup, G) // G) * G
        build_graph(G)

    def run(k0, count):
        if multi is not None:
            return run_multi(k0, count)
        if native is not None:
            check(lib.qg_comm_rollout(native, a_arr, len(pool), p_arr, g_arr, count, 0), "qg_comm_rollout")
            return
        if state["graph"] is None:
            return run_eager(k0, count)
        for _ in range(count // state["G"]):
            state["graph"].replay()

    def fence():
        if native is not None:
            check(lib.qg_comm_synchronize(native), "qg_comm_synchronize")
        if gatherer is not None:
            gatherer.drain()
        if args.resident:
            # the timed stream: a ring completes when its env-steps' rows are in memory, so this IS the fence of the measured work; a
            # device-wide wait would also wait for the resident kernel itself, i.e. until it leaves for lack of rings (2 ms)
            compute.synchronize()
        else:
            torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def timed(count):
        """EXACTLY `count` steps between two fences; returns (seconds, max over ranks; event-span ms per step on the kernel's stream)."""
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tstream = compute if state["graph"] is None else state["side"]
        if state["graph"] is not None:
            torch.cuda.set_stream(state["side"])
        t0 = time.perf_counter()
        ev0.record(tstream)
        run(args.warmup, count)
        ev1.record(tstream)
        fence()
        dt = time.perf_counter() - t0
        span_ms = ev0.elapsed_time(ev1) / count
        if use_dist:
            tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, span_ms

    # Device wake-up: after seconds of host-side set-up the GPU sits in its idle power state and needs sustained load of THIS kind to
    # settle at its running clocks -- far longer than W + K steps of 14 us when the driver asks for a handful of them (measured, same
    # box, W = 5, no wake-up: 14.9 us per launch over K = 20 steps, 14.5 over 400, 14.1 over 1000, 13.9 over 2000 after 200).  So
    # ~args.wakeup_ms of device work precede the W warm-up steps.  What that work is matters: a matrix-product chain (round 2's first
    # choice, QG_WAKEUP_KIND=alu) leaves the K = 20 launches at 14.2 us -- a heavy MFMA burst is not the load the governor then sees;
    # the step kernel itself on a SCRATCH handle (its own state: nothing the measurement touches) brings them to the 13.65 us of a long
    # run.  The line says so (config.device_wakeup); the W warm-up steps and the K timed steps on the measured handle follow as asked.
    wakeup_kind = None
    if args.wakeup_ms > 0:
        kind = wakeup_kind = os.environ.get("QG_WAKEUP_KIND", "sim")
        t_w = time.perf_counter()
        if kind == "sim":                                # the step kernel on a scratch handle
            scratch_sim = BatchedSim(n, device=local_rank, model=model, task=task, env_index_base=10 ** 9)
            scratch_sim.set_mapping(sim.mapping)
            scratch_sim.reset(seed=1, flags=task.reset_flags)
            sp = torch.empty((n, row), device=dev)
            while (time.perf_counter() - t_w) * 1e3 < args.wakeup_ms:
                for i in range(200):
                    scratch_sim.step_device_packed(pool[i & 15], sp, stream=compute)
                torch.cuda.synchronize(dev)
            scratch_sim.close()
            scratch = sp
        elif kind == "mem":
            scratch = torch.empty(1 << 24, device=dev)
            while (time.perf_counter() - t_w) * 1e3 < args.wakeup_ms:
                for _ in range(20):
                    scratch.add_(1.0)
                torch.cuda.synchronize(dev)
        else:
            scratch = torch.randn((2048, 2048), device=dev)
            while (time.perf_counter() - t_w) * 1e3 < args.wakeup_ms:
                for _ in range(10):
                    scratch = torch.tanh(scratch @ scratch * 1e-3)
                torch.cuda.synchronize(dev)
        del scratch
    run(0, args.warmup)
    fence()
    dt, kernel_ms = timed(args.steps)                    # HIP events on the stream the kernel runs on
    if native is not None:
        kernel_ms = dt / args.steps * 1e3                # the C loop runs on the library's own streams: wall clock per step
    exchange_mode = None
    if use_dist and native is None:
        exchange_mode = "hipGraph replay" if state["graph"] is not None else "eager"
        if walk is None:
            # with a collective per step the event span is the host's issue period, not the kernel: time the kernel by itself
            kernel_ms = sim.time_step_kernel(pool[0], packed[0], 200)

    qpos = sim.get_state()[0]
    healthy = bool(np.isfinite(qpos).all())

    MAP_KEY = {_abi.MAP_LANE: "lane", _abi.MAP_QUAD: "quad", _abi.MAP_PAIR: "pair", _abi.MAP_LINK: "link"}
    MAP_KERNEL = {_abi.MAP_LANE: "qg_step_kernel", _abi.MAP_QUAD: "qg_step_kernel_quad", _abi.MAP_PAIR: "qg_step_kernel_pair",
                  _abi.MAP_LINK: "qg_step_kernel_link"}
    MAP_ENVS_PER_WAVE = {_abi.MAP_LANE: 64, _abi.MAP_QUAD: 16, _abi.MAP_PAIR: 32, _abi.MAP_LINK: 4}
    def make_line(dt, kernel_ms, exchange_mode):
        # HBM traffic per launch: committed rocprofv3 PMC measurement of this same configuration, when there is one
        traffic, valu, flops = None, None, None
        stale, prof_id = None, None
        build_id = _abi.load_library().qg_build_id().decode()
        try:
            key = f"{MAP_KEY[sim.mapping]}_n{n}_fs{args.frame_skip}_obs{od}"
            # the walking layer and the table-driven kernels are other kernels with other traffic: their own entries, or none
            key += ("_walking" if args.walking else "") + ("" if sim.baked else "_generic")
            with open(os.path.join(ROOT, "profiles", "traffic_index.json")) as fh:
                idx = json.load(fh)
            ent = idx.get(key)
            if ent:
                traffic = ent["hbm_bytes_per_launch"]
                valu = ent.get("valu_insts_per_wave")
                # the counters are a committed measurement of this configuration, not of this run: flag an entry that was taken
                # on other sources than the library now loaded (entries are stamped by tools/update_traffic_index.py)
                prof_id = ent.get("build_id")
                stale = prof_id != build_id
            # counted FP32 flops per env-step (SQ_INSTS_VALU_FLOPS_FP32 of the step kernel / envs; FMA = 2): a property of the
            # kernel's instruction stream per substep, so the entry of the same mapping and frame_skip serves every batch size
            fkey = f"flops_{MAP_KEY[sim.mapping]}_fs{args.frame_skip}" + ("_walking" if args.walking else "") + ("" if sim.baked else "_generic")
            flops = idx.get(fkey)
            if flops is not None and flops.get("build_id") != build_id:
                stale = True
        except Exception:
            pass
        total_envs = n * world
        value = total_envs * args.steps / dt
        bytes_step = algorithmic_bytes_per_env_step(od)
        if multi is not None:
            # what these forms move per env-step: the action in, the packed row out; the state (and data.ctrl) once per LAUNCH
            traffic, valu, flops, stale, prof_id = None, None, None, None, None
            per_launch = 196 + 4 + 196 + 4 + 48
            bytes_step = 48 + 4 * od + 4 + 4 + (per_launch / multi["chunk"] if args.seq else 0)
        ach = bytes_step * n / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "env_steps_per_sec", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n} envs/GPU x {world} GPU, frame_skip={args.frame_skip}, flat ground, "
                                   f"forward+control_cost+alive rewards, fall(z<0.05)+time-limit terminations, auto-reset, "
                                   f"obs={od} f32, U(-1,1) actions resident in HBM"
                                   + (", random yaw at reset" if args.random_yaw else "")
                                   + (f", hinge jitter {args.joint_jitter} rad at reset" if args.joint_jitter > 0 else "")
                                   + (f", hipGraph of {args.graph} env-steps per replay" if args.graph > 0 else "")
                                   + (", cycled through a pool of 16 pre-generated action buffers")
                                   + (f"; OPT-IN sequence form: one launch per {args.seq} env-steps, state in registers in between" if args.seq else "")
                                   + ("; OPT-IN resident form: one launch for the run, every env-step handed over through a mailbox of "
                                      f"{multi['slots']} action / output slots, " + ("ONE ring kernel per env-step on the timed stream (closed loop)"
                                      if args.resident == "closed" else "one ring per 16 env-steps (run-ahead)") if args.resident else "")
                                   + (", WALKING task layer (estimator + 11-term reward + flip termination), one launch per env-step" if args.walking else "")
                                   + (f", per-step RCCL gather of [{n},{row}] f32 to rank 0 from the library's C loop (qg_comm_rollout, overlapped)" if native is not None else
                                      (f", per-step RCCL {args.gather_op} of [{n},{row}] f32 to rank 0 ({'sync' if args.sync_gather else 'overlapped'})" if use_dist else "")),
                       "envs_per_gpu": n, "frame_skip": args.frame_skip, "obs_dim": od, "mapping": mapping_name,
                       "constants": "baked literals" if sim.baked else "tables (LDS / scalar loads)"},
            "substeps_per_sec": value * args.frame_skip,
            "state_finite": healthy,
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": traffic, "profile_stale": stale, "profile_build_id": prof_id, "build_id": build_id,
                         "kernel": MAP_KERNEL[sim.mapping] if multi is None else
                                   {_abi.MAP_LINK: "qg_step_kernel_link_multi", _abi.MAP_PAIR: "qg_step_kernel_pair_multi", _abi.MAP_QUAD: "qg_step_kernel_quad_multi"}.get(
                                       sim.mapping, MAP_KERNEL[sim.mapping] + " (per-step launches: no one-launch form for this mapping)"),
                         "fence": "timed-stream synchronize (rings complete when their rows are out)" if args.resident else "device synchronize",
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": bytes_step * n,
                         "algorithmic_bytes_per_env_step": bytes_step,
                         "note": "VALU-issue-bound path (no dense contraction): ~0.6 KB of state traffic per env-step against "
                                 "tens of thousands of VALU lane-instructions; the HBM fraction is small by construction"},
        }
        if valu:
            # issue-rate view of the same launch: wave-instructions/s against 1024 SIMDs x (2.4 GHz / 2 cycles per wave64 VALU
            # instruction: SIMD-32; measured 2.2-2.5 with >= 2 resident waves, >= 4.5 for a lone wave, tools/ubench/valu_rate.hip)
            waves = -(-n // MAP_ENVS_PER_WAVE[sim.mapping])
            rate = valu * waves / (kernel_ms * 1e-3)
            line["roofline"]["valu_issue"] = {"insts_per_wave": valu, "waves": waves, "achieved_ginst_s": rate / 1e9,
                                              "peak_ginst_s": 1024 * 2.4 / 2, "frac": rate / (1024 * 1.2e9)}
        if flops:
            # the FP32 vector-ALU view -- counted flops per env-step x env-steps/s of the kernel against the 157.3 TFLOP/s
            # vector peak (which assumes every issue slot holds an FMA: a stream of dependent small solves cannot reach it)
            per = flops["flops_per_env_step"]
            tf = per * n / (kernel_ms * 1e-3) / 1e12
            line["roofline"]["fp32"] = {"flops_per_env_step": per, "achieved": tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                        "frac": tf / FP32_PEAK_TFLOPS, "flop_per_byte": per / bytes_step, "source": flops.get("source"),
                                        "note": "hardware-counted flops of the kernel that ran, every lane included (work the mapping "
                                                "replicates across lanes counts)"}
            base = idx.get(f"flops_quad_fs{args.frame_skip}")
            if base and sim.mapping != _abi.MAP_QUAD:
                # the same physics with the least replication (one leg per lane): the flops the env-step needs rather than the ones executed
                u = base["flops_per_env_step"] * n / (kernel_ms * 1e-3) / 1e12
                line["roofline"]["fp32"]["least_replicated"] = {"flops_per_env_step": base["flops_per_env_step"], "achieved": u,
                                                                "frac": u / FP32_PEAK_TFLOPS, "source": base.get("source")}
        line["config"]["ctrl_tracking"] = not args.no_track_ctrl
        line["config"]["device_wakeup_ms"] = args.wakeup_ms
        line["config"]["device_wakeup"] = {None: "none", "sim": "untimed launches of the step kernel on a scratch handle (separate state) before the W warm-up steps",
                                           "alu": "matrix-product chain on a scratch tensor", "mem": "streaming adds on a scratch tensor"}[wakeup_kind]
        if exchange_mode is not None:
            line["config"]["exchange"] = exchange_mode
        if group_info is not None:
            line["config"]["rccl"] = group_info
        return line

    line = make_line(dt, kernel_ms, exchange_mode)

    config4 = None
    # ---- N > 1: BASELINE config 4's shard in the same process (the driver only ever runs `bench.py --gpus N`) -- BEFORE the hipGraph
    # attempt below, so that a line printed by its watchdog carries it too ---------------------
    # 262 144 envs over 8 GPUs = 32 768 per GPU, random yaw at every (re)set, per env-step ONE gather of the packed [32 768, obs + 2]
    # f32 rows (4.59 MB per rank) to rank 0, double-buffered and overlapped exactly like the headline's (eager issue from the host).
    k4 = args.config4_steps if args.config4_steps >= 0 else max(args.steps, 200)
    if use_dist and native is None and walk is None and multi is None and k4 > 0 and gatherer is not None:
        n4 = 32768
        t4 = _abi.default_task()
        t4.frame_skip, t4.obs_mode, t4.use_fall, t4.fall_height, t4.auto_reset = args.frame_skip, args.obs_mode, 1, 0.05, 1
        t4.reset_flags = _abi.RESET_RANDOM_YAW
        torch.cuda.set_stream(compute)
        sim4 = BatchedSim(n4, device=local_rank, model=model, task=t4, env_index_base=rank * n4)
        sim4.set_track_ctrl(not args.no_track_ctrl)
        sim4.reset(seed=0, flags=t4.reset_flags)
        pool4 = [torch.rand((n4, 12), generator=gen, device=dev) * 2 - 1 for _ in range(4)]
        packed4 = [torch.empty((n4, row), device=dev) for _ in range(2)]
        from quadruped_gym_amd.dist import PackedGatherer
        g4 = PackedGatherer(n4, row, "cpu" if args.rehearse_shared_gpu else dev, dst=0, op=args.gather_op)
        step4 = sim4.bind_step_packed(pool4, packed4, stream=compute)

        def run4(count):
            for k in range(count):
                b = k & 1
                g4.wait_buffer_free(compute)
                step4(k & 3, b)
                g4.submit(packed4[b].cpu() if args.rehearse_shared_gpu else packed4[b])
                if len(g4.pending) > 64:
                    del g4.pending[:-2]

        def fence4():
            g4.drain()
            torch.cuda.synchronize(dev)
            dist.barrier()
            torch.cuda.synchronize(dev)
        run4(max(args.warmup, 10))
        fence4()
        t0 = time.perf_counter()
        run4(k4)
        fence4()
        dt4 = time.perf_counter() - t0
        tm = torch.tensor([dt4], device="cpu" if args.rehearse_shared_gpu else dev, dtype=torch.float64)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dt4 = float(tm.item())
        k_ms4 = sim4.time_step_kernel(pool4[0], packed4[0], 100)
        q4 = sim4.get_state()[0]
        config4 = {"workload": f"BASELINE config 4: {n4} envs/GPU x {world} GPU = {n4 * world} envs, random yaw at every (re)set, "
                                       f"frame_skip={args.frame_skip}, obs={od} f32, one {args.gather_op} of [{n4},{row}] f32 "
                                       f"({n4 * row * 4 / 1e6:.2f} MB per rank) to rank 0 per env-step, actions from a pool of 4 buffers",
                           "value": n4 * world * k4 / dt4, "unit": "env-steps/s", "steps": k4, "ms_per_step": dt4 / k4 * 1e3,
                           "exchange": "eager (one torch.distributed collective per env-step, issued asynchronously, double-buffered)",
                           "kernel_ms": k_ms4, "mapping": {_abi.MAP_QUAD: "quad", _abi.MAP_PAIR: "pair", _abi.MAP_LINK: "link",
                                                           _abi.MAP_LANE: "lane"}[sim4.mapping],
                           "state_finite": bool(np.isfinite(q4).all())}
        sim4.close()
        line["config4"] = config4
    # ---- multi-GPU, --exchange auto: the same K steps again as hipGraph replays, guarded by a watchdog -------------------------
    # The eager loop is bound by the host's cost of issuing one collective per step (29-39 us against a 14 us kernel); a graph of
    # 8 env-steps has no per-step host work.  (The same exchange from the library's C loop, --native-rccl, measured 32-36 us per step
    # with a 1-rank group: the cost is RCCL's own enqueue, not Python -- so it is not part of the automatic attempts.)  Graph replay of RCCL collectives across ranks cannot be validated on a one-GPU box,
    # so the eager measurement above is ALWAYS taken first and is what gets printed if the attempt raises, stalls or is slower.
    try_graph = (use_dist and native is None and walk is None and state["graph"] is None and args.exchange == "auto"
                 and not args.rehearse_shared_gpu and not args.sync_gather and args.gather_op == "gather")
    fault = os.environ.get("QG_BENCH_GRAPH_FAULT")     # test hook: "raise" / "stall" exercise the two fallbacks of the attempt
    # env-steps per graph: the largest even divisor of K up to 16 (whole double-buffer cycles; a replay costs ~10 us of launch whatever it holds)
    G = max([g for g in range(2, 17, 2) if args.steps % g == 0], default=0)
    if try_graph and G:
        import threading

        def give_up():
            # GPU work (a collective inside the replayed graph) did not finish: the eager measurement taken before it is still
            # valid and is printed, marked, but a stalled GPU is NOT a success -- the process ends with a non-zero status
            if rank == 0:
                line["graph_stalled"] = True
                line["config"]["exchange_note"] = (f"hipGraph replay of the per-step exchange did not complete within "
                                                   f"{args.graph_timeout:.0f} s on this node: eager result reported, exit status {STALL_EXIT}")
                print(json.dumps(line), flush=True)
            os._exit(STALL_EXIT)
        watchdog = threading.Timer(args.graph_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()

        from quadruped_gym_amd.dist import make_all_ok, negotiate_graph_replay

        def capture():
            if fault == "raise" or fault == f"raise@{rank}":
                raise RuntimeError("injected capture failure (QG_BENCH_GRAPH_FAULT=raise)")
            if fault == "stall":
                time.sleep(3600)
            build_graph(G)

        def local_fence():                               # no collective in here (the protocol's all-reduces are the barriers)
            if gatherer is not None:
                gatherer.drain()
            torch.cuda.synchronize(dev)

        def warm_replay():
            if fault == f"replay@{rank}":
                raise RuntimeError("injected replay failure (QG_BENCH_GRAPH_FAULT=replay@rank)")
            run(0, G)                                    # one replay as warm-up
            local_fence()

        def timed_replay():
            torch.cuda.set_stream(state["side"])
            run(args.warmup, args.steps)
            local_fence()

        def drop_graph():
            state["graph"] = None
            torch.cuda.set_stream(compute)

        def reduce_max(x):
            t = torch.tensor([x], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        # capture -> does EVERY rank hold a graph? -> one replay -> did every rank get through it? -> K steps -> closing all-reduce
        # -> MAX over ranks: quadruped_gym_amd.dist.negotiate_graph_replay (world-2 gloo tests with a rank that fails to capture /
        # to replay: tests/test_dist_gloo.py)
        att = negotiate_graph_replay(make_all_ok(None, dev), reduce_max, capture, warm_replay, timed_replay, drop_graph)
        ok, note, dt_g, agreed = att.ok, att.note, att.seconds, att.agreed
        watchdog.cancel()
        plausible = ok == 1 and dt_g / args.steps * 1e3 >= 0.9 * kernel_ms     # a step cannot take less than its own kernel
        tried = {"eager (one torch.distributed gather per step)": round(dt / args.steps * 1e6, 2)}      # us per step of every loop that completed
        if agreed and ok == 1:
            tried[f"hipGraph replay of {G} env-steps"] = round(dt_g / args.steps * 1e6, 2)
        if agreed and not plausible:
            note = f"hipGraph replay timed an implausible {dt_g / args.steps * 1e6:.1f} us per step (kernel alone: {kernel_ms * 1e3:.1f} us): discarded"
        if agreed and plausible and dt_g < dt:
            line = make_line(dt_g, kernel_ms, f"hipGraph replay of {G} env-steps (kernel + collective per step, no per-step host work); "
                                              f"eager loop measured first: {dt / args.steps * 1e6:.1f} us per step")
        else:
            line["graph_stalled"] = False
            line["config"]["exchange_note"] = (note or ("hipGraph attempt failed on another rank" if not agreed else
                                                        f"hipGraph replay measured slower ({dt_g / args.steps * 1e6:.1f} us per step)"))
    if try_graph and G:
        line["config"]["exchange_us_per_step"] = tried
    if config4 is not None:
        line["config4"] = config4

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(n, args.frame_skip, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if native is not None:
        lib.qg_comm_destroy(native)
    sim.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
