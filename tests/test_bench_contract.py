"""bench.py's bookkeeping that can be checked without a GPU."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_bytes_follow_survey_8d():
    b = _bench()
    # 196 + 48 + 4 read, 196 + 4 + 132 + 4 + 4 written = 588 B; 540 B with the 21-value pack
    assert b.algorithmic_bytes_per_env_step(33) == 588
    assert b.algorithmic_bytes_per_env_step(21) == 540
    assert b.HBM_PEAK_GBS == 8000.0


def test_traffic_index_is_consistent():
    import json
    idx = json.load(open(os.path.join(ROOT, "profiles", "traffic_index.json")))
    for key in ("quad_n4096_fs4_obs33", "link_n4096_fs4_obs33"):
        assert os.path.exists(os.path.join(ROOT, idx[key]["source"]))
    ent = idx["link_n4096_fs4_obs33"]
    assert os.path.exists(os.path.join(ROOT, ent["source"]))
    # writes match the algorithmic 340 B per env plus the 48 B of data.ctrl the step maintains (quadruped.py:164; tracking is on in
    # the timed loop since round 2) and the rare reset bookkeeping
    assert abs(ent["write"] - 388 * 4096) < 0.001 * 388 * 4096
    assert 0.9 * 588 * 4096 < ent["hbm_bytes_per_launch"] < 2.0 * 588 * 4096
    fl = idx["flops_quad_fs4"]
    assert os.path.exists(os.path.join(ROOT, fl["source"])) and 30e3 < fl["flops_per_env_step"] < 80e3      # SURVEY 8(d) estimated ~50 kflop
    assert 80e3 < idx["flops_link_fs4"]["flops_per_env_step"] < 160e3     # 16 lanes per env: the replicated base part and chain are counted


import json
import socket
import subprocess
import sys

import pytest

CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.gpu
def test_bench_prints_one_contract_line_on_one_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "200", "--warmup", "20", "--cpu-seconds", "1"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert CONTRACT_KEYS <= set(d) and "cpu_baseline" in d
    assert d["n_gpus"] == 1 and d["steps"] == 200 and d["warmup"] == 20 and d["higher_is_better"] is True
    assert d["metric"] == "env_steps_per_sec" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert abs(d["value"] - 4096 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["kernel"] == "qg_step_kernel_link"
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and d["state_finite"] is True
    # roofline.traffic / valu_issue / fp32 come from committed counter profiles, not from this run: the line says on which build of the
    # library they were taken and whether that is the one loaded now (profiles/traffic_index.json is stamped by
    # tools/update_traffic_index.py with qg_build_id(), a hash of the library's sources)
    from quadruped_gym_amd import _abi
    assert r["build_id"] == _abi.load_library().qg_build_id().decode() and len(r["build_id"]) == 16
    assert r["profile_stale"] == (r["profile_build_id"] != r["build_id"]) and isinstance(r["profile_stale"], bool)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [["--seq", "16"], ["--resident", "ahead"], ["--resident", "closed"]])
def test_bench_opt_in_forms_say_what_they_are(mode):
    """The opt-in forms that keep the state in registers across env-steps print the same contract line, name themselves in
    `config.workload`, restate the algorithmic bytes for the traffic they really do (action in, packed row out, state once per launch)
    and never change the default line's keys."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "320", "--warmup", "32", "--no-cpu-baseline"] + mode,
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert CONTRACT_KEYS <= set(d) and d["steps"] == 320 and d["state_finite"] is True
    assert abs(d["value"] - 4096 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "OPT-IN" in d["config"]["workload"] and "pool of 16" in d["config"]["workload"]
    r = d["roofline"]
    assert r["kernel"] == "qg_step_kernel_link_multi" and r["traffic"] is None
    assert 188 <= r["algorithmic_bytes_per_env_step"] < 588        # 48 B in, 140 B out; + 448 / 16 B for a 16-step launch's state
    assert 0.004 < d["ms_per_step"] < 0.03


@pytest.mark.gpu
@pytest.mark.parametrize("envs_per_gpu", [512, 32768])
def test_bench_two_ranks_rehearsal_on_one_gpu(envs_per_gpu):
    """The N > 1 code path of bench.py (rank-local shards, per-step gather, max-over-ranks timing, one line from rank 0),
    launched exactly as the driver launches it; both ranks share GPU 0 and the exchange goes through gloo, because a one-GPU
    box cannot host two RCCL ranks.  32 768 envs per rank is BASELINE config 4's per-GPU shard (262 144 envs over 8 GPUs, one
    gather of the packed [32 768, 35] f32 rows per step)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "60", "--warmup", "10",
           "--envs-per-gpu", str(envs_per_gpu), "--rehearse-shared-gpu", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                # rank 0 only
    d = json.loads(lines[0])
    assert CONTRACT_KEYS <= set(d)
    assert d["n_gpus"] == 2 and d["steps"] == 60 and d["scaling"] == "weak" and d["state_finite"] is True
    assert abs(d["value"] - 2 * envs_per_gpu / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]      # whole-job aggregate
    assert d["config"]["envs_per_gpu"] == envs_per_gpu and d["config"]["ctrl_tracking"] is True
    # the line explains itself (round 4): who the communicator saw -- here two ranks that share GPU 0, which must SHOW -- and BASELINE
    # config 4's shard timed in the same process (the driver only runs `bench.py --gpus N`, i.e. 4096 envs per GPU)
    r = d["config"]["rccl"]
    assert r["world_size"] == 2 and r["ranks_seen"] == 2 and r["distinct_gpus"] == 1 and r["backend"] == "gloo"
    assert [m["rank"] for m in r["members"]] == [0, 1] and all(m["pci_bus_id"] for m in r["members"])
    c4 = d["config4"]
    assert c4["steps"] == 200 and c4["state_finite"] is True and "32768 envs/GPU x 2 GPU" in c4["workload"]
    assert abs(c4["value"] - 2 * 32768 / (c4["ms_per_step"] * 1e-3)) < 1e-6 * c4["value"] and c4["mapping"] == "pair"
    assert "pool of 16" in d["config"]["workload"]


@pytest.mark.gpu
@pytest.mark.parametrize("fault", [None, "raise", "stall"])
def test_exchange_auto_reports_a_valid_line_whatever_the_graph_attempt_does(fault):
    """--exchange auto (the multi-GPU default) measures the eager loop first and then tries hipGraph replays under a watchdog.
    With a 1-rank RCCL group (--force-gather) on this box: the attempt succeeds; an injected exception falls back to the eager
    line; an injected stall makes the watchdog print the eager line, marked `graph_stalled`, and end the process with a NON-ZERO
    status (a hung GPU is not a success)."""
    env = dict(os.environ)
    if fault:
        env["QG_BENCH_GRAPH_FAULT"] = fault
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["MASTER_PORT"] = str(_free_port())
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-gather", "--steps", "400", "--warmup", "40",
                          "--no-cpu-baseline", "--graph-timeout", "20"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == (3 if fault == "stall" else 0), out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d.get("graph_stalled", False) is (fault == "stall")
    assert CONTRACT_KEYS <= set(d) and d["steps"] == 400 and d["n_gpus"] == 1
    assert abs(d["value"] - 4096 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    if fault is None:
        assert d["config"]["exchange"].startswith("hipGraph replay of 16 env-steps") and d["ms_per_step"] < 0.03
        assert len(d["config"]["exchange_us_per_step"]) == 2             # both loops completed and are on record
    else:
        assert d["config"]["exchange"] == "eager" and "exchange_note" in d["config"]
        assert ("injected" in d["config"]["exchange_note"]) == (fault == "raise")
