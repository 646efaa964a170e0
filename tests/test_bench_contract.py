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
    ent = idx["quad_n4096_fs4_obs33"]
    assert os.path.exists(os.path.join(ROOT, ent["source"]))
    assert abs(ent["write"] - 340 * 4096) < 0.001 * 340 * 4096   # writes match the algorithmic 340 B per env (plus the rare reset bookkeeping)
    assert 0.9 * 588 * 4096 < ent["hbm_bytes_per_launch"] < 2.0 * 588 * 4096
