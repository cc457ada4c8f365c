"""The driver's record keeps the first 24 keys of `config` and of `roofline` of bench.py's line (VERDICT r05 Weak 3): the digest of
the secondary legs and the once-per-step scalars must sit inside them.  Checked on the dict literals of bench.py itself (no GPU)."""
import ast
import os

from conftest import ROOT

KEEP = 24


def _dict_literals():
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.Dict):
            keys = [k.value for k in node.keys if isinstance(k, ast.Constant) and isinstance(k.value, str)]
            yield keys, node


def _find(first_key, must_have):
    for keys, node in _dict_literals():
        if keys and keys[0] == first_key and must_have in keys:
            return keys
    raise AssertionError(f"no dict literal starting with {first_key!r} that holds {must_have!r}")


def test_secondary_digest_is_inside_the_first_keys_of_config():
    keys = _find("workload", "secondary_digest")
    assert keys.index("secondary_digest") == 1 < KEEP
    # the scalars the digest does not carry but a reader of the record wants: still inside the kept keys
    for k in ("state_rmse_vs_oracle", "max_abs_diff_vs_oracle", "mean_detections_per_step"):
        assert keys.index(k) < KEEP, k


def test_once_per_step_scalars_follow_frac_in_roofline():
    keys = _find("bound", "once_per_step_frac")
    i = keys.index("frac")
    assert keys[i + 1] == "once_per_step_frac" and keys[i + 2] == "once_per_step_value"
    assert keys.index("once_per_step_value") < KEEP and keys.index("traffic") < KEEP and keys.index("kernel_ms") < KEEP


def test_digest_format_is_short_and_complete():
    import bench
    line = {"secondary": [
        {"name": "configs[2] UKF", "value": 5.32e6, "roofline": {"frac": 0.196}, "config": {"parity_check": {"max_abs_diff": 0.0}}},
        {"name": "configs[4] pgs", "value": 10400.0, "roofline": {"frac": 0.067},
         "config": {"parity_check": {"max_abs_diff_m": 4e-11, "lm_iteration_and_trial_counts_equal": True}}},
        {"name": "configs[4] every-iteration mode: pose graph", "value": 15200.0, "roofline": {"frac": 0.016},
         "config": {"parity_check": {"max_abs_diff_m": 4e-11, "lm_iteration_and_trial_counts_equal": True}}},
        {"name": "configs[3] f32", "value": 73.3e6, "roofline": {"frac": 0.40}, "config": {"parity_check": {"max_abs_diff": 0.0}}},
        {"name": "configs[1] L20", "value": 242e6, "roofline": {"frac": 0.11}, "config": {"parity_check": {"max_abs_diff": 0.0}}}],
        "roofline": {"once_per_step": {"value": 27.2e6, "frac": 0.63}}}
    d = bench.secondary_digest(line)
    assert len(d) <= 110 and d.count("|") >= 5, (len(d), d)
    for tag in ("ukf 5.32M f.20 p0", "pgs 10.4k f.07 p4e-11", "pgsit 15.2k p4e-11", "f32 73.3M", "L20 242M", "1step 27.2M"):
        assert tag in d, (tag, d)
