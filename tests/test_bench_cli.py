"""bench.py's launcher contract, checked without a GPU: a world size that differs from --gpus is refused, and the line the
driver parses stays compact whatever the full record holds."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=120,
                       cwd=ROOT, env=env)
    assert r.returncode != 0
    assert "--gpus 4" in r.stderr and "refusing" in r.stderr
    assert "{" not in r.stdout            # no JSON line for a run that did not happen


def test_compact_line_of_a_recorded_full_record_fits_the_drivers_tail():
    """round 5's full record was a 21 KB line and the driver's 8 KB tail cut its head off (BENCH_r05.json: parsed = null): the
    line printed last must stay under 4 KB and keep every key of the contract"""
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_line_final_code.json")))
    assert len(json.dumps(full)) > 8192
    text = bench.compact_line(full)
    assert len(text) < 4096 and "\n" not in text
    c = json.loads(text)
    for k in ("metric", "value", "value_one_stream", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in c, k
    assert c["value"] == full["value"] and c["roofline"]["frac"] == full["roofline"]["frac"]
    assert c["cpu_baseline"]["kind"] == "reference" and c["cpu_baseline"]["cores"] == 1 and c["cpu_baseline"]["allcores"]["cores"] > 1
    assert set(c["config"]) == {"workload", "frames_per_step", "ring_globes", "parallelism", "streams"}
    # a record bloated by any number of extra configurations still yields a line under the limit
    full["configs_extra"] = full["configs_extra"] * 40
    for i, e in enumerate(full["configs_extra"]):
        full["configs_extra"][i] = dict(e, name=f"{e['name']} #{i}")
    assert len(bench.compact_line(full)) < 4096


def test_the_hash_bench_check_uses_is_the_goldens_hash():
    """bench.py --check hashes frames with the product's bk_debug_fnv1a64 (the oracle is off limits there): it must be the FNV-1a-64
    tests/golden/lensmaps.json was recorded with"""
    import numpy as np
    sys.path.insert(0, ROOT)
    import blinky_amd
    import oracle_ffi as O
    rng = np.random.default_rng(11)
    for n in (0, 1, 17, 4096):
        a = rng.integers(0, 256, n, dtype=np.uint8)
        assert blinky_amd.ffi.fnv1a64(a) == O.fnv(a)
    assert blinky_amd.ffi.fnv1a64(np.zeros(0, np.uint8)) == "cbf29ce484222325"
