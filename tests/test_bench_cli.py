"""bench.py's launcher contract, checked without a GPU: a world size that differs from --gpus is refused."""
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
