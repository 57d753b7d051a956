"""bench.py end to end on the GPU box: the JSON contract at N=1, and the N>1 control flow (stripe-local
build, stripe buffers, rotating-root reassembly, double buffering) with two ranks sharing the one GPU over
gloo - `--check` makes every rank compare each frame it ends up holding with a full-height warp."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


COMPACT_KEYS = {"metric", "value", "value_one_stream", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"}


def _assert_compact_contract(c, n):
    assert COMPACT_KEYS <= set(c), COMPACT_KEYS - set(c)
    assert c["n_gpus"] == n and c["value"] > 0 and c["ms_per_step"] > 0 and c["higher_is_better"] is True
    assert c["unit"] == "Mpixels/s" and c["dtype"] == "u8" and c["data"] == "synthetic" and c["vs_baseline"] is None
    assert {"workload", "frames_per_step", "ring_globes", "parallelism", "streams"} <= set(c["config"])
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "frac_traffic", "kernel_ms_per_launch"} <= set(c["roofline"])
    assert c["roofline"]["bound"] == "hbm" and c["roofline"]["peak"] == 8000.0
    assert abs(c["roofline"]["frac"] - c["roofline"]["achieved"] / c["roofline"]["peak"]) < 1e-3


def test_bench_line_and_check_single_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--repeats", "3", "--check",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "[check] rank 0: OK" in r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1 and r.stdout.rstrip().splitlines()[-1] == line[0]
    # the line the driver parses is the LAST line of stdout, compact (r5's 21 KB line was cut by the driver's 8 KB tail), and
    # carries the whole contract; the full record sits in the file it names
    compact = json.loads(line[0])
    assert len(line[0]) < 4096, len(line[0])
    _assert_compact_contract(compact, 1)
    assert compact["steps"] == 3 and compact["warmup"] == 1
    out = json.load(open(os.path.join(ROOT, compact["detail"])))
    assert out["value"] == compact["value"] and out["roofline"]["frac"] == compact["roofline"]["frac"]
    assert set(compact["extras"]) == {"C2", "C2x64", "C3", "C5", "headline, rubix on", "4K cube/hammer"}, compact["extras"]
    # the metric's other half: lensmap build ms at the headline's size - its own lens, C3's, and two forward-map lenses (r6: under / near 1 ms)
    assert set(compact["build_ms"]) == {"panini", "quincuncial", "winkel2", "polyconic"}, compact["build_ms"]
    assert all(isinstance(v, float) and 0.05 < v < 10.0 for v in compact["build_ms"].values()), compact["build_ms"]
    assert out["build_ms"]["winkel2"]["map"] == "forward" and out["build_ms"]["panini"]["map"] == "inverse"
    assert out["build_ms"]["winkel2"]["call_ms_best"] < 1.5 and out["build_ms"]["panini"]["call_ms_best"] < 0.5
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in out
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["dtype"] == "u8" and out["value"] > 0
    # one workload for the whole 1/2/4/8 curve: 64 frames per step at N = 1 too (the 16-frame job of rounds 1-4 rides along)
    assert out["config"]["frames_per_step"] == 64 and ", 64 frames/step" in out["config"]["workload"]
    assert out["value_at_16_frames"] > 0 and out["scaling_reference_mpx_s"] is None
    rf = out["roofline"]
    assert set(rf) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "frac_traffic", "compulsory_bytes_per_launch",
                       "frac_compulsory", "ring_globes", "warm_ring"}
    # the calibration beside it: a streaming kernel with the apply's read : write ratio, measured in the same run
    sm = rf["stream_mix"]
    assert "error" not in sm, sm
    assert 2000 < sm["GB/s"] < 8000 and 0.5 < sm["apply_traffic_over_it"] < 1.5, sm
    # the timed ring is larger than the 256 MiB Infinity Cache, and what the kernel must move cannot exceed the HBM peak
    assert rf["ring_globes"] * 6 * 2160 * 2176 > 4 * 256 * 2 ** 20
    assert 0 < rf["frac_compulsory"] <= 1.0
    assert rf["traffic"] is None or rf["frac_traffic"] <= 1.0
    assert out["timed_regions"]["count"] == 3
    assert out["timed_regions"]["mpx_s_min"] <= out["value"] <= out["timed_regions"]["mpx_s_max"]
    # like for like: the kernel alone (HIP events, one stream) cannot take longer than a step of the one-stream job
    assert out["value_one_stream"] > 0 and rf["kernel_ms_per_launch"] <= out["ms_per_step_one_stream"] * 1.15, (rf["kernel_ms_per_launch"], out["ms_per_step_one_stream"])
    # the engine's real call inside `roofline`: one frame per launch, the contract's algorithmic formula, <= 1
    sf = rf["single_frame"]
    assert sf["us"] > 0 and 0 < sf["algorithmic_frac"] <= 1.0, sf
    # ... and the same frames as commands to the resident kernel (bk_apply_resident_*): pipelined submissions, host wall clock per frame
    rs = sf["resident"]
    assert "error" not in rs, rs
    assert 0 < rs["us"] < 1.5 * sf["us"] and rs["blocks_in_registers"] >= 1 and 0 < rs["algorithmic_frac"], (rs, sf)
    # the other BASELINE.json configurations that fit one GPU, timed in the same run: C2 (1080p stereographic), C3 (4K quincuncial),
    # C5 (8K hammer x 64 in one launch), the headline with rubix on (7 B/px), 4K hammer
    extras = {c["name"].split(" ")[0]: c for c in out["configs_extra"]}
    assert set(extras) == {"C2", "C2x64", "C3", "C5", "headline,", "4K"}, list(extras)
    for c in out["configs_extra"]:
        assert "error" not in c, c
        assert c["value"] > 0 and c["kernel_us_per_launch"] > 0 and 0 < c["frac_compulsory"] <= 1.0 and 0 < c["single_frame"]["algorithmic_frac"] <= 1.0, c
        assert "error" not in c["single_frame"]["resident"], c["single_frame"]
    c2, c3, c5, rbx = extras["C2"], extras["C3"], extras["C5"], extras["headline,"]
    assert c2["workload"].startswith("1920x1080 cube/stereographic") and c3["workload"].startswith("3840x2160 cube/quincuncial")
    assert c5["workload"].startswith("7680x4320 cube/hammer") and ", 64 frames/step from a ring of 64" in c5["workload"]
    assert rbx["algorithmic_bytes_per_px"] == 7 and "rubix on" in rbx["workload"]
    # ... and their HBM bytes counted the way the headline's are (FETCH_SIZE / WRITE_SIZE under rocprofv3): at least what the frames
    # of output weigh (mapped pixels), and no more than 1.5x the compulsory model
    assert "traffic" in c2 and (c2["traffic"] is None or 1920 * 1080 * 16 < c2["traffic"] < 1.5 * c2["compulsory_bytes_per_launch"]), c2
    for c in out["configs_extra"]:
        assert c["traffic"] is None or (0 < c["frac_traffic"] <= 1.0 and c["traffic"] < 1.6 * c["compulsory_bytes_per_launch"]), c
    # the scaling curve predicted on one GPU: rank r's stripe for N = 2 / 4 / 8, slowest rank
    ps = out["predicted_stripe_complete"]
    assert "error" not in ps, ps
    assert all(ps[n]["speedup_vs_1"] > 0.5 for n in ("2", "4", "8")), ps
    assert ps["frames_per_launch"] == 64 and all(sum(ps[n]["stripe_rows"]) == 2160 for n in ("2", "4", "8")), ps
    assert all(ps["C4_trism_panini"][n]["speedup_vs_1"] > 0.5 and ps["frames16"][n]["speedup_vs_1"] > 0.5 for n in ("2", "4", "8")), ps
    # ... and the per-frame pipeline: every stripe through its own resident kernel, one frame per command (frame stride)
    f1 = ps["frames1_resident"]
    assert all("error" not in f1[n] and f1[n]["speedup_vs_1"] > 0.5 for n in ("2", "4", "8")), f1


def test_bench_two_ranks_sharing_the_gpu_reassemble_every_frame():
    env = dict(os.environ, BLINKY_BENCH_BACKEND="gloo", BLINKY_BENCH_ONE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--frames", "5", "--ring", "10", "--repeats", "2", "--check"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "[check] rank 0: OK" in r.stderr and "[check] rank 1: OK" in r.stderr
    compact = json.loads(r.stdout.rstrip().splitlines()[-1])
    assert len(r.stdout.rstrip().splitlines()[-1]) < 4096
    _assert_compact_contract(compact, 2)
    # (the N > 1 compact line also carries the three figures SURVEY.md 8(e) asks for)
    for k in ("stripe_complete_mpx_s", "assembled_on_rank0_mpx_s", "exchange", "stripes", "first_step_check_ok", "scaling_reference_mpx_s"):
        assert k in compact, k
    assert compact["first_step_check_ok"] is True and compact["exchange"]["bound_mpx_s"] > 0
    out = json.load(open(os.path.join(ROOT, compact["detail"])))
    assert out["n_gpus"] == 2 and out["config"]["frames_per_step"] == 5
    # the same JSON keys the RCCL path prints (the driver's SCALE run parses this line at N = 2 / 4 / 8), stripes cut by work, and the
    # first step checked on every rank before anything was timed
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "stripe_complete_mpx_s", "assembled_on_rank0_mpx_s", "exchange", "stripes", "first_step_check"):
        assert k in out, k
    assert out["first_step_check"] == {"ranks": ["ok", "ok"], "ok": True}
    assert out["stripes"]["rebalanced"] and sum(out["stripes"]["rows_per_rank"]) == 2160 and all(r % 8 == 0 for r in out["stripes"]["rows_per_rank"][:-1])


def test_bench_eight_ranks_sharing_the_gpu():
    """First contact for N = 8 (no 8-GPU node is available before the driver tries one): `bench.py --gpus 8` with all eight ranks on
    the one GPU over gloo - the control flow the RCCL run takes (stripes cut by work, rotating roots, first-step check on every rank,
    the one-GPU scaling reference timed by rank 0) - and the N = 8 compact line the driver parses."""
    env = dict(os.environ, BLINKY_BENCH_BACKEND="gloo", BLINKY_BENCH_ONE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--frames", "5", "--ring", "10", "--repeats", "2", "--check"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    for k in range(8):
        assert f"[check] rank {k}: OK" in r.stderr, k
    last = r.stdout.rstrip().splitlines()[-1]
    compact = json.loads(last)
    assert len(last) < 4096
    _assert_compact_contract(compact, 8)
    for k in ("stripe_complete_mpx_s", "assembled_on_rank0_mpx_s", "exchange", "stripes", "first_step_check_ok", "scaling_reference_mpx_s",
              "speedup_vs_scaling_reference"):
        assert k in compact, k
    assert compact["first_step_check_ok"] is True and compact["scaling_reference_mpx_s"] > 0 and compact["exchange"]["bound_mpx_s"] > 0
    assert len(compact["stripes"]["rows_per_rank"]) == 8 and sum(compact["stripes"]["rows_per_rank"]) == 2160
    assert compact["scaling"] == "strong" and "row-stripes x8" in compact["config"]["parallelism"]
    out = json.load(open(os.path.join(ROOT, compact["detail"])))
    assert out["first_step_check"] == {"ranks": ["ok"] * 8, "ok": True}


def test_bench_gpus_flag_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun environment must start two ranks itself (the round-1 bench silently
    measured one GPU); here both ranks share the one GPU over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(BLINKY_BENCH_BACKEND="gloo", BLINKY_BENCH_ONE_GPU="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--repeats", "2"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    compact = json.loads(r.stdout.rstrip().splitlines()[-1])
    _assert_compact_contract(compact, 2)
    out = json.load(open(os.path.join(ROOT, compact["detail"])))
    assert out["n_gpus"] == 2
    # the N > 1 line runs the SAME frames per step as the N = 1 line (no --frames given: the default), names the transport that was
    # timed, and carries the one-GPU job of that very workload, timed by rank 0 in the same run, as the curve's denominator
    assert out["config"]["frames_per_step"] == 64 and "gloo host exchange" in out["config"]["parallelism"]
    assert out["scaling_reference_mpx_s"] > 0 and out["speedup_vs_scaling_reference"] > 0


def test_stream_mix_calibration_entry():
    import blinky_amd
    ctx = blinky_amd.Context()
    ro = ctx.stream_mix(256 << 20, 8, 0)          # read-only
    cp = ctx.stream_mix(256 << 20, 8, 8)          # a copy: as much written as read
    assert 2000 < ro < 8000 and 2000 < cp < 8000, (ro, cp)
    for bad in ((1 << 10, 8, 0), (256 << 20, 0, 0), (256 << 20, 8, 9)):
        with pytest.raises(blinky_amd.BlinkyError):
            ctx.stream_mix(*bad)
    ctx.close()
