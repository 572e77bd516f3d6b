"""`python bench.py --gpus N` must become N ranks by itself: the driver's command
carries no launcher.  --dry-run swaps the HIP step for a stub so that the spawn, the barriers, the max-reduce
over ranks and the JSON contract can be exercised on a box without GPUs (gloo)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "2", "--warmup", "1"] + extra,
                       capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    return p


def _json_lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.startswith("{")]


def test_gpus_2_spawns_two_ranks_and_prints_one_line():
    p = _run(["--gpus", "2"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout
    line = lines[0]
    assert line["n_gpus"] == 2 and line["dry_run"] is True
    assert line["config"]["global_batch"] == 2 * line["config"]["per_gpu_batch"]
    assert line["scaling"] == "weak" and line["higher_is_better"] is True and line["steps"] == 2
    assert line["timing"]["repeats"] >= 5 and line["timing"]["statistic"] == "median repeat"
    for key in ("metric", "value", "unit", "warmup", "ms_per_step", "vs_baseline", "dtype", "data", "config"):
        assert key in line


def test_gpus_8_dry_run_lists_every_rank_and_pins_cores():
    """Round 5: the driver's 8-GPU launch shape on gloo -- eight ranks, one line, every rank's own median step
    time and calibration slot in `per_rank` (a straggler is visible), each rank on its own slice of the host's cores."""
    p = _run(["--gpus", "8"], timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout
    line = lines[0]
    assert line["n_gpus"] == 8 and line["config"]["global_batch"] == 8 * line["config"]["per_gpu_batch"]
    pr = line["per_rank"]
    assert [r["rank"] for r in pr] == list(range(8)) and all(r["ms_per_step"] > 0 for r in pr)
    ncpu = len(os.sched_getaffinity(0))
    if ncpu >= 8:
        assert all(r["cpus"] == ncpu // 8 for r in pr), pr      # LOCAL_RANK's slice, not the whole host
    # the whole-job figure is the slowest rank's clock
    assert line["ms_per_step"] >= max(r["ms_per_step"] for r in pr) * 0.5


def test_single_rank_line_and_world_size_mismatch():
    p = _run(["--gpus", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert _json_lines(p.stdout)[0]["n_gpus"] == 1
    # a launcher that set WORLD_SIZE=1 while the command says --gpus 2 is an error, not a silent 1-GPU run
    p = _run(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_no_gpu_without_dry_run_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        return
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert p.returncode != 0 and "no CPU fallback" in (p.stderr + p.stdout)
