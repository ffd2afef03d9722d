"""The driver's plain command `python bench.py --gpus N --steps K --warmup W` must start its own N ranks (one process
per GPU under torch.distributed.run, 127.0.0.1, a free port) and have rank 0 print exactly one JSON line.
CPU leg: the launch / rendezvous / reduce plumbing alone (`--launch-check`, gloo, no training step).
GPU leg: the whole data-parallel bench with 2 gloo ranks sharing the test box's one GPU - a rehearsal of the N > 1
command line, not a measurement (RCCL wants one device per rank; the 2-device RCCL tests are in test_gpu_trainer.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra, timeout):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)                       # the driver's N = 1 form of the command: no launcher environment
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, cwd=ROOT, timeout=timeout,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


@pytest.mark.timeout(300)
def test_plain_command_launches_its_own_ranks():
    p, lines = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--launch-check"], {"SW_BENCH_BACKEND": "gloo"}, 280)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["launch_check"] is True and rec["value"] is None


@pytest.mark.timeout(120)
def test_world_size_mismatch_is_refused():
    p, lines = _run(["--gpus", "4", "--launch-check"], {"WORLD_SIZE": "2", "RANK": "0", "SW_BENCH_BACKEND": "gloo"}, 100)
    assert p.returncode != 0 and not lines
    assert "WORLD_SIZE" in p.stderr


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_rank_bench_line_on_one_box():
    """`python bench.py --gpus 2 --steps 4 --warmup 2` with SW_BENCH_BACKEND=gloo (both ranks on cuda:0): exit 0, one
    JSON line, n_gpus = 2, three all-reduces per step, replicas bit-identical after the run."""
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    p, lines = _run(["--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-other-workloads",
                     "--no-sustained"], {"SW_BENCH_BACKEND": backend, "SW_BENCH_SETTLE_STEPS": "8", "SW_BENCH_FORCE_PROBE": "1"}, 850)
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["steps"] == 4
    cfg = rec["config"]
    # the child job that tries in-graph collectives before the headline ran (forced here on gloo, where the answer is no)
    pr = cfg["collectives_probe"]
    assert pr is not None and "error" not in pr, pr
    assert pr["replicas_identical"] is True and pr["graph_collectives"] == (backend == "nccl")
    assert cfg["collectives"] == ("in-graph" if backend == "nccl" else "between graph segments")
    assert cfg["rccl_ranks"] == 2 and cfg["allreduces_per_step"] == 3 and cfg["replicas_identical"] is True
    assert cfg["backend"] == backend
    assert rec["value"] > 0 and abs(rec["value"] - 2 * 1e3 / rec["ms_per_step"]) < 1e-6 * rec["value"]


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_rank_bench_line_reports_both_gradient_exchanges():
    """N > 1 without --no-other-workloads: the line carries config.exchange - microseconds per all-reduce of the step's three
    buckets on the process group and on the library's direct exchange (SW_ALLREDUCE=direct), and the step on the direct
    form - so that a scaling run explains itself (2 ranks sharing cuda:0 here: a rehearsal of the plumbing)."""
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    p, lines = _run(["--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-sustained"],
                    {"SW_BENCH_BACKEND": backend, "SW_BENCH_SETTLE_STEPS": "8"}, 850)
    assert p.returncode == 0, p.stderr[-3000:]
    rec = json.loads(lines[-1])
    ex = rec["config"]["exchange"]
    assert ex is not None and "direct_error" not in ex, ex
    assert ex["buckets_floats"] == [27942, 27942, 86124]      # packed D, D, G gradient buffers (tensors on 4-float boundaries)
    assert len(ex["group_us"]) == 3 and len(ex["direct_us"]) == 3 and min(ex["direct_us"]) > 0 and ex["direct_status"] == 0
    assert ex["direct_step"]["steps_s"] > 0 and ex["direct_step"]["collectives"] == "in-graph" and ex["direct_step"]["status"] == 0
    assert ex["direct_headline"]["value"] > 0 and ex["direct_headline"]["steps"] == 4 and ex["direct_headline"]["status"] == 0
    # (RCCL ranks only: a child job tries in-graph collectives before the headline; gloo ranks sharing the device skip it)
    assert (rec["config"]["collectives_probe"] is None) == (backend != "nccl")


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_a_crash_in_the_exchange_report_cannot_lose_the_line():
    """r5 verdict item 1: the exchange report (the library's direct exchange has never crossed an xGMI link) runs in a CHILD
    JOB.  SW_COMM_FAULT_INJECT=1 makes sw_comm_ipc_import abort() the process - what a GPU fault in the peer mapping would
    do: the child job dies, the parent's ranks do not; the line comes out valid, with config.exchange.error set and the
    group's own all-reduce times (measured before the fault) kept."""
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    p, lines = _run(["--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-sustained"],
                    {"SW_BENCH_BACKEND": backend, "SW_BENCH_SETTLE_STEPS": "8", "SW_COMM_FAULT_INJECT": "1"}, 850)
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["config"]["replicas_identical"] is True
    ex = rec["config"]["exchange"]
    assert ex is not None and "error" in ex, ex
    assert len(ex["group_us"]) == 3 and "direct_us" not in ex


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_a_dead_side_leg_child_cannot_lose_the_single_gpu_line():
    """The N = 1 line's side legs run in child processes: a child that cannot even start its legs (here: its time limit is
    too short for the first one) leaves {"error": ...} records, the headline, roofline and cpu_baseline are untouched."""
    p, lines = _run(["--steps", "4", "--warmup", "2", "--no-sustained"],
                    {"SW_BENCH_SETTLE_STEPS": "8", "SW_BENCH_CHILD_TIMEOUT_S": "1"}, 850)
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and rec["roofline"]["frac"] > 0 and rec["cpu_baseline"]["value"] > 0
    ow = rec["config"]["other_workloads"]
    assert all("error" in ow["_%s_child" % g] for g in ("shapes", "dp1", "extra")), ow
