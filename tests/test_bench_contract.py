"""bench.py's control flow and output contract, checked WITHOUT a GPU through its --selftest mode (tiny model, ABI
simulator, gloo): one JSON line from rank 0 with the required keys; the 2-rank launch (the way the driver starts
`--gpus N`) goes through the barriers, the max-over-ranks timing and the gradient all-reduce without deadlocking."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def _json_line(out):
    lines = [line for line in out.splitlines() if line.startswith("{")]
    assert len(lines) == 1, f"expected exactly one JSON line, got {len(lines)}:\n{out[-2000:]}"
    return json.loads(lines[0])


def _check(d, n, steps, warmup, bs=1):
    assert REQUIRED <= set(d), REQUIRED - set(d)
    assert d["n_gpus"] == n and d["steps"] == steps and d["warmup"] == warmup
    assert d["unit"] == "images/sec" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - n * bs * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert f"bs={bs}/GPU" in d["metric"] and f"bs={bs}/GPU" in d["config"]["workload"]
    assert "SELFTEST" in d["data"]  # a self-test line can never be mistaken for a measurement


def _free_port():
    """a port nobody listens on right now (a fixed number can be busy on a shared host, and torchrun then waits for minutes)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_single_process_selftest():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    _check(_json_line(p.stdout), 1, 2, 1)


def test_single_process_selftest_with_a_per_gpu_batch():
    """`--bs 2` (the `secondary.c2_bs4` line runs the real thing at 4): the batch the step sees has that many prompts and the
    value counts every one of them"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest", "--steps", "2", "--warmup", "1", "--bs", "2"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    _check(_json_line(p.stdout), 1, 2, 1, bs=2)


def test_two_rank_launch_like_the_driver():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--selftest"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    _check(_json_line(p.stdout), 2, 2, 1)


import pytest  # noqa: E402


@pytest.mark.parametrize("fail", ["1:segments:1", "0:segments:1", "1:segments:0"])
def test_a_capture_failure_on_one_rank_sends_every_rank_to_eager_launches(fail):
    """fault injection (rank : stepper : call index; call 0 = building the stepper): rank 1's first capturing step raises.  The failing rank re-runs that step eagerly (the other rank
    is waiting in the step's two all-reduces), the outcome is agreed afterwards, and BOTH ranks time eager launches - no
    deadlock, no unmatched collective, one JSON line that says what happened."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["COMAT_SELFTEST_FAIL"] = fail
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--selftest"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    d = _json_line(p.stdout)
    _check(d, 2, 2, 1)
    assert "eager launches" in d["config"]["launch_mode"] and "failed" in d["config"]["launch_mode"], d["config"]["launch_mode"]
    assert "injected failure" in p.stderr


def test_precapture_plan_visits_every_segment_variant():
    """with attribute concentration every trained-call slot has two graph variants (attention maps captured or not) and the
    reference draws the capturing steps at random each optimisation step (training_script.py:589-590): the steps bench.py
    runs BEFORE its timed region must leave no variant to be captured inside it, whatever is drawn later."""
    import random
    sys.path.insert(0, ROOT)
    import bench
    from comat_amd.step import StepConfig, sample_training_steps

    scfg = StepConfig(total_step=50, K=5, attrcon=True, attrcon_train_steps=2)
    fixed = dict(crop=(1, 1, 510, 510))
    visited = set()
    for kw in list(bench.precapture_plan(scfg, fixed)) + [fixed]:
        ts = kw.get("training_steps")
        if ts is None:  # the stepper draws them itself: any draw visits the non-capturing variant of slots it does not pick
            continue
        order = sorted(ts)
        for slot, t in enumerate(order):
            visited.add((slot, t in kw.get("attrcon_steps", ())))
    assert visited == {(slot, cap) for slot in range(scfg.K) for cap in (False, True)}, sorted(visited)
    rng = random.Random(3)
    for _ in range(200):  # every later draw only needs variants that exist
        ts = sample_training_steps(scfg.total_step, scfg.K, rng)
        caps = rng.sample(ts, scfg.attrcon_train_steps)
        assert {(slot, t in caps) for slot, t in enumerate(sorted(ts))} <= visited
