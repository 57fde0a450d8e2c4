"""bench.py --gpus N must drive N ranks however it is started: a plain
`python bench.py --gpus 2` launches two ranks itself, and never times fewer GPUs than the flag says."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra, *flags, timeout=300):
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env.update({k: v for k, v in env_extra.items()})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_gpus2_refuses_on_a_node_with_fewer_devices():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return                      # a real multi-GPU node: nothing to refuse
    r = _run({}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "launching 2 ranks" in r.stderr and "refusing to time fewer GPUs" in r.stderr
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines()), "no bench line may be printed"


def test_gpus2_attempts_a_two_rank_launch():
    """with the device-count check overridden the script really starts two ranks under torch.distributed.run;
    without GPUs each rank fails loudly (no CPU fallback) and so does the launcher"""
    import torch
    if torch.cuda.is_available():
        return
    r = _run({"YL_BENCH_FORCE_LAUNCH": "1"}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "launching 2 ranks" in r.stderr
    assert "rank 0 of 2 needs GPU 0" in r.stderr and "rank 1 of 2 needs GPU 1" in r.stderr
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_world_size_must_match_the_flag():
    r = _run({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_PORT": "29999"}, "--gpus", "2")
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in r.stderr
